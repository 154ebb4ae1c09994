#!/bin/sh
# The reference's train.sh on this package's trainer (one process drives one GPU).
python -m cvae_gan_amd.starter --dataset "PoseUnit-stretch" \
                        --data_dir "./data" \
                        --epoch 10000 \
                        --batch_size 256 \
                        --data_size -1 \
                        --z_dim 32 \
                        --y_dim 16 \
                        --pix_dim 64 \
                        --gpus 0 \
                        --worker 16 \
                        --lrG 3e-5 \
                        --resume
