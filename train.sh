#!/bin/sh
# The reference's train.sh (same flags) on this package's trainer.  --gpus 0,1: one training process
# per listed GPU (only the visible ones are used), --batch_size is the global batch.
python -m cvae_gan_amd.starter --dataset "PoseUnit-stretch" \
                        --data_dir "./data" \
                        --epoch 10000 \
                        --batch_size 550 \
                        --data_size -1 \
                        --z_dim 32 \
                        --y_dim 16 \
                        --pix_dim 64 \
                        --gpus 0,1 \
                        --worker 48 \
                        --lrG 3e-5 \
                        --resume
