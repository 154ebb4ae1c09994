# The reference's train.sh shape (64x64, 16 classes, z 32, global batch 550) on synthetic data for two
# iterations: once on one rank, once on two ranks that share one GPU through the host-staged gloo exchange.
# Run on the GPU box from the repo root:   bash tools/trainer_rehearsal.sh
set -e
ARGS="--dataset Synth --data_dir /tmp/none --data_size 1100 --pix_dim 64 --y_dim 16 --z_dim 32 --batch_size 550 --epoch 1 --max_iters 2 --lrG 3e-5 --worker 0"
timeout -k 10 300 python -m cvae_gan_amd.starter $ARGS --gpus 0 --save_dir /tmp/ck1 --result_dir /tmp/res1 2>&1 | grep -v amdgpu | tail -4
export CVG_DDP_BACKEND=gloo CVG_DDP_SHARE_GPU=1
timeout -k 10 300 python -m cvae_gan_amd.starter $ARGS --gpus 0,0 --save_dir /tmp/ck2 --result_dir /tmp/res2 2>&1 | grep -v "amdgpu\|Gloo" | tail -4
