"""Command line of the trainer: the reference's starter.py flags (starter.py:17-35) plus
--dtype / --max_iters / --seed.

    python -m cvae_gan_amd.starter --dataset PoseUnit-stretch --data_dir ./data --z_dim 32 \\
        --y_dim 16 --pix_dim 64 --batch_size 256 --lrG 3e-5 [--resume | --testmode --pkl <prefix>]
"""
import argparse
import os


def build_parser():
    p = argparse.ArgumentParser(description="Training GANs or VAEs")
    p.add_argument("--dataset", type=str, required=True)
    p.add_argument("--data_dir", type=str, required=True)
    p.add_argument("--epoch", type=int, default=200)
    p.add_argument("--batch_size", type=int, default=50)
    p.add_argument("--data_size", type=int, default=-1)
    p.add_argument("--result_dir", default="result")
    p.add_argument("--pkl", default="")
    p.add_argument("--save_dir", default="checkpoint")
    p.add_argument("--z_dim", type=int, default=256)
    p.add_argument("--y_dim", type=int, default=1000)
    p.add_argument("--pix_dim", type=int, default=512)
    p.add_argument("--cls_num", type=int, default=12)
    p.add_argument("--gpus", type=str, default="0,1")
    p.add_argument("--gan_type", type=str, default="VAE")
    p.add_argument("--worker", type=int, default=10)
    p.add_argument("--lrG", type=float, default=1e-3)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--testmode", action="store_true")
    # additions
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--max_iters", type=int, default=0, help="stop after this many iterations (0: run the epochs)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--lam_gd", type=float, default=1e-3, help="weight of the discriminator feature-matching loss")
    p.add_argument("--lam_gc", type=float, default=1e-3, help="weight of the classifier feature-matching loss")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    from .cvae import CVAE
    if not torch.cuda.is_available():
        raise SystemExit("cvae_gan_amd needs an MI355X: no GPU is visible")
    # one process drives one GPU: the first index of --gpus, if it is visible.  (The visibility
    # variables are left alone: the HIP runtime is already loaded by the time the flags are parsed,
    # and changing them afterwards leaves it without a device.)
    first = int(args.gpus.split(",")[0] or 0)
    if 0 <= first < torch.cuda.device_count():
        torch.cuda.set_device(first)
    args.device = torch.device("cuda", torch.cuda.current_device())
    os.makedirs(args.result_dir, exist_ok=True)
    os.makedirs(args.save_dir, exist_ok=True)
    model = CVAE(args)
    if args.testmode:
        print("wrote", model.test())
    else:
        model.train()


if __name__ == "__main__":
    main()
