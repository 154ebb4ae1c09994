"""Command line of the trainer: the reference's starter.py flags (starter.py:17-35) plus
--dtype / --max_iters / --seed.

    python -m cvae_gan_amd.starter --dataset PoseUnit-stretch --data_dir ./data --z_dim 32 \\
        --y_dim 16 --pix_dim 64 --batch_size 256 --lrG 3e-5 [--resume | --testmode --pkl <prefix>]

--gpus 0,1,...: one training process per listed GPU, started by this command itself (or start the
ranks with torchrun: RANK / LOCAL_RANK / WORLD_SIZE are then taken from the environment and rank r
uses the r-th listed GPU).  --batch_size is the global batch, split evenly over the ranks.
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
    p.add_argument("--lam_pd", type=float, default=1e-3, help="weight of the pairwise discriminator-feature loss")
    p.add_argument("--lam_pc", type=float, default=1e-3, help="weight of the pairwise classifier-feature loss")
    return p


def _gpu_list(spec):
    return [int(g) for g in spec.split(",") if g.strip() != ""] or [0]


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    from . import ddp
    rank, local_rank, world = ddp.env_world()
    gpus = _gpu_list(args.gpus)
    n_dev = torch.cuda.device_count()             # counting devices does not initialise the GPU runtime
    if n_dev == 0:
        raise SystemExit("cvae_gan_amd needs an MI355X: no GPU is visible")
    if world == 1 and "RANK" not in os.environ:
        # --gpus 0,1 (the reference's train.sh:9; there nn.DataParallel over the listed devices,
        # starter.py:40) = one training process per listed GPU.  This process has not touched the GPU
        # yet: it starts the ranks as children of the launcher module and leaves with their exit code.
        usable = [g for g in gpus if 0 <= g < n_dev]
        if len(usable) < len(gpus):
            print("--gpus %s: only %d device(s) visible, using %s" % (args.gpus, n_dev, usable or [0]), flush=True)
        usable = usable or [0]
        if len(usable) > 1 and not args.testmode:
            import subprocess
            import sys
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(usable)),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "cvae_gan_amd.starter"]
            cmd += list(sys.argv[1:] if argv is None else argv)
            raise SystemExit(subprocess.call(cmd))
        gpus = usable
    # one process drives one GPU: rank r of a multi-process job takes the r-th entry of --gpus.  (The
    # visibility variables are left alone: the HIP runtime is already loaded by the time the flags are
    # parsed, and changing them afterwards leaves it without a device.)
    if world > 1:
        if len(gpus) < world and os.environ.get("CVG_DDP_SHARE_GPU") is None:
            raise SystemExit("--gpus %s lists %d device(s) for %d ranks" % (args.gpus, len(gpus), world))
        dev = gpus[local_rank % len(gpus)]
    else:
        dev = gpus[0]
    if not 0 <= dev < n_dev:
        raise SystemExit("--gpus names device %d, but %d device(s) are visible" % (dev, n_dev))
    torch.cuda.set_device(dev)
    args.device = torch.device("cuda", dev)
    if world > 1:
        ddp.init_process_group(args.device)
    from .cvae import CVAE
    if rank == 0:
        os.makedirs(args.result_dir, exist_ok=True)
        os.makedirs(args.save_dir, exist_ok=True)
    model = CVAE(args)
    if args.testmode:
        path = model.test()
        if rank == 0:
            print("wrote", path)
    else:
        model.train()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
