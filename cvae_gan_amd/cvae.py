"""Trainer with the reference's `CVAE` surface (models/cvae.py:23-247) on the fused engine.

Same constructor argument object (starter.py's flags), same attributes, `train()`, `test()`,
`save()`, `load(pkl=None)`, `model_dir`, and the same checkpoint files: `<dataset>_encoder.pkl` /
`<dataset>_decoder.pkl` hold state dicts with the reference's keys and layouts (cvae.py:234-235),
so checkpoints move between the two implementations in both directions.

What differs, on purpose:
* one engine call per step (forward, both losses, backward, AdamW on the device) instead of
  module forward + autograd + torch.optim; losses are read back every `log_every` iterations only;
* the one-hot label planes of cvae.py:129 are never built (labels go in as class indices);
* eps comes from the device RNG (the reference draws it with numpy on the host, networks.py:30);
* visdom plots are skipped when visdom is not importable; image grids are written with PIL;
* multi-GPU is one process per GPU with cvae_gan_amd.ddp, not nn.DataParallel: `--gpus 0,1` makes the
  starter launch one rank per listed GPU (or run it under torchrun yourself); --batch_size stays the
  GLOBAL batch as in the reference, rank r trains on rows [r*B/N, (r+1)*B/N) of every batch with
  per-replica BatchNorm statistics, gradients are all-reduced over RCCL while backward is still
  running, and rank 0 prints, samples and writes the checkpoints (DataParallel keeps replica 0's
  BatchNorm buffers too)."""
import os
import pickle
import random
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import ddp
from .datasets import make_dataset, DevicePrefetcher
from .engine import Engine, ADAM_BETAS, ADAM_EPS


def gaussian(batch_size, n_dim, mean=0.0, var=1.0):
    """utils.py:157: numpy normal with `var` used as the standard deviation."""
    return np.random.normal(mean, var, (batch_size, n_dim)).astype(np.float32)


def check_folder(path):
    os.makedirs(path, exist_ok=True)
    return path


def save_image_grid(images, rows, cols, path):
    """images [N,H,W,3] in [-1,1] (BGR, as the network sees them) -> one PNG grid (utils.py:18-53)."""
    from PIL import Image
    n, h, w, _ = images.shape
    grid = np.zeros((rows * h, cols * w, 3), dtype=np.float32)
    for i in range(min(n, rows * cols)):
        r, c = i // cols, i % cols
        grid[r * h:(r + 1) * h, c * w:(c + 1) * w] = images[i]
    rgb = ((grid[:, :, ::-1] + 1.0) * 127.5).clip(0, 255).astype(np.uint8)
    Image.fromarray(rgb).save(path)


class CVAE(object):
    def __init__(self, args):
        self.epoch = args.epoch
        self.sample_num = 10
        self.batch_size = args.batch_size
        self.save_dir = args.save_dir
        self.result_dir = args.result_dir
        self.dataset = args.dataset
        self.device = torch.device(getattr(args, "device", "cuda"))
        self.y_dim, self.z_dim, self.pix_dim = args.y_dim, args.z_dim, args.pix_dim
        self.lr = args.lrG
        self.log_every = getattr(args, "log_every", 3)           # cvae.py:159
        self.max_iters = getattr(args, "max_iters", 0)           # 0: whole epochs
        dtype = getattr(args, "dtype", "bf16")
        self.dtype = torch.float32 if dtype in ("f32", torch.float32) else torch.bfloat16
        # data parallel: the process group is up before the trainer is built (starter.py); without one
        # this is a single replica
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        if self.batch_size % self.world:
            raise ValueError("--batch_size %d does not split over %d ranks" % (self.batch_size, self.world))
        self.local_batch = self.batch_size // self.world
        seed = getattr(args, "seed", 0)

        # loader items are uint8 images + class indices; normalisation and layout happen on the GPU
        # behind a pinned, double-buffered copy (set args.device_preprocess = False for the
        # reference's host-side float pipeline)
        self.device_preprocess = getattr(args, "device_preprocess", True)
        self.train_data, self.synthetic = make_dataset(args.data_dir, args.dataset, args.pix_dim, args.y_dim,
                                                       max_size=args.data_size, raw=self.device_preprocess)
        if self.synthetic:
            self._print("data directory %r has no %r listing: using the synthetic stand-in"
                        % (args.data_dir, args.dataset))
        self._print("Trainset: %d \n" % len(self.train_data))
        if self.world == 1:
            self.trainset_loader = DataLoader(dataset=self.train_data, batch_size=args.batch_size, drop_last=True,
                                              shuffle=True, num_workers=getattr(args, "worker", 0), pin_memory=True)
        else:
            self.trainset_loader = DataLoader(
                dataset=self.train_data, num_workers=getattr(args, "worker", 0), pin_memory=True,
                batch_sampler=ddp.ShardedBatchSampler(len(self.train_data), self.batch_size, self.rank, self.world, seed=seed))

        # --gan_type CVAEGAN (starter.py:31 declares the flag, the reference never reads it): the adversarial step of cvae_gan_amd/gan.py (discriminator + classifier next to the
        # CVAE; no counterpart in the reference's trainer); self.engine stays the CVAE either way
        self.gan = None
        if getattr(args, "gan", False) or getattr(args, "gan_type", "VAE").upper() in ("CVAEGAN", "CVAE-GAN", "GAN"):
            from .gan import CVAEGAN
            self.gan = CVAEGAN(self.pix_dim, self.y_dim, self.z_dim, self.local_batch, dtype=self.dtype,
                               device=self.device, seed=seed,
                               lam_gd=getattr(args, "lam_gd", 1e-3), lam_gc=getattr(args, "lam_gc", 1e-3),
                               lam_pd=getattr(args, "lam_pd", 1e-3), lam_pc=getattr(args, "lam_pc", 1e-3))
            self.engine = self.gan.vae
        else:
            self.engine = Engine(self.pix_dim, self.y_dim, self.z_dim, self.local_batch, dtype=self.dtype,
                                 device=self.device, seed=seed)
        self._sample_engine = None
        self.sync = None
        # After two eager steps the whole step (dropout draw, forward, backward, gradient exchange, AdamW, weight
        # repack) is captured into HIP graphs and replayed on static input buffers.  Plain CVAE: one graph on one
        # replica, four with the exchanges between them on several (ddp.GraphedStep, the schedule bench.py times).
        # Adversarial step: one graph, or two around the all-reduce of the three arenas (gan.GraphedGanStep).
        # args.graph = False or CVG_TRAINER_EAGER: eager launches throughout.
        self._use_graph = getattr(args, "graph", True) and os.environ.get("CVG_TRAINER_EAGER") is None
        self._graph, self._steps_done = None, 0
        if self.world > 1 and self.gan is not None:
            self.gan.broadcast_state()                            # the three arenas are summed after backward
        elif self.world > 1:
            ddp.broadcast_state(self.engine)                      # every replica starts from rank 0's state
            self.sync = ddp.GradSync.for_engine(self.engine)
        self._eps_gen = torch.Generator(device=self.device).manual_seed(seed * 7919 + 17 + self.rank)
        if getattr(args, "resume", False):
            try:
                self.load()
            except FileNotFoundError:
                self._print("no checkpoint to resume from in", os.path.join(self.save_dir, self.model_dir))
        if getattr(args, "testmode", False):
            self.load(args.pkl if getattr(args, "pkl", "") else None)

        # fixed noise & condition (cvae.py:77-90): sample_num latent rows, each repeated for every class
        n = self.sample_num * self.y_dim
        self.sample_z_ = torch.zeros((n, self.z_dim))
        for i in range(self.sample_num):
            row = torch.from_numpy(gaussian(1, self.z_dim, mean=random.uniform(-0.5, 0.5), var=random.uniform(0.5, 1.5)))
            self.sample_z_[i * self.y_dim:(i + 1) * self.y_dim] = row
        self.sample_y_ = torch.zeros((n, self.y_dim))
        self.sample_y_[torch.arange(n), torch.arange(n) % self.y_dim] = 1
        self.train_hist = {}

    def _print(self, *a):
        if self.rank == 0:
            print(*a, flush=True)

    # ------------------------------------------------------------------ checkpoints
    @property
    def model_dir(self):
        return "VAE_data_{}_pix_{}_batch_{}_embed_{}".format(self.dataset, self.pix_dim, self.batch_size, self.z_dim)

    def save(self):
        if self.rank != 0:                                        # rank 0's replica is the checkpoint
            return
        save_dir = check_folder(os.path.join(self.save_dir, self.model_dir))
        torch.save(self.engine.state_dict("encoder"), os.path.join(save_dir, self.dataset + "_encoder.pkl"))
        torch.save(self.engine.state_dict("decoder"), os.path.join(save_dir, self.dataset + "_decoder.pkl"))
        if self.gan is not None:
            torch.save(self.gan.D.trunk.state_dict("critic"), os.path.join(save_dir, self.dataset + "_discriminator.pkl"))
            # the classifier's file also carries the per-class feature centres that L_GC pulls the samples towards
            # (state of the loss, not of a network: without them a resumed run starts that term from nothing)
            csd = self.gan.C.trunk.state_dict("critic")
            csd["gan.centres"] = self.gan.centres.detach().cpu().clone()
            csd["gan.centres_seen"] = self.gan.centres_seen.detach().cpu().clone()
            torch.save(csd, os.path.join(save_dir, self.dataset + "_classifier.pkl"))
        with open(os.path.join(save_dir, self.dataset + "_history.pkl"), "wb") as f:
            pickle.dump(self.train_hist, f)

    def load(self, pkl=None):
        base = os.path.join(self.save_dir, self.model_dir, self.dataset) if pkl is None else pkl
        for half in ("encoder", "decoder"):
            sd = torch.load(base + "_%s.pkl" % half, map_location="cpu", weights_only=True)
            self.engine.load_state(sd)
        if self.gan is not None:
            for name, critic in (("discriminator", self.gan.D), ("classifier", self.gan.C)):
                if os.path.exists(base + "_%s.pkl" % name):
                    sd = torch.load(base + "_%s.pkl" % name, map_location="cpu", weights_only=True)
                    centres, seen = sd.pop("gan.centres", None), sd.pop("gan.centres_seen", None)
                    if centres is not None and seen is not None and tuple(centres.shape) == tuple(self.gan.centres.shape):
                        self.gan.centres.copy_(centres)
                        self.gan.centres_seen.copy_(seen)
                    critic.trunk.load_state(sd)

    # ------------------------------------------------------------------ training
    def _to_device(self, imgs, labels):
        x = imgs.to(self.device, non_blocking=True).float().contiguous()
        cls = torch.max(labels, 1)[1].to(torch.int32).to(self.device, non_blocking=True)
        return x, cls

    def batches(self):
        """(x f32 NCHW in [-1, 1], class indices int32), both on the device."""
        if self.device_preprocess:
            return DevicePrefetcher(self.trainset_loader, self.local_batch, self.pix_dim, self.device, y_dim=self.y_dim)
        return (self._to_device(imgs, labels) for imgs, labels in self.trainset_loader)

    def train_step(self, x, cls, eps):
        """One step of models/cvae.py:132-157 on this rank's rows of the batch.  --batch_size (the
        global batch) divides both losses as in the reference (:137, :139); with N ranks every local
        loss is scaled by 1/N, so the summed gradients are those of the global-batch mean."""
        eng = self.engine
        self._steps_done += 1
        if self._use_graph and self._steps_done > 2:
            got = self._replay_step(x, cls, eps)
            if got is not None:
                return got
        eng.random_mask()
        if self.sync is None:
            return eng.train_step(x, cls, eps, float(self.batch_size), self.lr)
        self.sync.begin_step()
        eng.grad_ready = self.sync.on_ready              # exchanges start as backward reports its stages
        try:
            dec = eng.forward_backward(x, cls, eps, float(self.batch_size), loss_scale=1.0 / self.world)
        finally:
            eng.grad_ready = None
        self.sync.finish()
        eng.adamw_step(self.lr)
        return dec

    def _replay_step(self, x, cls, eps):
        """The step from HIP graphs; None (and eager launches from then on) if the capture fails, e.g. because
        another thread of the process made a call that cannot be captured."""
        if self._graph is None:
            step = ddp.GraphedStep(self.engine, self.batch_size, self.lr, 1.0 / self.world, sync=self.sync)
            try:
                step.capture(x, cls, eps)
            except Exception as exc:                     # noqa: BLE001 (whatever the runtime raised: fall back, do not die)
                torch.cuda.synchronize()
                self._use_graph = False
                self.engine._packed_version = -1
                self._print("HIP-graph capture of the train step failed (%s: %s); continuing with eager launches"
                            % (type(exc).__name__, exc))
                return None
            self._graph = step
        return self._graph.run(x, cls, eps)

    def gan_step(self, x, cls, eps, z_p, y_p):
        """One adversarial step (cvae_gan_amd/gan.py) on this rank's rows: eager for the first two steps, then
        replayed from HIP graphs; a failed capture leaves the trainer on eager launches."""
        self._steps_done += 1
        if self._use_graph and self._steps_done > 2:
            if self._graph is None:
                from .gan import GraphedGanStep
                step = GraphedGanStep(self.gan, self.batch_size, self.lr, world=self.world)
                try:
                    self._graph = step.capture(x, cls, eps, z_p, y_p)
                except Exception as exc:                 # noqa: BLE001 (whatever the runtime raised: fall back, do not die)
                    torch.cuda.synchronize()
                    self._use_graph = False
                    for e in (self.gan.vae, self.gan.gen2, self.gan.D.trunk, self.gan.C.trunk):
                        e._packed_version = -1
                    self._print("HIP-graph capture of the adversarial step failed (%s: %s); continuing with eager launches"
                                % (type(exc).__name__, exc))
            if self._graph is not None:
                return self._graph.run(x, cls, eps, z_p, y_p)
        self.gan.vae.random_mask()
        self.gan.gen2.random_mask()
        return self.gan.train_step(x, cls, eps, z_p, y_p, float(self.batch_size), self.lr, world=self.world)

    def read_losses(self):
        """The losses of the whole (global) batch: the mean of the ranks' values."""
        kl, ll, bad = ddp.allreduce_mean(self.engine.losses[:3]).tolist()
        if bad != 0.0:          # some rank's encoder met a class index outside [0, y_dim): every rank raises
            self.engine.losses[2:3].zero_()
            raise ValueError("a class label outside [0, %d) reached the encoder" % self.y_dim)
        return {"VAE": kl + ll, "KL": kl, "LL": ll}

    def train(self):
        hist = self.train_hist = {"VAE_loss": [], "KL_loss": [], "LL_loss": [], "per_epoch_time": [], "total_time": []}
        eng = self.engine
        eng.training = True
        self._print("training start!!")
        eps = torch.empty(self.local_batch, self.z_dim, device=self.device)
        if self.gan is not None:
            hist.update(D_loss=[], C_loss=[])
            z_p = torch.empty(self.local_batch, self.z_dim, device=self.device)
            y_p = torch.zeros(self.local_batch, self.y_dim, device=self.device)
            rows = torch.arange(self.local_batch, device=self.device)
        start_time = time.time()
        iters = 0
        for epoch in range(self.epoch):
            epoch_start = time.time()
            it = -1
            for it, (x, cls) in enumerate(self.batches()):
                eps.normal_(generator=self._eps_gen)
                if self.gan is not None:
                    z_p.normal_(generator=self._eps_gen)
                    y_p.zero_()
                    y_p[rows, torch.randint(0, self.y_dim, (self.local_batch,), device=self.device, generator=self._eps_gen)] = 1
                    self.gan_step(x, cls, eps, z_p, y_p)
                else:
                    self.train_step(x, cls, eps)
                iters += 1
                if (it + 1) % self.log_every == 0:
                    l = self.read_losses()
                    if self.gan is not None:
                        gl = self.gan.read_losses()
                        hist["D_loss"].append(gl["D"])
                        hist["C_loss"].append(gl["C"])
                        self._print("            D_loss: %.6f C_loss: %.6f GD: %.6f GC: %.6f PD: %.6f PC: %.6f" %
                                    (gl["D"], gl["C"], gl["GD"], gl["GC"], gl["PD"], gl["PC"]))
                    hist["VAE_loss"].append(l["VAE"])
                    hist["KL_loss"].append(l["KL"])
                    hist["LL_loss"].append(l["LL"])
                    self._print("Epoch: [%2d] [%4d/%4d] time: %4.4f VAE_loss: %.8f KL_loss: %.8f LL_loss: %.8f" %
                                (epoch + 1, it + 1, len(self.train_data) // self.batch_size, time.time() - start_time,
                                 l["VAE"], l["KL"], l["LL"]))
                if self.max_iters and iters >= self.max_iters:
                    break
            if self.rank == 0:
                samples = self.sample_grid()
                out_dir = check_folder(os.path.join(self.result_dir, self.model_dir))
                save_image_grid(samples, self.sample_num, self.y_dim,
                                os.path.join(out_dir, self.dataset + "_train_{:02d}_{:04d}.png".format(epoch, it + 1)))
            if epoch % 10 == 0:
                self.save()
            hist["per_epoch_time"].append(time.time() - epoch_start)
            if self.max_iters and iters >= self.max_iters:
                break
        hist["total_time"].append(time.time() - start_time)
        self._print("Avg one epoch time: %.2f, total %d epochs time: %.2f" %
                    (float(np.mean(hist["per_epoch_time"])), self.epoch, hist["total_time"][0]))
        self._print("Training finish!... save training results")
        self.save()
        if self.world > 1:
            torch.distributed.barrier()                            # the files are on disk before any rank leaves

    # ------------------------------------------------------------------ inference
    def sample_grid(self):
        """Eval-mode decoder on the fixed latents, every class per latent row (cvae.py:167-178).
        Returns [sample_num * y_dim, P, P, 3] float32 in [-1, 1]."""
        n = self.sample_num * self.y_dim
        if self._sample_engine is None:
            self._sample_engine = Engine(self.pix_dim, self.y_dim, self.z_dim, n, dtype=self.dtype,
                                         device=self.device, share=self.engine)
        se = self._sample_engine
        se.training = False
        out = se.decode(self.sample_z_.to(self.device), self.sample_y_.to(self.device))
        return out.detach().float().cpu().numpy().transpose(0, 2, 3, 1)

    def reconstruct(self, imgs, labels, eps=None):
        """Eval-mode encoder + decoder (cvae.py:209-213).  imgs [B,3,P,P] in [-1, 1]; labels one-hot
        [B,y] or class indices [B]."""
        if labels.dim() == 2:
            x, cls = self._to_device(imgs, labels)
        else:
            x, cls = imgs.to(self.device).float().contiguous(), labels.to(self.device).to(torch.int32)
        eng = self.engine
        assert x.shape[0] == eng.B, "batch size is fixed by --batch_size (divided by the number of ranks)"
        if eps is None:
            eps = torch.randn(eng.B, self.z_dim, device=self.device)
        was, eng.training = eng.training, False
        try:
            out = eng.reconstruct(x, cls, eps.to(self.device))
        finally:
            eng.training = was
        return out

    def test(self):
        iter_num = 10
        ins, outs = [], []
        for it, (x, cls) in enumerate(self.batches()):
            out = self.reconstruct(x, cls)
            outs.append(out.detach().float().cpu().numpy().transpose(0, 2, 3, 1))
            ins.append(x.detach().float().cpu().numpy().transpose(0, 2, 3, 1))
            if it + 1 >= iter_num:
                break
        if not ins:
            raise RuntimeError("the dataset is smaller than one batch")
        # rows alternate inputs and reconstructions, one batch per row pair (utils.py:26-33)
        cols = min(self.local_batch, 16)
        rows = []
        for a, b in zip(ins, outs):
            rows.append(a[:cols])
            rows.append(b[:cols])
        grid = np.concatenate(rows, 0)
        out_dir = check_folder(os.path.join(self.result_dir, self.model_dir))
        path = os.path.join(out_dir, self.dataset + "_test_{:03d}.png".format(iter_num))
        if self.rank == 0:
            save_image_grid(grid, len(rows), cols, path)
        return path
