"""Adversarial extension of the train step: CVAE-GAN (arXiv 1703.10155, the paper the reference's
README.md:3 names).  The reference repository itself contains no discriminator or classifier, so
nothing here has reference code to be checked against; the checker is oracle/gan_cpu.py, a PyTorch
autograd restatement of the same losses (parity unpinned).

Networks.  Encoder E and generator G are the CVAE's encoder and decoder (cvae_gan_amd.engine.Engine).
Discriminator D and classifier C are the same ResNetEncoder trunk (models/networks.py:164-196) without
label planes, with F output features, followed by one linear head (F -> 1, F -> y).

One step, for real images x with labels c, eps for the sampler and prior draws (z_p, c_p):
    x_f = G(E(x, c), c)            reconstruction            x_p = G(z_p, c_p)      sample
    D and C each see ONE joint batch [x | x_f | x_p] (BatchNorm statistics over all 3B rows)
    L_D  = BCE(D(x), 1) + BCE(D(x_f), 0) + BCE(D(x_p), 0)      (mean over the 3B rows)  -> D
    L_C  = CE(C(x), c)                                                                 -> C
    L_GD = 1/2 |mean f_D(x) - mean f_D(x_p)|^2          (mean feature matching; f = the trunk output; the
           real-image mean is a constant target)                                        -> G (and E)
    L_GC = sum_c 1/2 |centre_c - mean_{x_p of class c} f_C|^2   (per-class feature centres: centre_c is the
           moving average, momentum 0.1, of the mean f_C over the real images of class c; a constant
           target.  per_class=False: the class-blind form of L_GD instead)              -> G (and E)
    L_PD = mean_i 1/2 |f_D(x_i) - f_D(x_f,i)|^2,  L_PC likewise with f_C   (pairwise feature reconstruction,
           the feature terms of the paper's L_G; the real features are constant targets) -> G, E
    L_VAE = KL + LL as in models/cvae.py:137-140 (LL is the pixel term of the paper's L_G) -> E, G
E and G minimise L_VAE + lam_gd L_GD + lam_gc L_GC + lam_pd L_PD + lam_pc L_PC with D and C held constant;
the image gradients of the matching losses reach x_p and x_f (and, through the joint BatchNorm statistics,
each other).  Data parallel: every loss gradient takes loss_scale = 1 / world and the three gradient
arenas are summed over the ranks, each on a communication stream as soon as it is final (the discriminator's
after its own backward pass, the classifier's after its own, the CVAE's at the end: ddp.ArenaExchange);
BatchNorm statistics and class centres stay per replica.
"""
from collections import OrderedDict

import torch

from . import ops
from .engine import Engine


class Critic:
    """ResNetEncoder trunk + linear head over a joint batch of M rows."""

    def __init__(self, pix_dim, n_out, features, rows, dtype=torch.bfloat16, ngf=64, n_stage=4, device="cuda", seed=0):
        assert features % 8 == 0, "trunk features must be a multiple of 8"
        self.F, self.n_out, self.M = features, n_out, rows
        extra = OrderedDict([("head.weight", (n_out, features)), ("head.bias", (n_out,))])
        self.trunk = Engine(pix_dim, 0, features // 2, rows, dtype=dtype, ngf=ngf, n_stage=n_stage, device=device,
                            seed=seed, parts="encoder", extra_params=extra, stem_input_grad=True)
        t = self.trunk
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():       # init_weights('kaiming') semantics for the head (networks.py:283-304)
            t.params["head.weight"].copy_(torch.randn(n_out, features, generator=g) * (2.0 / features) ** 0.5)
            t.params["head.bias"].zero_()
        self.dh_own = torch.zeros(rows, features, dtype=dtype, device=t.device)     # from the critic's own loss
        self.dh_gen = torch.zeros(rows, features, dtype=dtype, device=t.device)     # from the matching loss
        self.ws = torch.zeros(rows * max(n_out, 1) + 2 * rows + 16, dtype=torch.float32, device=t.device)
        self.losses = torch.zeros(3, dtype=torch.float32, device=t.device)          # [own, matching, pairwise]

    @property
    def params(self):
        return self.trunk.params

    @property
    def grads(self):
        return self.trunk.grads

    def forward(self, x3):
        return self.trunk.enc_forward(x3, None)

    def backward_own(self):
        self.trunk.enc_backward(self.dh_own)

    def backward_generator(self):
        """dL_match / d(images) with this critic's parameters held constant: f32 [M,3,P,P]."""
        return self.trunk.enc_backward(self.dh_gen, param_grads=False, input_grad=True)

    def adamw_step(self, lr):
        self.trunk.adamw_step(lr)


class CVAEGAN:
    def __init__(self, pix_dim, y_dim, z_dim, batch, dtype=torch.bfloat16, ngf=64, n_stage=4, features=64,
                 lam_gd=1e-3, lam_gc=1e-3, lam_pd=1e-3, lam_pc=1e-3, per_class=True, centre_momentum=0.1,
                 device="cuda", seed=0, state=None):
        self.B, self.P, self.y, self.z = batch, pix_dim, y_dim, z_dim
        self.lam_gd, self.lam_gc, self.lam_pd, self.lam_pc = lam_gd, lam_gc, lam_pd, lam_pc
        self.per_class, self.centre_momentum = per_class, centre_momentum
        kw = dict(dtype=dtype, ngf=ngf, n_stage=n_stage, device=device)
        self.vae = Engine(pix_dim, y_dim, z_dim, batch, seed=seed, state=state, **kw)
        # second pass through the generator (the prior samples): same weights, gradients of its own
        self.gen2 = Engine(pix_dim, y_dim, z_dim, batch, share=self.vae, own_grads=True, seed=seed, **kw)
        self.D = Critic(pix_dim, 1, features, 3 * batch, seed=seed + 101, **kw)
        self.C = Critic(pix_dim, y_dim, features, 3 * batch, seed=seed + 202, **kw)
        dev = self.vae.device
        self.x3 = torch.zeros(3 * batch, 3, pix_dim, pix_dim, dtype=torch.float32, device=dev)
        self.labels_dev = torch.zeros(3 * batch, dtype=torch.int32, device=dev)
        self.labels_p = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.centres = torch.zeros(y_dim, features, dtype=torch.float32, device=dev)      # class centres of f_C over real images
        self.centres_seen = torch.zeros(y_dim, dtype=torch.int32, device=dev)
        self._exchange = None

    def forward_backward(self, x, labels, eps, z_p, y_p, batch_size, loss_scale=1.0, on_final=None):
        """x f32 NCHW [B,3,P,P]; labels int32 [B]; eps, z_p f32 [B,z]; y_p one-hot f32 [B,y].  The two
        dropout masks are the ones last set on self.vae / self.gen2.  Gradients (times loss_scale) land in
        the four networks' arenas; losses: self.vae.losses[0:2] = KL, LL; self.D.losses / self.C.losses =
        [own loss, matching loss, pairwise loss].
        on_final(arena): called as soon as a network's parameter gradients are final (the discriminator's, then
        the classifier's, then the CVAE's), with everything that produces them enqueued on the current stream: a
        data-parallel driver starts that arena's exchange there, beside the rest of the step."""
        out = None
        for piece, arena in zip(self.pieces(x, labels, eps, z_p, y_p, batch_size, loss_scale),
                                (self.D.trunk.g_arena, self.C.trunk.g_arena, self.vae.g_arena)):
            out = piece()
            if on_final is not None:
                on_final(arena)
        return out

    def pieces(self, x, labels, eps, z_p, y_p, batch_size, loss_scale=1.0):
        """forward_backward() as three consecutive callables (a driver that captures each into a graph of its own
        and exchanges gradients between them): after the first the discriminator's parameter gradients are final,
        after the second the classifier's, after the third (which returns (reconstruction, sample)) the CVAE's."""
        A, G2, D, C, B = self.vae, self.gen2, self.D, self.C, self.B
        ls = float(loss_scale)
        M, F = 3 * B, D.F
        st = {}

        def discriminator():
            # ---- E, G forward
            h = A.enc_forward(x, labels)
            zin = A.sample(h, eps, batch_size)
            st["dec_f"] = dec_f = A.dec_forward(zin, A.dec["mask"])
            st["dec_p"] = dec_p = G2.dec_forward(G2.set_decoder_input(z_p, y_p), G2.dec["mask"])
            self.x3[:B].copy_(x)
            self.x3[B:2 * B].copy_(dec_f)
            self.x3[2 * B:].copy_(dec_p)
            self.labels_dev[:B].copy_(A.enc["labels"])        # the copy enc_forward made: inside [0, y)
            self.labels_p.copy_(y_p.argmax(dim=1))
            # ---- the discriminator's own loss (parameter gradients) and matching losses (their image gradient
            # comes from backward_generator() in the next piece, with the parameters held constant)
            hD = D.forward(self.x3)
            ops.gan_d_head(hD, F, D.params["head.weight"], D.params["head.bias"], M, F, B, ls, D.grads["head.weight"],
                           D.grads["head.bias"], D.losses[0:1], D.dh_own, D.ws)
            ops.gan_feature_match(hD, F, M, F, 0, B, 2 * B, M, self.lam_gd * ls, D.losses[1:2], D.dh_gen)
            ops.gan_pair_feature_match(hD, F, F, 0, B, B, self.lam_pd * ls, D.losses[2:3], D.dh_gen, D.ws)
            D.backward_own()

        def classifier():
            st["dimg_d"] = D.backward_generator()
            hC = C.forward(self.x3)
            ops.gan_c_head(hC, F, C.params["head.weight"], C.params["head.bias"], self.labels_dev, M, F, self.y, B, ls,
                           C.grads["head.weight"], C.grads["head.bias"], C.losses[0:1], C.dh_own, C.ws)
            if self.per_class:
                C.dh_gen.zero_()
                ops.gan_class_feature_match(hC, F, F, self.labels_dev, self.labels_p, 0, B, 2 * B, M, self.y, self.centres,
                                            self.centres_seen, self.centre_momentum, self.lam_gc * ls, C.losses[1:2],
                                            C.dh_gen, C.ws)
            else:
                ops.gan_feature_match(hC, F, M, F, 0, B, 2 * B, M, self.lam_gc * ls, C.losses[1:2], C.dh_gen)
            ops.gan_pair_feature_match(hC, F, F, 0, B, B, self.lam_pc * ls, C.losses[2:3], C.dh_gen, C.ws)
            C.backward_own()

        def generator():
            dimg_d, dec_f = st["dimg_d"], st["dec_f"]
            dimg_c = C.backward_generator()
            # ---- generator / encoder backward
            d = A.dec
            ops.mse_fwd_bwd(dec_f, x, d["d_dec"], A.losses[1:2], dec_f.numel(), batch_size, ls, A.red_rows)
            ops.axpy(d["d_dec"], dimg_d[B:2 * B], 1.0)
            ops.axpy(d["d_dec"], dimg_c[B:2 * B], 1.0)
            G2.dec["d_dec"].copy_(dimg_d[2 * B:])
            ops.axpy(G2.dec["d_dec"], dimg_c[2 * B:], 1.0)
            dzin = A.dec_backward(d["d_dec"], d["mask"])
            A.sample_backward(dzin, eps, None, None, ls, batch_size)
            A.enc_backward(A.enc["dh"])
            G2.dec_backward(G2.dec["d_dec"], G2.dec["mask"])
            ops.axpy(A.g_arena, G2.g_arena, 1.0)
            return dec_f, st["dec_p"]

        return discriminator, classifier, generator

    def all_reduce_grads(self):
        """Data parallel: sum the gradient arenas of the CVAE, the discriminator and the classifier over
        the ranks (each rank ran forward_backward with loss_scale = 1 / world)."""
        from . import ddp
        for arena in (self.vae.g_arena, self.D.trunk.g_arena, self.C.trunk.g_arena):
            ddp.allreduce_sum_(arena)

    def broadcast_state(self, src=0):
        from . import ddp
        for eng in (self.vae, self.D.trunk, self.C.trunk):
            ddp.broadcast_state(eng, src=src)

    def adamw_step(self, lr):
        self.vae.adamw_step(lr)
        self.D.adamw_step(lr)
        self.C.adamw_step(lr)

    def exchange(self):
        """The stream-ordered exchange of the three gradient arenas (ddp.ArenaExchange), created on first use."""
        if self._exchange is None:
            from . import ddp
            self._exchange = ddp.ArenaExchange(self.vae.device)
        return self._exchange

    def train_step(self, x, labels, eps, z_p, y_p, batch_size, lr, world=1):
        if world > 1:
            ex = self.exchange()         # each arena travels as soon as it is final, beside the rest of the step
            out = self.forward_backward(x, labels, eps, z_p, y_p, batch_size, loss_scale=1.0 / world, on_final=ex.start)
            ex.finish()
        else:
            out = self.forward_backward(x, labels, eps, z_p, y_p, batch_size)
        self.adamw_step(lr)
        return out

    def read_losses(self):
        vae = self.vae.read_losses()         # raises if a class label outside [0, y) reached the encoder
        kl, ll = vae["KL"], vae["LL"]
        from . import ddp
        # like KL / LL in the trainer's log: the mean over the ranks (each rank's value is that of its shard)
        ld, lgd, lpd, lc, lgc, lpc = ddp.allreduce_mean(torch.cat([self.D.losses, self.C.losses])).tolist()
        return dict(KL=kl, LL=ll, D=ld, C=lc, GD=lgd, GC=lgc, PD=lpd, PC=lpc)


class GraphedGanStep:
    """The adversarial step (dropout draws, the four networks' forward and backward passes, three AdamW
    updates, weight repacks) replayed from HIP graphs on static input buffers.  One replica: ONE graph.
    Several replicas: one graph per piece of the step (CVAEGAN.pieces), each followed by the all-reduce of the
    arena it completed on a communication stream while the next piece runs, then [AdamW].  Captured in thread-local mode, like ddp.GraphedStep."""

    def __init__(self, net, batch_size, lr, world=1):
        self.net, self.batch_size, self.lr, self.world = net, float(batch_size), lr, world
        self.graphs, self.inputs, self.out = None, None, None

    def _engines(self):
        n = self.net
        return (n.vae, n.gen2, n.D.trunk, n.C.trunk)

    def capture(self, x, labels, eps, z_p, y_p):
        """Tensors of the step's shapes and dtypes (their values are not used)."""
        net = self.net
        self.inputs = tuple(torch.empty_like(t) for t in (x, labels, eps, z_p, y_p))
        gx, gl, ge, gz, gy = self.inputs
        owners = (net.vae, net.D.trunk, net.C.trunk)             # the engines that own an optimiser state
        for e in self._engines():
            e.grad_ready = None
            e.pack_weights(force=True)
        torch.cuda.synchronize()
        saved = [e.step_count for e in owners]
        mode = {"capture_error_mode": "thread_local"}

        def fwd_bwd():
            net.vae.random_mask()
            net.gen2.random_mask()
            for e in self._engines():
                e._packed_version = -1                           # the captured step repacks what it updated last time
            self.out = net.forward_backward(gx, gl, ge, gz, gy, self.batch_size, loss_scale=1.0 / self.world)

        if self.world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **mode):
                fwd_bwd()
                net.adamw_step(self.lr)
            self.graphs = [g]
        else:
            # one graph per piece of the step, so that each network's arena can travel while the next piece runs
            gs = [torch.cuda.CUDAGraph() for _ in range(4)]
            mode["pool"] = torch.cuda.graph_pool_handle()
            pieces = net.pieces(gx, gl, ge, gz, gy, self.batch_size, loss_scale=1.0 / self.world)
            with torch.cuda.graph(gs[0], **mode):
                net.vae.random_mask()
                net.gen2.random_mask()
                for e in self._engines():
                    e._packed_version = -1
                pieces[0]()
            with torch.cuda.graph(gs[1], **mode):
                pieces[1]()
            with torch.cuda.graph(gs[2], **mode):
                self.out = pieces[2]()
            with torch.cuda.graph(gs[3], **mode):
                net.adamw_step(self.lr)
            self.graphs = gs
        for e, c in zip(owners, saved):
            e.step_count = c
        return self

    def run(self, x, labels, eps, z_p, y_p):
        """Copies the inputs in and enqueues the step; returns (reconstruction, sample), static buffers."""
        for dst, src in zip(self.inputs, (x, labels, eps, z_p, y_p)):
            dst.copy_(src)
        net = self.net
        if self.world == 1:
            self.graphs[0].replay()
        else:
            ex = net.exchange()
            for g, arena in zip(self.graphs[:3], (net.D.trunk.g_arena, net.C.trunk.g_arena, net.vae.g_arena)):
                g.replay()
                ex.start(arena)
            ex.finish()
            self.graphs[3].replay()
        for e in (net.vae, net.D.trunk, net.C.trunk):
            e.step_count += 1
            e.params_changed()
        return self.out

