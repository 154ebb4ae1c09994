"""Shared test plumbing: load a golden case and replay it through the oracle."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import ref_cpu as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_p16_ngf8", "ragged_p32_ngf8", "c2_p32_ngf64", "deep5_p64_ngf8"]


class Case:
    def __init__(self, name):
        self.name = name
        self.npz = np.load(os.path.join(GOLDEN, name + ".npz"))
        m = self.npz["meta"]
        (self.P, self.y, self.z, self.B, self.bs_flag, self.n_steps, self.ngf, self.seed,
         small) = [int(v) for v in m[:9]]
        self.n_stage = int(m[9]) if len(m) > 9 else O.N_STAGE      # ndown = nup of the case
        self.small = bool(small)
        self.lr = float(self.npz["lr"])

    def init_state(self, dtype=torch.float32):
        """Initial encoder / decoder state_dicts under the checkpoint key names."""
        if not self.small:
            e, d = O.init_params(self.P, self.y, self.z, seed=self.seed, ngf=self.ngf, n_stage=self.n_stage)
        else:
            e, d = OrderedDict(), OrderedDict()
            es, ds = O.param_shapes(self.P, self.y, self.z, self.n_stage, self.ngf)
            for k in es:
                e[k] = torch.from_numpy(self.npz["init/" + k])
            for k in ds:
                d[k] = torch.from_numpy(self.npz["init/" + k])
        if dtype != torch.float32:
            e = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in e.items())
            d = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in d.items())
        return e, d

    def inputs(self, step, dtype=torch.float32):
        s = "s%d/" % step
        return (torch.from_numpy(self.npz[s + "x"]).to(dtype),
                torch.from_numpy(self.npz[s + "labels"]).long(),
                torch.from_numpy(self.npz[s + "eps"]).to(dtype),
                torch.from_numpy(self.npz[s + "mask"]).to(dtype))
