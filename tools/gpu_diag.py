#!/usr/bin/env python3
"""Print the layer-by-layer error of the HIP engine against the CPU oracle.
Usage: python tools/gpu_diag.py [case] [f32|bf16]   (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
from parity import run_parity, format_rows  # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny_p16_ngf8"
    dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
    rows, _, _ = run_parity(name, dtype)
    print("case %s dtype %s" % (name, dtype))
    print(format_rows(rows))
