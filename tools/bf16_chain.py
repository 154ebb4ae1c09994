#!/usr/bin/env python3
"""Where bf16 loses a gradient's direction: for every golden case, the cosine of each activation gradient and
parameter gradient of the bf16 engine against the f32 oracle, in backward order, next to the cosine that bf16
STORAGE of weights and input alone produces in the oracle itself (parity.bf16_storage_response).
    python tools/bf16_chain.py [raw|conditioned] [case ...]            (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
from helpers import CASES  # noqa: E402
from parity import run_parity, bf16_storage_response  # noqa: E402


def shrink(state):
    e, d = state
    for k in d:
        if k.endswith("weight"):
            d[k] = d[k] * 0.55
    return e, d


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "conditioned"
    names = sys.argv[2:] or list(CASES)
    fn = shrink if mode == "conditioned" else None
    for name in names:
        rows, eng, ref = run_parity(name, torch.bfloat16, state_fn=fn)
        store = bf16_storage_response(name, state_fn=fn)
        print("== %s (%s decoder weights): gradient cosines, backward order; [storage] = the f32 oracle with bf16-rounded "
              "weights and input against itself" % (name, mode))
        first = None
        for kind, k, r, m, c in rows:
            if kind not in ("dact", "grad"):
                continue
            s = store.get((kind, k), float("nan"))
            flag = ""
            if c < 0.99 and first is None and kind == "dact":
                first = k
                flag = "   <-- first activation gradient below 0.99"
            print("%-5s %-44s engine %8.5f   [storage %8.5f]%s" % (kind, k, c, s, flag))
        print("first activation gradient below 0.99: %s\n" % first)


if __name__ == "__main__":
    main()
