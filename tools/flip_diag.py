"""Where does an f32 step differ from the oracle?  For every activation gradient of a golden case: the
number of elements further than 1e-3 of the tensor's maximum from the oracle, and for the worst one
the oracle's forward value at that position (a LeakyReLU input within rounding of zero flips its gate)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from parity import run_parity, engine_tensors  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2_p32_ngf64"
rows, eng, ref = run_parity(name, torch.float32)
acts, agrads, pgrads = engine_tensors(eng)
for k in reversed(list(ref["agrads"].keys())):
    r, g = ref["agrads"][k].double(), agrads[k].double()
    err = (r - g).abs()
    mx = float(r.abs().max())
    nbad = int((err > 1e-3 * mx).sum())
    if nbad:
        i = int(err.argmax())
        fwd = ref["acts"][k].detach().flatten()[i]
        print("%-10s bad %6d / %8d  worst %.3e (max %.3e) ref %.4e got %.4e  forward value there %.3e (engine %.3e)" % (
            k, nbad, err.numel(), float(err.max()), mx, float(r.flatten()[i]), float(g.flatten()[i]), float(fwd),
            float(acts[k].flatten()[i])))
