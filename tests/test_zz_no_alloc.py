"""cvg_bn_stats_static (the entry point that allocates nothing: its block partials live in a static
device array of the library) against cvg_bn_stats with a caller-owned workspace, through the C ABI.
Same kernels, so the results must be bit-identical: batch statistics, running statistics and the batch
counter, on two calls in a row, the second with a smaller input (stale partials of the first call stay
in the array and must not be read), in both dtypes; and the size limit is an error, not a fault."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(x, M, C):
    from cvae_gan_amd import ops
    out = []
    for static in (False, True):
        stats = torch.full((2, C), float("nan"), device="cuda")
        rm = torch.linspace(-1, 1, C, device="cuda")
        rv = torch.linspace(0.5, 2, C, device="cuda")
        nbt = torch.full((), 7, dtype=torch.long, device="cuda")
        if static:
            ops.bn_stats_static(x, M, C, stats, rm, rv, nbt)
        else:
            from cvae_gan_amd._lib import WS_HEADER
            ws = torch.full((2 * ops.bn_stats_blocks(M, C) * C,), float("nan"), device="cuda")
            ws[:WS_HEADER] = 0          # the arrival counters start at zero; nothing else may be read before it is written
            ops.bn_stats(x, M, C, stats, rm, rv, nbt, ws)
        torch.cuda.synchronize()
        out.append((stats, rm, rv, nbt))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_static_scratch_equals_workspace_form(dtype):
    g = torch.Generator(device="cuda").manual_seed(5)
    for M, C in ((70000, 1024), (333, 64)):          # 512 blocks x 1024 channels fill the array; then a small call
        x = (torch.randn(M, C, device="cuda", generator=g) * 3 + 1).to(dtype)
        (s0, m0, v0, n0), (s1, m1, v1, n1) = _both(x, M, C)
        assert torch.isfinite(s1).all()
        assert torch.equal(s0, s1) and torch.equal(m0, m1) and torch.equal(v0, v1)
        assert int(n0) == int(n1) == 8
        # and against torch: mean / biased variance of the rows
        xf = x.float()
        assert torch.allclose(s1[0], xf.mean(0), atol=2e-5, rtol=1e-5)
        assert torch.allclose(s1[1], 1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5), atol=1e-5, rtol=2e-5)


def test_static_scratch_rejects_wide_rows():
    from cvae_gan_amd import ops
    x = torch.zeros(16, 1032, device="cuda")
    stats = torch.zeros(2, 1032, device="cuda")
    with pytest.raises(ops.CvgError, match="static scratch"):
        ops.bn_stats_static(x, 16, 1032, stats, None, None, None)
    ws = torch.zeros(2 * ops.bn_stats_blocks(16, 1032) * 1032, device="cuda")
    ops.bn_stats(x, 16, 1032, stats, None, None, None, ws)       # the workspace form takes it
    torch.cuda.synchronize()


_CAPTURE_FIRST = r"""
import sys, torch
sys.path.insert(0, %r)
from cvae_gan_amd import ops
M, C = 4096, 64
g = torch.Generator(device="cuda").manual_seed(3)
x = (torch.randn(M, C, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)
stats = torch.zeros(2, C, device="cuda")
rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
nbt = torch.zeros((), dtype=torch.long, device="cuda")
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):                      # the library's FIRST static-scratch call of this process
    ops.bn_stats_static(x, M, C, stats, rm, rv, nbt)
graph.replay()
graph.replay()
torch.cuda.synchronize()
ws = torch.zeros(2 * ops.bn_stats_blocks(M, C) * C, device="cuda")
s2 = torch.zeros(2, C, device="cuda")
ops.bn_stats(x, M, C, s2, None, None, None, ws)
torch.cuda.synchronize()
assert torch.equal(stats, s2), (stats - s2).abs().max()
assert int(nbt) == 2, int(nbt)
print("capture-first ok")
"""


def test_first_call_inside_graph_capture():
    """The header names HIP-graph capture as a use: here the process's first cvg_bn_stats_static call (the one that
    looks the scratch array's address up) happens INSIDE a capture, in a fresh process; the replays must equal the
    workspace form bit for bit and advance the batch counter once each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _CAPTURE_FIRST % root], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=170)
    assert p.returncode == 0 and "capture-first ok" in p.stdout, p.stdout[-3000:]
