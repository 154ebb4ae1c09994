"""Register budgets of the hot kernels (CPU: hipcc cross-compiles for gfx950 and reports per-kernel resources).
The bf16 convolution kernels sit at their register limits on purpose (256 per lane at two waves per SIMD); a change
that adds one live value to a loop makes the compiler spill to scratch memory, which halves such a kernel's rate
without failing any numerical test.  This test fails on the first spilled byte."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_bf16_convolution_kernels_do_not_spill(tmp_path):
    src = os.path.join(ROOT, "cvae_gan_amd", "csrc", "conv_patch.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "conv_patch.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    name, seen, spilled = None, 0, []
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            hot = ("patch_fwdIDF16b" in name or "patch_wgrad9" in name or "patch_dgrad_s2IDF16b" in name
                   or "patch_wgradIDF16b" in name or "lowk_fwd" in name)
            if hot:
                seen += 1
                if int(m.group(1)):
                    spilled.append((name, int(m.group(1))))
            name = None
    assert seen >= 20, "expected the bf16 patch kernels in the resource report, saw %d" % seen
    assert not spilled, "kernels spilling to scratch memory: %s" % spilled
