"""Static check of the shipped HIP sources' ISA (no GPU): the inline-asm global loads the compiler does not track.

The f16x3 kernels request some operands (residual rows, the folded res_conv's block input) through
`asm volatile("global_load_dwordx4 ...")` because a load hipcc tracks is awaited with `vmcnt(0)` while LDS-DMA traffic is
pending.  The compiler considers such a destination register written at the asm statement: it may copy
or spill it BEFORE the data has landed -- round 3 met exactly that (loop-carried registers copied ahead of the wait: NaN).
`tools/asm_load_audit.py` walks the `hipcc -S` dump and fails if any compiler instruction touches such a register between
the load and the wait statement that names it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical-image-denoising-using-diffusion_amd", "csrc")


@pytest.mark.parametrize("src", ["conv_mfma_f16x3.hip", "conv1x1_f16x3.hip"])
def test_untracked_loads_are_not_touched_before_their_wait(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_audit.py"), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "asm loads" in r.stdout
