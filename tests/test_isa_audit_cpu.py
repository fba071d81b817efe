"""Static checks of the shipped HIP sources' ISA (no GPU; hipcc cross-compiles gfx950 in the build container).

* `tools/asm_load_audit.py`: the inline-asm global loads the compiler does not track.  The f16x3 kernels request some operands
  (residual rows, the folded res_conv's block input) through `asm volatile("global_load_dwordx4 ...")` because a load hipcc
  tracks is awaited with `vmcnt(0)` while LDS-DMA traffic is pending.  The compiler considers such a
  destination register written at the asm statement: it may copy or spill it BEFORE the data has landed -- round 3 met exactly
  that.  The audit fails if any compiler instruction touches such a register between the load and the wait that names it.
* `tools/isa_hazard_audit.py` (round 4) over EVERY translation unit: no packed-fp32 op with an `op_sel:` bit (the form behind
  round 3's red split-sampler case), no `vmcnt` immediate beyond its 6-bit field.

hipcc is part of the image: a missing compiler FAILS these tests (no silent skip)."""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical-image-denoising-using-diffusion_amd", "csrc")
SOURCES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")))


@pytest.fixture(scope="module")
def isa_dumps(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc not found: the ISA audits cannot run (they must not be skipped)"
    out_dir = tmp_path_factory.mktemp("isa")

    def one(src):
        out = os.path.join(str(out_dir), src + ".s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                       check=True, capture_output=True, timeout=1200)
        return src, out

    def diagnostic_build():
        # the DMA-checking build of the 3x3 kernel (tools/dma_check.sh) must keep compiling, self-tests included
        out = os.path.join(str(out_dir), "conv_mfma_f16x3.dmacheck.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-DMIDD_DMA_CHECK", "-DMIDD_DMA_CHECK_BREAK",
                        "-DMIDD_DMA_CHECK_OLD_RES", os.path.join(CSRC, "conv_mfma_f16x3.hip"), "-o", out], check=True, capture_output=True, timeout=1200)
        return "conv_mfma_f16x3.hip:dmacheck", out

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        extra = ex.submit(diagnostic_build)
        dumps = dict(ex.map(one, SOURCES))
        dumps.update([extra.result()])
        return dumps


def test_every_translation_unit_is_audited():
    assert {"conv_mfma_f16x3.hip", "conv1x1_f16x3.hip", "attention_f16x3.hip", "pointwise.hip", "conv_mfma_f32.hip"} <= set(SOURCES)


@pytest.mark.parametrize("src", ["conv_mfma_f16x3.hip", "conv1x1_f16x3.hip"])
def test_untracked_loads_are_not_touched_before_their_wait(src, isa_dumps):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_audit.py"), isa_dumps[src]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "asm loads" in r.stdout


def test_no_packed_fp32_high_select_and_no_vmcnt_overflow(isa_dumps):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_audit.py")] + [isa_dumps[s] for s in SOURCES],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    assert f"{len(SOURCES)} file(s), 0 finding(s)" in r.stdout


def test_the_audit_catches_the_form(tmp_path):
    """The checker itself: the instruction hipcc emitted for tap 1 of in_conv1_kernel<32> in round 3 is flagged, the safe form is not."""
    p = tmp_path / "x.s"
    p.write_text("_Zfoo:\n\tv_pk_fma_f32 v[56:57], v[122:123], v[34:35], v[56:57] op_sel:[0,1,0]\n"
                 "\tv_pk_fma_f32 v[56:57], v[106:107], v[34:35], v[56:57] op_sel_hi:[1,0,1]\n\ts_waitcnt vmcnt(64)\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazard_audit.py"), str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "2 finding(s)" in r.stdout, r.stdout


def test_dma_checking_build_compiles_and_checks(isa_dumps):
    """tools/dma_check.sh's diagnostic build: it compiles, and it really contains the sentinel stores and comparisons."""
    text = open(isa_dumps["conv_mfma_f16x3.hip:dmacheck"]).read()
    assert text.count("0x7fff7fff") > 50 and text.count("0x7fc0dead") > 50
