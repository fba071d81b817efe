"""GPU: the opt-in kernel paths (environment knobs read once per process, so each set runs in a child process) must
stay parity-clean: they are measured alternatives, not dead code.
  MIDD_PREDMA_MAX_HW  conv3x3_pre_f16x3.hip (pre-activated input, DMA-only staging)
  MIDD_TILE_BIG / MIDD_TILE_NT6   16x16-pixel and 96-cout tiles
  MIDD_CONV1X1_DIRECT=0           1x1 convs through the general kernel
  MIDD_SPLIT=1 / MIDD_GRAPH=1     unsplit batch, hipGraph replay of the loop
The child runs shapes that SELECT those kernels (full network, B >= 4) and test_optional_kernels_are_reached asserts
from the library's own per-kernel profile that they were launched."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENV_SETS = [
    {"MIDD_PREDMA_MAX_HW": "1000000"},
    {"MIDD_TILE_BIG": "1", "MIDD_SPLIT": "1"},
    {"MIDD_CONV1X1_DIRECT": "0", "MIDD_TILE_NT6": "1", "MIDD_SPLIT": "1"},
    {"MIDD_GRAPH": "1"},
]


@pytest.mark.parametrize("knobs", ENV_SETS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_optional_path_keeps_parity(knobs):
    env = dict(os.environ, **knobs)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
           os.path.join(ROOT, "tests", "test_gpu_parity_r2.py"), "-x", "-q",
           "-k", "small_sampler or topologies or split_run or full_sampler_256 or optional_kernels or reloading or config4"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
