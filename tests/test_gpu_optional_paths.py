"""GPU: the execution switch a user can set (environment, read once per process, so each value runs in a child
process) must keep parity:
  MIDD_SPLIT=1 | 4   the batch as ONE program on the caller's stream / as four quarter-batches on four streams
                     (default 2: two half-batches on two streams)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENV_SETS = [{"MIDD_SPLIT": "1"}, {"MIDD_SPLIT": "4"}]


@pytest.mark.parametrize("knobs", ENV_SETS, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_switch_keeps_parity(knobs):
    env = dict(os.environ, **knobs)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
           os.path.join(ROOT, "tests", "test_gpu_parity_r2.py"), "-x", "-q",
           "-k", "small_sampler or topologies or split_run or full_sampler_256 or launched_kernels or reloading or config4"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
