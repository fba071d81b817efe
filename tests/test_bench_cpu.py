"""bench.py host logic that needs no GPU: started bare with --gpus N > 1 it must launch N ranks itself (as a child
process, before anything touches the GPU) and relay rank 0's JSON line (missing #2)."""
import importlib.util
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bare_multi_gpu_invocation_spawns_ranks(monkeypatch, capsys):
    bench = _bench()
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "m", "value": 1.0, "n_gpus": 4}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 4


def test_failed_child_gives_nonzero_exit(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=3, stdout=""))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 3


def test_stale_pmc_profile_is_refused():
    bench = _bench()
    traffic, why = bench.pmc_traffic("midd::conv_mfma_f16x3_kernel<3, 1, 16, 2, 3, 4, 1>")
    import glob
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
    doc = json.load(open(newest))
    if doc.get("kernel_source_hash") == bench.kernel_source_hash():
        assert traffic is None or traffic > 0
    else:
        assert traffic is None and "stale" in why
