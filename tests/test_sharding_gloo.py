"""N > 1 path on CPU: world_size-2 gloo processes exercise the shard/all-gather logic that
bench.py and denoise_sharded use (the GPU kernels are replaced by a per-image function, which
is all the collective cares about)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import midd_loader  # the spawned workers import this module without conftest.py

midd_loader.load()
from midd_amd.sharding import shard_bounds  # noqa: E402


def test_shard_bounds():
    assert [shard_bounds(256, 8, r) for r in (0, 1, 7)] == [(0, 32), (32, 64), (224, 256)]
    assert shard_bounds(8, 1, 0) == (0, 8)
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import midd_loader
    midd_loader.load()
    from midd_amd.sharding import denoise_sharded, gather_outputs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    full = torch.rand(8, 1, 16, 16, generator=g)

    def per_image(x):                                   # stands in for denoise(): strictly per-sample
        return (x * 0.5 + x.mean(dim=(1, 2, 3), keepdim=True)).clamp(0, 1)

    got = denoise_sharded(per_image, full)
    torch.save(got, os.path.join(out_dir, f"r{rank}.pt"))
    lo, hi = (rank * 4, rank * 4 + 4)
    again = gather_outputs(per_image(full[lo:hi]))
    assert torch.equal(again, got)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_all_gather_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    full = torch.rand(8, 1, 16, 16, generator=g)
    want = (full * 0.5 + full.mean(dim=(1, 2, 3), keepdim=True)).clamp(0, 1)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert torch.equal(got, want), r
