"""Host-only: print the execution program the planner builds (mi_debug_plan_dump) -- which kernel instantiations, tiles,
grids, ring depths and key splits a (topology, B, H, W, side-by-side) combination reaches.  No GPU needed.

    python tools/plan_dump.py --kw range --B 2 --H 104 --W 96 --side 1
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import midd_loader
midd_loader.load()
from midd_amd import native
from midd_amd.config import UNetConfig

KW = {
    "full": dict(),
    "range": dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32),
    "small": dict(model_channels=16, time_emb_dim=64),
}


def make_plan(kw, compute="f16x3", variant="ddim", batch_invariant=False):
    c = UNetConfig(**kw) if not isinstance(kw, UNetConfig) else kw
    cfg = native.UNetCfg()
    cfg.in_channels, cfg.model_channels, cfg.num_levels = c.in_channels, c.model_channels, len(c.channel_mult)
    for i, m in enumerate(c.channel_mult):
        cfg.channel_mult[i] = m
    cfg.num_res_blocks = c.num_res_blocks
    cfg.num_attention_levels = len(c.attention_resolutions)
    for i, a in enumerate(c.attention_resolutions):
        cfg.attention_levels[i] = a
    cfg.time_emb_dim, cfg.variant = c.time_emb_dim, native.MI_VARIANT[variant]
    cfg.compute_mode = native.MI_COMPUTE[compute] | (native.MI_COMPUTE_BATCH_INVARIANT if batch_invariant else 0)
    h = C.c_void_p()
    native.check(native.lib().mi_unet_plan_create(C.byref(cfg), C.byref(h)))
    return h


def dump(plan, B, H, W, side):
    lib = native.lib()
    n = lib.mi_debug_plan_dump(plan, B, H, W, int(side), None, 0)
    if n < 0:
        native.check(n)
    buf = C.create_string_buffer(n + 1)
    native.check(min(0, lib.mi_debug_plan_dump(plan, B, H, W, int(side), buf, n + 1)))
    return buf.value.decode()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kw", default="full", choices=sorted(KW))
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--W", type=int, default=256)
    ap.add_argument("--side", type=int, default=0)
    ap.add_argument("--compute", default="f16x3")
    a = ap.parse_args()
    print(dump(make_plan(KW[a.kw], a.compute), a.B, a.H, a.W, a.side), end="")
