"""Counts, for one library build (MIDD_LIBRARY), how often the B = 2 sub-batch forward run on two streams at once differs from
the same forward run alone -- per module (first differing one).  Cheap: DIAG_REPS rounds of 2 x 2 forwards."""
import os
import sys

os.environ.setdefault("MIDD_PLAN_AS_SIDE", "1")
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import midd_loader
midd_loader.load()
from midd_amd import UNetDiffusion, UNetConfig, topology, native
from midd_amd.weights import make_state_dict, synthetic_xray

KW = dict(model_channels=int(os.environ.get("DIAG_MC", "32")), channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32)
B, H, W = int(os.environ.get("DIAG_B", "2")), int(os.environ.get("DIAG_H", "104")), int(os.environ.get("DIAG_W", "96"))
REPS = int(os.environ.get("DIAG_REPS", "100"))
MODS = os.environ.get("DIAG_MODS", "in_conv,downs.0,downs.2,downs.3,downs.4,mid_attn,ups.0,ups.9").split(",")


def main():
    cfg = UNetConfig(**KW)
    sd = make_state_dict(cfg, seed=77)
    m = UNetDiffusion(compute="f16x3", **KW)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    m.check_status = False
    x = torch.from_numpy(synthetic_xray(B, H, W, seed=3, kind="uniform")).cuda()
    c = torch.from_numpy(synthetic_xray(B, H, W, seed=504)).cuda()
    t = torch.tensor([25] * B)
    snap = lambda: {n: m.debug_fetch(n, B, H, W).clone() for n in MODS}
    eps0 = m(x, c, t)
    ref = snap()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    hist, bad = {}, 0
    for r in range(REPS):
        torch.cuda.synchronize()
        outs = {}
        for st in (s1, s2):
            with torch.cuda.stream(st):
                for _ in range(2):
                    outs[st] = m(x, c, t)
        torch.cuda.synchronize()
        for st in (s1, s2):
            with torch.cuda.stream(st):
                sn = snap()
            torch.cuda.synchronize()
            order = [n for n in MODS if not torch.equal(ref[n], sn[n])]
            if order or not torch.equal(outs[st], eps0):
                bad += 1
                k = order[0] if order else "eps-only"
                hist[k] = hist.get(k, 0) + 1
    print(f"{os.environ.get('MIDD_LIBRARY', 'libmidd.so')} [{native.kernel_source_hash()}] B={B} {H}x{W} mc={KW['model_channels']}: "
          f"{bad} of {2 * REPS} concurrent forwards differ; first differing module: {hist}", flush=True)


if __name__ == "__main__":
    main()
