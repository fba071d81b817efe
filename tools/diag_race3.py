"""Characterises the corrupted in_conv outputs of concurrent B = 2 forwards: for every differing element, which lane of its
64-pixel pass, which channel, which of the 18 taps' terms is missing (or what else)."""
import os
import sys
from collections import Counter

os.environ.setdefault("MIDD_PLAN_AS_SIDE", "1")
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import midd_loader
midd_loader.load()
from midd_amd import UNetDiffusion, UNetConfig, native
from midd_amd.weights import make_state_dict, synthetic_xray

KW = dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32)
B, H, W = 2, 104, 96
REPS = int(os.environ.get("DIAG_REPS", "200"))


def main():
    cfg = UNetConfig(**KW)
    sd = make_state_dict(cfg, seed=77)
    m = UNetDiffusion(compute="f16x3", **KW)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    m.check_status = False
    xh = synthetic_xray(B, H, W, seed=3, kind="uniform"); ch = synthetic_xray(B, H, W, seed=504)
    x, c = torch.from_numpy(xh).cuda(), torch.from_numpy(ch).cuda()
    t = torch.tensor([25] * B)
    w = sd["in_conv.weight"].astype(np.float64)      # [32][2][3][3]
    pad = lambda a: np.pad(a[:, 0], ((0, 0), (1, 1), (1, 1))).astype(np.float64)
    xp, cp = pad(xh), pad(ch)

    def terms(b, cc, y, xx):
        v = np.concatenate([xp[b, y:y + 3, xx:xx + 3].reshape(9), cp[b, y:y + 3, xx:xx + 3].reshape(9)])
        return v * w[cc].reshape(18)

    m(x, c, t)
    ref = m.debug_fetch("in_conv", B, H, W).clone().cpu().numpy()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    lanes, chans, taps, waves, passes, other, events = Counter(), Counter(), Counter(), Counter(), Counter(), 0, 0
    for r in range(REPS):
        torch.cuda.synchronize()
        for st in (s1, s2):
            with torch.cuda.stream(st):
                for _ in range(2):
                    m(x, c, t)
        torch.cuda.synchronize()
        for st in (s1, s2):
            with torch.cuda.stream(st):
                got = m.debug_fetch("in_conv", B, H, W)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            nz = np.argwhere(got != ref)
            if len(nz) == 0:
                continue
            events += 1
            for b, cc, y, xx in nz:
                p = y * W + xx
                d = float(got[b, cc, y, xx]) - float(ref[b, cc, y, xx])
                tm = terms(b, cc, y, xx)
                hit = [i for i in range(18) if abs(-tm[i] - d) < 4e-6]
                lanes[(p % 64) // 16] += 1; chans[int(cc)] += 1; waves[(p % 256) // 64] += 1; passes[(p % 512) // 256] += 1
                if len(hit) == 1:
                    taps[hit[0]] += 1
                else:
                    other += 1
                    if other <= 6:
                        print(f"   unexplained: {(b, cc, y, xx)} ref {ref[b, cc, y, xx]:+.6f} got {got[b, cc, y, xx]:+.6f} terms {np.round(tm, 5).tolist()}")
    print(f"{os.environ.get('MIDD_LIBRARY', 'libmidd.so')}: {events} corrupted forwards of {2 * REPS}; elements by lane quarter {dict(lanes)}, "
          f"wave {dict(waves)}, pass {dict(passes)}, channel {dict(sorted(chans.items()))}, dropped tap {dict(sorted(taps.items()))}, unexplained {other}", flush=True)


def dump_weight_check():
    import ctypes as C
    lib = native.lib()
    if not hasattr(lib, "mi_debug_ic1_dump"):
        return
    n = 1 + 4096 * 8
    buf = (C.c_uint * n)()
    lib.mi_debug_ic1_dump(buf, n)
    cnt = buf[0]
    print(f"weight-check records: {cnt}")
    import struct
    f = lambda u: struct.unpack("<f", struct.pack("<I", u))[0]
    agg = Counter()
    for k in range(min(cnt, 4096)):
        d = buf[1 + k * 8: 1 + k * 8 + 8]
        blk, wl, code, got, want, base, again, clk = d
        i, h, q, e = code >> 16, (code >> 8) & 255, (code >> 4) & 15, code & 15
        agg[(i, (wl & 255) // 16, e)] += 1
        if k < 40:
            print(f"  block ({blk & 0xffff},{blk >> 16}) wave {wl >> 8} lane {wl & 255} tap {i} half {h} quad {q} elem {e}: got {f(got):+.6f} (0x{got:08x}) want {f(want):+.6f} re-read {f(again):+.6f} base {base} clk {clk}")
    print("  (tap, lane quarter, element) histogram:", dict(agg))


if __name__ == "__main__":
    main()
    dump_weight_check()
