"""Randomised parity soak (GPU box, one-off; not part of the test suite): random reference-valid topologies x batch x image size x
variant x arithmetic, forward (per-sample t) and a 2-iteration sampler against the CPU oracle, every call twice (identical bits).
Explores planner / tile-picker / key-split paths the fixed tests do not name.  python tests/diagnostics/soak_parity.py [cases] [seed]"""
import os
import sys
import time

os.environ.setdefault("MIDD_POISON_WS", "255")
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import midd_loader
midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig, topology, native
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
TOL_EPS, TOL_FINAL = 2e-4, 1e-3


def random_config(rng):
    while True:
        mc = int(rng.choice([16, 32, 48, 64]))
        mult = tuple(int(v) for v in rng.choice([(1, 2), (2, 4), (1, 2, 4), (1, 1, 2), (1, 2, 3, 4), (1, 4), (2, 2), (1, 2, 2)], 1)[0]) if False else \
            [(1, 2), (2, 4), (1, 2, 4), (1, 1, 2), (1, 2, 3, 4), (1, 4), (2, 2), (1, 2, 2)][int(rng.integers(0, 8))]
        levels = len(mult)
        att = int(rng.integers(0, levels))
        c_att = mc * mult[att]
        c_mid = mc * mult[-1]
        if c_att // 2 not in (32, 64, 96, 128) or c_mid // 2 not in (32, 64, 96, 128):
            continue
        kw = dict(model_channels=mc, channel_mult=mult, num_res_blocks=int(rng.integers(1, 4)), attention_resolutions=(att,),
                  time_emb_dim=int(rng.choice([32, 64, 192])), variant=str(rng.choice(["ddim", "cddpm"])))
        return kw


def main():
    rng = np.random.default_rng(SEED)
    done = bad = skipped = 0
    worst_f = worst_s = 0.0
    t0 = time.time()
    while done < CASES:
        kw = random_config(rng)
        variant = kw.pop("variant")
        cfg = UNetConfig(variant=variant, **kw)
        div = 1 << (len(kw["channel_mult"]) - 1)
        big = rng.random() < 0.2
        B = int(rng.integers(33, 160)) if big else int(rng.integers(1, 13))
        H, W = (int(rng.integers(1, 5 if big else 14)) * div * (1 if big else 1) for _ in range(2))
        H, W = max(H, div), max(W, div)
        if B * H * W * kw["model_channels"] * max(kw["channel_mult"]) > 6e7:
            continue
        compute = str(rng.choice(["f16x3", "f16x3", "f32"]))
        try:
            sd = make_state_dict(cfg, seed=int(rng.integers(0, 1 << 30)))
            m = UNetDiffusion(variant=variant, compute=compute, **kw)
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
            m = m.to("cuda").eval()
            x = torch.from_numpy(rng.random((B, 1, H, W), dtype=np.float32))
            c = torch.from_numpy(synthetic_xray(B, H, W, seed=int(rng.integers(0, 1 << 20))))
            t = torch.from_numpy(rng.integers(0, 50, B)).to(torch.int64)
            got = m(x.cuda(), c.cuda(), t)
        except native.MiddError as e:
            if "expected" in str(e) and "input channels" in str(e) or "skip stack" in str(e):
                skipped += 1              # not a reference-valid topology (the reference's own forward would fail the same way)
                continue
            raise
        sdt, topo = orc.to_torch(sd), topology(cfg)
        with torch.no_grad():
            want = orc.unet_forward(sdt, topo, x, c, t)
        d = float((got.cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))
        rep = torch.equal(m(x.cuda(), c.cuda(), t), got)
        den = DiffusionDenoiser(m, noise_steps=50)
        noise = None
        if variant == "cddpm":
            noise = 0.5 * torch.randn((2, B, 1, H, W), generator=torch.Generator().manual_seed(int(rng.integers(0, 1 << 30))))
        out = den.denoise(c.cuda(), inference_steps=2, step_noise=None if noise is None else noise.cuda())
        out2 = den.denoise(c.cuda(), inference_steps=2, step_noise=None if noise is None else noise.cuda())
        with torch.no_grad():
            want_den = orc.denoise(sdt, topo, c, noise_steps=50, inference_steps=2, step_noise=None if noise is None else [noise[0], noise[1]])
        dd = float((out.cpu() - want_den).abs().max())
        # the same forward beside another stream's kernels: identical bits (round 4: the co-residency property)
        xs, cs = x.cuda(), c.cuda()
        m.check_status = False
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for st in (s1, s2):
            with torch.cuda.stream(st):
                for _ in range(3):
                    o = m(xs, cs, t)
                outs.append(o)
        torch.cuda.synchronize()
        m.check_status = True
        rep = rep and all(torch.equal(o, got) for o in outs)
        ok = d < TOL_EPS and dd < TOL_FINAL and rep and torch.equal(out, out2)
        worst_f, worst_s = max(worst_f, d), max(worst_s, dd)
        done += 1
        if not ok:
            bad += 1
        print(f"{'ok ' if ok else 'BAD'} case {done}: {variant} {compute} mc={kw['model_channels']} mult={kw['channel_mult']} nres={kw['num_res_blocks']} att={kw['attention_resolutions']} "
              f"B={B} {H}x{W}: forward {d:.2e} sampler {dd:.2e} repeatable {rep and bool(torch.equal(out, out2))}", flush=True)
        del m
    print(f"{done} cases ({skipped} invalid topologies skipped), {bad} bad; worst forward {worst_f:.2e}, worst sampler {worst_s:.2e}; {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
