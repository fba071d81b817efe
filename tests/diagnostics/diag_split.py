"""One deterministic diagnostic of the red round-3 case (GPUTEST_r03: seeded random shapes, f16x3, case 4 = B 4, 104x96,
forward right, 2-iteration two-stream sampler off by 6.4e-3): replays cases 0..4 exactly as the test does, then runs case 4's
sampler in variants that separate orchestration, kernels and reads of scratch nobody wrote:
  split / MI_NO_SPLIT  x  workspace as left behind / poisoned 0xFF (NaN) / 0x00 / 0x7F (3.4e38)  x  f16x3 / f32,
each twice (bit-identical?), with the location of the largest error.  Run ONCE on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import midd_loader
midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig, topology, timestep_list, native
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc

RANGE_KW = dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32)
UPTO = int(os.environ.get("DIAG_UPTO", "4"))


def model_for(sd, compute):
    m = UNetDiffusion(compute=compute, **RANGE_KW)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m.to("cuda").eval()


def where(a, b):
    d = (a.double().cpu() - b.double().cpu()).abs()
    i = int(d.argmax())
    idx = np.unravel_index(i, tuple(d.shape))
    per_img = [float(d[k].max()) for k in range(d.shape[0])]
    return float(d.max()), tuple(int(v) for v in idx), per_img


def main():
    cfg = UNetConfig(**RANGE_KW)
    sd = make_state_dict(cfg, seed=77)
    sdt, topo = orc.to_torch(sd), topology(cfg)
    for compute in ("f16x3", "f32"):
        rng = np.random.default_rng(20260303)
        model = model_for(sd, compute)
        model.check_status = True
        den = DiffusionDenoiser(model, noise_steps=50)
        for case in range(UPTO + 1):
            B = int(rng.integers(1, 10))
            H, W = (int(rng.integers(2, 14)) * 8 for _ in range(2))
            x = torch.from_numpy(rng.random((B, 1, H, W), dtype=np.float32))
            c = torch.from_numpy(synthetic_xray(B, H, W, seed=500 + case))
            t = torch.from_numpy(rng.integers(0, 50, B)).to(torch.int64)
            with torch.no_grad():
                want = orc.unet_forward(sdt, topo, x, c, t)
                want_den = orc.denoise(sdt, topo, c, noise_steps=50, inference_steps=2)
            got = model(x.cuda(), c.cuda(), t.cuda())
            out = den.denoise(c.cuda(), inference_steps=2)
            d, _, _ = where(got, want)
            dd, at, per = where(out, want_den)
            print(f"[{compute}] case {case}: B={B} {H}x{W}: forward {d:.2e}, sampler {dd:.2e} at {at} per-image {['%.1e' % v for v in per]}", flush=True)
        # ---- variants on the last case ----
        steps = timestep_list(50, 2)
        cc = c.cuda()
        # the oracle's state after ONE iteration, to see in which iteration an error enters
        states = []
        with torch.no_grad():
            orc.denoise(sdt, topo, c, noise_steps=50, inference_steps=2, on_step=lambda i, eps, xn: states.append(xn.clone()))
        want1 = states[0]
        ref = None
        for poison in (None, 255, 0, 127):
            for no_split in (False, True):
                model.poison_workspace = poison
                outs = []
                for rep in range(2):
                    try:
                        o = model.run_sampler(cc, steps, den.beta, den.alpha, den.alpha_hat, clamp_eps=True, no_split=no_split)
                        outs.append(o.clone())
                        msg = ""
                    except native.MiddError as e:
                        outs.append(None)
                        msg = f" MiddError: {e}"
                    if outs[-1] is not None:
                        dd, at, per = where(outs[-1], want_den)
                        nan = int(torch.isnan(outs[-1]).sum())
                        print(f"[{compute}] poison={poison} no_split={int(no_split)} run {rep}: sampler {dd:.3e} at {at} per-image {['%.1e' % v for v in per]} nan={nan}", flush=True)
                    else:
                        print(f"[{compute}] poison={poison} no_split={int(no_split)} run {rep}:{msg}", flush=True)
                if outs[0] is not None and outs[1] is not None:
                    print(f"    two runs bit-identical: {bool(torch.equal(outs[0], outs[1]))}", flush=True)
                if not no_split and outs[0] is not None:
                    if ref is None:
                        ref = outs[0]
                    else:
                        print(f"    identical to the unpoisoned split run: {bool(torch.equal(ref, outs[0]))}", flush=True)
                # one-iteration runs: does the first iteration already differ?
                o1 = None
                try:
                    o1 = model.run_sampler(cc, steps[:1], den.beta, den.alpha, den.alpha_hat, clamp_eps=True, no_split=no_split)
                except native.MiddError as e:
                    print(f"    1-iteration run: MiddError {e}")
                if o1 is not None and want1 is not None:
                    dd, at, per = where(o1, want1)
                    print(f"    after 1 iteration: {dd:.3e} at {at}", flush=True)
        # forward on a poisoned workspace
        for poison in (255, 127):
            model.poison_workspace = poison
            try:
                got = model(x.cuda(), c.cuda(), t.cuda())
                d, at, _ = where(got, want)
                print(f"[{compute}] forward poison={poison}: {d:.3e} nan={int(torch.isnan(got).sum())}", flush=True)
            except native.MiddError as e:
                print(f"[{compute}] forward poison={poison}: MiddError {e}", flush=True)
        model.poison_workspace = None


if __name__ == "__main__":
    main()
