"""Round-2 golden fixtures, again FROM THE REFERENCE ITSELF (build container only):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_r2.py [clamp] [c3] [c5] [hybrid]

Kept apart from make_golden.py so the round-1 fixtures are never rewritten.  What each set pins:

  clamp   `torch.clamp(predicted_noise, -5, 5)` ACTIVE (DDIMModel.py:278): out_conv.2.weight / .bias scaled x15, so a
          sizeable fraction of |eps| exceeds 5 and the clamped value drives the update (:283-284).  Reduced UNet,
          every iteration's raw eps and x; full UNet at 64x64, 10 iterations (a horizon over which the reference
          reproduces itself across thread counts: the x15 loop amplifies rounding differences).
  c3      BASELINE.json configs[2]: noise_steps=100, inference_steps=100 at 256x256 (two images).
  c5      BASELINE.json configs[4]'s per-image shape: 512x512 x 50 iterations (one image; N=4096 attention).
  hybrid  the hybrid file's OWN copies of the classes (hybrid/hybrid3diffusionspeed.py:284-418): un-chunked attention
          with the scale applied after the matmul (:299), default inference_steps=10, served with 7 -> 8 steps
          (run.py:67,72), followed by nan_to_num + clamp (:619-620).  Also the key names / shapes of
          HybridDenoisingRouter.state_dict() under `diffusion_unet.` (what :597-598 loads).
Only arrays and generating parameters are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import midd_loader  # noqa: E402

midd_loader.load()
from midd_amd.config import UNetConfig, timestep_list  # noqa: E402
from midd_amd.weights import make_state_dict, synthetic_xray  # noqa: E402
from tests.golden.make_golden import SMALL, build, traced_denoise  # noqa: E402
from tests.golden.ref_import import import_reference, import_hybrid  # noqa: E402

CLAMP_GAIN = 15.0


def clamp_state_dict(cfg, seed, perturb):
    """The portable weights with the last conv amplified: same recipe on the GPU box."""
    sd = make_state_dict(cfg, seed=seed, perturb_norm=perturb)
    sd["out_conv.2.weight"] = (sd["out_conv.2.weight"] * CLAMP_GAIN).astype(np.float32)
    sd["out_conv.2.bias"] = (sd["out_conv.2.bias"] * CLAMP_GAIN).astype(np.float32)
    return sd


def build_from(refmod, cfg, sd):
    m = refmod.UNetDiffusion(in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                             channel_mult=cfg.channel_mult, num_res_blocks=cfg.num_res_blocks,
                             attention_resolutions=cfg.attention_resolutions, dropout=cfg.dropout,
                             time_emb_dim=cfg.time_emb_dim)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m.eval()


def gen_clamp(ref):
    cfg = UNetConfig(**SMALL)
    model = build_from(ref.ddim, cfg, clamp_state_dict(cfg, 42, True))
    den = ref.ddim.DiffusionDenoiser(model, noise_steps=50)
    B, H, W, S = 2, 32, 48, 8
    noisy = torch.from_numpy(synthetic_xray(B, H, W, seed=300))
    xf, eps_log, x_log = traced_denoise(den, noisy, S)
    eps = np.stack(eps_log)
    frac = float((np.abs(eps) > 5).mean())
    print(f"clamp small: {100 * frac:.1f} % of raw eps beyond +-5, |eps| max {np.abs(eps).max():.1f}")
    assert frac > 0.05
    np.savez_compressed(os.path.join(HERE, "small_ddim_clamp.npz"), den_out=xf, den_eps=eps, den_x=np.stack(x_log),
                        den_steps=np.array(timestep_list(50, S), np.int64), den_inference_steps=np.int64(S),
                        gain=np.float32(CLAMP_GAIN), frac_clamped=np.float64(frac))
    # Full network: 10 iterations.  With the x15 head the loop amplifies rounding differences (the reference's own
    # 50-iteration output moves by 9e-4 between 4 and 8 CPU threads), so the recorded horizon is kept short enough for
    # the reference to reproduce itself: checked here.
    cfg = UNetConfig()
    model = build_from(ref.ddim, cfg, clamp_state_dict(cfg, 42, False))
    den = ref.ddim.DiffusionDenoiser(model, noise_steps=50)
    noisy = torch.from_numpy(synthetic_xray(1, 64, 64, seed=1234))
    S = 10
    xf, eps_log, x_log = traced_denoise(den, noisy, S)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(max(1, nthreads // 2))
    xf2, _, _ = traced_denoise(den, noisy, S)
    torch.set_num_threads(nthreads)
    self_diff = float(np.abs(xf - xf2).max())
    eps = np.stack(eps_log)
    frac = float((np.abs(eps) > 5).mean())
    print(f"clamp full 64: {100 * frac:.1f} % of raw eps beyond +-5; reference vs itself at half the threads: {self_diff:.2e}")
    assert frac > 0.05 and self_diff < 1e-4
    np.savez_compressed(os.path.join(HERE, "full_ddim_64_clamp.npz"), den_out=xf, den_eps=eps, den_x=np.stack(x_log),
                        den_steps=np.array(timestep_list(50, S), np.int64), den_inference_steps=np.int64(S),
                        gain=np.float32(CLAMP_GAIN), frac_clamped=np.float64(frac), reference_self_diff=np.float64(self_diff))


def gen_c3(ref):
    cfg = UNetConfig()
    model = build(ref.ddim, cfg, seed=42, perturb=False)
    den = ref.ddim.DiffusionDenoiser(model, noise_steps=100)
    noisy = torch.from_numpy(synthetic_xray(2, 256, 256, seed=5100))
    xf, eps_log, x_log = traced_denoise(den, noisy, 100)
    assert len(eps_log) == 100
    np.savez_compressed(os.path.join(HERE, "full_ddim_256_n100.npz"), den_out=xf, den_eps_first=eps_log[0], den_eps_last=eps_log[-1],
                        den_x_after_1=x_log[0], den_x_after_50=x_log[49], seed_image=np.int64(5100))
    print("c3 done", float(xf.mean()))


def gen_c5(ref):
    cfg = UNetConfig()
    model = build(ref.ddim, cfg, seed=42, perturb=False)
    den = ref.ddim.DiffusionDenoiser(model, noise_steps=50)
    noisy = torch.from_numpy(synthetic_xray(1, 512, 512, seed=6200))
    xf, eps_log, x_log = traced_denoise(den, noisy, 50)
    np.savez_compressed(os.path.join(HERE, "full_ddim_512.npz"), den_out=xf, den_eps_first=eps_log[0], den_eps_last=eps_log[-1],
                        den_x_after_1=x_log[0], den_x_after_25=x_log[24], seed_image=np.int64(6200))
    print("c5 done", float(xf.mean()))


def gen_hybrid(ref):
    hyb = import_hybrid()
    cfg = UNetConfig()
    sd = make_state_dict(cfg, seed=42)
    unet = hyb.UNetDiffusion()
    unet.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)      # same 308 keys
    unet.eval()
    den = hyb.DiffusionDenoiser(unet, noise_steps=50)
    noisy = torch.from_numpy(synthetic_xray(2, 64, 64, seed=7300))
    arrays = {}
    for steps in (7, 8, 10):
        with torch.no_grad():
            hq = den.denoise(noisy, inference_steps=steps)
            hq = torch.clamp(torch.nan_to_num(hq, nan=0.0, posinf=1.0, neginf=0.0), 0, 1)     # :619-620
        arrays[f"hq_{steps}"] = hq.numpy().copy()
        arrays[f"iters_{steps}"] = np.int64(len(timestep_list(50, steps)))
    # the whole router with small side networks: state-dict layout under `diffusion_unet.`
    router = hyb.HybridDenoisingRouter(dict(width=8, middle_blk_num=1, enc_blk_nums=[1, 1], dec_blk_nums=[1, 1]), {},
                                       inference_diffusion_steps=7)
    keys = [k for k in router.state_dict() if k.startswith("diffusion_unet.")]
    arrays["router_unet_keys"] = np.array(keys)
    arrays["router_unet_shapes"] = np.array([",".join(map(str, router.state_dict()[k].shape)) for k in keys])
    arrays["router_other_prefixes"] = np.array(sorted({k.split(".")[0] for k in router.state_dict()}))
    np.savez_compressed(os.path.join(HERE, "hybrid_ddim_64.npz"), seed_image=np.int64(7300), **arrays)
    print("hybrid done", [float(arrays[f"hq_{s}"].mean()) for s in (7, 8, 10)])


def main():
    want = set(sys.argv[1:]) or {"clamp", "c3", "c5", "hybrid"}
    ref = import_reference()
    torch.manual_seed(0)
    if "clamp" in want:
        gen_clamp(ref)
    if "hybrid" in want:
        gen_hybrid(ref)
    if "c3" in want:
        gen_c3(ref)
    if "c5" in want:
        gen_c5(ref)


if __name__ == "__main__":
    main()
