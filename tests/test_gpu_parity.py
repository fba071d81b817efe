"""HIP path vs the CPU oracle and the golden fixtures — the parity tests proper.

All calls go through the C ABI (ctypes -> libmidd.so).  Tolerance: north_star states
|delta| < 1e-3 per pixel in fp32 for the sampler output; per-layer and single-forward checks
use tighter bounds (printed so regressions are visible long before the 1e-3 gate).
"""
import os

import numpy as np
import pytest
import torch

from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig, topology, timestep_list
from midd_amd import native
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
SMALL = dict(model_channels=16, time_emb_dim=64)
TOL_FINAL = 1e-3          # north_star: |delta| < 1e-3 fp32 per pixel
TOL_LAYER = 2e-4          # per-module activations (values are O(1))
TOL_EPS = 2e-4            # single forward


COMPUTE_MODES = ["f16x3", "f32"]


def _model(cfg_kw, sd_np, variant="ddim", compute="f16x3"):
    m = UNetDiffusion(variant=variant, compute=compute, **cfg_kw)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return m.to("cuda").eval()


def _maxdiff(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.fixture(scope="module", params=COMPUTE_MODES)
def full_model(request):
    cfg = UNetConfig()
    sd = make_state_dict(cfg, seed=42)
    return cfg, sd, _model({}, sd, compute=request.param)


# ------------------------------------------------------------------------------ small network
@pytest.mark.parametrize("compute", COMPUTE_MODES)
@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_small_per_layer_vs_golden_and_oracle(variant, compute):
    g = np.load(os.path.join(G, f"small_{variant}.npz"))
    cfg = UNetConfig(variant=variant, **SMALL)
    sd = make_state_dict(cfg, seed=42, perturb_norm=True)
    model = _model(SMALL, sd, variant, compute)
    x, cond = torch.from_numpy(g["fwd_x"]).cuda(), torch.from_numpy(g["fwd_cond"]).cuda()
    B, _, H, W = x.shape
    t = torch.full((B,), int(g["fwd_t"]), dtype=torch.long, device="cuda")
    eps = model(x, cond, t)
    torch.cuda.synchronize()
    worst = {}
    for key in [k for k in g.files if k.startswith("layer/")]:
        name = key[6:]
        if name in ("time_mlp", "out_conv"):
            continue
        try:
            got = model.debug_fetch(name, B, H, W).cpu().numpy()
        except native.MiddError:
            # a ConvTranspose folded into its consumer has no materialised output
            assert any(m.name == name and m.kind == "up" for m in topology(cfg).ups), name
            continue
        assert got.shape == g[key].shape, name
        worst[name] = _maxdiff(got, g[key])
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < TOL_LAYER, max(worst, key=worst.get)
    assert _maxdiff(eps.cpu().numpy(), g["fwd_eps"]) < TOL_EPS
    # and against the oracle evaluated here
    with torch.no_grad():
        want = orc.unet_forward(orc.to_torch(sd), topology(cfg), x.cpu(), cond.cpu(), t.cpu())
    assert _maxdiff(eps.cpu().numpy(), want.numpy()) < TOL_EPS


@pytest.mark.parametrize("compute", COMPUTE_MODES)
@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_small_sampler_vs_golden(variant, compute):
    g = np.load(os.path.join(G, f"small_{variant}.npz"))
    cfg = UNetConfig(variant=variant, **SMALL)
    sd = make_state_dict(cfg, seed=42, perturb_norm=True)
    model = _model(SMALL, sd, variant, compute)
    den = DiffusionDenoiser(model, noise_steps=50)
    noisy = torch.from_numpy(g["den_noisy"]).cuda()
    keep = noisy.clone()
    kw = {}
    if variant == "cddpm":
        kw["step_noise"] = torch.from_numpy(g["den_noise_scaled"]).cuda()
    out = den.denoise(noisy, inference_steps=int(g["den_inference_steps"]), **kw)
    torch.cuda.synchronize()
    assert torch.equal(noisy, keep), "denoise must not mutate its input (DDIMModel.py:271)"
    assert out.data_ptr() != noisy.data_ptr() and out.device == noisy.device
    d = _maxdiff(out.cpu().numpy(), g["den_out"])
    print(f"small {variant} sampler max|d| = {d:.2e}")
    assert d < TOL_FINAL
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0


@pytest.mark.parametrize("compute", COMPUTE_MODES)
def test_per_sample_timesteps_and_batch_independence(compute):
    cfg = UNetConfig(**SMALL)
    sd = make_state_dict(cfg, seed=3, perturb_norm=True)
    model = _model(SMALL, sd, compute=compute)
    x = torch.from_numpy(synthetic_xray(3, 24, 40, seed=50, kind="uniform")).cuda()
    c = torch.from_numpy(synthetic_xray(3, 24, 40, seed=60)).cuda()
    t = torch.tensor([0, 17, 49], device="cuda")
    eps = model(x, c, t)
    with torch.no_grad():
        want = orc.unet_forward(orc.to_torch(sd), topology(cfg), x.cpu(), c.cpu(), t.cpu())
    assert _maxdiff(eps.cpu().numpy(), want.numpy()) < TOL_EPS
    # each sample alone gives the same bits (no cross-sample reductions, fixed summation order)
    for i in range(3):
        one = model(x[i:i + 1], c[i:i + 1], t[i:i + 1])
        assert torch.equal(one[0], eps[i]), i


@pytest.mark.parametrize("case", [(50, 8, 5, 24, 40), (20, 7, 6, 32, 16), (100, 100, 1, 16, 16), (10, 25, 2, 8, 8),
                                  (50, 1, 3, 40, 24), (1000, 3, 4, 16, 32)])
def test_sampler_schedules_and_batch_shapes_vs_oracle(case):
    """`denoise` over noise_steps / inference_steps / batch / size combinations the serving paths do not use: more
    inference steps than noise steps (stride clamps to 1), one step, 1000-row time table, odd batches (3 and 5 do
    not split over two streams, 6 splits 3 + 3), non-square images (DDIMModel.py:268-289)."""
    noise_steps, inference_steps, B, H, W = case
    cfg = UNetConfig(**SMALL)
    sd = make_state_dict(cfg, seed=11, perturb_norm=True)
    model = _model(SMALL, sd)
    noisy = torch.from_numpy(synthetic_xray(B, H, W, seed=noise_steps + B, kind="uniform"))
    ref = orc.denoise(orc.to_torch(sd), topology(cfg), noisy, noise_steps=noise_steps, inference_steps=inference_steps)
    out = DiffusionDenoiser(model, noise_steps=noise_steps).denoise(noisy.cuda(), inference_steps=inference_steps)
    n_it = len(timestep_list(noise_steps, inference_steps))
    d = _maxdiff(out.cpu().numpy(), ref.numpy())
    print(f"schedule {case}: {n_it} iterations, max|d| = {d:.2e}")
    assert d < TOL_FINAL


# ------------------------------------------------------------------------------ full network
def test_full_forward_64_vs_golden(full_model):
    cfg, sd, model = full_model
    g = np.load(os.path.join(G, "full_ddim_64.npz"))
    noisy = torch.from_numpy(synthetic_xray(1, 64, 64, seed=1234)).cuda()
    eps = model(noisy, noisy, torch.tensor([49]))
    d = _maxdiff(eps.cpu().numpy(), g["fwd_eps_t49"])
    print(f"full 64 forward max|d| = {d:.2e}")
    assert d < TOL_EPS
    # per-layer checksums recorded from the reference
    for key in [k for k in g.files if k.startswith("cksum/")]:
        name = key[6:]
        if name in ("time_mlp", "out_conv"):
            continue
        try:
            got = model.debug_fetch(name, 1, 64, 64).cpu().numpy().astype(np.float64)
        except native.MiddError:
            continue
        ck = g[key]
        flat = got.reshape(-1)
        idx = np.linspace(0, flat.size - 1, 16).astype(np.int64)
        mine = np.concatenate([[got.mean(), got.std(), np.abs(got).max()], flat[idx]])
        np.testing.assert_allclose(mine, ck, rtol=0, atol=TOL_LAYER, err_msg=name)


def test_full_sampler_64_vs_golden(full_model):
    cfg, sd, model = full_model
    g = np.load(os.path.join(G, "full_ddim_64.npz"))
    noisy = torch.from_numpy(synthetic_xray(1, 64, 64, seed=1234)).cuda()
    den = DiffusionDenoiser(model, noise_steps=50)
    for steps, key in ((50, "den_out"),):
        out = den.denoise(noisy, inference_steps=steps)
        d = _maxdiff(out.cpu().numpy(), g[key])
        print(f"full 64 x 50 steps max|d| = {d:.2e}")
        assert d < TOL_FINAL
    # the served setting: inference_steps=8 -> 9 iterations (run.py:107)
    g8 = np.load(os.path.join(G, "full_ddim_64_s8.npz"))
    noisy2 = torch.from_numpy(synthetic_xray(2, 64, 64, seed=1234)).cuda()
    out = den.ddim_sample(noisy2, inference_steps=8)
    assert _maxdiff(out.cpu().numpy(), g8["den_out"]) < TOL_FINAL
    # noise_steps=100 (BASELINE.json config 3 needs DiffusionDenoiser(model, noise_steps=100))
    g100 = np.load(os.path.join(G, "full_ddim_64_n100.npz"))
    den100 = DiffusionDenoiser(model, noise_steps=100)
    noisy3 = torch.from_numpy(synthetic_xray(1, 64, 64, seed=4321)).cuda()
    out = den100.denoise(noisy3, inference_steps=100)
    d = _maxdiff(out.cpu().numpy(), g100["den_out"])
    print(f"full 64 x 100 steps max|d| = {d:.2e}")
    assert d < TOL_FINAL


def test_full_sampler_256_vs_golden(full_model):
    """BASELINE.json's resolution: 256x256, 50 steps, against the reference's own output."""
    cfg, sd, model = full_model
    g = np.load(os.path.join(G, "full_ddim_256.npz"))
    noisy = torch.from_numpy(synthetic_xray(1, 256, 256, seed=1234)).cuda()
    eps = model(noisy, noisy, torch.tensor([49]))
    d0 = _maxdiff(eps.cpu().numpy(), g["den_eps_first"])
    out = DiffusionDenoiser(model, noise_steps=50).denoise(noisy, inference_steps=50)
    d = _maxdiff(out.cpu().numpy(), g["den_out"])
    print(f"full 256: first eps max|d| = {d0:.2e}, after 50 steps max|d| = {d:.2e}")
    assert d0 < TOL_EPS and d < TOL_FINAL


def test_full_size_properties(full_model):
    """Size-independent properties at BASELINE's batch size: determinism, per-image independence
    (what makes sharding over GPUs safe), range."""
    cfg, sd, model = full_model
    den = DiffusionDenoiser(model, noise_steps=50)
    noisy = torch.from_numpy(synthetic_xray(8, 256, 256, seed=77)).cuda()
    a = den.denoise(noisy, inference_steps=5)
    b = den.denoise(noisy, inference_steps=5)
    assert torch.equal(a, b), "two runs must be bit-identical"
    # an image's result does not depend on WHICH other images share its batch: permuting the
    # batch permutes the output bit for bit (fixed per-image summation order, no atomics)
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device="cuda")
    c = den.denoise(noisy[perm], inference_steps=5)
    assert torch.equal(c, a[perm])
    # equal-size shards (what each rank runs under weak scaling) reproduce each other ...
    lo = den.denoise(noisy[:4], inference_steps=5)
    hi = den.denoise(noisy[4:], inference_steps=5)
    assert torch.equal(den.denoise(noisy[4:], inference_steps=5), hi)
    # ... and agree with the full batch to rounding: the conv tile (and with it the grouping of the
    # fused GroupNorm partial sums) is chosen per batch size, so this is 1e-5, not bit-exact
    assert float((torch.cat([lo, hi]) - a).abs().max()) < 1e-5
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and torch.isfinite(a).all()


@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_split_run_matches_oracle(variant):
    """B >= 4 runs as two half-batches on two streams (mi_denoise): same results as the oracle,
    including the per-iteration noise slices of the stochastic variant."""
    cfg = UNetConfig(variant=variant, **SMALL)
    sd = make_state_dict(cfg, seed=9, perturb_norm=True)
    model = _model(SMALL, sd, variant)
    B, H, W, S = 6, 40, 32, 5
    noisy = torch.from_numpy(synthetic_xray(B, H, W, seed=400))
    steps = timestep_list(50, S)
    kw, okw = {}, {}
    if variant == "cddpm":
        g = np.random.Generator(np.random.Philox(key=5))
        noise = torch.from_numpy(np.stack([0.5 * g.standard_normal((B, 1, H, W), dtype=np.float32) for _ in steps]))
        kw["step_noise"] = noise.cuda()
        okw["step_noise"] = list(noise)
    out = DiffusionDenoiser(model, noise_steps=50).denoise(noisy.cuda(), inference_steps=S, **kw)
    want = orc.denoise(orc.to_torch(sd), topology(cfg), noisy, 50, S, **okw)
    assert _maxdiff(out.cpu().numpy(), want.numpy()) < TOL_FINAL
    assert _maxdiff(out.cpu().numpy(), want.numpy()) < 5e-5


# ------------------------------------------------------------------------------ errors
CONFIG_SWEEP = [
    # (constructor kwargs, variant, B, H, W) -- only topologies whose skip stack closes in the reference itself and
    # whose output keeps the input resolution (attention at the last level only), with head_dim in {32, 64, 96, 128}
    (dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32), "ddim", 2, 24, 24),
    # 32..256 channels: concat inputs of up to 512 channels (LDS sized at launch beyond the nominal 384), head_dim 128,
    # 2x2 maps at the lowest level
    (dict(model_channels=32, channel_mult=(1, 2, 4, 8), num_res_blocks=2, attention_resolutions=(3,), time_emb_dim=64), "ddim", 1, 16, 16),
    (dict(model_channels=64, channel_mult=(1, 1), num_res_blocks=3, attention_resolutions=(1,), time_emb_dim=48), "ddim", 3, 10, 14),
    (dict(model_channels=32, channel_mult=(2, 2, 2, 2), num_res_blocks=3, attention_resolutions=(3,), time_emb_dim=40), "ddim", 2, 16, 24),
    (dict(model_channels=32, channel_mult=(1, 2, 4), num_res_blocks=1, attention_resolutions=(2,), time_emb_dim=48), "cddpm", 3, 16, 24),
    (dict(model_channels=48, channel_mult=(1, 2, 4), num_res_blocks=1, attention_resolutions=(2,), time_emb_dim=96), "cddpm", 2, 20, 12),
    (dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(1,), time_emb_dim=32), "cddpm", 1, 6, 10),
    (dict(model_channels=16, channel_mult=(1, 2, 4, 8), num_res_blocks=1, attention_resolutions=(3,), time_emb_dim=32), "cddpm", 2, 16, 16),
    # more than one image channel (the reference's constructor argument; its call sites use 1): the general in_conv kernel and
    # the out_conv instance whose output count is a run-time value
    (dict(in_channels=2, model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32), "ddim", 2, 24, 16),
    (dict(in_channels=3, model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(1,), time_emb_dim=32), "cddpm", 1, 16, 16),
]


@pytest.mark.parametrize("compute", COMPUTE_MODES)
@pytest.mark.parametrize("case", range(len(CONFIG_SWEEP)))
def test_other_topologies_forward_vs_oracle(case, compute):
    """Planner + kernels on topologies other than the shipped one (the same classes serve the cddpm and hybrid
    copies of the UNet, cddpmModels.py:176-265, hybrid3diffusionspeed.py:308-388): eps vs the oracle."""
    kw, variant, B, H, W = CONFIG_SWEEP[case]
    cfg = UNetConfig(variant=variant, **kw)
    sd = make_state_dict(cfg, seed=100 + case)
    m = _model(kw, sd, variant=variant, compute=compute)
    rng = np.random.default_rng(case)
    ic = kw.get("in_channels", 1)
    x = torch.from_numpy(rng.random((B, ic, H, W), dtype=np.float32))
    cond = torch.from_numpy(np.concatenate([synthetic_xray(B, H, W, seed=case + 31 * c) for c in range(ic)], axis=1))
    t = torch.from_numpy(rng.integers(0, 50, B)).to(torch.int64)
    ref = orc.unet_forward(orc.to_torch(sd), topology(cfg), x, cond, t).numpy()
    assert ref.shape == (B, ic, H, W)
    got = m(x.cuda(), cond.cuda(), t.cuda()).cpu().numpy()
    d = _maxdiff(got, ref)
    print(f"topology {case} {variant} {compute}: max|eps - oracle| = {d:.3e} (|eps| up to {np.abs(ref).max():.2f})")
    assert d < TOL_EPS * max(1.0, float(np.abs(ref).max()))


def test_error_behaviour():
    cfg = UNetConfig(**SMALL)
    model = _model(SMALL, make_state_dict(cfg, seed=1))
    x = torch.zeros(1, 1, 32, 32, device="cuda")
    with pytest.raises(RuntimeError):
        model(x.cpu(), x.cpu(), torch.tensor([0]))                 # no CPU fallback
    with pytest.raises(native.MiddError):
        bad = torch.zeros(1, 1, 36, 32, device="cuda")             # H not a multiple of 8
        model(bad, bad, torch.tensor([0]))
    with pytest.raises(ValueError):
        model(x, torch.zeros(1, 1, 32, 40, device="cuda"), torch.tensor([0]))
    with pytest.raises(ZeroDivisionError):
        DiffusionDenoiser(model).denoise(x, inference_steps=0)     # the reference divides by it too
    with pytest.raises(RuntimeError):
        sd = model.state_dict(); sd.pop("in_conv.bias")
        UNetDiffusion(**SMALL).load_state_dict(sd)                 # strict load, like nn.Module
