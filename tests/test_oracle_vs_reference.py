"""Pins the oracle: the CPU restatement must equal the imported reference.

Runs only in the build container (needs /root/reference).  The reference has no tests or
golden vectors of its own, so this import is the pin.
"""
import numpy as np
import pytest
import torch

from midd_amd.config import UNetConfig, topology, param_shapes, timestep_list
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc

pytestmark = pytest.mark.reference

SMALL = dict(model_channels=16, time_emb_dim=32)


def _load_ref(ref_mod, cfg, sd_np):
    kw = dict(in_channels=cfg.in_channels, model_channels=cfg.model_channels,
              channel_mult=cfg.channel_mult, num_res_blocks=cfg.num_res_blocks,
              attention_resolutions=cfg.attention_resolutions, dropout=cfg.dropout,
              time_emb_dim=cfg.time_emb_dim)
    model = ref_mod.UNetDiffusion(**kw)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return model.eval()


@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_state_dict_names_and_shapes_match_reference(reference_module, variant):
    cfg = UNetConfig(variant=variant)
    ref = getattr(reference_module, variant).UNetDiffusion()
    ref_sd = ref.state_dict()
    ours = param_shapes(cfg)
    assert [n for n, _ in ours] == list(ref_sd.keys())
    for n, s in ours:
        assert tuple(ref_sd[n].shape) == s, n
    nparam = sum(int(np.prod(s)) for _, s in ours)
    assert nparam == (12_823_489 if variant == "ddim" else 12_526_273)


def test_schedule_matches_reference(reference_module):
    for steps in (50, 100):
        d = reference_module.ddim.DiffusionDenoiser(None, noise_steps=steps)
        b, a, ah = orc.schedule(steps)
        assert torch.equal(b, d.beta) and torch.equal(a, d.alpha) and torch.equal(ah, d.alpha_hat)


def test_iteration_counts():
    # SURVEY.md section 0 fact 5 / App. B
    assert [len(timestep_list(50, s)) for s in (5, 8, 15, 25, 50, 100)] == [5, 9, 17, 25, 50, 50]
    assert timestep_list(50, 8)[0] == 48 and timestep_list(50, 8)[-1] == 0
    assert orc.timestep_list(50, 8) == timestep_list(50, 8)


@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_forward_blockwise_equal(reference_module, variant):
    cfg = UNetConfig(variant=variant, **SMALL)
    topo = topology(cfg)
    sd_np = make_state_dict(cfg, seed=7, perturb_norm=True)
    model = _load_ref(getattr(reference_module, variant), cfg, sd_np)
    sd = orc.to_torch(sd_np)
    x = torch.from_numpy(synthetic_xray(2, 32, 40, seed=3, kind="uniform"))
    cond = torch.from_numpy(synthetic_xray(2, 32, 40, seed=5))
    t = torch.tensor([13, 13])

    captured = {}
    hooks = []
    for name, mod in model.named_modules():
        top = (name.startswith(("downs.", "ups.")) and name.count(".") == 1) or \
              name in ("mid_block1", "mid_attn", "mid_block2", "in_conv", "out_conv", "time_mlp")
        if top:
            hooks.append(mod.register_forward_hook(lambda m, i, o, n=name: captured.__setitem__(n, o.detach())))
    with torch.no_grad():
        ref_out = model(x, cond, t)
    for h in hooks:
        h.remove()

    ours = {}
    with torch.no_grad():
        out = orc.unet_forward(sd, topo, x, cond, t, trace=lambda n, v: ours.__setitem__(n, v))
    assert set(ours) == set(captured)
    for n in ours:
        torch.testing.assert_close(ours[n], captured[n], rtol=0, atol=2e-6, msg=lambda m, n=n: f"{n}: {m}")
    torch.testing.assert_close(out, ref_out, rtol=0, atol=2e-6)


def test_denoise_equal_ddim(reference_module):
    cfg = UNetConfig(**SMALL)
    topo = topology(cfg)
    sd_np = make_state_dict(cfg, seed=11, perturb_norm=True)
    model = _load_ref(reference_module.ddim, cfg, sd_np)
    den = reference_module.ddim.DiffusionDenoiser(model, noise_steps=50)
    noisy = torch.from_numpy(synthetic_xray(2, 32, 32, seed=21))
    keep = noisy.clone()
    ref = den.denoise(noisy, inference_steps=8)
    assert torch.equal(noisy, keep)                  # the reference does not mutate its input
    ours = orc.denoise(orc.to_torch(sd_np), topo, noisy, noise_steps=50, inference_steps=8)
    torch.testing.assert_close(ours, ref, rtol=0, atol=5e-6)


def test_denoise_equal_cddpm(reference_module):
    cfg = UNetConfig(variant="cddpm", **SMALL)
    topo = topology(cfg)
    sd_np = make_state_dict(cfg, seed=12, perturb_norm=True)
    model = _load_ref(reference_module.cddpm, cfg, sd_np)
    den = reference_module.cddpm.DiffusionDenoiser(model, noise_steps=50)
    noisy = torch.from_numpy(synthetic_xray(1, 32, 32, seed=22))
    steps = timestep_list(50, 5)
    torch.manual_seed(1234)
    ref = den.denoise(noisy, inference_steps=5)
    # replay the reference's RNG consumption: one randn_like per iteration with i > 0
    torch.manual_seed(1234)
    noises = [torch.randn_like(noisy) * 0.5 if i > 0 else None for i in steps]
    ours = orc.denoise(orc.to_torch(sd_np), topo, noisy, noise_steps=50, inference_steps=5, step_noise=noises)
    torch.testing.assert_close(ours, ref, rtol=0, atol=5e-6)


def test_convtranspose_mean_fold_identity():
    """ConvT(4,2,1) followed by the bilinear half-size resample equals
    one 3x3 conv with W_eff — the identity the HIP planner relies on."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 12, 10, generator=g)
    w = torch.randn(8, 8, 4, 4, generator=g) * 0.1
    b = torch.randn(8, generator=g)
    up = torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=1)
    ref = torch.nn.functional.interpolate(up, size=x.shape[2:], mode="bilinear", align_corners=False)
    weff = torch.zeros(8, 8, 3, 3)
    for d in range(3):
        for e in range(3):
            for a in range(2):
                for bb in range(2):
                    ky, kx = a - 2 * d + 3, bb - 2 * e + 3
                    if 0 <= ky < 4 and 0 <= kx < 4:
                        weff[:, :, d, e] += 0.25 * w[:, :, ky, kx].t()
    got = torch.nn.functional.conv2d(x, weff, b, padding=1)
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-6)
