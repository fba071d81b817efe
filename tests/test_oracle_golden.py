"""Oracle vs the committed golden fixtures (outputs of the reference itself,
tests/golden/make_golden.py).  CPU only; runs in the container and on the GPU box."""
import os

import numpy as np
import pytest
import torch

from midd_amd.config import UNetConfig, topology, timestep_list
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
SMALL = dict(model_channels=16, time_emb_dim=64)


def _load(name):
    return np.load(os.path.join(G, name))


def test_schedule_tables():
    g = _load("schedule.npz")
    for steps in (50, 100):
        b, a, ah = orc.schedule(steps)
        assert np.array_equal(b.numpy(), g[f"beta_{steps}"])
        assert np.array_equal(a.numpy(), g[f"alpha_{steps}"])
        assert np.array_equal(ah.numpy(), g[f"alpha_hat_{steps}"])
    # SURVEY.md App. B spot values
    assert abs(float(g["beta_50"][0]) - 9.999999747e-05) < 1e-12
    assert abs(float(g["alpha_hat_50"][49]) - 0.60295159) < 1e-7


@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_small_unet_layers_and_sampler(variant):
    g = _load(f"small_{variant}.npz")
    cfg = UNetConfig(variant=variant, **SMALL)
    topo = topology(cfg)
    sd = orc.to_torch(make_state_dict(cfg, seed=42, perturb_norm=True))
    x, cond = torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_cond"])
    # inputs are formula-generated: the fixtures must agree with the generator
    assert np.array_equal(g["fwd_x"], synthetic_xray(2, 32, 48, seed=100, kind="uniform"))
    assert np.array_equal(g["fwd_cond"], synthetic_xray(2, 32, 48, seed=200))
    t = torch.full((2,), int(g["fwd_t"]), dtype=torch.long)
    got = {}
    with torch.no_grad():
        eps = orc.unet_forward(sd, topo, x, cond, t, trace=lambda n, v: got.__setitem__(n, v.numpy()))
    layer_keys = [k for k in g.files if k.startswith("layer/")]
    assert len(layer_keys) == len(got)
    for k in layer_keys:
        np.testing.assert_allclose(got[k[6:]], g[k], rtol=0, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(eps.numpy(), g["fwd_eps"], rtol=0, atol=2e-6)

    steps = list(g["den_steps"])
    assert steps == timestep_list(50, int(g["den_inference_steps"]))
    noise = None
    if variant == "cddpm":
        noise = [torch.from_numpy(n) for n in g["den_noise_scaled"]]
    eps_log, x_log = [], []
    out = orc.denoise(sd, topo, torch.from_numpy(g["den_noisy"]), noise_steps=50,
                      inference_steps=int(g["den_inference_steps"]), step_noise=noise,
                      on_step=lambda i, e, xx: (eps_log.append(e.numpy()), x_log.append(xx.numpy())))
    # fixture eps are the raw network outputs; the +-5 clamp is inactive for these weights
    assert np.abs(g["den_eps"]).max() < 5
    np.testing.assert_allclose(np.stack(eps_log), g["den_eps"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(np.stack(x_log), g["den_x"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out.numpy(), g["den_out"], rtol=0, atol=1e-5)


def test_full_unet_64():
    g = _load("full_ddim_64.npz")
    cfg = UNetConfig()
    topo = topology(cfg)
    sd = orc.to_torch(make_state_dict(cfg, seed=int(g["seed_weights"])))
    noisy = torch.from_numpy(synthetic_xray(1, 64, 64, seed=int(g["seed_image"])))
    t = torch.full((1,), 49, dtype=torch.long)
    with torch.no_grad():
        eps = orc.unet_forward(sd, topo, noisy, noisy, t)
    np.testing.assert_allclose(eps.numpy(), g["fwd_eps_t49"], rtol=0, atol=5e-6)
    xs = []
    out = orc.denoise(sd, topo, noisy, 50, 50, on_step=lambda i, e, xx: xs.append(xx.numpy()))
    np.testing.assert_allclose(xs[0], g["den_x_after_1"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(xs[4], g["den_x_after_5"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out.numpy(), g["den_out"], rtol=0, atol=2e-5)


def test_full_unet_256_single_forward():
    g = _load("full_ddim_256.npz")
    cfg = UNetConfig()
    sd = orc.to_torch(make_state_dict(cfg, seed=42))
    noisy = torch.from_numpy(synthetic_xray(1, 256, 256, seed=1234))
    with torch.no_grad():
        eps = orc.unet_forward(sd, topology(cfg), noisy, noisy, torch.full((1,), 49, dtype=torch.long))
    np.testing.assert_allclose(eps.numpy(), g["fwd_eps_t49"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(eps.numpy(), g["den_eps_first"], rtol=0, atol=5e-6)
