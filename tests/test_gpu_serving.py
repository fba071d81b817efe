"""GPU tests of the callers either side of the sampler: the /denoise
diffusion branch (run.py:103-111,185-213) and the single-image CLI helper
(cddpmModels.py:470-504), each against the oracle run through the same pre/post-processing."""
import base64
import io

import numpy as np
import pytest
import torch
from PIL import Image

from midd_amd import UNetConfig, topology
from midd_amd.cli import denoise_image_diffusion
from midd_amd.server import (DiffusionService, create_app, preprocess, preprocess_device, tensor_to_base64,
                             tensor_to_base64_device)
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc

pytestmark = pytest.mark.gpu


def _png_bytes(w, h, seed):
    arr = (synthetic_xray(1, h, w, seed=seed)[0, 0] * 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(arr, mode="L").save(buf, format="PNG")
    return buf.getvalue()


def test_denoise_route_matches_oracle_pipeline(tmp_path):
    from fastapi.testclient import TestClient
    cfg = UNetConfig()
    sd = make_state_dict(cfg, seed=42)
    ckpt = tmp_path / "ddimdiffusion.pth"                      # the dict run.py:37-41 loads
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "noise_steps": 50,
                "best_psnr": 0.0, "best_ssim": 0.0, "epoch": 0}, ckpt)
    svc = DiffusionService(checkpoint=str(ckpt))
    data = _png_bytes(200, 152, seed=5)
    with TestClient(create_app(service=svc)) as client:
        assert client.get("/health").json()["models_loaded"]["diffusion"] is True
        r = client.post("/denoise", files={"file": ("xray.png", data, "image/png")})
        assert r.status_code == 200
        got = np.asarray(Image.open(io.BytesIO(base64.b64decode(r.json()["diffusion"]))))
    assert got.shape == (152, 200)
    # oracle through the same recipe: 512x512, inference_steps=8 (9 iterations), clamp, truncate, resize back
    x, size = preprocess(data)
    ref = orc.denoise(orc.to_torch(sd), topology(cfg), x, noise_steps=50, inference_steps=8).clamp(0, 1)
    ref_img = Image.fromarray((ref[0, 0].numpy() * 255).astype("uint8"), mode="L").resize(size, Image.BICUBIC)
    diff = np.abs(got.astype(np.int32) - np.asarray(ref_img).astype(np.int32))
    # |delta| < 1e-3 in [0,1] can move a truncated 8-bit level by at most one step
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02, (diff.max(), (diff > 0).mean())


def test_device_pre_and_post_processing_equal_the_host_recipe():
    """run.py:143-149,193-201 on the GPU (csrc/prepost.hip) vs PIL / numpy on the host: same bytes."""
    for (w, h) in [(200, 152), (1024, 768), (512, 512), (333, 517)]:
        data = _png_bytes(w, h, seed=w + h)
        x_host, size_host = preprocess(data)
        x_dev, size_dev = preprocess_device(data, torch.device("cuda"))
        assert size_host == size_dev == (w, h)
        assert torch.equal(x_dev.cpu(), x_host)
        out = torch.from_numpy(synthetic_xray(1, 512, 512, seed=w)).clamp(0, 1) * 1.1 - 0.05   # leaves [0,1] in places
        out = torch.clamp(out, 0, 1)
        assert tensor_to_base64_device(out.cuda(), (w, h)) == tensor_to_base64(out, (w, h))


def test_cli_helper_cddpm_replay(tmp_path):
    cfg = UNetConfig(variant="cddpm")
    sd = make_state_dict(cfg, seed=7)
    ckpt = tmp_path / "best_diffusion_denoiser_new.pth"
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "noise_steps": 50}, ckpt)
    png = tmp_path / "in.png"
    png.write_bytes(_png_bytes(120, 88, seed=9))
    size, steps = 64, 5
    g = np.random.Generator(np.random.Philox(key=11))
    noise = torch.from_numpy(np.stack([0.5 * g.standard_normal((1, 1, size, size), dtype=np.float32) for _ in range(steps)]))
    out = denoise_image_diffusion(str(ckpt), str(png), device_type="cuda", img_size=size, inference_steps=steps,
                                  variant="cddpm", step_noise=noise.cuda())
    assert out.size == (120, 88)
    img = Image.open(png).convert("L")
    x = torch.from_numpy(np.asarray(img.resize((size, size), Image.BICUBIC), np.uint8).astype(np.float32) / 255.0)[None, None]
    ref = orc.denoise(orc.to_torch(sd), topology(cfg), x, 50, steps, step_noise=list(noise))
    ref_img = Image.fromarray((ref[0, 0].numpy() * 255).astype(np.uint8), mode="L").resize(img.size, Image.BICUBIC)
    diff = np.abs(np.asarray(out).astype(np.int32) - np.asarray(ref_img).astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02
