"""CPU tests of the /denoise diffusion branch: HTTP contract and pre/post-processing
(run.py:143-149,185-226).  The sampler is replaced by a stand-in here; the GPU test runs the real one."""
import base64
import io

import numpy as np
import pytest
import torch
from PIL import Image

from midd_amd.server import (DiffusionService, SERVE_SIZE, create_app, extract_multipart_file, preprocess,
                             tensor_to_base64)


def _png(w=96, h=64, seed=0):
    rng = np.random.default_rng(seed)
    img = Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), mode="L")
    buf = io.BytesIO()
    img.save(buf, format="PNG")
    return buf.getvalue()


def test_preprocess_matches_reference_recipe():
    data = _png(96, 64)
    x, size = preprocess(data)
    assert size == (96, 64) and x.shape == (1, 1, 512, 512) and x.dtype == torch.float32
    ref = Image.open(io.BytesIO(data)).convert("L").resize((512, 512), Image.BICUBIC)
    assert np.array_equal((x[0, 0].numpy() * 255).round().astype(np.uint8), np.asarray(ref))
    assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0


def test_postprocess_truncates_and_resizes_back():
    t = torch.full((1, 1, 512, 512), 0.999)
    out = Image.open(io.BytesIO(base64.b64decode(tensor_to_base64(t, (96, 64)))))
    assert out.size == (96, 64) and out.mode == "L"
    assert int(np.asarray(out).max()) == 254          # (0.999*255).astype(uint8) truncates, run.py:145


def test_multipart_parser():
    boundary = "XyZ"
    payload = b"\x89PNG fake \r\n bytes"
    body = (f"--{boundary}\r\nContent-Disposition: form-data; name=\"other\"\r\n\r\nv\r\n"
            f"--{boundary}\r\nContent-Disposition: form-data; name=\"file\"; filename=\"a.png\"\r\n"
            f"Content-Type: image/png\r\n\r\n").encode() + payload + f"\r\n--{boundary}--\r\n".encode()
    assert extract_multipart_file(body, f"multipart/form-data; boundary={boundary}") == payload
    with pytest.raises(ValueError):
        extract_multipart_file(body, "application/json")


def test_http_contract_with_stand_in_sampler():
    from fastapi.testclient import TestClient
    svc = DiffusionService(device=torch.device("cpu"), denoise_fn=lambda x: 1.0 - x)
    with TestClient(create_app(service=svc)) as client:
        assert client.get("/").json()["endpoints"] == {"denoise": "/denoise", "health": "/health"}
        h = client.get("/health").json()
        assert h["status"] == "healthy" and h["models_loaded"]["diffusion"] is True
        r = client.post("/denoise", files={"file": ("x.png", _png(80, 48), "image/png")})
        assert r.status_code == 200
        js = r.json()
        assert set(js) == {"diffusion", "nafnet", "expert", "hybrid"}
        assert js["nafnet"] is None and js["expert"] is None and js["hybrid"] is None
        img = Image.open(io.BytesIO(base64.b64decode(js["diffusion"])))
        assert img.size == (80, 48)
        assert client.post("/denoise", content=b"junk", headers={"content-type": "text/plain"}).status_code == 500

    # a failing branch yields null, never a crash (run.py:90,96-101)
    def boom(x):
        raise RuntimeError("sampler failed")
    svc2 = DiffusionService(device=torch.device("cpu"), denoise_fn=boom)
    with TestClient(create_app(service=svc2)) as client:
        assert client.post("/denoise", files={"file": ("x.png", _png(), "image/png")}).json()["diffusion"] is None
