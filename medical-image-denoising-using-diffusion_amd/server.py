"""The reference's HTTP service, diffusion branch only, on the MI355X sampler.

Contract kept from /root/reference/Backend/run.py:
  * ``POST /denoise`` with a multipart field ``file`` (run.py:185-186) -> JSON whose ``"diffusion"``
    value is a base64 PNG string, or ``null`` when that branch failed (run.py:96-101).  The other
    three keys of the reference's response (``nafnet`` / ``expert`` / ``hybrid``, models outside this
    repository's scope) are present and ``null`` so the React client (frontend/src/services/api.js:20-25)
    keeps working.
  * pre/post-processing of ``_process_diffusion`` / ``_tensor_to_base64`` (run.py:103-111,143-149):
    grayscale, bicubic resize to 512x512, ``ToTensor`` scaling, ``denoise(x, inference_steps=8)`` (9
    iterations), clamp, ``(x*255).astype(uint8)`` (truncation), bicubic resize back, PNG, base64.
  * ``GET /health`` (run.py:215-226) and ``GET /`` (run.py:166-175).
  * checkpoint dict ``{'model_state_dict', 'noise_steps', ...}`` (run.py:37-41), loaded with
    ``weights_only=True`` (nothing from the file is executed).
The sampler call runs in a worker thread (``asyncio.to_thread``, as run.py:85) on the GPU.  On a GPU service the
resizes, the ToTensor scaling and the uint8 conversion also run on the device (``prepost``; bit-identical to the
PIL / numpy recipe, which stays as ``preprocess`` / ``tensor_to_base64`` for CPU tensors and tests).

``python-multipart`` is not available in this image, so the multipart body is parsed with the
standard library instead of FastAPI's ``UploadFile``; the wire format is the same.
"""


import asyncio
import base64
import io
import time
from email.parser import BytesParser
from email.policy import HTTP
from typing import Callable, Optional, Tuple

import numpy as np
import torch
from PIL import Image

from .modules import UNetDiffusion
from .sampler import DiffusionDenoiser

SERVE_SIZE = (512, 512)          # run.py:198
SERVE_INFERENCE_STEPS = 8        # run.py:107 (-> 9 iterations with noise_steps=50)


# ------------------------------------------------------------------------------ pre / post
def preprocess(image_bytes: bytes) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """bytes -> (fp32 [1,1,512,512] in [0,1], original (width, height)) — run.py:193-201.
    ``transforms.Resize`` on a PIL image is ``Image.resize(..., BICUBIC)``; ``ToTensor`` is uint8/255."""
    image = Image.open(io.BytesIO(image_bytes)).convert("L")
    original_size = image.size
    resized = image.resize(SERVE_SIZE[::-1], Image.BICUBIC)
    arr = np.asarray(resized, dtype=np.uint8).astype(np.float32) / 255.0
    return torch.from_numpy(arr)[None, None], original_size


def tensor_to_base64(tensor: torch.Tensor, size: Tuple[int, int]) -> str:
    """[1,1,H,W] in [0,1] -> base64 PNG at the original size — run.py:143-149."""
    output_np = tensor.squeeze(0).squeeze(0).cpu().numpy()
    output_img = Image.fromarray((output_np * 255).astype("uint8"), mode="L")
    output_img = output_img.resize(size, Image.BICUBIC)
    buffered = io.BytesIO()
    output_img.save(buffered, format="PNG")
    return base64.b64encode(buffered.getvalue()).decode()


def preprocess_device(image_bytes: bytes, device: torch.device) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """`preprocess` with the resize and the ToTensor scaling on the GPU (csrc/prepost.hip; bit-identical to the
    host recipe): only the PNG/JPEG decode stays on the host."""
    from . import prepost
    image = Image.open(io.BytesIO(image_bytes)).convert("L")
    original_size = image.size
    raw = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).to(device, non_blocking=True)
    resized = prepost.resize_bicubic_u8(raw, SERVE_SIZE)
    return prepost.to_unit_float(resized)[None, None], original_size


def tensor_to_base64_device(tensor: torch.Tensor, size: Tuple[int, int]) -> str:
    """`tensor_to_base64` with clamp / x255 truncation / resize-back on the GPU; PNG encoding on the host."""
    from . import prepost
    u8 = prepost.to_u8(tensor.reshape(tensor.shape[-2], tensor.shape[-1]).float())
    back = prepost.resize_bicubic_u8(u8, (size[1], size[0])).cpu().numpy()      # PIL size is (width, height)
    buffered = io.BytesIO()
    Image.fromarray(back, mode="L").save(buffered, format="PNG")
    return base64.b64encode(buffered.getvalue()).decode()


def extract_multipart_file(body: bytes, content_type: str, field: str = "file") -> bytes:
    """Returns the payload of multipart form field ``field`` (stdlib parser)."""
    if "multipart/form-data" not in (content_type or ""):
        raise ValueError("expected multipart/form-data")
    msg = BytesParser(policy=HTTP).parsebytes(b"Content-Type: " + content_type.encode() + b"\r\n\r\n" + body)
    for part in msg.iter_parts():
        if part.get_param("name", header="content-disposition") == field:
            return part.get_payload(decode=True)
    raise ValueError(f"multipart field '{field}' missing")


# ------------------------------------------------------------------------------ service
class DiffusionService:
    """Counterpart of ModelManager's diffusion members (run.py:20-42,103-111)."""

    def __init__(self, checkpoint: Optional[str] = None, device: Optional[torch.device] = None,
                 denoise_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.checkpoint = checkpoint
        self.diffusion_model = None
        self.diffusion_denoiser = None
        self.random_init = False
        self._denoise_fn = denoise_fn          # tests inject a stand-in; production uses the HIP sampler

    def load_models(self) -> None:
        model = UNetDiffusion(in_channels=1, model_channels=48, channel_mult=(1, 2, 3, 4), num_res_blocks=2,
                              attention_resolutions=(3,), dropout=0.0, time_emb_dim=192)
        noise_steps = 50
        if self.checkpoint:
            ckpt = torch.load(self.checkpoint, map_location="cpu", weights_only=True)
            model.load_state_dict(ckpt["model_state_dict"])
            noise_steps = int(ckpt.get("noise_steps", 50))
        else:
            self.random_init = True            # the trained weights are not distributed with the reference
        self.diffusion_model = model.to(self.device).eval()
        self.diffusion_denoiser = DiffusionDenoiser(self.diffusion_model, noise_steps=noise_steps)

    def process_diffusion(self, input_tensor: torch.Tensor, original_size: Tuple[int, int]) -> str:
        """run.py:103-111."""
        start = time.time()
        with torch.no_grad():
            if self._denoise_fn is not None:
                output = self._denoise_fn(input_tensor)
            else:
                output = self.diffusion_denoiser.denoise(input_tensor, inference_steps=SERVE_INFERENCE_STEPS)
            output = torch.clamp(output, 0, 1)
            result = (tensor_to_base64_device if output.is_cuda else tensor_to_base64)(output, original_size)
        print(f"  Diffusion: {time.time() - start:.2f}s")
        return result

    async def process_all_models(self, input_tensor: torch.Tensor, original_size: Tuple[int, int]) -> dict:
        """run.py:80-101 with the three out-of-scope branches reported as null."""
        results = await asyncio.gather(asyncio.to_thread(self.process_diffusion, input_tensor, original_size),
                                       return_exceptions=True)
        return {"diffusion": results[0] if not isinstance(results[0], Exception) else None,
                "nafnet": None, "expert": None, "hybrid": None}

    def denoise_bytes(self, image_bytes: bytes) -> dict:
        """Synchronous helper: the whole request path without HTTP."""
        x, size = self.preprocess(image_bytes)
        return asyncio.run(self.process_all_models(x, size))

    def preprocess(self, image_bytes: bytes) -> Tuple[torch.Tensor, Tuple[int, int]]:
        """Decode + resize + scale; on the GPU when the service runs there (same bytes either way)."""
        if self.device.type == "cuda" and self._denoise_fn is None:
            return preprocess_device(image_bytes, self.device)
        x, size = preprocess(image_bytes)
        return x.to(self.device), size


def create_app(service: Optional[DiffusionService] = None, checkpoint: Optional[str] = None):
    """FastAPI application with the reference's routes (run.py:159-226)."""
    from contextlib import asynccontextmanager

    from fastapi import FastAPI, HTTPException, Request
    from fastapi.middleware.cors import CORSMiddleware
    from fastapi.responses import JSONResponse

    svc = service or DiffusionService(checkpoint=checkpoint)

    @asynccontextmanager
    async def lifespan(app):
        if svc.diffusion_model is None and svc._denoise_fn is None:
            svc.load_models()
        yield

    app = FastAPI(title="X-Ray Denoising API", description="diffusion branch on MI355X", version="2.0.0", lifespan=lifespan)
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"], allow_headers=["*"])
    app.state.service = svc

    @app.get("/")
    async def root():
        return {"message": "X-Ray Denoising API with Hybrid Routing", "status": "running",
                "endpoints": {"denoise": "/denoise", "health": "/health"}}

    @app.post("/denoise")
    async def denoise_xray(request: Request):
        try:
            total_start = time.time()
            image_data = extract_multipart_file(await request.body(), request.headers.get("content-type", ""))
            input_tensor, original_size = svc.preprocess(image_data)
            results = await svc.process_all_models(input_tensor, original_size)
            print(f"Total request time: {time.time() - total_start:.2f}s")
            return JSONResponse(content=results)
        except Exception as e:                                   # run.py:210-213
            raise HTTPException(status_code=500, detail=str(e))

    @app.get("/health")
    async def health_check():
        return {"status": "healthy", "device": str(svc.device),
                "models_loaded": {"diffusion": svc.diffusion_model is not None or svc._denoise_fn is not None,
                                  "nafnet": False, "expert": False, "hybrid": False}}

    return app


if __name__ == "__main__":      # python -m midd_amd.server [checkpoint.pth]
    import sys

    import uvicorn
    uvicorn.run(create_app(checkpoint=sys.argv[1] if len(sys.argv) > 1 else None), host="0.0.0.0", port=8000, log_level="info")
