"""Pre/post-processing either side of the sampler, on the GPU.

Host-side mirror of what the reference does around `denoise` with PIL / torchvision / numpy on the host
(Backend/run.py:143-149,193-201; Backend/cddpm/cddpmModels.py:485-503) and of `compute_metrics`
(Backend/DDIM/DDIMModel.py:290-300), backed by libmidd.so (csrc/prepost.hip).  GPU tensors only: like the sampler
there is no CPU fallback here -- the reference's own host recipe (`server.preprocess` / `tensor_to_base64`)
remains the path for CPU tensors.
"""
from typing import Tuple

import torch

from . import native


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(t: torch.Tensor, dtype: torch.dtype, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a tensor on the GPU (the HIP kernels are the only implementation)")
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def _workspace(nbytes: int, device) -> Tuple[torch.Tensor, int]:
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return buf, (buf.data_ptr() + 255) & ~255


def resize_bicubic_u8(images: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """uint8 [N,H,W] (or [H,W]) -> uint8 [N,size[0],size[1]]; bit-identical to
    `Image.fromarray(img, 'L').resize((size[1], size[0]), Image.BICUBIC)` per image
    (= `transforms.Resize(size, BICUBIC)` on a PIL 'L' image, run.py:198)."""
    squeeze = images.dim() == 2
    x = _need_cuda(images[None] if squeeze else images, torch.uint8, "resize_bicubic_u8")
    if x.dim() != 3:
        raise ValueError("resize_bicubic_u8: expected [N,H,W] or [H,W]")
    n, sh, sw = x.shape
    dh, dw = int(size[0]), int(size[1])
    lib = native.lib()
    nbytes = lib.mi_resize_workspace_bytes(n, sw, sh, dw, dh)
    if nbytes == 0:
        raise ValueError(f"resize_bicubic_u8: bad sizes {tuple(x.shape)} -> {(dh, dw)}")
    out = torch.empty((n, dh, dw), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        ws, wptr = _workspace(nbytes, x.device)
        native.check(lib.mi_resize_bicubic_u8(x.data_ptr(), n, sw, sh, out.data_ptr(), dw, dh, wptr, nbytes, _stream(x)))
        ws.record_stream(torch.cuda.current_stream(x.device))
    return out[0] if squeeze else out


def to_unit_float(images_u8: torch.Tensor) -> torch.Tensor:
    """`transforms.ToTensor()` scaling: uint8 -> float32 / 255 (same shape)."""
    x = _need_cuda(images_u8, torch.uint8, "to_unit_float")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        native.check(native.lib().mi_u8_to_unit_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream(x)))
    return out


def to_u8(images: torch.Tensor) -> torch.Tensor:
    """`(clamp(x, 0, 1) * 255).astype('uint8')` (run.py:107,145): fp32 multiply, truncation."""
    x = _need_cuda(images, torch.float32, "to_u8")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        native.check(native.lib().mi_unit_f32_to_u8(x.data_ptr(), out.data_ptr(), x.numel(), _stream(x)))
    return out


def image_metrics(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Per-image (PSNR, SSIM) of `compute_metrics`: float64 [N,2] on the GPU; inputs [N,1,H,W] or [N,H,W] fp32."""
    p = _need_cuda(pred, torch.float32, "image_metrics")
    t = _need_cuda(target, torch.float32, "image_metrics")
    if p.shape != t.shape:
        raise ValueError("image_metrics: shapes differ")
    if p.dim() == 4:
        if p.shape[1] != 1:
            raise ValueError("image_metrics: single-channel images expected")
        p, t = p[:, 0].contiguous(), t[:, 0].contiguous()
    n, h, w = p.shape
    lib = native.lib()
    nbytes = lib.mi_metrics_workspace_bytes(n, h)
    out = torch.empty((n, 2), dtype=torch.float64, device=p.device)
    with torch.cuda.device(p.device):
        ws, wptr = _workspace(nbytes, p.device)
        native.check(lib.mi_image_metrics(t.data_ptr(), p.data_ptr(), n, h, w, out.data_ptr(), wptr, nbytes, _stream(p)))
        ws.record_stream(torch.cuda.current_stream(p.device))
    return out


def compute_metrics(pred: torch.Tensor, target: torch.Tensor) -> Tuple[float, float]:
    """Drop-in for the reference's `compute_metrics(pred, target)` (DDIMModel.py:290-300): batch means of PSNR, SSIM."""
    m = image_metrics(pred, target).mean(dim=0).cpu()
    return float(m[0]), float(m[1])
