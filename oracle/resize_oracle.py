"""TEST INFRASTRUCTURE ONLY (see oracle/ddim_oracle.py header): CPU restatement of the pre/post-processing
either side of the sampler in the reference's serving path, for checking the on-device kernels.

What the reference does (Backend/run.py:143-149,193-201; Backend/cddpm/cddpmModels.py:485-503):
  * `transforms.Resize((S, S), interpolation=BICUBIC)` on a PIL 'L' image  ==  `Image.resize((S, S), Image.BICUBIC)`
  * `transforms.ToTensor()`                                                 ==  uint8 -> float32 / 255
  * `(output_np * 255).astype('uint8')` after `clamp(0, 1)`                 ==  float32 multiply, truncation
  * `Image.resize(original_size, Image.BICUBIC)` back
  * `compute_metrics` (Backend/DDIM/DDIMModel.py:290-300): skimage PSNR / SSIM with data_range = 1.

The resize lives in a third-party dependency, Pillow (12.2.0 in this image; the reference pins Pillow 10.0.1 in
Backend/requirements.txt -- the 8-bit resampler, src/libImaging/Resample.c, is unchanged between them).  Its
published algorithm for 8-bit single-band images is restated below: separable convolution, horizontal pass first
(only over the rows the vertical pass needs), bicubic kernel a = -0.5 with support 2 * max(1, scale), coefficients
normalised in double precision then rounded to 22-bit fixed point, accumulation from 1 << 21, `>> 22`, clip to
[0, 255], uint8 intermediate between the passes.  PINNED: tests/test_prepost_cpu.py checks this restatement
bit for bit against Pillow itself on random and structured images (up- and down-scaling, odd sizes).

scikit-image is not installed here, so psnr()/ssim() restate its published defaults (PSNR = 10 log10(R^2 / mse) in
float64; SSIM: 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, mean over the interior crop) with
scipy.ndimage.uniform_filter -- the routine skimage itself calls.  PARITY UNPINNED against skimage proper.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (bounds [out][2] int32 (first tap, tap count), coeffs [out][ksize] int32 fixed point, ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """Resamples the LAST axis of a uint8 array."""
    out = np.empty(img.shape[:-1] + (bounds.shape[0],), np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (src[..., x0:x0 + n] * kk[xx, :n].astype(np.int64)).sum(axis=-1) + (1 << (PRECISION_BITS - 1))
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """uint8 [H][W] -> uint8 [out_h][out_w], as Image.fromarray(img, 'L').resize((out_w, out_h), Image.BICUBIC)."""
    assert img.dtype == np.uint8 and img.ndim == 2
    h, w = img.shape
    cur = img
    if out_w != w:
        bh, kh, _ = precompute_coeffs(w, out_w)
        cur = _pass(cur, bh, kh)
    if out_h != h:
        bv, kv, _ = precompute_coeffs(h, out_h)
        cur = np.ascontiguousarray(_pass(np.ascontiguousarray(cur.T), bv, kv).T)
    return cur


def to_unit_float(img_u8: np.ndarray) -> np.ndarray:
    """transforms.ToTensor(): uint8 -> float32 / 255."""
    return img_u8.astype(np.float32) / np.float32(255.0)


def to_u8(x: np.ndarray) -> np.ndarray:
    """clamp(0, 1) then (x * 255).astype('uint8') (run.py:107,145)."""
    return (np.clip(x.astype(np.float32), np.float32(0), np.float32(1)) * np.float32(255.0)).astype(np.uint8)


def psnr(target: np.ndarray, pred: np.ndarray) -> float:
    t = np.clip(target, 0, 1).astype(np.float64)
    p = np.clip(pred, 0, 1).astype(np.float64)
    mse = np.mean((t - p) ** 2)
    return float(10.0 * np.log10(1.0 / mse))


def ssim(target: np.ndarray, pred: np.ndarray) -> float:
    from scipy.ndimage import uniform_filter
    x = np.clip(target, 0, 1).astype(np.float64)
    y = np.clip(pred, 0, 1).astype(np.float64)
    win, k1, k2, rng = 7, 0.01, 0.03, 1.0
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())
