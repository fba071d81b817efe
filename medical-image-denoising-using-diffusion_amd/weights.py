"""Portable random-init weights and synthetic X-ray-like inputs (numpy only).

The reference's trained checkpoints are git-ignored and absent, so every
measurement uses random-init weights "of that architecture".  torch's RNG streams differ
between CPU and GPU builds, therefore weights come from ``numpy.random.Philox`` keyed by
the parameter *name*: the container, the GPU box and the golden-vector script all
regenerate bit-identical tensors from (seed, name, shape) alone.

Scale follows PyTorch's default init for the layers the reference uses (Conv2d / Linear /
ConvTranspose2d: U(-1/sqrt(fan_in), +1/sqrt(fan_in)) for weight and bias; GroupNorm
gamma=1, beta=0).  ``perturb_norm=True`` additionally jitters gamma/beta so parity tests
exercise the affine path.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Tuple

import numpy as np

from .config import UNetConfig, param_shapes


def _rng(seed: int, name: str) -> np.random.Generator:
    digest = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = np.frombuffer(digest[:16], dtype=np.uint64).copy()
    return np.random.Generator(np.random.Philox(key=key))


def _fan_in(name: str, shape: Tuple[int, ...]) -> int:
    if len(shape) == 4:
        # Conv2d [Cout,Cin,k,k] -> Cin*k*k ; ConvTranspose2d [Cin,Cout,k,k] -> torch uses size(1)*k*k
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    raise ValueError(name)


def make_state_dict(cfg: UNetConfig, seed: int = 42, perturb_norm: bool = False) -> Dict[str, np.ndarray]:
    """name -> float32 ndarray for every tensor of ``param_shapes(cfg)``."""
    shapes = dict(param_shapes(cfg))
    sd: Dict[str, np.ndarray] = {}
    for name, shape in shapes.items():
        g = _rng(seed, name)
        is_norm = (".block1.0." in name or ".block2.0." in name or ".norm." in name
                   or name.startswith("out_conv.0."))
        if is_norm:
            if name.endswith(".weight"):
                v = np.ones(shape, np.float32)
                if perturb_norm:
                    v = v + 0.1 * g.standard_normal(shape, dtype=np.float32)
            else:
                v = np.zeros(shape, np.float32)
                if perturb_norm:
                    v = 0.1 * g.standard_normal(shape, dtype=np.float32)
            sd[name] = v.astype(np.float32)
            continue
        if name.endswith(".weight"):
            bound = 1.0 / np.sqrt(_fan_in(name, shape))
        else:
            wshape = shapes[name[:-len("bias")] + "weight"]
            bound = 1.0 / np.sqrt(_fan_in(name, wshape))
        sd[name] = g.uniform(-bound, bound, size=shape).astype(np.float32)
    return sd


def synthetic_xray(batch: int, height: int, width: int, seed: int = 1234,
                   kind: str = "xray") -> np.ndarray:
    """float32 [B,1,H,W] in [0,1].

    kind='xray'   : smooth blobs/gradients with speckle + Gaussian noise (the dataset
                    directories named at /root/reference/Backend/cddpm/cddpmTrain.py:3 are
                    speckle/Gaussian-corrupted X-rays) — image i uses seed+i.
    kind='uniform': U[0,1) — used by pure parity tests.
    """
    out = np.empty((batch, 1, height, width), np.float32)
    yy, xx = np.meshgrid(np.linspace(0, 1, height, dtype=np.float32),
                         np.linspace(0, 1, width, dtype=np.float32), indexing="ij")
    for i in range(batch):
        g = _rng(seed + i, "image")
        if kind == "uniform":
            out[i, 0] = g.random((height, width), dtype=np.float32)
            continue
        img = 0.25 + 0.3 * yy + 0.1 * xx
        for _ in range(6):
            cy, cx = g.random(2)
            s = 0.05 + 0.2 * g.random()
            a = 0.5 * (g.random() - 0.3)
            img = img + a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
        clean = np.clip(img, 0, 1).astype(np.float32)
        n1 = g.standard_normal((height, width), dtype=np.float32)
        n2 = g.standard_normal((height, width), dtype=np.float32)
        out[i, 0] = np.clip(clean * (1 + 0.2 * n1) + 0.05 * n2, 0, 1)
    return out


def psnr(a: np.ndarray, b: np.ndarray, data_range: float = 1.0) -> float:
    """10*log10(R^2 / mse) — the skimage formula the reference reports (DDIMModel.py:298)."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    if mse == 0:
        return float("inf")
    return 10.0 * np.log10(data_range * data_range / mse)
