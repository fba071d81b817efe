"""CPU: the pre/post-processing oracle (oracle/resize_oracle.py) pinned against Pillow itself -- the third-party
dependency that implements `transforms.Resize(..., BICUBIC)` / `Image.resize(..., BICUBIC)` for the reference
(Backend/run.py:146,198) -- and against direct numpy formulas for the conversions and PSNR."""
import os
import sys

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resize_oracle as ro  # noqa: E402


def _images(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    return [rng.integers(0, 256, (h, w), dtype=np.uint8),                                   # noise: every clip path
            ((np.sin(xx / 7.0) * np.cos(yy / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8),     # smooth
            ((xx // 8 + yy // 8) % 2 * 255).astype(np.uint8),                               # checkerboard: overshoot
            np.full((h, w), 255, np.uint8), np.zeros((h, w), np.uint8)]


@pytest.mark.parametrize("shape", [((300, 400), (512, 512)), ((1024, 768), (512, 512)), ((512, 512), (300, 400)),
                                   ((512, 512), (1024, 768)), ((37, 53), (512, 512)), ((512, 512), (37, 53)),
                                   ((512, 512), (512, 512)), ((600, 512), (512, 512)), ((512, 700), (512, 512)),
                                   ((1, 9), (4, 4)), ((5, 5), (1, 1))])
def test_resize_restatement_is_pillow_bit_for_bit(shape):
    (h, w), (oh, ow) = shape
    for i, img in enumerate(_images(h, w, seed=h * 131 + w)):
        ref = np.asarray(Image.fromarray(img, "L").resize((ow, oh), Image.BICUBIC))
        got = ro.resize_bicubic_u8(img, ow, oh)
        assert got.shape == ref.shape and np.array_equal(got, ref), f"image {i}: {np.abs(got.astype(int) - ref).max()}"


def test_coefficients_sum_to_one_and_bounds_cover_the_support():
    for in_size, out_size in [(400, 512), (1024, 512), (53, 512), (512, 53)]:
        bounds, kk, ksize = ro.precompute_coeffs(in_size, out_size)
        assert kk.shape == (out_size, ksize)
        assert np.all(np.abs(kk.sum(axis=1) - (1 << ro.PRECISION_BITS)) <= ksize)      # rounding of each tap only
        assert np.all(bounds[:, 0] >= 0) and np.all(bounds[:, 0] + bounds[:, 1] <= in_size)


def test_conversions_match_the_reference_recipes():
    u8 = np.arange(256, dtype=np.uint8)
    np.testing.assert_array_equal(ro.to_unit_float(u8), u8.astype(np.float32) / 255.0)      # ToTensor
    x = np.array([-0.5, 0.0, 0.0039, 0.00393, 0.5, 0.999, 1.0, 1.7], np.float32)
    np.testing.assert_array_equal(ro.to_u8(x), (np.clip(x, 0, 1) * 255).astype("uint8"))  # run.py:107,145
    np.testing.assert_array_equal(ro.to_u8(ro.to_unit_float(u8)), u8)                       # round trip is exact


def test_psnr_ssim_known_answers():
    rng = np.random.default_rng(3)
    t = rng.random((64, 80)).astype(np.float32)
    assert ro.ssim(t, t) == pytest.approx(1.0, abs=1e-12)
    p = np.clip(t + 0.1, 0, 1).astype(np.float32)
    assert ro.psnr(t, p) == pytest.approx(10 * np.log10(1.0 / np.mean((t.astype(np.float64) - p.astype(np.float64)) ** 2)))
    const = np.full((32, 32), 0.25, np.float32)
    assert ro.psnr(const, const + 0.1) == pytest.approx(20.0, abs=1e-5)                     # mse = 0.01
    noisy = np.clip(t + 0.2 * rng.standard_normal(t.shape), 0, 1).astype(np.float32)
    assert 0.0 < ro.ssim(t, noisy) < ro.ssim(t, np.clip(t + 0.02 * rng.standard_normal(t.shape), 0, 1).astype(np.float32)) < 1.0
