"""GPU: the on-device pre/post-processing kernels (csrc/prepost.hip, through the C ABI) against the oracle
(oracle/resize_oracle.py, itself pinned to Pillow on the CPU) and against Pillow directly.
Bar: bit-exact for the byte work (resize, conversions); 1e-9 relative for the fp64 metrics."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import midd_loader  # noqa: E402

midd_loader.load()
from midd_amd import prepost  # noqa: E402
from oracle import resize_oracle as ro  # noqa: E402

pytestmark = pytest.mark.gpu


def _batch(n, h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8),
            ((np.sin(xx / 7.0) * np.cos(yy / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8),
            ((xx // 8 + yy // 8) % 2 * 255).astype(np.uint8)]
    return np.stack([imgs[i % 3] for i in range(n)])


@pytest.mark.parametrize("shape", [((300, 400), (512, 512)), ((1024, 768), (512, 512)), ((512, 512), (300, 400)),
                                   ((512, 512), (1024, 768)), ((37, 53), (512, 512)), ((512, 512), (37, 53)),
                                   ((512, 512), (512, 512)), ((600, 512), (512, 512)), ((512, 700), (512, 512)),
                                   ((1, 9), (4, 4)), ((2000, 1500), (512, 512))])
def test_resize_is_pillow_bit_for_bit(shape):
    (h, w), (oh, ow) = shape
    batch = _batch(3, h, w, seed=h * 131 + w)
    got = prepost.resize_bicubic_u8(torch.from_numpy(batch).cuda(), (oh, ow)).cpu().numpy()
    for i in range(batch.shape[0]):
        ref = np.asarray(Image.fromarray(batch[i], "L").resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(got[i], ref), f"image {i}: max diff {np.abs(got[i].astype(int) - ref).max()}"
        assert np.array_equal(got[i], ro.resize_bicubic_u8(batch[i], ow, oh))
    single = prepost.resize_bicubic_u8(torch.from_numpy(batch[0]).cuda(), (oh, ow)).cpu().numpy()
    assert np.array_equal(single, got[0])                                    # [H,W] form, batch independence


def test_conversions_are_exact():
    u8 = torch.arange(256, dtype=torch.uint8).repeat(5).cuda()
    f = prepost.to_unit_float(u8)
    np.testing.assert_array_equal(f.cpu().numpy(), ro.to_unit_float(u8.cpu().numpy()))
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-0.2, 1.2, 100000).astype(np.float32),
                        np.arange(256, dtype=np.float32) / np.float32(255.0),
                        np.nextafter(np.arange(1, 256, dtype=np.float32) / np.float32(255.0), np.float32(0))])
    got = prepost.to_u8(torch.from_numpy(x).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, ro.to_u8(x))
    np.testing.assert_array_equal(prepost.to_u8(f).cpu().numpy(), u8.cpu().numpy())          # round trip


def test_metrics_match_the_restated_skimage_defaults():
    rng = np.random.default_rng(5)
    n, h, w = 3, 96, 130
    t = rng.random((n, 1, h, w)).astype(np.float32) * 1.2 - 0.1                  # exercises the clip
    p = (t + rng.standard_normal(t.shape).astype(np.float32) * np.array([0.01, 0.1, 0.3], np.float32)[:, None, None, None])
    m = prepost.image_metrics(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
    for i in range(n):
        assert m[i, 0] == pytest.approx(ro.psnr(t[i, 0], p[i, 0]), rel=1e-9)
        assert m[i, 1] == pytest.approx(ro.ssim(t[i, 0], p[i, 0]), rel=1e-9, abs=1e-12)
    ps, ss = prepost.compute_metrics(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda())
    assert ps == pytest.approx(np.mean([ro.psnr(t[i, 0], p[i, 0]) for i in range(n)]), rel=1e-9)
    assert ss == pytest.approx(np.mean([ro.ssim(t[i, 0], p[i, 0]) for i in range(n)]), rel=1e-9)
    again = prepost.image_metrics(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
    assert np.array_equal(m, again)                                                # fixed summation order


def test_errors_are_loud():
    with pytest.raises(RuntimeError):
        prepost.resize_bicubic_u8(torch.zeros((4, 4), dtype=torch.uint8), (8, 8))           # CPU tensor: no fallback
    with pytest.raises(TypeError):
        prepost.to_u8(torch.zeros(4, dtype=torch.float64).cuda())
    from midd_amd.native import MiddError
    with pytest.raises(MiddError):
        prepost.image_metrics(torch.zeros((1, 5, 5)).cuda(), torch.zeros((1, 5, 5)).cuda())  # smaller than the SSIM window
