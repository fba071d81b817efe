"""CPU-only checks of the C-ABI library and host logic (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig, param_shapes, native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "midd.h")).read()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = native.lib()
    bound = {n for n, _, _ in native.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.mi_version()


def _cfg_struct(cfg: UNetConfig):
    s = native.UNetCfg()
    s.in_channels, s.model_channels, s.num_levels = cfg.in_channels, cfg.model_channels, len(cfg.channel_mult)
    for i, m in enumerate(cfg.channel_mult):
        s.channel_mult[i] = m
    s.num_res_blocks, s.num_attention_levels = cfg.num_res_blocks, len(cfg.attention_resolutions)
    for i, a in enumerate(cfg.attention_resolutions):
        s.attention_levels[i] = a
    s.time_emb_dim, s.variant = cfg.time_emb_dim, native.MI_VARIANT[cfg.variant]
    return s


@pytest.mark.parametrize("variant", ["ddim", "cddpm"])
def test_native_topology_matches_python(variant):
    lib = native.lib()
    cfg = UNetConfig(variant=variant)
    h = C.c_void_p()
    native.check(lib.mi_unet_plan_create(C.byref(_cfg_struct(cfg)), C.byref(h)))
    names = [lib.mi_unet_weight_name(h, i).decode() for i in range(lib.mi_unet_num_weights(h))]
    assert names == [n for n, _ in param_shapes(cfg)]
    assert len(names) == 308 if variant == "ddim" else len(names) > 0
    # load-time validation mirrors load_state_dict's errors
    w = np.zeros((48, 2, 3, 3), np.float32)
    shp = (C.c_int64 * 4)(48, 2, 3, 3)
    native.check(lib.mi_unet_load_weights(h, b"in_conv.weight", w.ctypes.data_as(C.c_void_p), shp, 4))
    bad = (C.c_int64 * 4)(48, 3, 3, 3)
    assert lib.mi_unet_load_weights(h, b"in_conv.weight", w.ctypes.data_as(C.c_void_p), bad, 4) == -1
    assert b"size mismatch" in lib.mi_last_error()
    assert lib.mi_unet_load_weights(h, b"nope.weight", w.ctypes.data_as(C.c_void_p), shp, 4) == -1
    assert b"unexpected key" in lib.mi_last_error()
    # finalize with missing weights is a state error, not a crash
    assert lib.mi_unet_finalize(h, 50) == -2
    assert b"missing key" in lib.mi_last_error()
    lib.mi_plan_destroy(h)


def test_plan_create_rejects_unsupported_configs():
    lib = native.lib()
    h = C.c_void_p()
    s = _cfg_struct(UNetConfig())
    s.model_channels = 24
    assert lib.mi_unet_plan_create(C.byref(s), C.byref(h)) == -1
    assert b"multiple of 16" in lib.mi_last_error()
    s = _cfg_struct(UNetConfig())
    s.variant = 7
    assert lib.mi_unet_plan_create(C.byref(s), C.byref(h)) == -1


def test_container_state_dict_and_schedule():
    m = UNetDiffusion()
    sd = m.state_dict()
    assert list(sd.keys()) == [n for n, _ in param_shapes(UNetConfig())]
    assert sum(v.numel() for v in sd.values()) == 12_823_489
    m2 = UNetDiffusion()
    m2.load_state_dict(sd)
    d = DiffusionDenoiser(m2, noise_steps=50)
    assert d.model is m2 and d.noise_steps == 50
    g = np.load(os.path.join(ROOT, "tests", "golden", "schedule.npz"))
    assert np.array_equal(d.beta.cpu().numpy(), g["beta_50"])
    assert np.array_equal(d.alpha.cpu().numpy(), g["alpha_50"])
    assert np.array_equal(d.alpha_hat.cpu().numpy(), g["alpha_hat_50"])
    with pytest.raises(RuntimeError):       # CPU tensors never silently fall back
        d.denoise(torch.zeros(1, 1, 32, 32), inference_steps=2)


def test_attention_key_split_rule_never_leaves_an_empty_split():
    """the doubling rule produced splits that own no tile (N = 784 = 224x224 / 64: 25 tiles, 8 splits of
    4 tiles -> the launch returned an error).  Host-only sweep over every N and the per-program batches."""
    lib = native.lib()
    k, t = C.c_int(), C.c_int()
    for B in (1, 2, 3, 4, 8, 16, 32):
        for N in list(range(1, 1100)) + [24 * 28, 28 * 28, 36 * 36, 2048, 4095, 4096, 4097, 16384]:
            assert lib.mi_debug_attention_split(N, B, C.byref(k), C.byref(t)) == 0
            tiles = (N + 31) // 32
            assert 1 <= k.value <= 8 and t.value >= 1
            assert (k.value - 1) * t.value < tiles <= k.value * t.value, (N, B, k.value, t.value)
    assert lib.mi_debug_attention_split(0, 1, C.byref(k), C.byref(t)) != 0


def test_library_carries_the_hash_of_the_sources_it_was_built_from():
    """profiles are matched against the BUILT binary (mi_source_hash), not the working tree."""
    assert native.kernel_source_hash() == native.tree_source_hash(), "libmidd.so is older than csrc/: rebuild"
