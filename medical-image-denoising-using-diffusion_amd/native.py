"""ctypes binding of libmidd.so (C ABI: include/midd.h).

There is deliberately no fallback: if the shared library is missing or cannot be loaded,
``lib()`` raises and every compute entry point of the package fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIDD_LIBRARY: development only (A/B runs of two builds on one box); there is no other implementation behind it
LIB_PATH = os.environ.get("MIDD_LIBRARY") or os.path.join(_HERE, "libmidd.so")

MI_MAX_LEVELS = 8
MI_VARIANT = {"ddim": 0, "cddpm": 1}
MI_CLAMP_EPS = 1
MI_NO_SPLIT = 2
MI_COMPUTE = {"f32": 0, "f16x3": 1}
MI_STATUS_NONFINITE, MI_STATUS_FP16_RANGE = 1, 2
MI_COMPUTE_BATCH_INVARIANT = 0x100          # include/midd.h: OR into compute_mode


class NativeLibraryError(RuntimeError):
    pass


class MiddError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmidd error {code}: {message}")
        self.code = code


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("model_channels", C.c_int32), ("num_levels", C.c_int32),
                ("channel_mult", C.c_int32 * MI_MAX_LEVELS), ("num_res_blocks", C.c_int32),
                ("num_attention_levels", C.c_int32), ("attention_levels", C.c_int32 * MI_MAX_LEVELS),
                ("time_emb_dim", C.c_int32), ("variant", C.c_int32), ("compute_mode", C.c_int32)]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/midd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mi_unet_plan_create", C.c_int, [C.POINTER(UNetCfg), C.POINTER(C.c_void_p)]),
    ("mi_unet_load_weights", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    ("mi_unet_num_weights", C.c_int, [C.c_void_p]),
    ("mi_unet_weight_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("mi_unet_finalize", C.c_int, [C.c_void_p, C.c_int]),
    ("mi_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("mi_unet_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi_denoise", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                             C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                             C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi_debug_fetch", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    ("mi_status", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    ("mi_debug_attention_split", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("mi_debug_plan_dump", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    ("mi_debug_conv16_geometry", C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_int)] * 4),
    ("mi_source_hash", C.c_char_p, []),
    ("mi_profile_begin", C.c_int, [C.c_void_p]),
    ("mi_profile_end", C.c_int, [C.c_void_p, C.POINTER(ProfileEntry), C.c_int, C.POINTER(C.c_int)]),
    ("mi_resize_workspace_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("mi_resize_bicubic_u8", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi_u8_to_unit_f32", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi_unit_f32_to_u8", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi_metrics_workspace_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("mi_image_metrics", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi_plan_destroy", None, [C.c_void_p]),
    ("mi_last_error", C.c_char_p, []),
    ("mi_version", C.c_char_p, []),
]

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Loads libmidd.so once; raises NativeLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeLibraryError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C medical-image-denoising-using-diffusion_amd/csrc`). "
                    "This package has no CPU fallback.")
            try:
                handle = C.CDLL(LIB_PATH)
            except OSError as exc:
                raise NativeLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
            for name, restype, argtypes in SYMBOLS:
                fn = getattr(handle, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise MiddError(rc, lib().mi_last_error().decode("utf-8", "replace"))


def tree_source_hash() -> str:
    """sha256 over the csrc/ sources in the working tree (what the Makefile embeds at build time)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h*"))):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def kernel_source_hash() -> str:
    """Hash of the kernel sources the LOADED library was built from (embedded at build time, mi_source_hash): identifies
    the binary a profile (profiles/*_pmc_traffic.json) describes, whatever the working tree or MIDD_LIBRARY says."""
    return lib().mi_source_hash().decode()
