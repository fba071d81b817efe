"""``UNetDiffusion``: the reference's noise-prediction network as a parameter container whose
forward pass runs entirely in libmidd.so.

Interface kept from /root/reference/Backend/DDIM/DDIMModel.py:169-248:
  * same constructor arguments and defaults (:169-170),
  * ``forward(x, condition, t) -> eps`` (:219),
  * an ``nn.Module`` whose ``state_dict()`` has the reference's 308 key names and shapes, so
    ``model.load_state_dict(ckpt['model_state_dict'])`` (run.py:37-39) works unchanged.
``variant='cddpm'`` selects the module lists of /root/reference/Backend/cddpm/cddpmModels.py:176-232.

The nn.Module holds only parameters; there are no torch ops on the compute path and no CPU
fallback (a CPU tensor raises).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import native
from .config import UNetConfig, param_shapes, topology

DEFAULT_TIME_ROWS = 1000     # rows of the precomputed timestep table (t in [0, rows))


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    *path, leaf = dotted.split(".")
    mod = root
    for part in path:
        if part not in mod._modules:
            mod.add_module(part, _Node())
        mod = mod._modules[part]
    mod.register_parameter(leaf, param)


def _default_init(name: str, shape: Tuple[int, ...], all_shapes: Dict[str, Tuple[int, ...]]) -> torch.Tensor:
    """PyTorch's default initialisation for the layer types the reference instantiates
    (Conv2d / ConvTranspose2d / Linear: U(+-1/sqrt(fan_in)); GroupNorm: ones / zeros)."""
    norm = (".block1.0." in name or ".block2.0." in name or ".norm." in name or name.startswith("out_conv.0."))
    if norm:
        return torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
    wshape = shape if name.endswith("weight") else all_shapes[name[:-4] + "weight"]
    fan_in = wshape[1] * (wshape[2] * wshape[3] if len(wshape) == 4 else 1)
    bound = 1.0 / math.sqrt(fan_in)
    return torch.empty(shape).uniform_(-bound, bound)


class UNetDiffusion(nn.Module):
    def __init__(self, in_channels=1, model_channels=48, channel_mult=(1, 2, 3, 4), num_res_blocks=2,
                 attention_resolutions=(3,), dropout=0.0, time_emb_dim=192, variant="ddim", compute=None,
                 batch_invariant=None):
        super().__init__()
        # batch_invariant (not a reference argument; env MIDD_BATCH_INVARIANT=1): a sample's result does not depend on the
        # batch it is computed in, bit for bit (denoise(x[:k]) == denoise(x)[:k]) -- every launch is planned as for a batch of
        # one, which costs throughput at large batches.  Default off: results are then reproducible per (batch size, image size).
        self.batch_invariant = bool(int(os.environ.get("MIDD_BATCH_INVARIANT", "0"))) if batch_invariant is None else bool(batch_invariant)
        # arithmetic of the MFMA contractions: "f16x3" (split-fp16, default) or "f32" (fp32-input MFMA)
        self.compute = compute or os.environ.get("MIDD_COMPUTE", "f16x3")
        if self.compute not in native.MI_COMPUTE:
            raise ValueError(f"compute must be one of {sorted(native.MI_COMPUTE)}")
        self.cfg = UNetConfig(in_channels, model_channels, tuple(channel_mult), num_res_blocks,
                              tuple(attention_resolutions), dropout, time_emb_dim, variant)
        self.topology = topology(self.cfg)
        shapes = param_shapes(self.cfg)
        lookup = dict(shapes)
        for name, shape in shapes:
            _attach(self, name, nn.Parameter(_default_init(name, shape, lookup)))
        self._names = [n for n, _ in shapes]
        # native state (not part of the state dict)
        self._plan: Optional[int] = None
        self._stamp = None
        self._time_rows = DEFAULT_TIME_ROWS
        self._lock = threading.RLock()
        self._workspaces: Dict[Tuple[int, int, int, int, int], torch.Tensor] = {}
        # After every native call the status word of its workspace is read back (mi_status: one 4-byte copy, synchronises the
        # stream): NaN / Inf activations or an operand beyond the split-fp16 range raise MiddError instead of returning garbage.
        # Set to False (env MIDD_CHECK_STATUS=0) to keep forward() / denoise() asynchronous; the output is NaN then, as torch's.
        self.check_status = bool(int(os.environ.get("MIDD_CHECK_STATUS", "1")))
        # Debug / test knob (env MIDD_POISON_WS = a byte value 0..255, e.g. 255: every float reads as NaN, every statistics limb
        # as -1): the workspace is filled with that byte before EVERY native call, so a kernel that reads scratch the call did
        # not write first shows up as NaN / MiddError / a different answer instead of depending on what an earlier call with
        # another layout left behind (the workspace is torch.empty and reused across programs of different layouts).
        env_poison = os.environ.get("MIDD_POISON_WS")
        self.poison_workspace: Optional[int] = int(env_poison) & 255 if env_poison not in (None, "") else None

    # ------------------------------------------------------------------ native plumbing
    @property
    def variant(self) -> str:
        return self.cfg.variant

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _param_stamp(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _ensure_plan(self, time_rows: Optional[int] = None) -> int:
        """Creates the native plan and (re)uploads weights when parameters changed."""
        lib = native.lib()
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("UNetDiffusion runs only on a ROCm GPU: move the model with .to('cuda') "
                               "(there is no CPU fallback)")
        if time_rows is not None and time_rows > self._time_rows:
            self._time_rows = int(time_rows)
            self._stamp = None
        stamp = (self._param_stamp(), dev.index, self._time_rows)
        if self._plan is not None and stamp == self._stamp:
            return self._plan
        if self._plan is None:
            cfg = native.UNetCfg()
            c = self.cfg
            cfg.in_channels, cfg.model_channels, cfg.num_levels = c.in_channels, c.model_channels, len(c.channel_mult)
            for i, m in enumerate(c.channel_mult):
                cfg.channel_mult[i] = m
            cfg.num_res_blocks = c.num_res_blocks
            cfg.num_attention_levels = len(c.attention_resolutions)
            for i, a in enumerate(c.attention_resolutions):
                cfg.attention_levels[i] = a
            cfg.time_emb_dim, cfg.variant = c.time_emb_dim, native.MI_VARIANT[c.variant]
            cfg.compute_mode = native.MI_COMPUTE[self.compute] | (native.MI_COMPUTE_BATCH_INVARIANT if self.batch_invariant else 0)
            handle = C.c_void_p()
            native.check(lib.mi_unet_plan_create(C.byref(cfg), C.byref(handle)))
            self._plan = handle.value
            n = lib.mi_unet_num_weights(self._plan)
            theirs = [lib.mi_unet_weight_name(self._plan, i).decode() for i in range(n)]
            if theirs != self._names:
                raise RuntimeError("native plan and Python container disagree on the state-dict layout")
        sd = self.state_dict()
        for name in self._names:
            host = sd[name].detach().to("cpu", torch.float32).contiguous().numpy()
            shape = (C.c_int64 * host.ndim)(*host.shape)
            native.check(lib.mi_unet_load_weights(self._plan, name.encode(), host.ctypes.data_as(C.c_void_p),
                                                  shape, host.ndim))
        with torch.cuda.device(dev):
            native.check(lib.mi_unet_finalize(self._plan, self._time_rows))
        self._stamp = stamp
        self._workspaces.clear()
        return self._plan

    MAX_WORKSPACES = 4       # resident (shape, stream) workspaces; least recently used is dropped first

    def _workspace(self, B: int, H: int, W: int, dev: torch.device) -> torch.Tensor:
        """Scratch for one call.  Keyed by the CURRENT STREAM as well as the shape: the library call only
        enqueues work, so two threads that run the same model on different streams (run.py:85-91 runs the
        models of one request concurrently) must not share activations; calls on one stream are ordered by
        the stream itself.  Allocated while that stream is current, so the caching allocator ties the block
        to it."""
        key = (B, H, W, dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspaces.pop(key, None)
        if ws is None:
            nbytes = native.lib().mi_workspace_bytes(self._plan, B, H, W)
            if nbytes == 0:
                native.check(-1)
            while len(self._workspaces) >= self.MAX_WORKSPACES:
                self._workspaces.pop(next(iter(self._workspaces)))
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        self._workspaces[key] = ws                # most recently used last
        return ws

    @staticmethod
    def _aligned_ptr(ws: torch.Tensor) -> Tuple[int, int]:
        ptr = ws.data_ptr()
        aligned = (ptr + 255) & ~255
        return aligned, ws.numel() - (aligned - ptr)

    def _check_image(self, x: torch.Tensor, what: str) -> None:
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != self.cfg.in_channels:
            raise ValueError(f"{what} must be a [B,{self.cfg.in_channels},H,W] tensor")
        if x.device.type != "cuda":
            raise RuntimeError(f"{what} is on {x.device}: the MI355X path has no CPU fallback")
        if x.dtype != torch.float32:
            raise TypeError(f"{what} must be float32 (got {x.dtype})")

    def _raise_on_status(self, wptr: int, stream: int) -> None:
        if self.check_status:
            flags = C.c_int()
            native.check(native.lib().mi_status(wptr, stream, C.byref(flags)))

    # ------------------------------------------------------------------ reference interface
    @torch.no_grad()
    def forward(self, x: torch.Tensor, condition: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """eps = model(x, condition, t) — DDIMModel.py:219-248."""
        self._check_image(x, "x")
        self._check_image(condition, "condition")
        if condition.shape != x.shape or condition.device != x.device:
            raise ValueError("condition must match x in shape and device")
        B, _, H, W = x.shape
        tt = torch.as_tensor(t).reshape(-1).to("cpu", torch.int64)
        if tt.numel() != B:
            raise ValueError(f"t must have {B} entries")
        t_host = np.ascontiguousarray(tt.numpy().astype(np.int32))
        with self._lock, torch.cuda.device(x.device):
            plan = self._ensure_plan(time_rows=int(t_host.max()) + 1 if B else None)
            xc, cc = x.contiguous(), condition.contiguous()
            eps = torch.empty_like(xc)
            ws = self._workspace(B, H, W, x.device)
            if self.poison_workspace is not None:
                ws.fill_(self.poison_workspace)
            wptr, wbytes = self._aligned_ptr(ws)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            native.check(native.lib().mi_unet_forward(
                plan, xc.data_ptr(), cc.data_ptr(), t_host.ctypes.data_as(C.POINTER(C.c_int32)), eps.data_ptr(),
                B, H, W, wptr, wbytes, stream))
            self._raise_on_status(wptr, stream)
        return eps

    @torch.no_grad()
    def run_sampler(self, noisy: torch.Tensor, t_list, beta: torch.Tensor, alpha: torch.Tensor,
                    alpha_hat: torch.Tensor, clamp_eps: bool, step_noise: Optional[torch.Tensor] = None,
                    no_split: bool = False) -> torch.Tensor:
        """The whole reverse loop in one native call (used by DiffusionDenoiser.denoise)."""
        self._check_image(noisy, "noisy_img")
        B, _, H, W = noisy.shape
        steps = np.ascontiguousarray(np.asarray(list(t_list), dtype=np.int32))
        tabs = [np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy()) for v in (beta, alpha, alpha_hat)]
        noise_steps = int(tabs[0].shape[0])
        with self._lock, torch.cuda.device(noisy.device):
            plan = self._ensure_plan(time_rows=noise_steps)
            src = noisy.contiguous()
            out = torch.empty_like(src)
            nptr = None
            if step_noise is not None:
                if step_noise.shape != (len(steps),) + tuple(src.shape) or step_noise.device != src.device:
                    raise ValueError("step_noise must be [n_iters,B,C,H,W] on the image's device")
                step_noise = step_noise.to(torch.float32).contiguous()
                nptr = step_noise.data_ptr()
            ws = self._workspace(B, H, W, noisy.device)
            if self.poison_workspace is not None:
                ws.fill_(self.poison_workspace)
            wptr, wbytes = self._aligned_ptr(ws)
            stream = torch.cuda.current_stream(noisy.device).cuda_stream
            fp = C.POINTER(C.c_float)
            native.check(native.lib().mi_denoise(
                plan, src.data_ptr(), out.data_ptr(), B, H, W,
                steps.ctypes.data_as(C.POINTER(C.c_int32)), len(steps),
                tabs[0].ctypes.data_as(fp), tabs[1].ctypes.data_as(fp), tabs[2].ctypes.data_as(fp), noise_steps,
                nptr, (native.MI_CLAMP_EPS if clamp_eps else 0) | (native.MI_NO_SPLIT if no_split else 0), wptr, wbytes, stream))
            self._raise_on_status(wptr, stream)
        return out

    @torch.no_grad()
    def debug_fetch(self, module_name: str, B: int, H: int, W: int) -> torch.Tensor:
        """Output of a top-level module from the last forward at this shape, as NCHW (tests)."""
        lib = native.lib()
        dev = self._device()
        with self._lock, torch.cuda.device(dev):
            c, h, w = C.c_int(), C.c_int(), C.c_int()
            native.check(lib.mi_debug_fetch(self._plan, module_name.encode(), B, H, W, None, None,
                                            C.byref(c), C.byref(h), C.byref(w), None))
            out = torch.empty(B, c.value, h.value, w.value, device=dev)
            ws = self._workspace(B, H, W, dev)
            wptr, _ = self._aligned_ptr(ws)
            native.check(lib.mi_debug_fetch(self._plan, module_name.encode(), B, H, W, wptr, out.data_ptr(),
                                            C.byref(c), C.byref(h), C.byref(w),
                                            torch.cuda.current_stream(dev).cuda_stream))
        return out

    def profile_begin(self) -> None:
        """Bracket every kernel of subsequent calls with HIP events (bench.py roofline leg)."""
        with self._lock:
            self._ensure_plan()
            native.check(native.lib().mi_profile_begin(self._plan))

    def profile_end(self):
        """-> list of dicts {name, launches, total_ms, flops, bytes}, one per kernel symbol."""
        with self._lock:
            cap = 512
            buf = (native.ProfileEntry * cap)()
            n = C.c_int()
            native.check(native.lib().mi_profile_end(self._plan, buf, cap, C.byref(n)))
            return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                         flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(min(n.value, cap))]

    def workspace_bytes(self, B: int, H: int, W: int) -> int:
        with self._lock, torch.cuda.device(self._device()):
            self._ensure_plan()
            return int(native.lib().mi_workspace_bytes(self._plan, B, H, W))

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan is not None:
            try:
                native.lib().mi_plan_destroy(plan)
            except Exception:
                pass
