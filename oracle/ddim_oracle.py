"""TEST INFRASTRUCTURE ONLY — CPU oracle for the reverse-diffusion sampler path.

This is a fresh CPU (PyTorch fp32, functional) restatement of the reference algorithm:
``DiffusionDenoiser.denoise`` -> ``UNetDiffusion.forward``
(/root/reference/Backend/DDIM/DDIMModel.py:94-289, cddpm variant
/root/reference/Backend/cddpm/cddpmModels.py:176-308).  It exists so the HIP path can be
checked on the GPU box, where /root/reference does not exist.

Rules (task statement, section 3):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import this file, and only as the checker / reported baseline;
  * the product package never imports it and has no CPU fallback.

Pinning: ``tests/test_oracle_vs_reference.py`` (runs only where /root/reference exists)
asserts this restatement equals the imported reference per block and end to end, and
``tests/golden/make_golden.py`` stores reference outputs as fixtures that
``tests/test_oracle_golden.py`` re-checks everywhere (the reference has no tests or
golden vectors of its own — SURVEY.md section 4).

The state dict is a plain ``{name: torch.Tensor}`` with the reference's key names.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- schedule
def schedule(noise_steps: int = 50, beta_start: float = 1e-4, beta_end: float = 0.02):
    """beta / alpha / alpha_hat tables — DDIMModel.py:255-257."""
    beta = torch.linspace(beta_start, beta_end, noise_steps)
    alpha = 1.0 - beta
    alpha_hat = torch.cumprod(alpha, dim=0)
    return beta, alpha, alpha_hat


def timestep_list(noise_steps: int, inference_steps: int) -> List[int]:
    """DDIMModel.py:272-274."""
    step = max(1, noise_steps // inference_steps)
    return list(reversed(range(0, noise_steps, step)))


# --------------------------------------------------------------------------- blocks
def sinusoidal(t: torch.Tensor, dim: int) -> torch.Tensor:
    """SinusoidalPositionEmbeddings.forward — DDIMModel.py:99-106."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -k)
    arg = t[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_embedding(sd: SD, t: torch.Tensor, model_channels: int) -> torch.Tensor:
    """UNetDiffusion.time_mlp — DDIMModel.py:173-178."""
    e = sinusoidal(t, model_channels)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = F.silu(e)
    return F.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def residual_block(sd: SD, p: str, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
    """ResidualBlock.forward — DDIMModel.py:128-133 (Dropout(0.0) in eval is identity)."""
    h = F.group_norm(x, 8, sd[f"{p}.block1.0.weight"], sd[f"{p}.block1.0.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.block1.2.weight"], sd[f"{p}.block1.2.bias"], padding=1)
    te = F.linear(F.silu(temb), sd[f"{p}.time_mlp.1.weight"], sd[f"{p}.time_mlp.1.bias"])
    h = h + te[:, :, None, None]
    h = F.group_norm(h, 8, sd[f"{p}.block2.0.weight"], sd[f"{p}.block2.0.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.block2.3.weight"], sd[f"{p}.block2.3.bias"], padding=1)
    if f"{p}.res_conv.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.res_conv.weight"], sd[f"{p}.res_conv.bias"])
    return h + x


def attention_block(sd: SD, p: str, x: torch.Tensor, heads: int = 2) -> torch.Tensor:
    """AttentionBlock.forward — DDIMModel.py:143-166.

    The reference walks 512-query chunks with a full softmax over all keys per chunk, which
    is exact; the oracle therefore computes all queries at once.
    """
    b, c, h, w = x.shape
    xn = F.group_norm(x, 8, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], eps=1e-5)
    qkv = F.conv2d(xn, sd[f"{p}.qkv.weight"], sd[f"{p}.qkv.bias"]).reshape(b, 3, heads, c // heads, h * w)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    q = q * (c // heads) ** -0.5
    att = torch.softmax(torch.matmul(q.transpose(-2, -1), k), dim=-1)       # [b,heads,Nq,Nk]
    out = torch.matmul(att, v.transpose(-2, -1)).transpose(-2, -1)          # [b,heads,d,Nq]
    out = out.reshape(b, c, h, w)
    out = F.conv2d(out, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])
    return out + x


# --------------------------------------------------------------------------- network
def unet_forward(sd: SD, topo, x: torch.Tensor, condition: torch.Tensor, t: torch.Tensor,
                 trace: Optional[Callable[[str, torch.Tensor], None]] = None) -> torch.Tensor:
    """UNetDiffusion.forward — DDIMModel.py:219-248.  ``topo`` = package config.topology(cfg).

    ``trace(name, tensor)`` is called with every module output (tests use it for per-layer
    parity of the HIP path).
    """
    def tr(name, v):
        if trace is not None:
            trace(name, v)
        return v

    temb = tr("time_mlp", time_embedding(sd, t, topo.cfg.model_channels))
    h = torch.cat([x, condition], dim=1)
    h = tr("in_conv", F.conv2d(h, sd["in_conv.weight"], sd["in_conv.bias"], padding=1))
    skips = []

    def run(m, h):
        if m.kind == "rb":
            return residual_block(sd, m.name, h, temb)
        if m.kind == "attn":
            return attention_block(sd, m.name, h)
        if m.kind == "down":
            return F.conv2d(h, sd[f"{m.name}.weight"], sd[f"{m.name}.bias"], stride=2, padding=1)
        if m.kind == "up":
            return F.conv_transpose2d(h, sd[f"{m.name}.weight"], sd[f"{m.name}.bias"], stride=2, padding=1)
        raise ValueError(m.kind)

    for m in topo.downs:
        h = tr(m.name, run(m, h))
        skips.append(h)                              # every down module pushes (DDIMModel.py:232)
    for m in topo.mid:
        h = tr(m.name, run(m, h))
    for m in topo.ups:
        if m.kind == "rb":
            skip = skips.pop()                       # only residual blocks pop (DDIMModel.py:240)
            if h.shape[2:] != skip.shape[2:]:
                h = F.interpolate(h, size=skip.shape[2:], mode="bilinear", align_corners=False)
            h = torch.cat([h, skip], dim=1)
        h = tr(m.name, run(m, h))
    h = F.group_norm(h, 8, sd["out_conv.0.weight"], sd["out_conv.0.bias"], eps=1e-5)
    return tr("out_conv", F.conv2d(F.silu(h), sd["out_conv.2.weight"], sd["out_conv.2.bias"], padding=1))


# --------------------------------------------------------------------------- sampler
@torch.no_grad()
def denoise(sd: SD, topo, noisy: torch.Tensor, noise_steps: int = 50, inference_steps: int = 25,
            beta_start: float = 1e-4, beta_end: float = 0.02,
            step_noise: Optional[List[torch.Tensor]] = None,
            on_step: Optional[Callable[[int, torch.Tensor, torch.Tensor], None]] = None) -> torch.Tensor:
    """DiffusionDenoiser.denoise — DDIMModel.py:268-289 (deterministic, eps clamped to +-5).

    With ``topo.cfg.variant == 'cddpm'`` the update of cddpmModels.py:281-308 is used instead:
    no eps clamp and ``+ sqrt(beta_t) * noise`` where ``noise = 0.5*randn`` for i > 0; the
    already-scaled noise tensors are passed in as ``step_noise`` (one per iteration, the
    last may be None) because RNG streams are not portable.
    ``on_step(i, eps, x_next)`` observes every iteration.
    """
    beta, alpha, alpha_hat = schedule(noise_steps, beta_start, beta_end)
    cddpm = topo.cfg.variant == "cddpm"
    x = noisy.clone()
    for n, i in enumerate(timestep_list(noise_steps, inference_steps)):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps = unet_forward(sd, topo, x, noisy, t)
        if not cddpm:
            eps = torch.clamp(eps, -5, 5)
        a = alpha[t][:, None, None, None]
        ah = alpha_hat[t][:, None, None, None]
        x = (1 / torch.sqrt(a)) * (x - ((1 - a) / torch.sqrt(1 - ah)) * eps)
        if cddpm and i > 0 and step_noise is not None and step_noise[n] is not None:
            x = x + torch.sqrt(beta[t][:, None, None, None]) * step_noise[n]
        x = torch.clamp(x, 0, 1)
        if on_step is not None:
            on_step(i, eps, x)
    return x


def to_torch(sd_np) -> SD:
    """numpy state dict (package weights.make_state_dict) -> torch CPU tensors."""
    return {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
