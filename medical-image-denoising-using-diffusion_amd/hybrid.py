"""``HybridDenoisingRouter`` — the hybrid caller of the sampler path.

The reference's router (/root/reference/Backend/hybrid/hybrid3diffusionspeed.py:560-628) owns four
networks: a NAFNet, a copy of the diffusion UNet + sampler (:308-418), a ``NoiseAnalyzer`` router and a
``FusionModule``.  Only the diffusion branch is this repository's hot path; the other three are out of
scope and stay ordinary torch modules, so they are INJECTED here instead of being
rebuilt:

    router = HybridDenoisingRouter(nafnet=EnhancedNAFNet(...), router=NoiseAnalyzer(...),
                                   fusion=FusionModule(...), diffusion_params=ckpt['diffusion_params'],
                                   inference_diffusion_steps=7)
    router.load_state_dict(ckpt['model_state_dict'])       # run.py:69 — same keys: nafnet.* diffusion_unet.* router.* fusion.*

Kept from the reference:
  * attribute names ``nafnet / diffusion_unet / diffusion_wrapper / router / fusion`` and
    ``training_diffusion_steps / inference_diffusion_steps`` (run.py:71-72 overwrites both after loading),
    hence the state-dict layout (``diffusion_unet.<308 reference keys>``; the sampler is not a module, :581);
  * ``forward(noisy_input)`` (:608-628): both backends under no_grad, each followed by
    ``nan_to_num(nan=0, posinf=1, neginf=0)`` + ``clamp(0, 1)``, then router mask and fusion;
  * ``load_pretrained_models`` (:594-600) for the two backend checkpoints — loaded with
    ``weights_only=True`` (the reference passes ``False``; the files are plain state dicts);
  * ``freeze_backends`` (:602-606).
The hybrid file's UNet copy differs from DDIMModel.py only in its attention (no 512-query chunking, scale
applied after QK^T, :295-301) — the same mathematics, so the HIP ``UNetDiffusion`` serves it unchanged
(pinned by tests/golden/hybrid_ddim_64.npz, generated from the hybrid file's own classes).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .modules import UNetDiffusion
from .sampler import DiffusionDenoiser


def _sanitize(x: torch.Tensor) -> torch.Tensor:
    """hybrid3diffusionspeed.py:615-616, 619-620, 623-624."""
    return torch.clamp(torch.nan_to_num(x, nan=0.0, posinf=1.0, neginf=0.0), 0, 1)


class HybridDenoisingRouter(nn.Module):
    def __init__(self, nafnet: nn.Module, router: nn.Module, fusion: nn.Module, diffusion_params: Optional[dict] = None,
                 training_diffusion_steps: int = 10, inference_diffusion_steps: int = 10, compute: Optional[str] = None):
        super().__init__()
        p = diffusion_params or {}
        self.nafnet = nafnet
        self.diffusion_unet = UNetDiffusion(                                  # :572-579, same defaults
            in_channels=p.get("in_channels", 1), model_channels=p.get("model_channels", 48),
            channel_mult=tuple(p.get("channel_mult", (1, 2, 3, 4))), num_res_blocks=p.get("num_res_blocks", 2),
            attention_resolutions=tuple(p.get("attention_resolutions", (3,))), time_emb_dim=p.get("time_emb_dim", 192),
            compute=compute)
        self._noise_steps = p.get("noise_steps", 50)                          # :583
        self.diffusion_wrapper = DiffusionDenoiser(self.diffusion_unet, noise_steps=self._noise_steps)
        self.router = router
        self.fusion = fusion
        self.training_diffusion_steps = training_diffusion_steps
        self.inference_diffusion_steps = inference_diffusion_steps

    def _apply(self, fn, *args, **kwargs):
        # the sampler's schedule tables live outside the module tree (as in the reference, :581-584):
        # rebuild them on the parameters' device after .to(...) / .cuda()
        out = super()._apply(fn, *args, **kwargs)
        self.diffusion_wrapper = DiffusionDenoiser(self.diffusion_unet, noise_steps=self._noise_steps)
        return out

    def _backends(self):
        return (("nafnet", self.nafnet), ("diffusion_unet", self.diffusion_unet))

    def load_pretrained_models(self, nafnet_path: str, diffusion_path: str) -> None:
        """Same method surface as hybrid3diffusionspeed.py:592-598: each backend takes the `model_state_dict` of its own checkpoint
        file.  Restated, not transcribed: one loop over (module, path), tensors-only unpickling (the reference passes
        weights_only=False), strict key check by load_state_dict, no console output."""
        for (name, module), path in zip(self._backends(), (nafnet_path, diffusion_path)):
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
            if "model_state_dict" not in ckpt:
                raise KeyError(f"{path}: checkpoint for {name} has no 'model_state_dict' entry")
            module.load_state_dict(ckpt["model_state_dict"])

    def freeze_backends(self) -> None:
        """hybrid3diffusionspeed.py:600-606: both backends become inference-only (no gradients, eval mode); router and fusion stay trainable."""
        for _, module in self._backends():
            module.requires_grad_(False)
            module.eval()

    def hq_denoised(self, noisy_input: torch.Tensor, diffusion_steps: Optional[int] = None) -> torch.Tensor:
        """The diffusion branch alone (:618-620): HIP sampler, then nan_to_num + clamp."""
        if diffusion_steps is None:
            diffusion_steps = self.training_diffusion_steps if self.training else self.inference_diffusion_steps
        with torch.no_grad():
            return _sanitize(self.diffusion_wrapper.denoise(noisy_input, inference_steps=diffusion_steps))

    def forward(self, noisy_input: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            fast_denoised = _sanitize(self.nafnet(noisy_input))
        hq = self.hq_denoised(noisy_input)
        routing_mask = _sanitize(self.router(noisy_input))
        return self.fusion(fast_denoised, hq, routing_mask)
