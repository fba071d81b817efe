"""``DiffusionDenoiser``: schedule + reverse loop, same surface as the reference class
(/root/reference/Backend/DDIM/DDIMModel.py:250-289; stochastic cddpm variant
/root/reference/Backend/cddpm/cddpmModels.py:263-308).

The schedule tables are built with the same torch calls as the reference so they are bit
identical; the loop itself (UNet forward + fused x_{t-1} update per timestep) is a single
call into libmidd.so.
"""
from __future__ import annotations

from typing import Optional

import torch

from .config import timestep_list

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")    # DDIMModel.py:10


class DiffusionDenoiser:
    def __init__(self, model, noise_steps=50, beta_start=1e-4, beta_end=0.02):
        self.model = model
        self.noise_steps = noise_steps
        dev = device
        try:
            dev = next(model.parameters()).device
        except (AttributeError, StopIteration):
            pass
        # DDIMModel.py:255-257 (linspace is evaluated on the CPU there too, then moved)
        self.beta = torch.linspace(beta_start, beta_end, noise_steps).to(dev)
        self.alpha = 1.0 - self.beta
        self.alpha_hat = torch.cumprod(self.alpha, dim=0)

    @torch.no_grad()
    def denoise(self, noisy_img: torch.Tensor, inference_steps: int = 25,
                step_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x = denoiser.denoise(noisy_img, inference_steps) — DDIMModel.py:268-289.

        Starts from the noisy image itself, conditions every step on it, never mutates it and
        returns a new tensor on the same device.  For ``model.variant == 'cddpm'`` the update
        adds ``sqrt(beta_t) * 0.5 * randn`` for t > 0 and does not clamp eps
        (cddpmModels.py:290-303); ``step_noise`` ([n_iters,B,C,H,W], already scaled by 0.5)
        overrides the on-device draw so a run can be replayed exactly.
        """
        self.model.eval()
        steps = timestep_list(self.noise_steps, inference_steps)
        stochastic = getattr(self.model, "variant", "ddim") == "cddpm"
        if stochastic and step_noise is None:
            step_noise = 0.5 * torch.randn((len(steps),) + tuple(noisy_img.shape), device=noisy_img.device)
        if not stochastic:
            step_noise = None
        return self.model.run_sampler(noisy_img, steps, self.beta, self.alpha, self.alpha_hat,
                                      clamp_eps=not stochastic, step_noise=step_noise)

    # north_star's wording for the same call
    ddim_sample = denoise
