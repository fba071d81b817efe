"""MI355X-native reverse-diffusion sampler (UNet noise predictor + sampler update).

Drop-in for the reference's ``UNetDiffusion`` / ``DiffusionDenoiser``
(/root/reference/Backend/DDIM/DDIMModel.py:169-289): same constructor arguments, same
``forward(x, condition, t)`` / ``denoise(noisy_img, inference_steps)`` signatures, same
state-dict key names; the arithmetic runs in hand-written HIP kernels for gfx950 behind
the C ABI declared in ``include/midd.h``.  There is no CPU fallback: without the built
``libmidd.so`` and a GPU every compute call raises.
"""
from .config import UNetConfig, topology, param_shapes, timestep_list  # noqa: F401
from .modules import UNetDiffusion  # noqa: F401
from .sampler import DiffusionDenoiser, device  # noqa: F401
