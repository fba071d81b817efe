"""Single-image inference harness — the MI355X counterpart of
``denoise_image_diffusion`` (/root/reference/Backend/cddpm/cddpmModels.py:470-504, DDIM twin
/root/reference/Backend/DDIM/DDIMModel.py:470-504) and of the script that calls it
(/root/reference/Backend/cddpminference.py:13-18): load checkpoint -> grayscale -> bicubic resize to
img_size -> sampler -> uint8 -> bicubic resize back -> PNG, with the reference's timing print.

    python -m midd_amd.cli --image in.png --out out.png [--checkpoint ckpt.pth] [--variant cddpm|ddim]
                           [--img-size 512] [--inference-steps 25]

Without a checkpoint (the trained weights are not distributed with the reference) the network is
random-init, which exercises the path but does not denoise.  The reference helper has a latent
bug — it builds the sampler on a module-global device instead of ``device_type``
(cddpmModels.py:472 vs :268) — which is not reproduced: everything runs on the model's device.
"""
from __future__ import annotations

import argparse
import time
from typing import Optional

import numpy as np
import torch
from PIL import Image

from .modules import UNetDiffusion
from .sampler import DiffusionDenoiser


def denoise_image_diffusion(model_path: Optional[str], test_image_path: str, device_type: str = "cuda",
                            img_size: int = 512, inference_steps: int = 50, variant: str = "cddpm",
                            step_noise: Optional[torch.Tensor] = None) -> Image.Image:
    device = torch.device(device_type)
    model = UNetDiffusion(in_channels=1, model_channels=48, channel_mult=(1, 2, 3, 4), num_res_blocks=2,
                          attention_resolutions=(3,), dropout=0.0, time_emb_dim=192, variant=variant)
    checkpoint = {}
    if model_path:
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
        model.load_state_dict(checkpoint["model_state_dict"])
    model = model.to(device).eval()
    diffusion = DiffusionDenoiser(model, noise_steps=int(checkpoint.get("noise_steps", 50)))
    print(f"Loaded model - PSNR: {checkpoint.get('best_psnr', 'N/A')} dB | SSIM: {checkpoint.get('best_ssim', 'N/A')}")

    img = Image.open(test_image_path).convert("L")
    on_gpu = device.type == "cuda"
    if on_gpu:      # resize + ToTensor scaling on the device (prepost: bit-identical to the PIL / numpy recipe below)
        from . import prepost
        raw = torch.from_numpy(np.asarray(img, np.uint8).copy()).to(device)
        input_tensor = prepost.to_unit_float(prepost.resize_bicubic_u8(raw, (img_size, img_size)))[None, None]
    else:
        resized = img.resize((img_size, img_size), Image.BICUBIC)       # transforms.Resize on a PIL image
        input_tensor = torch.from_numpy(np.asarray(resized, np.uint8).astype(np.float32) / 255.0)[None, None].to(device)

    start_time = time.time()
    kw = {"step_noise": step_noise} if step_noise is not None else {}
    denoised = diffusion.denoise(input_tensor, inference_steps=inference_steps, **kw)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    print(f"Inference time: {time.time() - start_time:.2f} seconds")

    if on_gpu:
        u8 = prepost.to_u8(denoised.reshape(img_size, img_size).float())     # denoise() already clamped to [0, 1]
        return Image.fromarray(prepost.resize_bicubic_u8(u8, (img.size[1], img.size[0])).cpu().numpy(), mode="L")
    output_np = denoised.squeeze().cpu().numpy()
    output_img = Image.fromarray((output_np * 255).astype(np.uint8), mode="L")
    return output_img.resize(img.size, Image.BICUBIC)


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--image", required=True)
    ap.add_argument("--out", default="denoised_diffusion_result.png")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--variant", default="cddpm", choices=["cddpm", "ddim"])
    ap.add_argument("--img-size", type=int, default=512)
    ap.add_argument("--inference-steps", type=int, default=25)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    restored = denoise_image_diffusion(args.checkpoint, args.image, device_type=args.device, img_size=args.img_size,
                                       inference_steps=args.inference_steps, variant=args.variant)
    restored.save(args.out, quality=95)
    print(f"\nResult saved: {args.out}")


if __name__ == "__main__":
    main()
