"""Single-image latency (the serving regime, run.py:107): python tools/latency_b1.py  -- 256x256 x 50 iterations and the served
recipe 512x512, inference_steps = 8 (9 iterations)."""
import sys, time, torch
sys.path.insert(0, '.')
import midd_loader; midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig
from midd_amd.weights import make_state_dict, synthetic_xray
m = UNetDiffusion(); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(UNetConfig(), 42).items()}); m = m.cuda().eval()
d = DiffusionDenoiser(m)
for S, it in ((256, 50), (512, 8)):
    x = torch.from_numpy(synthetic_xray(1, S, S)).cuda()
    d.denoise(x, it); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): d.denoise(x, it)
    torch.cuda.synchronize()
    print(f"B=1 {S}^2 inference_steps={it}: {1e3*(time.perf_counter()-t)/5:.1f} ms", end="   ")
print()
