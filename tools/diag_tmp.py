import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import midd_loader; midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig, topology
from midd_amd.weights import make_state_dict, synthetic_xray
from oracle import ddim_oracle as orc
SMALL = dict(model_channels=16, time_emb_dim=64)
cfg = UNetConfig(**SMALL)
sd = make_state_dict(cfg, seed=42, perturb_norm=True)
m = UNetDiffusion(**SMALL); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m = m.cuda().eval()
m.check_status = False
x = torch.from_numpy(synthetic_xray(2, 32, 32, seed=1, kind="uniform")); c = torch.from_numpy(synthetic_xray(2, 32, 32, seed=2)); t = torch.tensor([3, 40])
cap = {}
with torch.no_grad():
    ref = orc.unet_forward(orc.to_torch(sd), topology(cfg), x, c, t, trace=lambda n, v: cap.__setitem__(n, v.numpy().copy()))
eps = m(x.cuda(), c.cuda(), t.cuda()); torch.cuda.synchronize()
for name, want in cap.items():
    if name in ("time_mlp", "out_conv"): continue
    try: got = m.debug_fetch(name, 2, 32, 32).cpu().numpy()
    except Exception as e: print(name, "n/a"); continue
    print(f"{name:12s} max|d| {np.abs(got - want).max():.3e} finite {np.isfinite(got).all()} max {np.abs(want).max():.3f}")
print("eps", float((eps.cpu() - ref).abs().max()))
# split run
full = UNetDiffusion(); full.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(UNetConfig(), 42).items()}); full = full.cuda().eval()
den = DiffusionDenoiser(full)
xx = torch.from_numpy(synthetic_xray(8, 256, 256)).cuda()
try:
    out = den.denoise(xx, 3); torch.cuda.synchronize(); print("split ok", float(out.mean()))
except Exception as e: print("split run failed:", e)
