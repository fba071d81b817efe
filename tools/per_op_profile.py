"""Per-op timing of one sampler iteration (MIDD_PROFILE_PER_OP=1): python tools/per_op_profile.py [B] [size]"""
import os, sys
os.environ["MIDD_PROFILE_PER_OP"] = "1"
os.environ.setdefault("MIDD_SPLIT", "1")      # one stream: spans are per-kernel times, not contended ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import midd_loader; midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig
from midd_amd.weights import make_state_dict, synthetic_xray
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = UNetDiffusion(); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(UNetConfig(), 42).items()}); m = m.cuda().eval()
d = DiffusionDenoiser(m); x = torch.from_numpy(synthetic_xray(B, S, S)).cuda()
d.denoise(x, 5); torch.cuda.synchronize()
m.profile_begin(); d.denoise(x, 10); prof = m.profile_end()
tot = sum(p["total_ms"] for p in prof)
print(f"total {tot/10:.3f} ms per iteration, {len(prof)} ops")
for p in prof:
    us = 1e3 * p["total_ms"] / p["launches"]
    tf = p["flops"] / (p["total_ms"] * 1e-3) / 1e12 if p["flops"] else 0
    gb = p["bytes"] / (p["total_ms"] * 1e-3) / 1e9
    print(f"{p['name'][:96]:96s} {us:8.1f} us {tf:7.1f} TF {gb:7.0f} GB/s")
