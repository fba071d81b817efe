"""In-kernel stamp shares of the f16x3 conv kernel (diagnostic build: make -C <pkg>/csrc timing).
python tools/conv_timing.py [B] [size]   -- prints, per launch shape, where wave 0 of a workgroup spends its cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import midd_loader; midd_loader.load()
import midd_amd.native as native
native.LIB_PATH = os.path.join(ROOT, "libmidd_timing.so")
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig
from midd_amd.weights import make_state_dict, synthetic_xray
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = UNetDiffusion(); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(UNetConfig(), 42).items()}); m = m.cuda().eval()
d = DiffusionDenoiser(m); x = torch.from_numpy(synthetic_xray(B, S, S)).cuda()
d.denoise(x, 5); torch.cuda.synchronize()
lib = ctypes.CDLL(native.LIB_PATH)
lib.mi_debug_conv_timing_dump()          # discard warm-up
d.denoise(x, 10); torch.cuda.synchronize()
lib.mi_debug_conv_timing_dump()
