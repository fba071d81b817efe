"""In-kernel stamps of the wave-specialised 3x3 kernel (diagnostic build: make -C .../csrc timing): one line per layer shape.
MIDD_LIBRARY=$PWD/libmidd_timing.so python tools/ws_timing.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIDD_SPLIT", "1")
import torch
import midd_loader; midd_loader.load()
from midd_amd import UNetDiffusion, UNetConfig, native
from midd_amd.weights import make_state_dict, synthetic_xray
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = native.lib()
dump = lib.mi_debug_ws_timing_dump; dump.argtypes = [ctypes.c_char_p]; dump.restype = None
# single-layer probes: a UNet whose 3x3 convs all have one shape is not available, so time whole forwards per resolution
for size in (256, 128, 64, 32):
    m = UNetDiffusion(); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(UNetConfig(), 42).items()}); m = m.cuda().eval()
    x = torch.from_numpy(synthetic_xray(B, size, size)).cuda()
    t = torch.full((B,), 7, dtype=torch.long)
    m(x, x, t); torch.cuda.synchronize(); dump(b"warmup (discard)")
    for _ in range(3): m(x, x, t)
    dump(f"B={B} input {size}x{size} (all WS layers)".encode())
