"""Probe: does running two half-batches on two streams (two host threads) beat one full batch?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import midd_loader; midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig
from midd_amd.weights import make_state_dict, synthetic_xray
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(UNetConfig(), 42).items()}
models = []
for i in range(nsplit):
    m = UNetDiffusion(); m.load_state_dict(sd); models.append(m.cuda().eval())
dens = [DiffusionDenoiser(m) for m in models]
x = torch.from_numpy(synthetic_xray(B, 256, 256)).cuda()
parts = list(x.chunk(nsplit))
streams = [torch.cuda.Stream() for _ in range(nsplit)]
outs = [None] * nsplit
def work(i):
    with torch.cuda.stream(streams[i]):
        outs[i] = dens[i].denoise(parts[i], inference_steps=50)
def run():
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nsplit)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
run()
t0 = time.perf_counter(); run(); run(); dt = (time.perf_counter() - t0) / 2
print(f"B={B} split into {nsplit} streams: {dt*1e3:.1f} ms per step -> {B/dt:.2f} img/s")
ref = dens[0].denoise(x, inference_steps=50); torch.cuda.synchronize()
t0 = time.perf_counter(); ref = dens[0].denoise(x, inference_steps=50); torch.cuda.synchronize(); dt1 = time.perf_counter() - t0
print(f"single stream: {dt1*1e3:.1f} ms -> {B/dt1:.2f} img/s; max diff split vs full {float((torch.cat(outs)-ref).abs().max()):.2e}")
