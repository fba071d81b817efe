"""Runs tools/mb/pk_opsel_probe.hip (a) alone and (b) while the library's forward runs on a second stream, and prints per lane
quarter how many v_pk_fma_f32 results differed from the scalar v_fma_f32 chain, for the two operand-selection forms.
Build first:  hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/mb/pk_opsel_probe.hip -o tools/mb/libpk_opsel_probe.so"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import midd_loader
midd_loader.load()
from midd_amd import UNetDiffusion, UNetConfig
from midd_amd.weights import make_state_dict, synthetic_xray

probe = C.CDLL(os.path.join(ROOT, "tools", "mb", "libpk_opsel_probe.so"))
probe.pk_opsel_probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]

KW = dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=2, attention_resolutions=(1,), time_emb_dim=32)


def report(tag, counters, launches):
    c = counters.cpu().numpy().reshape(64, 4).astype(np.int64)
    q = c.reshape(4, 16, 4).sum(1)
    print(f"{tag}: {launches} probe launches; mismatching results per lane quarter [0-15, 16-31, 32-47, 48-63]:")
    for k, name in enumerate(["op_sel:[0,1,0]   low result ", "op_sel:[0,1,0]   high result", "op_sel_hi:[1,0,1] low result ", "op_sel_hi:[1,0,1] high result"]):
        print(f"    {name}: {q[:, k].tolist()}")


def main():
    rng = np.random.default_rng(5)
    w = torch.from_numpy(rng.standard_normal(18 * 32).astype(np.float32)).cuda()
    n = 1 << 16
    x = torch.from_numpy(rng.random(n, dtype=np.float32)).cuda()
    blocks, rounds, launches = 768, 8, int(os.environ.get("PROBE_LAUNCHES", "400"))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run_probe(k):
        with torch.cuda.stream(s1):
            for _ in range(k):
                rc = probe.pk_opsel_probe_launch(w.data_ptr(), x.data_ptr(), counters.data_ptr(), blocks, rounds, n, s1.cuda_stream)
                assert rc == 0, rc

    counters = torch.zeros(256, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    run_probe(launches)
    torch.cuda.synchronize()
    report("alone", counters, launches)

    cfg = UNetConfig(**KW)
    sd = make_state_dict(cfg, seed=77)
    m = UNetDiffusion(compute="f16x3", **KW)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    m.check_status = False
    B, H, W = 2, 104, 96
    xi = torch.from_numpy(synthetic_xray(B, H, W, seed=3, kind="uniform")).cuda()
    ci = torch.from_numpy(synthetic_xray(B, H, W, seed=504)).cuda()
    t = torch.tensor([25] * B)
    m(xi, ci, t)
    counters.zero_()
    torch.cuda.synchronize()
    done = 0
    while done < launches:
        with torch.cuda.stream(s2):
            for _ in range(4):
                m(xi, ci, t)
        run_probe(20)
        done += 20
    torch.cuda.synchronize()
    report("beside the library's forward on a second stream", counters, done)

    # which kind of neighbour does it take?  torch kernels on the second stream instead of the library's
    a16 = torch.randn(2048, 2048, device="cuda", dtype=torch.float16)
    a32 = torch.randn(2048, 2048, device="cuda", dtype=torch.float32)
    big = torch.randn(1 << 24, device="cuda")
    neighbours = {
        "torch fp16 matmul (MFMA)": lambda: torch.matmul(a16, a16),
        "torch fp32 matmul": lambda: torch.matmul(a32, a32),
        "torch elementwise (exp, no MFMA, no LDS)": lambda: torch.exp(big),
        "torch sort (LDS-heavy, no MFMA)": lambda: torch.sort(big[: 1 << 20]),
    }
    for name, fn in neighbours.items():
        fn()
        counters.zero_()
        torch.cuda.synchronize()
        done = 0
        while done < launches:
            with torch.cuda.stream(s2):
                for _ in range(6):
                    fn()
            run_probe(20)
            done += 20
        torch.cuda.synchronize()
        report("beside " + name, counters, done)


if __name__ == "__main__":
    main()
