"""Hazard map of the packed-fp32 VALU ops on this MI355X: for every (op, op_sel, op_sel_hi) combination of v_pk_fma_f32 /
v_pk_mul_f32 / v_pk_add_f32 on VGPR pairs, how many results differ from the scalar arithmetic on the selected halves -- alone, and
while a torch fp16 matmul (MFMA) runs on a second stream.  Build: python tools/mb/gen_pk_opsel_map.py && hipcc -O3 --offload-arch=gfx950
-shared -fPIC tools/mb/pk_opsel_map.hip -o tools/mb/libpk_opsel_map.so"""
import ctypes as C
import os

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
lib = C.CDLL(os.path.join(ROOT, "tools", "mb", "libpk_opsel_map.so"))
lib.pk_map_name.restype = C.c_char_p
lib.pk_map_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]


def main():
    n = 1 << 16
    x = torch.from_numpy(np.random.default_rng(3).random(n, dtype=np.float32) + 0.25).cuda()
    a16 = torch.randn(2048, 2048, device="cuda", dtype=torch.float16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    blocks, rounds, launches = 768, 16, 60
    results = {}
    for mode in ("alone", "beside fp16 MFMA"):
        for i in range(lib.pk_map_count()):
            name = lib.pk_map_name(i).decode()
            counters = torch.zeros(128, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            done = 0
            while done < launches:
                if mode != "alone":
                    with torch.cuda.stream(s2):
                        for _ in range(3):
                            torch.matmul(a16, a16)
                with torch.cuda.stream(s1):
                    for _ in range(10):
                        assert lib.pk_map_launch(i, x.data_ptr(), counters.data_ptr(), blocks, rounds, n, s1.cuda_stream) == 0
                done += 10
            torch.cuda.synchronize()
            c = counters.cpu().numpy().reshape(64, 2).astype(np.int64)
            results[(mode, name)] = (c[:, 0].reshape(4, 16).sum(1).tolist(), c[:, 1].reshape(4, 16).sum(1).tolist())
    total = blocks * 256 * rounds * launches
    print(f"results per lane quarter [0-15, 16-31, 32-47, 48-63] of {total} evaluations per combination; only combinations with a mismatch are listed")
    for mode in ("alone", "beside fp16 MFMA"):
        bad = {k[1]: v for k, v in results.items() if k[0] == mode and (sum(v[0]) or sum(v[1]))}
        print(f"--- {mode}: {len(bad)} of {lib.pk_map_count()} combinations show mismatches")
        for name, (lo, hi) in sorted(bad.items()):
            op, sel, selhi = name.split("_")[1:]
            nb = 3 if op == "fma" else 2
            bits = lambda v: "[" + ",".join(str((int(v) >> i) & 1) for i in range(nb)) + "]"
            print(f"    v_pk_{op}_f32 op_sel:{bits(sel)} op_sel_hi:{bits(selhi)}: low result {lo}  high result {hi}")


if __name__ == "__main__":
    main()
