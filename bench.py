#!/usr/bin/env python3
"""Throughput benchmark of the reverse-diffusion sampler path on MI355X.

Contract (task statement): ``python bench.py --gpus N --steps K --warmup W`` — one "step" is one
pass of the hot path over one batch: ``DiffusionDenoiser.denoise`` of B images through the
50-iteration reverse loop (BASELINE.json configs[1]: batch 8, 256x256, 50 steps, one MI355X).
For N > 1 it runs one rank per GPU (launched by torch.distributed.run; started bare, it launches those
ranks itself as a child process); the batch is sharded (B per GPU fixed -> weak scaling) and one RCCL
all-gather collects the outputs.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import midd_loader  # noqa: E402

midd_loader.load()
from midd_amd import UNetDiffusion, DiffusionDenoiser, UNetConfig, topology, timestep_list  # noqa: E402
from midd_amd.native import kernel_source_hash  # noqa: E402
from midd_amd.sharding import gather_outputs  # noqa: E402
from midd_amd.weights import make_state_dict, synthetic_xray  # noqa: E402

# Work model, SURVEY.md section 8(d): reference-graph FLOPs and ideal-fusion bytes per image-step.
GF_PER_IMAGE_STEP = {256: 91.669e9, 512: 424.656e9}
BYTES_PER_IMAGE_STEP_F32 = {256: 846.2e6, 512: 3384.8e6}
PEAK_MFMA_F32 = 157.3e12        # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_MFMA_F16 = 2.5e15          # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
PEAK_HBM = 8.0e12


def pmc_traffic(kernel_name):
    """(HBM bytes per launch of `kernel_name`, provenance) from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 corrections applied as
    MI355X_MICROARCH.md prescribes; tools/profile_round.sh).  A profile taken from other kernel sources than the
    ones this library was built from is refused: (None, reason)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not found:
        return None, "no profiles/*_pmc_traffic.json"
    path = found[-1]                      # newest round's passes
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("kernel_source_hash") != kernel_source_hash():
            return None, f"{os.path.basename(path)} was taken from other kernel sources (stale): not used"
        k = doc["kernels"].get(kernel_name)
        return (k["hbm_bytes_per_launch"] if k else None), os.path.basename(path)
    except (OSError, ValueError, KeyError) as exc:
        return None, f"{os.path.basename(path)}: {exc}"


def rocprof_avg_us(kernel_name):
    """(average duration in us of `kernel_name`, provenance) from the newest committed profiles/*_kernel_stats.csv whose
    sibling *_pmc_traffic.json carries the source hash of the running library; (None, reason) otherwise."""
    import csv
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")))
    if not found:
        return None, "no profiles/*_kernel_stats.csv"
    path = found[-1]
    tag = os.path.basename(path).split("_kernel_stats")[0]
    try:
        with open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")) as f:
            if json.load(f).get("kernel_source_hash") != kernel_source_hash():
                return None, f"{os.path.basename(path)} was taken from other kernel sources (stale): not used"
        with open(path) as f:
            for row in csv.DictReader(f):
                name = (row.get("Name") or row.get("KernelName") or "").replace("void ", "").split("(")[0]
                if name == kernel_name:
                    return float(row.get("AverageNs") or row.get("Average")) / 1e3, os.path.basename(path)
    except (OSError, ValueError, KeyError, TypeError) as exc:
        return None, f"{os.path.basename(path)}: {exc}"
    return None, f"{os.path.basename(path)}: kernel not listed"


def cpu_baseline(sd_np, cfg, size, noise_steps, iters, batch):
    """The oracle (a port of the reference's CPU path, pinned against it in the build container) timed on this
    host's cores on a bounded sample (B = 1 and the workload's batch, 3 repetitions, median):
    `iters` of the iterations for one image, one iteration for the batch."""
    from oracle import ddim_oracle as orc
    sd = orc.to_torch(sd_np)
    topo = topology(cfg)
    beta, alpha, alpha_hat = orc.schedule(noise_steps)
    all_steps = timestep_list(noise_steps, noise_steps)

    def run(noisy, steps):
        x = noisy.clone()
        t0 = time.perf_counter()
        for i in steps:
            t = torch.full((noisy.shape[0],), i, dtype=torch.long)
            eps = torch.clamp(orc.unet_forward(sd, topo, x, noisy, t), -5, 5)
            a, ah = alpha[t][:, None, None, None], alpha_hat[t][:, None, None, None]
            x = torch.clamp((1 / torch.sqrt(a)) * (x - ((1 - a) / torch.sqrt(1 - ah)) * eps), 0, 1)
        return x, time.perf_counter() - t0

    reps = 3
    with torch.no_grad():
        one = torch.from_numpy(synthetic_xray(1, size, size, seed=1234))
        orc.unet_forward(sd, topo, one, one, torch.tensor([all_steps[0]]))      # warm-up (thread pool, oneDNN primitives)
        steps1 = all_steps[:iters]
        t1 = []
        for _ in range(reps):
            x1, dt = run(one, steps1)
            t1.append(dt / len(steps1))
        many = torch.from_numpy(synthetic_xray(batch, size, size, seed=1234))
        tb = []
        if batch > 1:
            run(many, all_steps[:1])
            for _ in range(reps):
                tb.append(run(many, all_steps[:1])[1])
    per_iter_1 = float(np.median(t1))
    rate_1 = 1.0 / (per_iter_1 * noise_steps)
    rate_b = batch / (float(np.median(tb)) * noise_steps) if tb else rate_1
    base = dict(value=rate_b, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"batch {batch} {size}x{size}: 1 of {noise_steps} iterations x {reps} repetitions (median {np.median(tb) if tb else per_iter_1:.2f} s); "
                       f"batch 1: {len(steps1)} iterations x {reps} repetitions (median {per_iter_1:.2f} s per iteration); "
                       f"torch {torch.__version__} CPU fp32 oracle (oracle/ddim_oracle.py), value = the batch-{batch} rate",
                value_batch1=rate_1, reps=reps)
    return x1, steps1, base


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N fresh ranks with torch.distributed.run
    as a CHILD process (this process has not touched the GPU and never will), relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or not lines:
        raise SystemExit(proc.returncode or 1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def baseline_config_label(Bp, S, noise_steps, n_iters):
    """Which BASELINE.json configuration a (per-GPU batch, size, schedule) corresponds to."""
    if (Bp, S, noise_steps, n_iters) == (8, 256, 50, 50):
        return "BASELINE.json configs[1]"
    if (Bp, S, noise_steps, n_iters) == (32, 256, 100, 100):
        return "BASELINE.json configs[2]"
    if (Bp, S, noise_steps, n_iters) == (32, 256, 50, 50):
        return "per-GPU shard of BASELINE.json configs[3]"
    if (Bp, S, noise_steps, n_iters) == (8, 512, 50, 50):
        return "per-GPU shard of BASELINE.json configs[4]"
    return "not a BASELINE.json configuration"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--noise-steps", type=int, default=50)
    ap.add_argument("--cpu-iters", type=int, default=5, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--latency-reps", type=int, default=3,
                    help="single-image latency leg after the timed region (0 = skip: profile runs, so that rocprofv3's "
                         "per-kernel averages cover the bench workload only)")
    ap.add_argument("--compute", default=None, choices=["f16x3", "f32"],
                    help="MFMA arithmetic: split-fp16 x3 (default) or fp32-input MFMA")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                   # before anything initialises the GPU in this process
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sampler path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)        # "nccl" is RCCL on ROCm

    cfg = UNetConfig()
    sd_np = make_state_dict(cfg, seed=42)                     # random-init weights of the architecture
    model = UNetDiffusion(compute=args.compute)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model = model.to(dev).eval()
    den = DiffusionDenoiser(model, noise_steps=args.noise_steps)
    Bp, S = args.batch_per_gpu, args.size
    # inputs resident in HBM before the timed region; image seeds continue across ranks
    noisy = torch.from_numpy(synthetic_xray(Bp, S, S, seed=1234 + rank * Bp)).to(dev)
    n_iters = len(timestep_list(args.noise_steps, args.inference_steps))

    def step():
        out = den.denoise(noisy, inference_steps=args.inference_steps)
        return gather_outputs(out) if world > 1 else out

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert out.shape[0] == Bp * world and torch.isfinite(out).all()

    images = Bp * world * args.steps
    value = images / elapsed
    result = {
        "metric": "denoised images/sec, 256x256 x50 DDIM steps" if (S == 256 and n_iters == 50)
                  else f"denoised images/sec, {S}x{S} x{n_iters} DDIM steps",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (split-fp16 x3 MFMA, fp32 accumulate)" if model.compute == "f16x3" else "f32",
        "data": "synthetic",
        "config": {"workload": f"batch={Bp}/GPU {S}x{S} grayscale, {n_iters}-iteration reverse loop "
                               f"(noise_steps={args.noise_steps}, inference_steps={args.inference_steps}), "
                               f"random-init 12.8M-param UNet ({baseline_config_label(Bp, S, args.noise_steps, n_iters)})",
                   "batch_per_gpu": Bp, "global_batch": Bp * world, "image": [S, S], "iterations": n_iters,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "collective": "one all_gather_into_tensor (RCCL) per step" if world > 1 else "none"},
    }

    if rank == 0:
        # ---- roofline leg: per-kernel HIP-event timing of one more step on the same stream ----
        model.profile_begin()
        den.denoise(noisy, inference_steps=args.inference_steps)
        prof = model.profile_end()
        total_ms = sum(p["total_ms"] for p in prof)
        dom = max((p for p in prof if p["flops"] > 0), key=lambda p: p["total_ms"])
        achieved = dom["flops"] / (dom["total_ms"] * 1e-3)
        if model.compute == "f16x3":
            # every fp32-equivalent product costs three fp16 MFMA products (hi*hi + hi*lo + lo*hi), so the
            # attainable ALGORITHMIC rate is the dense fp16 MFMA peak / 3
            peak, peak_note = PEAK_MFMA_F16 / 3, "2.5 PFLOP/s dense fp16 MFMA / 3 split passes"
        else:
            peak, peak_note = PEAK_MFMA_F32, "fp32-input MFMA dense peak"
        traffic, traffic_src = pmc_traffic(dom["name"]) if (Bp == 8 and S == 256) else (None, "profiled for the default workload only")
        result["roofline"] = {
            "bound": "mfma", "kernel": dom["name"], "achieved": achieved / 1e12, "peak": peak / 1e12,
            "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src, "peak_note": peak_note,
            "measured_under": "HIP events around each launch on the stream it goes to; the batch runs as two half-batch "
                              "programs on two streams, so another kernel is usually co-resident during a span",
            "launches": dom["launches"], "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
            "alg_flops_per_launch": dom["flops"] / dom["launches"],
            "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
            "alg_GBps": dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9,
            "frac_hbm_peak": dom["bytes"] / (dom["total_ms"] * 1e-3) / PEAK_HBM,
            "share_of_kernel_time": dom["total_ms"] / total_ms,
        }
        # the same fraction from the committed rocprofv3 --kernel-trace --stats summary of this command, when it describes
        # the library that is running (source hash): the tracer's averages sit several % above the in-bench events
        rp_us, rp_src = rocprof_avg_us(dom["name"]) if (Bp == 8 and S == 256) else (None, "profiled for the default workload only")
        result["roofline"]["frac_rocprof"] = (dom["flops"] / dom["launches"] / (rp_us * 1e-6) / peak) if rp_us else None
        result["roofline"]["rocprof_avg_launch_us"] = rp_us
        result["roofline"]["rocprof_source"] = rp_src
        # the same kernel with nothing else on the chip: one more pass as ONE program on one stream (MI_NO_SPLIT)
        model.profile_begin()
        model.run_sampler(noisy, timestep_list(args.noise_steps, args.inference_steps), den.beta, den.alpha, den.alpha_hat, clamp_eps=True, no_split=True)
        alone = {p["name"]: p for p in model.profile_end()}
        big = max((p for p in alone.values() if p["flops"] > 0), key=lambda p: p["total_ms"])
        result["roofline"]["alone"] = {
            "what": "the dominant kernel of the same batch run as one program on one stream (no co-resident kernel)",
            "kernel": big["name"], "achieved": big["flops"] / (big["total_ms"] * 1e-3) / 1e12,
            "frac": big["flops"] / (big["total_ms"] * 1e-3) / peak, "avg_launch_us": 1e3 * big["total_ms"] / big["launches"],
            "launches": big["launches"]}
        gf = GF_PER_IMAGE_STEP.get(S)
        if gf:
            per_gpu_rate = (value / world) * n_iters          # image-steps per second per GPU
            result["whole_loop"] = {
                "ref_graph_tflops": per_gpu_rate * gf / 1e12,
                # fraction of the peak of the arithmetic this run computes in (never of a peak it does not use)
                "frac_mfma_peak": per_gpu_rate * gf / peak, "mfma_peak_note": peak_note,
                "alg_GBps": per_gpu_rate * BYTES_PER_IMAGE_STEP_F32[S] / 1e9,
                "frac_hbm_peak": per_gpu_rate * BYTES_PER_IMAGE_STEP_F32[S] / PEAK_HBM,
            }
        result["kernels"] = sorted(
            [dict(name=p["name"], launches=p["launches"], ms=round(p["total_ms"], 3),
                  tflops=round(p["flops"] / (p["total_ms"] * 1e-3) / 1e12, 2) if p["flops"] else None,
                  GBps=round(p["bytes"] / (p["total_ms"] * 1e-3) / 1e9, 1)) for p in prof],
            key=lambda d: -d["ms"])[:24]

        # ---- CPU baseline (rank 0, N = 1 only) + a parity spot check on the same sample ----
        if world == 1 and args.latency_reps > 0:
            # serving regime (run.py:107 denoises one image per request): latency of a single image
            one = torch.from_numpy(synthetic_xray(1, S, S, seed=1234)).to(dev)
            den.denoise(one, inference_steps=args.inference_steps)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.latency_reps):
                den.denoise(one, inference_steps=args.inference_steps)
            torch.cuda.synchronize()
            result["latency_batch1"] = {"ms_per_image": 1e3 * (time.perf_counter() - t1) / args.latency_reps, "iterations": n_iters, "image": [S, S]}
        if world == 1 and args.cpu_iters > 0 and S <= 512:
            x_cpu, steps, base = cpu_baseline(sd_np, cfg, S, args.noise_steps, args.cpu_iters, Bp)
            result["cpu_baseline"] = base
            # row 0 of the TIMED batch (image seed 1234, the oracle's sample) through the same iterations, computed IN the
            # batch the number above was measured on (same execution programs: per-program batch, tiles, key split)
            x_gpu = model.run_sampler(noisy, steps, den.beta, den.alpha, den.alpha_hat, clamp_eps=True)
            result["parity_max_abs_err_vs_oracle"] = float((x_gpu[:1].cpu() - x_cpu).abs().max())
            result["parity_sample"] = f"row 0 of the timed batch of {Bp}, first {len(steps)} iterations"
            result["speedup_vs_cpu_baseline"] = value / base["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()                      # rank 0's roofline leg runs after the timed region; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
