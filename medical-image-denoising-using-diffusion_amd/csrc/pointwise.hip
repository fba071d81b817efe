// HBM-bound direct kernels at the two ends of the UNet and the resampling helpers.
//
//   in_conv_kernel   : Conv3x3(cat[x, condition]) 2*ic -> Cout        (DDIMModel.py:222-223)
//   out_conv_kernel  : GroupNorm-apply + SiLU + Conv3x3 C -> ic       (DDIMModel.py:213-217,248)
//                      fused with the sampler update of DiffusionDenoiser.denoise
//                      (DDIMModel.py:278-284; cddpm noise term cddpmModels.py:297-303)
//   resize_bilinear  : F.interpolate(mode='bilinear', align_corners=False) (DDIMModel.py:242)
//   conv_transpose   : ConvTranspose2d(C,C,4,2,1) (DDIMModel.py:211) for topologies where the
//                      planner cannot fold it into a 3x3 (never on the default networks)
// K = 18 and N = 1 are degenerate GEMM shapes: these stay on the vector ALU and are judged
// against the HBM roofline.
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// x * sigmoid(x) on v_exp_f32 / v_rcp_f32 (~1 ulp each)
__device__ __forceinline__ float silu_pw(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

// ------------------------------------------------------------------------------ statistics of a pointwise producer
// The two HBM-bound producers below also leave the GroupNorm totals of their output (stats_common.h) -- round 1/2 read the
// tensor a second time for that (chan_total_kernel: 4 launches per forward).  A workgroup owns a pixel range of ONE sample;
// thread = (pixel lane pl < ppi, channel group); each thread sums ITS NV channels over the pixels it walks (fp32, fixed
// order), the lanes are folded per channel in fp64 in lane order, rounded once to fp32, and published with exact integer
// atomics.  scratch: 2 * C * ppi floats of LDS; acc: (C + 2) * 6 words.
template <int NV>
__device__ __forceinline__ void pointwise_publish(const float (&sum)[NV], const float (&sq)[NV], bool active, int pl, int ppi, int cgrp,
                                                  int C, float* scratch, stat_word* acc, stat_word* tot, int b, int bs, int rep,
                                                  int replica, int tid, int nthreads = 256, int c0 = 0, int ncol = -1) {
    if (ncol < 0) ncol = C;                       // the workgroup's channels: [c0, c0 + ncol) of the tensor's C; cgrp counts inside them
    if (active) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            scratch[(size_t)(cgrp * NV + e) * ppi + pl] = sum[e];
            scratch[(size_t)(ncol + cgrp * NV + e) * ppi + pl] = sq[e];
        }
    }
    auto fold = [&](int i) {
        double t = 0;
        for (int l = 0; l < ppi; ++l) t += (double)scratch[(size_t)i * ppi + l];
        return (float)t;
    };
    stat_publish(tot, b, C, bs, rep, replica, c0, ncol, fold, acc, tid, nthreads);
}

// pixels a workgroup of 256 threads walks: ~4 per pixel lane, at most 1024 workgroups per sample
static int pointwise_rows(int HW, int ppi) {
    int r = (HW + ppi * 4 - 1) / (ppi * 4);
    return r < 1 ? 1 : (r > 1024 ? 1024 : r);
}

// ------------------------------------------------------------------------------ in_conv
// thread = (pixel lane, 16 consecutive couts): the 18 input taps are loaded once per 16 outputs; weights
// [9][2ic][Cout] and bias staged in LDS; a pixel's lanes write one contiguous 4*Cout-byte run.  grid (rows, B).
__global__ __launch_bounds__(256)
void in_conv_kernel(const float* __restrict__ x, const float* __restrict__ cond, const float* __restrict__ w,
                    const float* __restrict__ bias, float* __restrict__ out, stat_word* __restrict__ tot, int rep, int bs,
                    int ic, int H, int W, int Cout, int rows, int blocked) {
    extern __shared__ float wl[];                 // 9*2ic*Cout + Cout, then the statistics scratch
    const int nw = 9 * 2 * ic * Cout;
    for (int i = threadIdx.x; i < nw + Cout; i += 256) wl[i] = (i < nw) ? w[i] : bias[i - nw];
    __syncthreads();
    const int CG = Cout >> 4;                     // groups of 16 couts
    const int ppi = 256 / CG;
    const int tid = threadIdx.x;
    const int pl = tid / CG, cg = tid - pl * CG;
    const bool active = pl < ppi;
    const int b = blockIdx.y, row = blockIdx.x;
    const int HW = H * W;
    const int per = (HW + rows - 1) / rows;
    const int p0 = row * per, p1 = min(HW, p0 + per);
    float ssum[16], ssq[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
    if (active) {
        for (int p = p0 + pl; p < p1; p += ppi) {
            const int oy = p / W, ox = p - oy * W;
            f32x4 acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = *reinterpret_cast<const f32x4*>(&wl[nw + cg * 16 + k * 4]);
            for (int ci = 0; ci < 2 * ic; ++ci) {
                const float* plane = (ci < ic) ? x + ((size_t)b * ic + ci) * HW
                                               : cond + ((size_t)b * ic + (ci - ic)) * HW;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int gy = oy + dy - 1;
                    if (gy < 0 || gy >= H) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int gx = ox + dx - 1;
                        if (gx < 0 || gx >= W) continue;
#if defined(PW_ABL) && PW_ABL == 3      // ablation (tools/mb/pw_abl.hip, wrong results): no input loads
                        const float v = 1.0f + (float)gx;
#else
                        const float v = plane[(size_t)gy * W + gx];
#endif
                        const float* wr = &wl[((dy * 3 + dx) * 2 * ic + ci) * Cout + cg * 16];
#if defined(PW_ABL) && PW_ABL == 4      // ablation: no weight reads / multiply-adds
                        acc[dx] += v;
#else
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] += v * *reinterpret_cast<const f32x4*>(wr + k * 4);
#endif
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#if defined(PW_ABL) && PW_ABL == 1      // ablation: no output stores
                if (acc[k][0] == 12345.678f)
#endif
                *reinterpret_cast<f32x4*>(out + act_index(blocked, b, Cout, HW, p, cg * 16 + k * 4)) = acc[k];
#if !(defined(PW_ABL) && PW_ABL == 2)   // ablation: no statistics
#pragma unroll
                for (int e = 0; e < 4; ++e) { ssum[k * 4 + e] += acc[k][e]; ssq[k * 4 + e] += acc[k][e] * acc[k][e]; }
#endif
            }
        }
    }
    if (tot == nullptr) return;
    float* const scratch = wl + nw + Cout;
    stat_word* const acc_lds = reinterpret_cast<stat_word*>(scratch + 2 * (size_t)Cout * ppi + (((nw + Cout) & 1) ? 1 : 0));   // 8-byte aligned
    pointwise_publish<16>(ssum, ssq, active, pl, ppi, cg, Cout, scratch, acc_lds, tot, b, bs, rep, row % rep, tid);
}


// ------------------------------------------------------------------------------ in_conv, one input channel (the reference's grayscale case)
// Round 3.  Ablations (tools/mb/pw_abl.hip) of the kernel above at B = 4: 36 us, 23 of them with the stores removed -- it
// is bound by its instruction stream and its latencies (72 per-lane LDS weight reads and a branch per tap for 288
// multiply-adds; three threads load each pixel's taps; 1.5 waves per SIMD), and its stores, 16 bytes per lane 64 bytes
// apart, reach 2.4 TB/s where contiguous ones reach 4 (tools/mb/hbm_rate.hip).  Here every WAVE works on its own:
//   lane = pixel, all Cout outputs of it: the weights are uniform -- broadcast reads from LDS, half of the couts at a time with
//   the next tap's quads requested before this tap's multiply-adds (as scalar loads with SGPR operands they
//   cost 620 cycles per tap: SMEM returns out of order, every use waits for all of it) -- the 18 taps are loaded once per
//   pixel, branch-free from clamped addresses, coalesced, one pass ahead;
//   the wave's [64 pixels][Cout] tile is turned through its OWN LDS patch (no workgroup barrier in the loop) and
//   leaves as whole contiguous rows, every lane 16 bytes next to its neighbour's;
//   the statistics are summed on the values in store order: a lane meets NQ / gcd(64, NQ) different channel quads.
// Same accumulation order per output as above: bias, then (channel, dy, dx).  grid (rows, B), 4 waves.
template <int COUT, bool BLOCKED>
__global__ __launch_bounds__(256, 3)
void in_conv1_kernel(const float* __restrict__ x, const float* __restrict__ cond, const float* __restrict__ w,
                     const float* __restrict__ bias, float* __restrict__ out, stat_word* __restrict__ tot, int rep, int bs,
                     int H, int W, int per) {
    constexpr int NQ = COUT / 4;                                  // 16-byte pieces per pixel
    constexpr int G64 = (NQ % 16 == 0) ? 16 : (NQ % 8 == 0) ? 8 : (NQ % 4 == 0) ? 4 : (NQ % 2 == 0) ? 2 : 1;   // gcd(64, NQ)
    // NHWC output: store order = pixel-major quads, a lane meets NQ / gcd(64, NQ) different quads; channel-blocked output
    // [B][COUT/16][HW][16] (BLOCKED, midd_internal.h): store order = block, pixel, quad -- a lane meets quad lane % 4 of every block
    constexpr int SETS = BLOCKED ? COUT / 16 : NQ / G64;          // channel quads a lane meets in store order
    constexpr int PS = COUT + 4;                                  // padded pixel stride (words): 16-byte stores of 8 lanes on distinct banks
    constexpr int SLOTS = BLOCKED ? 64 : 4 * SETS * ((64 + NQ - 1) / NQ);   // contributors per channel: (wave, lane / 4) | (wave, set, lane / NQ)
    extern __shared__ __attribute__((aligned(16))) float ic1_lds[];   // [4 waves][64][PS] tiles; afterwards the statistics scratch [2][COUT][SLOTS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tile = ic1_lds + wave * (64 * PS);
    float* const wlds = ic1_lds + 4 * 64 * PS + ((COUT + 2) * STAT_WORDS * 2 + 4);      // behind the tiles and the publish accumulators: [18][COUT] weights, [COUT] bias
    for (int i = tid; i < 19 * COUT; i += 256) wlds[i] = (i < 18 * COUT) ? w[i] : bias[i - 18 * COUT];
    __syncthreads();
    const int b = blockIdx.y, HW = H * W;
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    const float* const xp = x + (size_t)b * HW;
    const float* const cp = cond + (size_t)b * HW;
    float ssum[SETS][4], ssq[SETS][4];
#pragma unroll
    for (int j = 0; j < SETS; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssum[j][e] = 0.f; ssq[j][e] = 0.f; }
    auto load_taps = [&](int base, float (&v)[18]) {
        const int pc = min(base + lane, HW - 1);
        const int oy = pc / W, ox = pc - oy * W;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int gy = oy + dy - 1, cy = min(max(gy, 0), H - 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int gx = ox + dx - 1, cx = min(max(gx, 0), W - 1);
                const bool in = (gy == cy) && (gx == cx);
#if defined(PW_ABL) && PW_ABL == 6      // ablation (tools/mb/pw_abl.hip, wrong results): no input loads
                const float a0 = (float)cx, a1 = (float)cy;
#else
                const float a0 = xp[cy * W + cx], a1 = cp[cy * W + cx];
#endif
                v[dy * 3 + dx] = in ? a0 : 0.f;
                v[9 + dy * 3 + dx] = in ? a1 : 0.f;
            }
        }
    };
    float vn[18];
    int base = p0 + wave * 64;
    if (base < p1) load_taps(base, vn);
    for (; base < p1; base += 256) {
        float v[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = vn[i];
        if (base + 256 < p1) load_taps(base + 256, vn);           // next pass's taps fly under this pass's arithmetic
        // Weights from LDS (uniform address: a broadcast read), half of the couts at a time so that the NEXT tap's quads fit in
        // registers beside the accumulators: LDS reads return in order, the compiler's counted waits keep several in flight.
        // (Scalar loads return out of order: every use waits for ALL of them, lgkmcnt(0) -- one tap in flight, ~620 cycles per tap.)
        int wofs = 0;                                             // opaque per pass: otherwise the weights are hoisted out of the pass loop
        asm volatile("" : "+v"(wofs));
        const float* const wl = wlds + wofs;
        constexpr int NQH = NQ / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 acc[NQH], wv[2][NQH];
            auto wload = [&](int i, f32x4 (&d)[NQH]) {
#pragma unroll
                for (int q = 0; q < NQH; ++q) d[q] = *reinterpret_cast<const f32x4*>(wl + ((i % 9) * 2 + i / 9) * COUT + (h * NQH + q) * 4);
            };
#pragma unroll
            for (int q = 0; q < NQH; ++q) acc[q] = *reinterpret_cast<const f32x4*>(wl + 18 * COUT + (h * NQH + q) * 4);
            wload(0, wv[0]);
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                if (i + 1 < 18) wload(i + 1, wv[(i + 1) & 1]);
                // The multiply-adds are written out as v_pk_fma_f32 with the tap in the LOW dword of an aligned register pair and
                // `op_sel_hi:[1,0,1]` (both results read src1's low dword).  hipcc's own choice for a tap that sits in the HIGH dword
                // of a pair (the taps are consecutive registers: v[1] was the one) is `op_sel:[0,1,0]` -- the low result selects
                // src1's high dword -- and THAT form intermittently drops its low-half product (D.lo = C.lo) in lanes 48..63 when the
                // workgroup shares its CU with another kernel's waves: round 3's red split-sampler case (DESIGN.md section 2a;
                // A/B of nothing but the operand selection, 600 concurrent forwards each: 0 vs 367 corrupted, every tap hit).
                // tests/test_isa_audit_cpu.py fails the build if any shipped kernel carries a packed-fp32 op with an `op_sel:` bit set.
                {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 tp = {v[i], v[i]};
#pragma unroll
                    for (int q = 0; q < NQH; ++q) {
                        f32x2 a0 = {acc[q][0], acc[q][1]}, a1 = {acc[q][2], acc[q][3]};
                        const f32x2 w0 = {wv[i & 1][q][0], wv[i & 1][q][1]}, w1 = {wv[i & 1][q][2], wv[i & 1][q][3]};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a0) : "v"(w0), "v"(tp));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a1) : "v"(w1), "v"(tp));
                        acc[q] = (f32x4){a0[0], a0[1], a1[0], a1[1]};
                    }
                }
#pragma unroll
                for (int q = 0; q < NQH; ++q) asm volatile("" : "+v"(acc[q]));      // pinned: not sunk to the tile stores with all weights live
            }
#pragma unroll
            for (int q = 0; q < NQH; ++q) *reinterpret_cast<f32x4*>(&tile[lane * PS + (h * NQH + q) * 4]) = acc[q];
        }
        // the wave's own patch: its LDS operations execute in order, no barrier
        const int npx = min(64, p1 - base);
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            int px, c, set; float* dst;
            if constexpr (BLOCKED) {                              // block r / 4: 64 pixels x 64 bytes, contiguous
                const int k = r >> 2, idx = (r & 3) * 64 + lane;
                px = idx >> 2; c = k * 4 + (idx & 3); set = k;
                dst = out + (((size_t)b * (COUT / 16) + k) * HW + base) * 16 + (size_t)idx * 4;
            } else {
                const int idx = r * 64 + lane;
                px = idx / NQ; c = idx - px * NQ; set = r % SETS;
                dst = out + ((size_t)b * HW + base) * COUT + (size_t)idx * 4;
            }
            const f32x4 val = *reinterpret_cast<const f32x4*>(&tile[px * PS + c * 4]);
            if (px < npx) {
#if defined(PW_ABL) && PW_ABL == 5      // ablation: no output stores
                if (val[0] == 12345.678f)
#endif
                *reinterpret_cast<f32x4*>(dst) = val;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ssum[set][e] += val[e]; ssq[set][e] += val[e] * val[e]; }
            }
        }
    }
    if (tot == nullptr) return;
#if defined(PW_ABL) && PW_ABL == 8      // ablation: sums accumulated, never published
    if (ssum[0][0] != 12345.678f) return;
#endif
    // statistics: scratch [2][COUT][SLOTS]; contributor (wave, set j, lane) of quad (lane + 64 j) % NQ sits in slot (wave * SETS + j) * ceil(64 / NQ) + lane / NQ
    __syncthreads();                          // every wave is done with its tile
    float* const scratch = ic1_lds;
    for (int i = tid; i < 2 * COUT * SLOTS; i += 256) scratch[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SETS; ++j) {
        const int quad = BLOCKED ? j * 4 + (lane & 3) : (lane + 64 * j) % NQ;
        const int slot = BLOCKED ? wave * 16 + (lane >> 2) : (wave * SETS + j) * ((64 + NQ - 1) / NQ) + lane / NQ;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            scratch[(size_t)(quad * 4 + e) * SLOTS + slot] = ssum[j][e];
            scratch[(size_t)(COUT + quad * 4 + e) * SLOTS + slot] = ssq[j][e];
        }
    }
    auto fold = [&](int i) {
        double t = 0;
        for (int l = 0; l < SLOTS; ++l) t += (double)scratch[(size_t)i * SLOTS + l];
        return (float)t;
    };
    stat_word* const acc_lds = reinterpret_cast<stat_word*>(ic1_lds + 4 * 64 * PS);
    stat_publish(tot, b, COUT, bs, rep, blockIdx.x % rep, 0, COUT, fold, acc_lds, tid, 256);
}

template <int COUT, bool BLOCKED>
static hipError_t in_conv1_launch(const float* x, const float* cond, const float* w, const float* bias, float* out,
                                  stat_word* tot, int rep, int bs, int B, int H, int W, hipStream_t s) {
#ifndef IC1_PASSES
#define IC1_PASSES 2
#endif
    const int HW = H * W;
    int per = IC1_PASSES * 256;             // pixels per workgroup (4 waves x 64 pixels per pass), at most 1024 workgroups per sample
    while ((HW + per - 1) / per > 1024) per += 256;
    const int rows = (HW + per - 1) / per;
    constexpr int NQ = COUT / 4, PS = COUT + 4;
    constexpr int G64 = (NQ % 16 == 0) ? 16 : (NQ % 8 == 0) ? 8 : (NQ % 4 == 0) ? 4 : (NQ % 2 == 0) ? 2 : 1;
    constexpr int SLOTS = BLOCKED ? 64 : 4 * (NQ / G64) * ((64 + NQ - 1) / NQ);
    static_assert(2 * COUT * SLOTS <= 4 * 64 * PS, "the statistics scratch aliases the tiles");
    const size_t lds = (size_t)4 * 64 * PS * sizeof(float) + (size_t)(COUT + 2) * STAT_WORDS * sizeof(stat_word) + 16 + (size_t)19 * COUT * sizeof(float);
    {                                       // COUT = 64: 77.7 KB of dynamic LDS
        static int raised[MIDD_MAX_DEVICES] = {};
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&in_conv1_kernel<COUT, BLOCKED>), (int)lds, raised);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((in_conv1_kernel<COUT, BLOCKED>), dim3(rows, B), dim3(256), lds, s, x, cond, w, bias, out, tot, rep, bs, H, W, per);
    return hipGetLastError();
}

hipError_t in_conv_launch(const float* x, const float* cond, const float* w, const float* bias, float* out,
                          stat_word* tot, int rep, int bs, int B, int ic, int H, int W, int Cout, int blocked, hipStream_t s) {
    if (Cout % 16 || Cout / 16 > 256) return hipErrorInvalidValue;
    if (ic == 1) {                          // the grayscale case: in_conv1_kernel for the widths it is instantiated for
#define MIDD_IC1(N) if (Cout == N) return blocked ? in_conv1_launch<N, true>(x, cond, w, bias, out, tot, rep, bs, B, H, W, s) \
                                                 : in_conv1_launch<N, false>(x, cond, w, bias, out, tot, rep, bs, B, H, W, s);
        MIDD_IC1(48) MIDD_IC1(32) MIDD_IC1(64)
#undef MIDD_IC1
    }
    const int ppi = 256 / (Cout / 16);
    const int rows = pointwise_rows(H * W, ppi);
    const size_t lds = (size_t)(9 * 2 * ic * Cout + Cout + 2 * Cout * ppi + 2) * sizeof(float) + (size_t)(Cout + 2) * STAT_WORDS * sizeof(stat_word);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(in_conv_kernel, dim3(rows, B), dim3(256), lds, s, x, cond, w, bias, out, tot, rep, bs, ic, H, W, Cout, rows, blocked);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ out_conv (+ sampler update)
// Workgroup = 16x16 output pixels.  The 18x18 halo is staged through LDS in 16-channel chunks
// with GroupNorm-apply + SiLU on the way in; pixel stride 20 floats keeps the float4 reads of
// 16 neighbouring lanes on distinct banks.
constexpr int OC_T = 16;
constexpr int OC_I = OC_T + 2;
constexpr int OC_PS = 20;         // padded pixel stride in floats

// IC: output channels at compile time (1: the reference's grayscale case; 0: a.ic at run time, <= 4).  With the count
// only known at run time hipcc indexes the accumulators through select chains and splits the 16-byte LDS reads:
// 7 340 instructions, 989 v_cndmask among them, for a loop of 432 multiply-adds (round 3; tools/isa_count.py).
template <int IC>
__global__ __launch_bounds__(256)
void out_conv_kernel(const OutConvArgs a, const float* __restrict__ wglob /* == a.w: a restrict parameter of its own, so that uniform reads become scalar loads */) {
    __shared__ __attribute__((aligned(16))) float tile[OC_I * OC_I * OC_PS];
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [ic][9][C], then [2][C] GroupNorm scale / shift of this sample
    const int ic = IC ? IC : a.ic;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    float* const gnp = wl + ic * 9 * a.C;
    const int tiles_x = (a.W + OC_T - 1) / OC_T, tiles_y = (a.H + OC_T - 1) / OC_T;
    const int b = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - b * tiles_x * tiles_y;
    const int oy0 = (trem / tiles_x) * OC_T, ox0 = (trem % tiles_x) * OC_T;
    const int C = a.C;
    if constexpr (IC == 0) for (int i = tid; i < ic * 9 * C; i += 256) wl[i] = a.w[i];     // (IC > 0 reads the weights through scalar loads)

    float acc[4] = {0.f, 0.f, 0.f, 0.f};          // ic <= 4 output channels
    // Staging, round 3: a thread's slots (halo pixel, channel quad) are the same for every 16-channel chunk, so their
    // addresses are formed once, branch-free (clamped; out-of-image and surplus slots load a valid dummy and store
    // zeros), and the NEXT chunk's quads are requested before this chunk's taps: the loads fly under the arithmetic.
    constexpr int NSL = (OC_I * OC_I * 4 + 255) / 256;
    unsigned soff[NSL];                           // float offset of the slot's quad from the chunk's base
    unsigned live = 0;                            // bit s: slot exists and lies in the image
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
        const int slot = min(tid + s * 256, OC_I * OC_I * 4 - 1);
        const int pix = slot >> 2, q = slot & 3;
        const int iy = pix / OC_I, ix = pix - iy * OC_I;
        const int gy = oy0 + iy - 1, gx = ox0 + ix - 1;
        const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
        if (tid + s * 256 < OC_I * OC_I * 4 && gy == cy && gx == cx) live |= 1u << s;
        soff[s] = (unsigned)((size_t)(cy * a.W + cx) * (a.blocked ? 16 : C) + q * 4);       // inside the sample (NHWC) / inside a block plane
    }
    const size_t plane = (size_t)a.H * a.W;
    auto chunk_base = [&](int c0) {               // first element of the sample's 16-channel chunk c0
        return a.blocked ? ((size_t)b * (C >> 4) + (c0 >> 4)) * plane * 16 : (size_t)b * plane * C + c0;
    };
    f32x4 pre[NSL];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
#if defined(PW_ABL) && PW_ABL == 11     // ablation (tools/mb/pw_abl.hip, wrong results): no input loads
            pre[s] = (f32x4){(float)soff[s], 1.f, 2.f, 3.f};
#else
            pre[s] = *reinterpret_cast<const f32x4*>(a.src + chunk_base(c0) + soff[s]);
#endif
        }
    };
    prefetch(0);                                  // the first chunk's quads fly while the GroupNorm scale / shift are derived
    // (second source: C1 = 0, never read; a literal nullptr there crashes hipcc 7.2's inliner)
    gn_prologue_lds(a.gn_tot, C, a.gn_bs, a.gn_tot, 0, 1, a.stat_rep, a.gn_gamma, a.gn_beta, a.gn_eps, 1.0 / ((double)a.H * a.W * (C / GN_GROUPS_C)), b, 1.0f, gnp, tid, 256);
    for (int c0 = 0; c0 < C; c0 += 16) {
        __syncthreads();                          // the previous chunk's taps are done with the tile
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const int slot = tid + s * 256;
            if (slot < OC_I * OC_I * 4) {
                const int q = slot & 3;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(gnp + c0 + q * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(gnp + C + c0 + q * 4);
                f32x4 v = pre[s] * sc + sh;
#if !(defined(PW_ABL) && PW_ABL == 12)  // ablation: no SiLU
                v.x = silu_pw(v.x); v.y = silu_pw(v.y); v.z = silu_pw(v.z); v.w = silu_pw(v.w);
#endif
                if (!((live >> s) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};      // the conv's zero padding
                *reinterpret_cast<f32x4*>(&tile[(slot >> 2) * OC_PS + q * 4]) = v;
            }
        }
        if (c0 + 16 < C) prefetch(c0 + 16);
        __syncthreads();
#if defined(PW_ABL) && PW_ABL == 13     // ablation: no taps
        acc[0] += tile[(ty * OC_I + tx) * OC_PS];
#else
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const float* px = &tile[((ty + dy) * OC_I + tx + dx) * OC_PS];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
                if constexpr (IC > 0) {
#pragma unroll
                    for (int oc = 0; oc < IC; ++oc) {
                        // uniform address: scalar loads, the weights are SGPR operands (no LDS read, no register)
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wglob + (((oc * 9 + tap) * (C >> 4)) << 4) + c0 + q * 4);
                        acc[oc] = __builtin_fmaf(v.x, wv.x, __builtin_fmaf(v.y, wv.y, __builtin_fmaf(v.z, wv.z, __builtin_fmaf(v.w, wv.w, acc[oc]))));
                    }
                } else {
                    for (int oc = 0; oc < ic; ++oc) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[(oc * 9 + tap) * C + c0 + q * 4]);
                        acc[oc] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                    }
                }
            }
        }
#endif
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= a.H || ox >= a.W) return;
#pragma unroll
    for (int oc = 0; oc < (IC ? IC : 4); ++oc) {
        if (oc >= ic) break;
        const size_t o = (((size_t)b * ic + oc) * a.H + oy) * a.W + ox;
        float eps = acc[oc] + a.bias[oc];
        if (a.eps_out) a.eps_out[o] = eps;
        if (a.x) {
            const float c1 = a.c1, c2 = a.c2, c3 = a.c3;
            const float* noise = a.noise;
            // x <- clamp( (1/sqrt(alpha)) * (x - ((1-alpha)/sqrt(1-alpha_hat)) * eps) [+ sqrt(beta)*noise], 0, 1 )
            // evaluated with the reference's operation order and no fused multiply-add.
            if (a.clamp_eps) eps = fminf(fmaxf(eps, -5.0f), 5.0f);
            float xn = __fmul_rn(c1, __fsub_rn(a.x[o], __fmul_rn(c2, eps)));
            if (noise) xn = __fadd_rn(xn, __fmul_rn(c3, noise[o]));
            a.x[o] = fminf(fmaxf(xn, 0.0f), 1.0f);
        }
    }
}

hipError_t out_conv_launch(const OutConvArgs& a, hipStream_t s) {
    if (a.ic > 4 || a.C % 16) return hipErrorInvalidValue;
    const size_t lds = ((size_t)a.ic * 9 * a.C + 2 * a.C) * sizeof(float);
    const int tiles = ((a.W + OC_T - 1) / OC_T) * ((a.H + OC_T - 1) / OC_T);
    if (a.ic == 1) hipLaunchKernelGGL(out_conv_kernel<1>, dim3(a.B * tiles), dim3(256), lds, s, a, a.w);
    else hipLaunchKernelGGL(out_conv_kernel<0>, dim3(a.B * tiles), dim3(256), lds, s, a, a.w);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ bilinear resize (NHWC; the channel-blocked variant follows)
// Same index/weight arithmetic as ATen's upsample_bilinear2d with align_corners=False:
//   src = max(0, scale*(dst+0.5)-0.5), scale = in/out;  i0 = floor(src), i1 = i0 + (i0 < in-1), l1 = src - i0.
// grid (rows, B); also leaves the GroupNorm totals of its output (pointwise_publish)
__global__ __launch_bounds__(256)
void resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, stat_word* __restrict__ tot, int rep, int bs,
                            int H, int W, int C, int OH, int OW, float sy, float sx, int rows) {
    extern __shared__ float rs_lds[];             // statistics scratch
    const int CQ = C >> 2;
    const int ppi = 256 / CQ;
    const int tid = threadIdx.x;
    const int pl = tid / CQ, cq = tid - pl * CQ;
    const bool active = pl < ppi;
    const int b = blockIdx.y, row = blockIdx.x;
    const int OHW = OH * OW;
    const int per = (OHW + rows - 1) / rows;
    const int p0 = row * per, p1 = min(OHW, p0 + per);
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const float* base = src + (size_t)b * H * W * C + cq * 4;
        for (int p = p0 + pl; p < p1; p += ppi) {
            const int oy = p / OW, ox = p - oy * OW;
            float fy = sy * ((float)oy + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
            float fx = sx * ((float)ox + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
            const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * C);
            const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * C);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * C);
            const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * C);
            const f32x4 r = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            *reinterpret_cast<f32x4*>(dst + ((size_t)b * OHW + p) * C + cq * 4) = r;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[e] += r[e]; ssq[e] += r[e] * r[e]; }
        }
    }
    if (tot == nullptr) return;
    stat_word* const acc_lds = reinterpret_cast<stat_word*>(rs_lds + 2 * (size_t)C * ppi);       // C % 4 == 0: 8-byte aligned
    pointwise_publish<4>(ssum, ssq, active, pl, ppi, cq, C, rs_lds, acc_lds, tot, b, bs, rep, row % rep, tid);
}

// Channel-blocked tensors [B][C/16][H][W][16] (midd_internal.h): grid (rows, B, C/16), thread = (pixel lane 0..63, quad 0..3 of
// the block) -- a wave reads and writes whole contiguous runs of 16 pixels x 64 bytes.  Same arithmetic and statistics as above.
__global__ __launch_bounds__(256)
void resize_bilinear_blocked_kernel(const float* __restrict__ src, float* __restrict__ dst, stat_word* __restrict__ tot, int rep, int bs,
                                    int H, int W, int C, int OH, int OW, float sy, float sx, int rows) {
    __shared__ float rs_scratch[2 * 16 * 64];
    __shared__ stat_word rs_acc[(16 + 2) * STAT_WORDS];
    const int tid = threadIdx.x;
    const int pl = tid >> 2, q = tid & 3;
    const int b = blockIdx.y, row = blockIdx.x, blk = blockIdx.z;
    const int OHW = OH * OW;
    const int per = (OHW + rows - 1) / rows;
    const int p0 = row * per, p1 = min(OHW, p0 + per);
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const float* base = src + (((size_t)b * (C >> 4) + blk) * (size_t)(H * W)) * 16 + q * 4;
    float* const obase = dst + (((size_t)b * (C >> 4) + blk) * (size_t)OHW) * 16 + q * 4;
    for (int p = p0 + pl; p < p1; p += 64) {
        const int oy = p / OW, ox = p - oy * OW;
        float fy = sy * ((float)oy + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
        float fx = sx * ((float)ox + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * 16);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * 16);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * 16);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * 16);
        const f32x4 r = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        *reinterpret_cast<f32x4*>(obase + (size_t)p * 16) = r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssum[e] += r[e]; ssq[e] += r[e] * r[e]; }
    }
    if (tot == nullptr) return;
    pointwise_publish<4>(ssum, ssq, true, pl, 64, q, C, rs_scratch, rs_acc, tot, b, bs, rep, row % rep, tid, 256, blk * 16, 16);
}

hipError_t resize_bilinear_launch(const float* src, float* dst, stat_word* tot, int rep, int bs, int B, int H, int W, int C, int OH, int OW, int blocked, hipStream_t s) {
    if (blocked) {
        if (C % 16) return hipErrorInvalidValue;
        const int rows_b = pointwise_rows(OH * OW, 64);
        hipLaunchKernelGGL(resize_bilinear_blocked_kernel, dim3(rows_b, B, C / 16), dim3(256), 0, s,
                           src, dst, tot, rep, bs, H, W, C, OH, OW, (float)H / (float)OH, (float)W / (float)OW, rows_b);
        return hipGetLastError();
    }
    if (C % 4 || C / 4 > 256) return hipErrorInvalidValue;
    const int ppi = 256 / (C / 4);
    const int rows = pointwise_rows(OH * OW, ppi);
    const size_t lds = (size_t)2 * C * ppi * sizeof(float) + (size_t)(C + 2) * STAT_WORDS * sizeof(stat_word);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(rows, B), dim3(256), lds, s,
                       src, dst, tot, rep, bs, H, W, C, OH, OW, (float)H / (float)OH, (float)W / (float)OW, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ ConvTranspose2d(4,2,1) direct
// out[oy][ox][co] = bias[co] + sum_{ky,kx,ci} in[(oy+1-ky)/2][(ox+1-kx)/2][ci] * w[ky][kx][ci][co]
// over taps with (oy+1-ky), (ox+1-kx) even and in range.  Thread = (output pixel, 4 couts).
__global__ __launch_bounds__(256)
void conv_transpose_kernel(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
                           float* __restrict__ dst, int B, int H, int W, int Cin, int Cout, int blocked) {
    const int OH = 2 * H, OW = 2 * W, CQ = Cout >> 2;
    const long total = (long)B * OH * OW * CQ;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cq = (int)(gid % CQ);
    const long pix = gid / CQ;
    const int ox = (int)(pix % OW);
    const int oy = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((long)OW * OH));
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + cq * 4);
    for (int ky = 0; ky < 4; ++ky) {
        const int ny = oy + 1 - ky;
        if (ny < 0 || (ny & 1) || (ny >> 1) >= H) continue;
        for (int kx = 0; kx < 4; ++kx) {
            const int nx = ox + 1 - kx;
            if (nx < 0 || (nx & 1) || (nx >> 1) >= W) continue;
            const int ipix = (ny >> 1) * W + (nx >> 1);
            const float* wp = w + ((size_t)(ky * 4 + kx) * Cin) * Cout + cq * 4;
            for (int ci = 0; ci < Cin; ++ci)
                acc += src[act_index(blocked, b, Cin, H * W, ipix, ci)] * *reinterpret_cast<const f32x4*>(wp + (size_t)ci * Cout);
        }
    }
    *reinterpret_cast<f32x4*>(dst + act_index(blocked, b, Cout, OH * OW, oy * OW + ox, cq * 4)) = acc;
}

hipError_t conv_transpose_launch(const float* src, const float* w, const float* bias, float* dst,
                                 int B, int H, int W, int Cin, int Cout, int blocked, hipStream_t s) {
    if (Cout % 4) return hipErrorInvalidValue;
    const long total = (long)B * 4 * H * W * (Cout / 4);
    hipLaunchKernelGGL(conv_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       src, w, bias, dst, B, H, W, Cin, Cout, blocked);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ helpers
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C, int blocked) {
    const long total = (long)B * H * W * C;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const long pix = gid / C;
    const long hw = pix % ((long)H * W);
    const int b = (int)(pix / ((long)H * W));
    dst[((size_t)b * C + c) * H * W + hw] = src[act_index(blocked, b, C, H * W, (int)hw, c)];
}

hipError_t nhwc_to_nchw_launch(const float* src, float* dst, int B, int H, int W, int C, int blocked, hipStream_t s) {
    const long total = (long)B * H * W * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, B, H, W, C, blocked);
    return hipGetLastError();
}

struct I32x32 { int v[32]; };
__global__ void fill_i32_kernel(int* dst, I32x32 vals, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

hipError_t fill_i32_launch(int* dst, const int* host_vals, int n, hipStream_t s) {
    for (int i = 0; i < n; i += 32) {
        I32x32 v;
        const int m = (n - i < 32) ? n - i : 32;
        for (int j = 0; j < 32; ++j) v.v[j] = (j < m) ? host_vals[i + j] : 0;
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(32), 0, s, dst + i, v, m);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace midd
