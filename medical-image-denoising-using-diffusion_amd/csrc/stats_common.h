// GroupNorm statistics without a kernel, a reduction pass or a hand-off of their own (replaces gn_from_partial_kernel:
// 51 launches per forward).
//
// nn.GroupNorm(8, C) (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214) needs, per sample and group, the
// mean and variance of a tensor that a previous kernel produced.  Here
//   PRODUCER  every workgroup adds its per-channel partial sums (sum, sum of squares; fp32, folded over its waves
//             in a fixed order) to per-channel TOTALS with integer atomics.  A total is a 120-bit fixed-point
//             number in three int64 limbs of 40 value bits each (resolution 2^-60, range +-2^59; 24 spare bits per
//             limb absorb up to 2^23 additions without a carry): the fp32 partial converts EXACTLY, integer addition
//             is associative, so the totals are exact and independent of the order in which workgroups finish --
//             bit-deterministic without a fixed-order reduction pass;
//   CONSUMER  derives scale = rstd * gamma, shift = beta - mean * rstd * gamma of ITS sample in its prologue from the
//             totals of up to two (torch.cat) sources (gn_prologue_lds): limbs -> fp64, a fixed-order wave
//             reduction per group, so every workgroup of every launch gets identical bits.
// Producer and consumers are different kernels (kernel boundary = visibility); one memset of the whole statistics
// arena per forward pass zeroes the totals.
#pragma once
#include <hip/hip_runtime.h>

namespace midd {

constexpr int GN_GROUPS_C = 8;                    // nn.GroupNorm(8, C) everywhere in the reference
constexpr int STAT_LIMBS = 3;                     // int64 limbs per total
constexpr int STAT_WORDS = 2 * STAT_LIMBS;        // per replica: sum, sum of squares
// Same-address atomics serialise at the memory side (~25 ns each, measured: a 640-workgroup launch at batch 1 spent
// 13-25 us in them), so a channel keeps `rep` copies of its totals (1..STAT_MAX_REPLICAS, chosen per execution program
// from the batch size: ~48 workgroups per copy); a producer workgroup adds to copy (its index mod rep), a consumer adds
// the copies' limbs (integers: exact, order-free) before converting.  Layout [B][C][rep][sum | sumsq][limb].
constexpr int STAT_MAX_REPLICAS = 8;
typedef unsigned long long stat_word;

__device__ __forceinline__ stat_word* stat_slot(stat_word* tot, size_t b, int C, int c, int rep, int replica, int which) {
    return tot + ((b * C + c) * rep + replica) * STAT_WORDS + which * STAT_LIMBS;
}

// totals[k] += limb k of v * 2^60 (exact for 2^-37 <= |v| < 2^59; smaller magnitudes are truncated towards zero at
// 2^-60, far below fp32 resolution of any sum they could matter in)
__device__ __forceinline__ void stat_atomic_add(stat_word* limbs, float v) {
    const unsigned u = __float_as_uint(v);
    const int ex = (int)((u >> 23) & 0xffu);
    if (ex == 0) return;                                         // zero (denormals are flushed: < 2^-126)
    unsigned long long m = (unsigned long long)((u & 0x7fffffu) | 0x800000u);
    int s = ex - 150 + 60;                                       // bit position of the mantissa's LSB in the fixed-point number
    if (s < 0) { m = (s > -24) ? (m >> (-s)) : 0ull; s = 0; }
    if (s > 95) s = 95;                                          // |v| >= 2^59 (never a finite activation statistic): pinned, no limb 3
    const int k = s / 40, r = s - k * 40;
    const unsigned long long x = m << r;                         // < 2^63
    unsigned long long lo = x & ((1ull << 40) - 1ull), hi = x >> 40;
    if (u >> 31) { lo = 0ull - lo; hi = 0ull - hi; }             // two's complement: limbs are signed accumulators
    if (lo) atomicAdd(limbs + k, lo);
    if (hi) atomicAdd(limbs + k + 1, hi);                        // k == 2 => r <= 15 => hi == 0
}

// 1 / sqrt(x) to fp32 accuracy without the fp64 sqrt / divide sequence: v_rsq_f32 and one Newton step
__device__ __forceinline__ float stat_rstd(double var_plus_eps) {
    const float x = (float)var_plus_eps;
    float y = __builtin_amdgcn_rsqf(x);
    y = y * (1.5f - 0.5f * x * y * y);
    return y;
}

__device__ __forceinline__ double stat_total(const stat_word* limbs) {
    const long long l0 = (long long)limbs[0], l1 = (long long)limbs[1], l2 = (long long)limbs[2];
    return (double)l0 * 0x1p-60 + (double)l1 * 0x1p-20 + (double)l2 * 0x1p20;       // each limb is exact in fp64 (|l| < 2^53)
}

// one channel's (sum, sum of squares) from its `rep` copies: 16-byte loads, integer limb sums, one conversion
__device__ __forceinline__ void stat_channel(const stat_word* __restrict__ p, int rep, double& s1, double& s2) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    stat_word acc[STAT_WORDS];
#pragma unroll
    for (int i = 0; i < STAT_WORDS; ++i) acc[i] = 0;
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < STAT_WORDS; i += 2) {
            const u64x2 v = *reinterpret_cast<const u64x2*>(p + r * STAT_WORDS + i);
            acc[i] += v[0]; acc[i + 1] += v[1];
        }
    }
    s1 = stat_total(acc); s2 = stat_total(acc + STAT_LIMBS);
}

// GroupNorm scale/shift of sample b into LDS: gnp[c] = mult * rstd * gamma[c], gnp[Cin + c] = mult * (beta[c] - mean * rstd * gamma[c]).
// tot0 / tot1: totals [B][C0][rep][2][3] / [B][C1][rep][2][3] of the two concatenated sources (tot1 is not read when C1 == 0);
// hw = pixels per channel.  Wave w handles groups w, w + nwaves, ...: lane l takes channel g*cg + l (+64, ...), a
// 64-lane butterfly (commutative adds: every lane ends with the same bits) gives the group sums, mean / rstd in fp64.
// Visible to the workgroup after its next barrier.  Called by all threads.
__device__ __forceinline__ void gn_prologue_lds(const stat_word* __restrict__ tot0, int C0, const stat_word* __restrict__ tot1, int C1, int rep,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                int hw, int b, float mult, float* gnp, int tid, int nthreads) {
    const int STAT_CH_WORDS = rep * STAT_WORDS;
    const int Cin = C0 + C1, cg = Cin / GN_GROUPS_C;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    if (cg <= 64 && GN_GROUPS_C <= 2 * nwaves) {
        // common case: at most two groups per wave, one channel per lane.  Every load (totals live at the memory
        // side after the producers' atomics: ~1 us away; gamma / beta) is requested before any of them is used.
        const int g0 = wave, g1 = wave + nwaves;
        const bool on0 = g0 < GN_GROUPS_C && lane < cg, on1 = g1 < GN_GROUPS_C && lane < cg;
        const int c0 = g0 * cg + lane, c1 = g1 * cg + lane;
        float ga0 = 0.f, be0 = 0.f, ga1 = 0.f, be1 = 0.f;
        double s1a = 0, s2a = 0, s1b = 0, s2b = 0;
        if (on0) { ga0 = gamma[c0]; be0 = beta[c0]; }
        if (on1) { ga1 = gamma[c1]; be1 = beta[c1]; }
        if (on0) stat_channel((c0 < C0) ? tot0 + ((size_t)b * C0 + c0) * STAT_CH_WORDS : tot1 + ((size_t)b * C1 + (c0 - C0)) * STAT_CH_WORDS, rep, s1a, s2a);
        if (on1) stat_channel((c1 < C0) ? tot0 + ((size_t)b * C0 + c1) * STAT_CH_WORDS : tot1 + ((size_t)b * C1 + (c1 - C0)) * STAT_CH_WORDS, rep, s1b, s2b);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s1a += __shfl_xor(s1a, off, 64); s2a += __shfl_xor(s2a, off, 64);
            s1b += __shfl_xor(s1b, off, 64); s2b += __shfl_xor(s2b, off, 64);
        }
        const double inv_n = 1.0 / ((double)hw * cg);
        {
            const double mean = s1a * inv_n;
            double var = s2a * inv_n - mean * mean;          // biased variance, as torch's group_norm
            if (var < 0) var = 0;
            const float rstd = stat_rstd(var + (double)eps), meanf = (float)mean;
            if (on0) { const float sc = rstd * ga0; gnp[c0] = mult * sc; gnp[Cin + c0] = mult * (be0 - meanf * sc); }
        }
        if (g1 < GN_GROUPS_C) {
            const double mean = s1b * inv_n;
            double var = s2b * inv_n - mean * mean;
            if (var < 0) var = 0;
            const float rstd = stat_rstd(var + (double)eps), meanf = (float)mean;
            if (on1) { const float sc = rstd * ga1; gnp[c1] = mult * sc; gnp[Cin + c1] = mult * (be1 - meanf * sc); }
        }
        return;
    }
    for (int g = wave; g < GN_GROUPS_C; g += nwaves) {
        double s1 = 0, s2 = 0;
        for (int l = lane; l < cg; l += 64) {
            const int c = g * cg + l;
            double c1v, c2v;
            stat_channel((c < C0) ? tot0 + ((size_t)b * C0 + c) * STAT_CH_WORDS : tot1 + ((size_t)b * C1 + (c - C0)) * STAT_CH_WORDS, rep, c1v, c2v);
            s1 += c1v; s2 += c2v;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
        const double inv_n = 1.0 / ((double)hw * cg);
        const double mean = s1 * inv_n;
        double var = s2 * inv_n - mean * mean;           // biased variance, as torch's group_norm
        if (var < 0) var = 0;
        const float rstd = stat_rstd(var + (double)eps);
        const float meanf = (float)mean;
        for (int l = lane; l < cg; l += 64) {
            const int c = g * cg + l;
            const float sc = rstd * gamma[c];
            gnp[c] = mult * sc;
            gnp[Cin + c] = mult * (beta[c] - meanf * sc);
        }
    }
}

}  // namespace midd
