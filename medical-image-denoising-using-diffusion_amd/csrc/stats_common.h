// GroupNorm statistics without a kernel, a reduction pass or a hand-off of their own (replaces gn_from_partial_kernel:
// 51 launches per forward).
//
// nn.GroupNorm(8, C) (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214) needs, per sample and group, the
// mean and variance of a tensor that a previous kernel produced.  Here
//   PRODUCER  every workgroup adds its partial sums (sum, sum of squares; fp32 per channel, folded over its waves in a
//             fixed order) to TOTALS with integer atomics.  A total is a 120-bit fixed-point number in three int64
//             limbs of 40 value bits each (resolution 2^-60, range +-2^59; 24 spare bits per limb absorb up to 2^23
//             additions without a carry): the fp32 partial converts EXACTLY, integer addition is associative, so the
//             totals are exact and independent of the order in which workgroups finish -- bit-deterministic without a
//             fixed-order reduction pass.  Totals are kept per BLOCK of `bs` consecutive channels, bs = the largest
//             size that every consuming GroupNorm's groups are whole multiples of (the planner knows the consumers:
//             C/8 on the default network): a workgroup first adds its channels' limbs per block in LDS (exact), then
//             issues one global atomic per block limb -- 6x fewer than per channel;
//   CONSUMER  every thread derives scale = rstd * gamma, shift = beta - mean * rstd * gamma of ITS channel in the
//             prologue: it adds the limbs of its group's blocks (one or two on the default network; both torch.cat
//             sources) and copies, converts once, and needs no cross-lane step (the lanes of a group read the same
//             addresses: one request).  Integer sums are order-free, so every workgroup of every launch gets the
//             same bits.
// Producer and consumers are different kernels (kernel boundary = visibility); one memset of the whole statistics
// arena per forward pass zeroes the totals.  What this replaced, measured: per-channel totals
// with a 64-lane fp64 butterfly in every consumer workgroup cost 11 % of the sampler, the per-channel atomics 5 %.
#pragma once
#include <hip/hip_runtime.h>

namespace midd {

constexpr int GN_GROUPS_C = 8;                    // nn.GroupNorm(8, C) everywhere in the reference
constexpr int STAT_LIMBS = 3;                     // int64 limbs per total
constexpr int STAT_WORDS = 2 * STAT_LIMBS;        // per replica: sum, sum of squares
// Same-address atomics serialise at the memory side (~25 ns each, measured: a 640-workgroup launch at batch 1 spent
// 13-25 us in them), so a channel keeps `rep` copies of its totals (1..STAT_MAX_REPLICAS, chosen per execution program
// from the batch size: ~48 workgroups per copy); a producer workgroup adds to copy (its index mod rep), a consumer adds
// the copies' limbs (integers: exact, order-free) before converting.  Layout [B][C / bs][rep][sum | sumsq][limb].
constexpr int STAT_MAX_REPLICAS = 8;
typedef unsigned long long stat_word;

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

__device__ __forceinline__ int stat_gcd(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// GroupNorm scale/shift of sample b into LDS: gnp[c] = mult * rstd * gamma[c], gnp[Cin + c] = mult * (beta[c] - mean * rstd * gamma[c]).
// tot0 / tot1: block totals [B][C0/bs0][rep][2][3] / [B][C1/bs1][rep][2][3] of the two concatenated sources (tot1 is not
// read when C1 == 0); every group of cg = (C0+C1)/8 channels is a whole number of blocks of each source (the planner
// chose bs0 / bs1 that way).  inv_n = 1 / (pixels * cg).  Visible to the workgroup after its next barrier.  Called by
// all threads; no cross-lane operation.
__device__ __forceinline__ void gn_prologue_lds(const stat_word* __restrict__ tot0, int C0, int bs0,
                                                const stat_word* __restrict__ tot1, int C1, int bs1, int rep,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                double inv_n, int b, float mult, float* gnp, int tid, int nthreads) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int Cin = C0 + C1, cg = Cin / GN_GROUPS_C;
    const int nb0 = C0 / bs0, nb1 = (C1 > 0) ? C1 / bs1 : 0;
    for (int c = tid; c < Cin; c += nthreads) {
        const float ga = gamma[c], be = beta[c];               // requested before the totals are used
        const int g = c / cg;
        stat_word acc[STAT_WORDS];
#pragma unroll
        for (int i = 0; i < STAT_WORDS; ++i) acc[i] = 0;
        int ch = g * cg;
        const int ch_end = ch + cg;
        while (ch < ch_end) {
            const stat_word* p;
            if (ch < C0) { p = tot0 + ((size_t)b * nb0 + ch / bs0) * rep * STAT_WORDS; ch += bs0; }
            else         { p = tot1 + ((size_t)b * nb1 + (ch - C0) / bs1) * rep * STAT_WORDS; ch += bs1; }
            for (int r = 0; r < rep; ++r) {
#pragma unroll
                for (int i = 0; i < STAT_WORDS; i += 2) {
                    const u64x2 v = *reinterpret_cast<const u64x2*>(p + r * STAT_WORDS + i);
                    acc[i] += v[0]; acc[i + 1] += v[1];
                }
            }
        }
        const double mean = stat_total(acc) * inv_n;
        double var = stat_total(acc + STAT_LIMBS) * inv_n - mean * mean;      // biased variance, as torch's group_norm
        if (var < 0) var = 0;
        const float sc = stat_rstd(var + (double)eps) * ga;
        gnp[c] = mult * sc;
        gnp[Cin + c] = mult * (be - (float)mean * sc);
    }
}

// Producer side, called by ALL threads of the workgroup (two barriers inside).  Thread i < 2 * ncol holds t = the
// workgroup's partial sum (which = i / ncol: 0 sum, 1 sum of squares) of channel c0 + i % ncol of sample b.  The channels'
// limbs are added per block of bs channels in LDS (ds_add_u64: exact), then one global atomic per non-zero block limb goes to
// copy `replica` of tot [B][C/bs][rep][2][3].  lds_acc: >= (ncol + 1) * STAT_WORDS words of LDS nobody else is using.
__device__ __forceinline__ void stat_publish(stat_word* __restrict__ tot, int b, int C, int bs, int rep, int replica,
                                             int c0, int ncol, float t, stat_word* lds_acc, int tid, int nthreads) {
    const int blk_first = c0 / bs, nblk = (c0 + ncol - 1) / bs - blk_first + 1;
    for (int j = tid; j < nblk * STAT_WORDS; j += nthreads) lds_acc[j] = 0;
    __syncthreads();
    if (tid < 2 * ncol) {
        const int which = tid / ncol, c = c0 + tid - which * ncol;
        stat_word* slot = lds_acc + ((c / bs - blk_first) * 2 + which) * STAT_LIMBS;
        const unsigned u = __float_as_uint(t);
        const int ex = (int)((u >> 23) & 0xffu);
        if (ex != 0) {                                              // zero (denormals are flushed: < 2^-126)
            unsigned long long m = (unsigned long long)((u & 0x7fffffu) | 0x800000u);
            int s = ex - 150 + 60;                                  // bit position of the mantissa's LSB in the fixed-point number
            if (s < 0) { m = (s > -24) ? (m >> (-s)) : 0ull; s = 0; }
            if (s > 95) s = 95;                                     // |t| >= 2^59 (never a finite activation statistic): pinned, no limb 3
            const int k = s / 40, r = s - k * 40;
            const unsigned long long x = m << r;                    // < 2^63
            unsigned long long lo = x & ((1ull << 40) - 1ull), hi = x >> 40;
            if (u >> 31) { lo = 0ull - lo; hi = 0ull - hi; }        // two's complement: limbs are signed accumulators
            if (lo) atomicAdd(slot + k, lo);
            if (hi) atomicAdd(slot + k + 1, hi);                    // k == 2 => r <= 15 => hi == 0
        }
    }
    __syncthreads();
    for (int j = tid; j < nblk * STAT_WORDS; j += nthreads) {
        const stat_word v = lds_acc[j];
        if (v) atomicAdd(tot + (((size_t)b * (C / bs) + blk_first + j / STAT_WORDS) * rep + replica) * STAT_WORDS + j % STAT_WORDS, v);
    }
}

// The same for a workgroup's whole slice: vals = LDS floats [2][ncol] (sums, then sums of squares, of channels c0 .. c0+ncol-1),
// taken nthreads/2 columns at a time.  lds_acc must not overlap vals.
__device__ __forceinline__ void stat_publish_cols(stat_word* __restrict__ tot, int b, int C, int bs, int rep, int replica,
                                                  int c0, int ncol, const float* vals, stat_word* lds_acc, int tid, int nthreads) {
    const int step = nthreads / 2;
    for (int cb = 0; cb < ncol; cb += step) {
        const int n = min(step, ncol - cb);
        float t = 0.f;
        if (tid < 2 * n) { const int which = tid / n; t = vals[which * ncol + cb + tid - which * n]; }
        stat_publish(tot, b, C, bs, rep, replica, c0 + cb, n, t, lds_acc, tid, nthreads);
    }
}

}  // namespace midd
