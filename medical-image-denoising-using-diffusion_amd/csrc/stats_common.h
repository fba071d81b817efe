// GroupNorm statistics without a kernel, a reduction pass or a hand-off of their own (replaces gn_from_partial_kernel:
// 51 launches per forward).
//
// nn.GroupNorm(8, C) (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214) needs, per sample and group, the
// mean and variance of a tensor that a previous kernel produced.  Here
//   PRODUCER  every workgroup adds its partial sums (sum, sum of squares; fp32 per channel, folded over its waves in a
//             fixed order) to TOTALS with integer atomics.  A total is a 120-bit fixed-point number in three int64
//             limbs of 40 value bits each (resolution 2^-60, range +-2^59; 24 spare bits per limb absorb up to 2^23
//             additions without a carry): the fp32 partial converts EXACTLY, integer addition is associative, so the
//             totals are exact (for partials of magnitude >= 2^-37; smaller ones lose their low bits, nothing a
//             statistic of this network can see) and independent of the order in which workgroups finish --
//             bit-deterministic without a fixed-order reduction pass.  A partial that is Inf / NaN (or >= 2^59) adds a
//             sentinel far above any legitimate limb value instead: the total then reads as NaN, so a non-finite
//             activation makes its whole GroupNorm group NaN, as torch's group_norm does.  Totals are kept per BLOCK of `bs` consecutive channels, bs = the largest
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

// Sentinel a non-finite (or out-of-range) partial adds to the top limb, and the magnitude from which a top limb counts as
// "not a number": legitimate top limbs stay below 2^40 * (a few hundred additions); fewer than 2^13 sentinels ever meet
// in one total (workgroups x copies), so the sum neither wraps nor falls below the threshold.
constexpr unsigned long long STAT_NAN_SENTINEL = 1ull << 50;
constexpr long long STAT_NAN_THRESHOLD = 1ll << 49;

__device__ __forceinline__ double stat_total(const stat_word* limbs) {
    const long long l0 = (long long)limbs[0], l1 = (long long)limbs[1], l2 = (long long)limbs[2];
    if (l2 >= STAT_NAN_THRESHOLD || l2 <= -STAT_NAN_THRESHOLD) return __builtin_nan("");
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
                                                double inv_n, int b, float mult, float* gnp, int tid, int nthreads, int* status = nullptr) {
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
        if (status != nullptr && !(mean == mean && var == var)) atomicOr(status, 1);     // NaN / Inf activations (STATUS_NONFINITE): the group becomes NaN, as in torch
        gnp[c] = mult * sc;
        gnp[Cin + c] = mult * (be - (float)mean * sc);
    }
}

// LDS stores / adds / barrier the compiler does not see as such.  While an LDS-DMA transfer is pending in hipcc's bookkeeping
// (and an inline-asm s_waitcnt never clears it) every wait for a tracked vector load becomes vmcnt(0), a compiler-visible LDS
// store may be preceded by one, and __syncthreads() carries a vmcnt(0) of its own -- and vmcnt counts stores too: at the end
// of a convolution that is a wait for the tile's output stores to retire.  Only for LDS areas that are no DMA destination.
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_store_raw(void* p, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(lds_offset(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_raw(void* p, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" ::"v"(lds_offset(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_raw(void* p, float __attribute__((ext_vector_type(4))) v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_offset(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_add_raw(unsigned long long* p, unsigned long long v) { asm volatile("ds_add_u64 %0, %1" ::"v"(lds_offset(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_barrier_raw() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Conversion of an fp32 partial sum to the fixed-point limbs (exact for 2^-37 <= |t| < 2^59), added into LDS block accumulators: slot[0..2].
__device__ __forceinline__ void stat_add_lds(stat_word* slot, float t) {
    const unsigned u = __float_as_uint(t);
    const int ex = (int)((u >> 23) & 0xffu);
    if (ex == 0) return;                                        // zero (denormals are flushed: < 2^-126)
    unsigned long long m = (unsigned long long)((u & 0x7fffffu) | 0x800000u);
    int s = ex - 150 + 60;                                      // bit position of the mantissa's LSB in the fixed-point number
    if (ex == 255 || s > 95) { lds_add_raw(slot + 2, STAT_NAN_SENTINEL); return; }     // Inf / NaN, or |t| >= 2^59: the total reads as NaN
    if (s < 0) { m = (s > -24) ? (m >> (-s)) : 0ull; s = 0; }  // |t| < 2^-37: low bits dropped
    const int k = s / 40, r = s - k * 40;
    const unsigned long long x = m << r;                        // < 2^63
    unsigned long long lo = x & ((1ull << 40) - 1ull), hi = x >> 40;
    if (u >> 31) { lo = 0ull - lo; hi = 0ull - hi; }            // two's complement: limbs are signed accumulators
    if (lo) lds_add_raw(slot + k, lo);
    if (hi) lds_add_raw(slot + k + 1, hi);                      // k == 2 => r <= 15 => hi == 0
}

// ---- power-of-two prescale of a RAW (not GroupNorm-ed) split-fp16 operand ------------------------------------------
// The fp16 halves of an operand x * 2^a lose bits below fp16's subnormal floor (2^-24) and overflow above 65504, so a raw
// operand (stride-2 / folded-ConvT / res_conv input: nothing bounds it) gets a per-(sample, conv) exponent a from the
// sum of squares S of its source tensor(s), which the producers left in the statistics arena:  max|x| <= sqrt(S), so
//     a = 15 - ceil(e / 2),  S = f * 2^e, f in [0.5, 1)     =>     2^a * max|x| < 2^15
// -- no finite input can overflow, and the operand's rms sits at 2^15 / sqrt(elements) or just below (4 .. 150 for the
// tensors of this network).  2^a is exact and is undone exactly in the epilogue (out_scale * 2^-a).
//
// raw_sumsq_lds: called by the 64 lanes of ONE wave; adds the sum-of-squares limbs of sample b's block totals (both
// concatenated sources, all copies) into acc3[0..2] (LDS).  LDS instructions of one wave execute in order, so the zeroing
// store needs no barrier before the adds; the caller's next workgroup barrier publishes acc3.
__device__ __forceinline__ void raw_sumsq_lds(const stat_word* __restrict__ tot0, int C0, int bs0,
                                              const stat_word* __restrict__ tot1, int C1, int bs1, int rep, int b,
                                              stat_word* acc3, int lane) {
    if (lane < STAT_LIMBS) lds_store_raw(acc3 + lane, 0ull);
    const int nb0 = C0 / bs0, nb1 = (C1 > 0) ? C1 / bs1 : 0;
    for (int e = lane; e < (nb0 + nb1) * rep; e += 64) {
        const int blk = e / rep, r = e - blk * rep;
        const stat_word* p = (blk < nb0) ? tot0 + (((size_t)b * nb0 + blk) * rep + r) * STAT_WORDS
                                         : tot1 + (((size_t)b * nb1 + (blk - nb0)) * rep + r) * STAT_WORDS;
#pragma unroll
        for (int i = 0; i < STAT_LIMBS; ++i) {
            const stat_word v = p[STAT_LIMBS + i];
            if (v) lds_add_raw(acc3 + i, v);
        }
    }
}
// exponent a from the published limbs; *bad is set when the sum of squares is not finite (NaN / Inf activations)
__device__ __forceinline__ int raw_prescale_exp(const stat_word* acc3, bool* bad) {
    const float S = (float)stat_total(acc3);
    *bad = !(S < __builtin_inff());                              // NaN or +Inf
    if (!(S > 0.f) || *bad) return 0;                            // all-zero tensor: any exponent
    const int e = (int)((__float_as_uint(S) >> 23) & 0xffu) - 126;          // S = f * 2^e, f in [0.5, 1)  (denormal S: e = -126)
    int a = 15 - ((e + 1) >> 1);                                 // ceil(e / 2) = floor((e + 1) / 2)
    if (a > 60) a = 60;
    return a;                                                    // >= 15 - 64 = -49
}
__device__ __forceinline__ float pow2f(int a) { return __uint_as_float((unsigned)(127 + a) << 23); }      // -126 <= a <= 127

// Producer side, called by ALL threads of the workgroup; TWO barriers inside (the caller's LDS rows must have been written
// with lds_store_raw or be otherwise complete: the barriers wait for lgkmcnt(0) only).  fold(i), i in [0, 2 * ncol), returns the
// workgroup's partial sum (i / ncol: 0 sum, 1 sum of squares) of channel c0 + i % ncol of sample b -- typically a
// fixed-order sum over the waves' rows in LDS, which the caller wrote BEFORE the call (the first barrier here publishes
// them).  The channels' limbs are added per block of bs channels in LDS (ds_add_u64: exact), then one global atomic per
// non-zero block limb goes to copy `replica` of tot [B][C/bs][rep][2][3].  lds_acc: >= (ncol / bs + 2) * STAT_WORDS words
// of LDS nobody else is using (and not what fold() reads).
template <class Fold>
__device__ __forceinline__ void stat_publish(stat_word* __restrict__ tot, int b, int C, int bs, int rep, int replica,
                                             int c0, int ncol, Fold fold, stat_word* lds_acc, int tid, int nthreads) {
    const int blk_first = c0 / bs, nblk = (c0 + ncol - 1) / bs - blk_first + 1;
    // raw LDS stores / barriers throughout: a compiler-visible LDS store or __syncthreads() after the kernel's LDS-DMA
    // traffic would first wait for vmcnt(0) -- here that is the retirement of the tile's output stores (1-2 us per workgroup)
    for (int j = tid; j < nblk * STAT_WORDS; j += nthreads) lds_store_raw(lds_acc + j, 0ull);
    lds_barrier_raw();
    for (int i = tid; i < 2 * ncol; i += nthreads) {
        const int which = i / ncol, c = c0 + i - which * ncol;
        stat_add_lds(lds_acc + ((c / bs - blk_first) * 2 + which) * STAT_LIMBS, fold(i));
    }
    lds_barrier_raw();
    for (int j = tid; j < nblk * STAT_WORDS; j += nthreads) {
        const stat_word v = lds_acc[j];
#if defined(C16_ABL) && C16_ABL == 7      // ablation 7 (wrong results): plain stores instead of the global atomics
        if (v) tot[(((size_t)b * (C / bs) + blk_first + j / STAT_WORDS) * rep + replica) * STAT_WORDS + j % STAT_WORDS] = v;
#else
        if (v) atomicAdd(tot + (((size_t)b * (C / bs) + blk_first + j / STAT_WORDS) * rep + replica) * STAT_WORDS + j % STAT_WORDS, v);
#endif
    }
}

}  // namespace midd
