// GroupNorm statistics without a kernel of their own (replaces gn_from_partial_kernel: 51 launches per forward).
//
// nn.GroupNorm(8, C) (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214) needs, per sample and group, the
// mean and variance of a tensor that a previous kernel produced.  Here
//   PRODUCER  every workgroup writes ONE row of per-channel partial sums (sum, sum of squares; fp32) of the pixels it
//             produced, [B][rows][2][C]; the workgroup of a (sample, cout slice) that arrives LAST folds that slice's
//             rows in a fixed order in fp64 into per-channel totals [B][C][2] (stats_arrive_and_fold);
//   CONSUMER  derives scale = rstd * gamma, shift = beta - mean * rstd * gamma of ITS sample in its prologue from the
//             per-channel totals of up to two (torch.cat) sources (gn_prologue_lds): one 16-byte load per channel,
//             a fixed-order wave reduction per group, so every workgroup of every launch gets identical bits.
// Hand-off inside the producer launch (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md
// Guideline 16, recipe R1 in its counter form): rows are stored write-through (agent-scope relaxed atomic store =
// global_store ... sc1), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane does the
// agent-scope fetch-add on the slice's arrival counter; the workgroup whose add returns expected-1 is last: one lane
// runs the agent-scope acquire (buffer_inv sc1), waits for it, workgroup barrier, then plain loads.  The totals are
// read by LATER kernels only (kernel boundary).  The last arriver resets the counter, so one memset of the counter
// block per library call (not per launch) keeps them initialised.
#pragma once
#include <hip/hip_runtime.h>

namespace midd {

constexpr int GN_GROUPS_C = 8;                    // nn.GroupNorm(8, C) everywhere in the reference

__device__ __forceinline__ void stat_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // write-through (sc1)
}

// Called by ALL threads of the workgroup after it has issued the stat_store()s of its row.
//   rows     [nrows][2][Cfull] fp32 rows of this sample;   this workgroup's slice = channels [c0, c0 + ncol)
//   tot      [Cfull][2] fp64 totals of this sample (written for the slice by the last arriver)
//   counter  arrival counter of (sample, slice); `expected` workgroups arrive per launch
//   scratch  LDS, >= (NTHREADS / (2 * ncol / 4)) * 2 * ncol doubles is NOT required: sized as (NTHREADS * 4) doubles
// ncol % 4 == 0.  Fold order: thread (row lane rl, column quad q) adds rows rl, rl + RL, ... in fp64; the RL lane
// sums of a column are then added in lane order -- fixed for a given (nrows, ncol, NTHREADS).
template <int NTHREADS>
__device__ __forceinline__ void stats_arrive_and_fold(const float* rows, int nrows, int Cfull, int c0, int ncol,
                                                      double* tot, int* counter, int expected, double* scratch) {
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    int* const flag = reinterpret_cast<int*>(scratch);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (prev == expected - 1);
        if (last) {
            __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop this CU's stale lines of the rows
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();                                                 // flag is read before scratch is reused
    if (!last) return;
    const int nq = (2 * ncol) >> 2;                                  // column quads: [sum | sumsq] x ncol / 4
    const int RL = NTHREADS / nq;                                    // row lanes
    const int q = tid % nq, rl = tid / nq;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (rl < RL) {
        const int which = (q * 4) / ncol, cc = q * 4 - which * ncol;
        const float* p = rows + (size_t)which * Cfull + c0 + cc;
        const size_t rstride = (size_t)2 * Cfull;
        int r = rl;
        for (; r + 3 * RL < nrows; r += 4 * RL) {                    // four independent 16-byte loads in flight
            const f32x4_ v0 = *reinterpret_cast<const f32x4_*>(p + (size_t)r * rstride);
            const f32x4_ v1 = *reinterpret_cast<const f32x4_*>(p + (size_t)(r + RL) * rstride);
            const f32x4_ v2 = *reinterpret_cast<const f32x4_*>(p + (size_t)(r + 2 * RL) * rstride);
            const f32x4_ v3 = *reinterpret_cast<const f32x4_*>(p + (size_t)(r + 3 * RL) * rstride);
            a0 += (double)v0[0]; a1 += (double)v0[1]; a2 += (double)v0[2]; a3 += (double)v0[3];
            a0 += (double)v1[0]; a1 += (double)v1[1]; a2 += (double)v1[2]; a3 += (double)v1[3];
            a0 += (double)v2[0]; a1 += (double)v2[1]; a2 += (double)v2[2]; a3 += (double)v2[3];
            a0 += (double)v3[0]; a1 += (double)v3[1]; a2 += (double)v3[2]; a3 += (double)v3[3];
        }
        for (; r < nrows; r += RL) {
            const f32x4_ v = *reinterpret_cast<const f32x4_*>(p + (size_t)r * rstride);
            a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
        }
        double* s = scratch + ((size_t)rl * nq + q) * 4;
        s[0] = a0; s[1] = a1; s[2] = a2; s[3] = a3;
    }
    __syncthreads();
    for (int i = tid; i < 2 * ncol; i += NTHREADS) {                 // column i of [sum | sumsq]
        double t = 0;
        for (int l = 0; l < RL; ++l) t += scratch[(size_t)l * nq * 4 + i];
        const int which = i / ncol, cc = i - which * ncol;
        tot[(size_t)(c0 + cc) * 2 + which] = t;
    }
}
constexpr int stats_scratch_doubles(int nthreads) { return nthreads * 4; }

// GroupNorm scale/shift of sample b into LDS: gnp[c] = mult * rstd * gamma[c], gnp[Cin + c] = mult * (beta[c] - mean * rstd * gamma[c]).
// tot0 / tot1: per-channel fp64 (sum, sumsq) totals [B][C0][2] / [B][C1][2] of the two concatenated sources (tot1 may
// be null when C1 == 0); hw = pixels per channel.  Wave w handles groups w, w + nwaves, ...: lane l loads channel
// g*cg + l (+64, ...), a 64-lane butterfly (commutative adds: every lane ends with the same bits) gives the group
// sums, mean / rstd in fp64.  Visible to the workgroup after its next barrier.  Called by all threads.
__device__ __forceinline__ void gn_prologue_lds(const double* __restrict__ tot0, int C0, const double* __restrict__ tot1, int C1,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                int hw, int b, float mult, float* gnp, int tid, int nthreads) {
    const int Cin = C0 + C1, cg = Cin / GN_GROUPS_C;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    for (int g = wave; g < GN_GROUPS_C; g += nwaves) {
        double s1 = 0, s2 = 0;
        for (int l = lane; l < cg; l += 64) {
            const int c = g * cg + l;
            const double* p = (c < C0) ? tot0 + ((size_t)b * C0 + c) * 2 : tot1 + ((size_t)b * C1 + (c - C0)) * 2;
            const double2 v = *reinterpret_cast<const double2*>(p);
            s1 += v.x; s2 += v.y;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
        const double n = (double)hw * cg;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;               // biased variance, as torch's group_norm
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
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
