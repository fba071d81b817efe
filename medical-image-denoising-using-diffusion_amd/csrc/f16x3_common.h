// Device helpers shared by the f16x3 convolution kernels (conv_mfma_f16x3.hip, conv1x1_f16x3.hip).
#pragma once
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float ACT_PRESCALE = 16.0f;             // 2^s, s = 4, of GroupNorm-ed operands (see header); must match midd_api.hip
constexpr float RAW_PRESCALE = 1.0f;              // operands no GroupNorm bounds keep fp16's full range: |x| < 65504

__device__ __forceinline__ float silu16(float v) {
    // x * 1/(1+2^(-x*log2 e)) on v_exp_f32 / v_rcp_f32 (~1 ulp each)
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

// Sum over the 16 lanes of a DPP row (lanes 16r..16r+15) with four rotate-and-add steps on the
// VALU (row_ror:8,4,2,1): every lane ends with the row total; the order is fixed per lane.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * ACT_PRESCALE;
        const _Float16 h = (_Float16)x;
        hi[e] = h;
        lo[e] = (_Float16)(x - (float)h);
    }
}

// Wait until at most N of this wave's vector-memory operations (all of them LDS-DMA inside the
// K loop) are outstanding and all its LDS accesses are done, then the workgroup barrier.
template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// GroupNorm finalize inside a consumer's prologue (ConvArgs::gn_part0 != nullptr; replaces a launch of
// gn_from_partial_kernel per GroupNorm): per channel the partial rows of sample b are added in row order in
// fp64, per group the channel sums in channel order in fp64 -- the same order in every workgroup, so all
// workgroups of a launch (and every launch) see bit-identical scale/shift.  gnp: LDS [2][Cin] floats + 16
// (mean, rstd per group); on return gnp[c] = mult * rstd * gamma[c], gnp[Cin + c] = mult * (beta[c] - mean * rstd * gamma[c])
// -- visible after the caller's next barrier.  Called by all threads of the workgroup.
__device__ __forceinline__ void gn_finalize_lds(const ConvArgs& a, int b, float* gnp, float mult, int tid, int nthreads) {
    const int Cin = a.C0 + a.C1;
    const int cg = Cin / GN_GROUPS_;
    float* const mr = gnp + 2 * Cin;
    for (int c = tid; c < Cin; c += nthreads) {
        const float* p; int rows, Cs;
        if (c < a.C0) { rows = a.gn_rows0; Cs = a.C0; p = a.gn_part0 + (size_t)b * rows * 2 * Cs + c; }
        else          { rows = a.gn_rows1; Cs = a.C1; p = a.gn_part1 + (size_t)b * rows * 2 * Cs + (c - a.C0); }
        double s1 = 0, s2 = 0;
        int r = 0;
        for (; r + 8 <= rows; r += 8) {                 // 16 loads in flight; fixed summation order
            float v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v1[u] = p[(size_t)(r + u) * 2 * Cs]; v2[u] = p[(size_t)(r + u) * 2 * Cs + Cs]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
        }
        for (; r < rows; ++r) { s1 += (double)p[(size_t)r * 2 * Cs]; s2 += (double)p[(size_t)r * 2 * Cs + Cs]; }
        gnp[c] = (float)s1; gnp[Cin + c] = (float)s2;
    }
    lds_barrier();
    if (tid < GN_GROUPS_) {
        double t1 = 0, t2 = 0;
        for (int i = 0; i < cg; ++i) { t1 += (double)gnp[tid * cg + i]; t2 += (double)gnp[Cin + tid * cg + i]; }
        const double n = (double)a.gn_hw * cg;
        const double mean = t1 / n;
        double var = t2 / n - mean * mean;
        if (var < 0) var = 0;
        mr[2 * tid] = (float)mean;
        mr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)a.gn_eps));
    }
    lds_barrier();
    for (int c = tid; c < Cin; c += nthreads) {
        const int g = c / cg;
        const float sc = mr[2 * g + 1] * a.gn_gamma[c];
        gnp[c] = mult * sc;
        gnp[Cin + c] = mult * (a.gn_beta[c] - mr[2 * g] * sc);
    }
}

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

}  // namespace midd
