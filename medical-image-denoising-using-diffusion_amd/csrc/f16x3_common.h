// Device helpers shared by the f16x3 convolution kernels (conv_mfma_f16x3.hip, conv1x1_f16x3.hip).
#pragma once
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float ACT_PRESCALE = 16.0f;             // 2^s, s = 4, of GroupNorm-ed operands (see header); must match midd_api.hip
// GroupNorm + SiLU operands of the 3x3 kernel: the transform forms t = -log2(e) y and u = 16 t / (1 + 2^t) = -16 log2(e) silu(y);
// the packed weights of those convolutions carry the factor -ln 2 that makes w'' . u = 16 w . silu(y) (midd_api.hip).
constexpr float SILU_ARG_FACTOR = -1.4426950408889634f;
constexpr float SILU_WEIGHT_FACTOR = -0.6931471805599453f;
constexpr float RAW_PRESCALE = 1.0f;              // raw operands without statistics of their own (none on the default networks)
constexpr float ATT_PRESCALE = 16.0f;             // the attention output entering proj: |att| <= max|v| and 16 |v| < 65504 was checked

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

// The two fp16 halves of a pair of fp32 values: hi = fp16(x) (round to nearest even: v_cvt_pk_f16_f32), lo = fp16(x - hi).
// x - hi is exact in fp32 (hi is x rounded to 11 bits), and v_fma_mix{lo,hi}_f16 computes (-1) * hi + x in fp32 with hi
// read as the fp16 it is and rounds the result to fp16 into one half of the destination: ONE instruction per element for
// lo instead of convert-back, subtract, convert (checked bit for bit against the three-instruction form: tools/mb).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h;
    h[0] = (_Float16)x0; h[1] = (_Float16)x1;
    hi = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
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

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_dst_wave_base, 16, 0, 0);
}

}  // namespace midd
