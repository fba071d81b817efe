// GroupNorm(8, C) support kernels for NHWC fp32 activations.
//
// nn.GroupNorm(8, C) on the reference's hot path (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214; eps =
// 1e-5, affine) never runs as a kernel of its own here (stats_common.h): the PRODUCER of a tensor leaves per-channel
// fixed-point (sum, sum of squares) totals, the CONSUMER derives scale = rstd * gamma, shift = beta - mean * rstd * gamma in
// its prologue and applies x * scale + shift (+SiLU) while staging its input.  This file holds
//   chan_total_kernel  the totals of tensors no MFMA convolution produced (in_conv output, bilinear 2x outputs,
//                      unfolded ConvTranspose outputs): one read of the tensor, HBM-bound;
//   preact_kernel      opt-in pre-activation pass for conv3x3_pre_f16x3.hip (PRO_PRE_DMA).
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GN_THREADS = 256;

// grid (rows, B): block `row` sums a contiguous pixel range of sample b per channel in fp64 (products of fp32 values
// are exact in fp64), rounds the block's sums to fp32 and adds them to tot [B][C][2][3] with exact integer atomics.
__global__ __launch_bounds__(GN_THREADS)
void chan_total_kernel(const float* __restrict__ src, stat_word* __restrict__ tot, int HW, int C, int rows) {
    extern __shared__ double red[];               // [ppi][C][2]
    const int CQ = C >> 2;
    const int ppi = GN_THREADS / CQ;
    const int tid = threadIdx.x;
    const int b = blockIdx.y, row = blockIdx.x;
    const int pl = tid / CQ, q = tid - pl * CQ;
    const int per = (HW + rows - 1) / rows;
    const int p0 = row * per, p1 = min(HW, p0 + per);
    if (pl < ppi) {
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        const float* base = src + (size_t)b * HW * C + q * 4;
        for (int p = p0 + pl; p < p1; p += ppi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; s[e] += d; ss[e] = fma(d, d, ss[e]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[((size_t)pl * C + q * 4 + e) * 2 + 0] = s[e];
            red[((size_t)pl * C + q * 4 + e) * 2 + 1] = ss[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += GN_THREADS) {
        double cs = 0, css = 0;
        for (int l = 0; l < ppi; ++l) { cs += red[((size_t)l * C + c) * 2]; css += red[((size_t)l * C + c) * 2 + 1]; }
        stat_word* o = tot + ((size_t)b * C + c) * STAT_WORDS;
        stat_atomic_add(o, (float)cs); stat_atomic_add(o + STAT_LIMBS, (float)css);
    }
}

int chan_partial_rows(int HW, int C) {
    const int ppi = GN_THREADS / (C / 4);
    int r = HW / (ppi * 8);
    if (r < 1) r = 1;
    if (r > 128) r = 128;
    return r;
}

hipError_t chan_total_launch(const float* src, stat_word* tot, int B, int HW, int C, int rows, hipStream_t s) {
    if (C % 4 || C / 4 > GN_THREADS) return hipErrorInvalidValue;
    const int ppi = GN_THREADS / (C / 4);
    const size_t lds = (size_t)ppi * C * 2 * sizeof(double);
    hipLaunchKernelGGL(chan_total_kernel, dim3(rows, B), dim3(GN_THREADS), lds, s, src, tot, HW, C, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ pre-activation pass (opt-in)
// grid (chunks, B).  One thread = 4 channels of one pixel.  Same arithmetic as the conv's in-kernel transform
// (conv_mfma_f16x3.hip): v = x * (16 sc) + 16 sh, SiLU on the 16x-scaled value, hi = fp16(v), lo = fp16(v - hi); output
// per pixel and 16-channel block: 16 high halves (32 B), then 16 low halves.
__global__ __launch_bounds__(256)
void preact_kernel(const PreactArgs a) {
    extern __shared__ float gnp[];                 // [2][C]
    const int C = a.C0 + a.C1, CQ = C >> 2;
    const int b = blockIdx.y;
    gn_prologue_lds(a.gn_tot0, a.C0, a.gn_tot1, a.C1, a.gn_gamma, a.gn_beta, a.gn_eps, a.HW, b, 16.0f, gnp, threadIdx.x, 256);
    __syncthreads();
    const size_t per = ((size_t)a.HW * CQ + gridDim.x - 1) / gridDim.x;
    const size_t i0 = blockIdx.x * per, i1 = min((size_t)a.HW * CQ, i0 + per);
    for (size_t idx = i0 + threadIdx.x; idx < i1; idx += 256) {
        const int q = (int)(idx % CQ);
        const size_t pix = (size_t)b * a.HW + idx / CQ;
        const int c = q * 4;
        const f32x4 x = (c < a.C0) ? *reinterpret_cast<const f32x4*>(a.src0 + pix * a.C0 + c)
                                   : *reinterpret_cast<const f32x4*>(a.src1 + pix * a.C1 + (c - a.C0));
        f32x4 v = x * *reinterpret_cast<const f32x4*>(gnp + c) + *reinterpret_cast<const f32x4*>(gnp + C + c);
        if (a.silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[e] * (-1.4426950408889634f / 16.0f)));
        }
        unsigned short hb[4], lb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16 h = (_Float16)v[e];
            const _Float16 l = (_Float16)(v[e] - (float)h);
            hb[e] = __builtin_bit_cast(unsigned short, h); lb[e] = __builtin_bit_cast(unsigned short, l);
        }
        char* o = reinterpret_cast<char*>(a.out) + (pix * (C >> 4) + (c >> 4)) * 64 + (c & 15) * 2;
        *reinterpret_cast<uint2*>(o) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
        *reinterpret_cast<uint2*>(o + 32) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
    }
}

hipError_t preact_launch(const PreactArgs& a, hipStream_t s) {
    const int C = a.C0 + a.C1;
    if (a.C0 % 4 || a.C1 % 4 || C % 16) return hipErrorInvalidValue;
    const size_t per_img = (size_t)a.HW * (C / 4);
    int chunks = (int)((per_img + 4095) / 4096);           // >= 16 elements of work per thread and block
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(preact_kernel, dim3(chunks, a.B), dim3(256), 2 * C * sizeof(float), s, a);
    return hipGetLastError();
}

}  // namespace midd
