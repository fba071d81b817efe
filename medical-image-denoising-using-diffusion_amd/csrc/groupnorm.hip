// GroupNorm(8, C) support kernels for fp32 activations (NHWC or channel-blocked, midd_internal.h: act_index).
//
// nn.GroupNorm(8, C) on the reference's hot path (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214; eps =
// 1e-5, affine) never runs as a kernel of its own here (stats_common.h): the PRODUCER of a tensor leaves per-channel
// fixed-point (sum, sum of squares) totals, the CONSUMER derives scale = rstd * gamma, shift = beta - mean * rstd * gamma in
// its prologue and applies x * scale + shift (+SiLU) while staging its input.  This file holds
//   chan_total_kernel  the totals of tensors no MFMA convolution produced (in_conv output, bilinear 2x outputs,
//                      unfolded ConvTranspose outputs): one read of the tensor, HBM-bound.
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GN_THREADS = 256;

// grid (rows, B): block `row` sums a contiguous pixel range of sample b per channel in fp64 (products of fp32 values
// are exact in fp64), rounds the block's sums to fp32 and adds them to the totals with exact integer atomics (stats_common.h).
__global__ __launch_bounds__(GN_THREADS)
void chan_total_kernel(const float* __restrict__ src, stat_word* __restrict__ tot, int rep, int bs, int HW, int C, int rows, int blocked) {
    extern __shared__ double red[];               // [ppi][C][2] doubles, then the block accumulators of stat_publish
    const int CQ = C >> 2;
    const int ppi = GN_THREADS / CQ;
    const int tid = threadIdx.x;
    const int b = blockIdx.y, row = blockIdx.x;
    const int pl = tid / CQ, q = tid - pl * CQ;
    const int per = (HW + rows - 1) / rows;
    const int p0 = row * per, p1 = min(HW, p0 + per);
    if (pl < ppi) {
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        for (int p = p0 + pl; p < p1; p += ppi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + act_index(blocked, b, C, HW, p, q * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; s[e] += d; ss[e] = fma(d, d, ss[e]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[((size_t)pl * C + q * 4 + e) * 2 + 0] = s[e];
            red[((size_t)pl * C + q * 4 + e) * 2 + 1] = ss[e];
        }
    }
    // fixed-order fp64 sum over the pixel lanes, rounded once to fp32 (stat_publish's first barrier publishes `red`)
    auto fold = [&](int i) {
        const int which = i / C, c = i - which * C;
        double t = 0;
        for (int l = 0; l < ppi; ++l) t += red[((size_t)l * C + c) * 2 + which];
        return (float)t;
    };
    stat_publish(tot, b, C, bs, rep, row % rep, 0, C, fold, reinterpret_cast<stat_word*>(red + (size_t)ppi * C * 2), tid, GN_THREADS);
}

int chan_partial_rows(int HW, int C) {
    const int ppi = GN_THREADS / (C / 4);
    int r = HW / (ppi * 8);
    if (r < 1) r = 1;
    if (r > 128) r = 128;
    return r;
}

hipError_t chan_total_launch(const float* src, stat_word* tot, int rep, int bs, int B, int HW, int C, int rows, int blocked, hipStream_t s) {
    if (C % 4 || C / 4 > GN_THREADS) return hipErrorInvalidValue;
    const int ppi = GN_THREADS / (C / 4);
    const size_t lds = (size_t)ppi * C * 2 * sizeof(double) + (size_t)(C + 2) * STAT_WORDS * sizeof(stat_word);
    hipLaunchKernelGGL(chan_total_kernel, dim3(rows, B), dim3(GN_THREADS), lds, s, src, tot, rep, bs, HW, C, rows, blocked);
    return hipGetLastError();
}

}  // namespace midd
