// GroupNorm(8, C) statistics for NHWC fp32 activations (torch.cat of two sources allowed).
//
// Replaces the statistics half of nn.GroupNorm(8, C) on the reference's hot path
// (/root/reference/Backend/DDIM/DDIMModel.py:116,121,139,214; eps = 1e-5, affine).  The
// normalise/affine half is never a separate pass: this kernel emits, per (sample, channel),
//     scale = rstd * gamma            shift = beta - mean * rstd * gamma
// and the consuming convolution applies x*scale+shift (+SiLU) while staging its input.
//
// Two launches, both deterministic (fixed summation order, no atomics):
//   gn_partial : grid (nsplit, B); each block sums a contiguous pixel range in fp64
//                (products of fp32 values are exact in fp64, so E[x^2]-mean^2 does not cancel)
//   gn_finalize: grid (B); folds the nsplit partials in order, writes scale/shift.
// HBM-bound: one read of the tensor.
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GN_THREADS = 256;
// GN_GROUPS_ (8, nn.GroupNorm(8, C)) is defined in midd_internal.h

__global__ __launch_bounds__(GN_THREADS)
void gn_partial_kernel(const GnArgs a) {
    extern __shared__ double red[];               // [ppi][C][2]
    const int C = a.C0 + a.C1;
    const int CQ = C >> 2;
    const int ppi = GN_THREADS / CQ;              // pixels per iteration
    const int tid = threadIdx.x;
    const int b = blockIdx.y, split = blockIdx.x;
    const int pl = tid / CQ, q = tid - pl * CQ;
    const int per = (a.HW + a.nsplit - 1) / a.nsplit;
    const int p0 = split * per;
    const int p1 = min(a.HW, p0 + per);

    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (pl < ppi) {
        const int ch = q * 4;
        const float* src; int Cs, coff;
        if (ch < a.C0) { src = a.src0; Cs = a.C0; coff = ch; }
        else           { src = a.src1; Cs = a.C1; coff = ch - a.C0; }
        const float* base = src + (size_t)b * a.HW * Cs + coff;
        for (int p = p0 + pl; p < p1; p += ppi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * Cs);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)v[e];
                s[e] += d;
                ss[e] = fma(d, d, ss[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[((size_t)pl * C + ch + e) * 2 + 0] = s[e];
            red[((size_t)pl * C + ch + e) * 2 + 1] = ss[e];
        }
    }
    __syncthreads();
    // per-channel fold over the pixel lanes (fixed order), then per-group fold over channels
    for (int c = tid; c < C; c += GN_THREADS) {
        double cs = 0, css = 0;
        for (int l = 0; l < ppi; ++l) { cs += red[((size_t)l * C + c) * 2]; css += red[((size_t)l * C + c) * 2 + 1]; }
        red[(size_t)c * 2] = cs; red[(size_t)c * 2 + 1] = css;      // lane-0 row is only read by its own channel thread
    }
    __syncthreads();
    if (tid < GN_GROUPS_) {
        const int cg = C / GN_GROUPS_;
        double gs = 0, gss = 0;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { gs += red[(size_t)c * 2]; gss += red[(size_t)c * 2 + 1]; }
        double* o = a.partial + (((size_t)b * a.nsplit + split) * GN_GROUPS_ + tid) * 2;
        o[0] = gs; o[1] = gss;
    }
}

__global__ __launch_bounds__(GN_THREADS)
void gn_finalize_kernel(const GnArgs a) {
    __shared__ float s_mean_rstd[GN_GROUPS_][2];
    const int C = a.C0 + a.C1;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < GN_GROUPS_) {
        double gs = 0, gss = 0;
        for (int sp = 0; sp < a.nsplit; ++sp) {
            const double* p = a.partial + (((size_t)b * a.nsplit + sp) * GN_GROUPS_ + tid) * 2;
            gs += p[0]; gss += p[1];
        }
        const double n = (double)a.HW * (C / GN_GROUPS_);
        const double mean = gs / n;
        double var = gss / n - mean * mean;          // biased variance, as torch's group_norm
        if (var < 0) var = 0;
        s_mean_rstd[tid][0] = (float)mean;
        s_mean_rstd[tid][1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    const int cg = C / GN_GROUPS_;
    for (int c = tid; c < C; c += GN_THREADS) {
        const int g = c / cg;
        const float sc = s_mean_rstd[g][1] * a.gamma[c];
        a.scale[(size_t)b * C + c] = sc;
        a.shift[(size_t)b * C + c] = a.beta[c] - s_mean_rstd[g][0] * sc;
    }
}

int gn_pick_nsplit(int B, int HW, int C) {
    // Independent of B on purpose: the partial-sum order (hence the bits of mean/rstd) of an
    // image must not change with the batch it is processed in (multi-GPU shards == single GPU).
    (void)B;
    const int ppi = GN_THREADS / (C / 4);
    int ns = HW / (ppi * 8);          // >= 8 pixels per lane-row per block
    if (ns < 1) ns = 1;
    if (ns > 128) ns = 128;
    return ns;
}

// ------------------------------------------------------------------------------ fused-statistics path
// Per-channel partial sums [B][rows][2][C] come from the producing convolution's epilogue
// (conv_mfma_*.hip) or, for tensors no MFMA conv produced (in_conv output, bilinear 2x
// outputs), from chan_partial_kernel.  gn_from_partial_kernel folds them in a fixed order in
// fp64 and emits scale/shift; one block per (group, sample).  Channels are resolved one by one
// to (source, local channel), so groups may straddle the torch.cat seam (cddpm up path).
constexpr int GNF_THREADS = 256;              // many row lanes: the row loop is L2-latency bound
__global__ __launch_bounds__(GNF_THREADS)
void gn_from_partial_kernel(const GnFromPartialArgs a) {
    __shared__ double red[GNF_THREADS][2];
    __shared__ float s_mr[2];
    const int C = a.C0 + a.C1;
    const int cg = C / GN_GROUPS_;
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nrl = GNF_THREADS / cg;                 // row lanes
    const int ci = tid % cg, rl = tid / cg;
    double s1 = 0, s2 = 0;
    float gam = 0.f, bet = 0.f;                        // requested up front: off the dependent chain at the end
    if (tid < cg) { gam = a.gamma[g * cg + tid]; bet = a.beta[g * cg + tid]; }
    if (rl < nrl) {
        const int c = g * cg + ci;
        const float* p; int rows, Cs, cl;
        if (c < a.C0) { p = a.part0; rows = a.rows0; Cs = a.C0; cl = c; }
        else          { p = a.part1; rows = a.rows1; Cs = a.C1; cl = c - a.C0; }
        p += (size_t)b * rows * 2 * Cs + cl;
        // 8 rows in flight per thread (the loads are L2-latency bound); fixed summation order
        int r = rl;
        for (; r + 7 * nrl < rows; r += 8 * nrl) {
            float v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v1[u] = p[(size_t)(r + u * nrl) * 2 * Cs];
                v2[u] = p[(size_t)(r + u * nrl) * 2 * Cs + Cs];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
        }
        for (; r < rows; r += nrl) {
            s1 += (double)p[(size_t)r * 2 * Cs];
            s2 += (double)p[(size_t)r * 2 * Cs + Cs];
        }
    }
    red[tid][0] = s1; red[tid][1] = s2;
    __syncthreads();
    if (cg <= 64) {
        // latency path (every configuration of this model): the first wave folds the row lanes per channel,
        // then the channels with fixed-order cross-lane adds; every lane ends with the group totals, so there
        // is no second barrier and no broadcast through LDS
        if (tid < 64) {
            double t1 = 0, t2 = 0;
            if (tid < cg)
                for (int l = 0; l < nrl; ++l) { t1 += red[l * cg + tid][0]; t2 += red[l * cg + tid][1]; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { t1 += __shfl_xor(t1, off, 64); t2 += __shfl_xor(t2, off, 64); }
            if (tid < cg) {
                const double n = (double)a.HW * cg;
                const double mean = t1 / n;
                double var = t2 / n - mean * mean;
                if (var < 0) var = 0;
                const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
                const int c = g * cg + tid;
                const float sc = rstd * gam;
                a.scale[(size_t)b * C + c] = sc;
                a.shift[(size_t)b * C + c] = bet - (float)mean * sc;
            }
        }
        return;
    }
    if (tid < cg) {                                    // fold row lanes, fixed order
        double t1 = 0, t2 = 0;
        for (int l = 0; l < nrl; ++l) { t1 += red[l * cg + tid][0]; t2 += red[l * cg + tid][1]; }
        red[tid][0] = t1; red[tid][1] = t2;
    }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0, t2 = 0;
        for (int c = 0; c < cg; ++c) { t1 += red[c][0]; t2 += red[c][1]; }
        const double n = (double)a.HW * cg;
        const double mean = t1 / n;
        double var = t2 / n - mean * mean;
        if (var < 0) var = 0;
        s_mr[0] = (float)mean;
        s_mr[1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    if (tid < cg) {
        const int c = g * cg + tid;
        const float sc = s_mr[1] * gam;
        a.scale[(size_t)b * C + c] = sc;
        a.shift[(size_t)b * C + c] = bet - s_mr[0] * sc;
    }
}

hipError_t gn_from_partial_launch(const GnFromPartialArgs& a, hipStream_t s) {
    const int C = a.C0 + a.C1;
    if (C % GN_GROUPS_ || C / GN_GROUPS_ > GNF_THREADS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_from_partial_kernel, dim3(GN_GROUPS_, a.B), dim3(GNF_THREADS), 0, s, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(GN_THREADS)
void chan_partial_kernel(const float* __restrict__ src, float* __restrict__ part, int HW, int C, int rows) {
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
        float* o = part + ((size_t)(b * rows + row) * 2) * C + c;
        o[0] = (float)cs; o[C] = (float)css;
    }
}

int chan_partial_rows(int HW, int C) {
    const int ppi = GN_THREADS / (C / 4);
    int r = HW / (ppi * 8);
    if (r < 1) r = 1;
    if (r > 128) r = 128;
    return r;
}

hipError_t chan_partial_launch(const float* src, float* part, int B, int HW, int C, int rows, hipStream_t s) {
    if (C % 4 || C / 4 > GN_THREADS) return hipErrorInvalidValue;
    const int ppi = GN_THREADS / (C / 4);
    const size_t lds = (size_t)ppi * C * 2 * sizeof(double);
    hipLaunchKernelGGL(chan_partial_kernel, dim3(rows, B), dim3(GN_THREADS), lds, s, src, part, HW, C, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ pre-activation pass
// One thread = 4 channels of one pixel.  Same arithmetic as the conv's in-kernel transform (conv_mfma_f16x3.hip):
// v = x * (16 sc) + 16 sh, SiLU on the 16x-scaled value, hi = fp16(v), lo = fp16(v - hi).
__global__ __launch_bounds__(256)
void preact_kernel(const float* __restrict__ src0, int C0, const float* __restrict__ src1, int C1,
                   const float* __restrict__ scale, const float* __restrict__ shift, int silu, int planar,
                   unsigned* __restrict__ out, int B, int HW) {
    const int C = C0 + C1, CQ = C >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * HW * CQ) return;
    const int q = (int)(idx % CQ);
    const size_t pix = idx / CQ;                       // b * HW + p
    const int b = (int)(pix / HW);
    const int c = q * 4;
    const f32x4 x = (c < C0) ? *reinterpret_cast<const f32x4*>(src0 + pix * C0 + c)
                             : *reinterpret_cast<const f32x4*>(src1 + pix * C1 + (c - C0));
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c) * 16.0f;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (size_t)b * C + c) * 16.0f;
    f32x4 v = x * sc + sh;
    if (silu) {
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
    if (planar) {       // per pixel and 16-channel block: 16 high halves (32 B), then 16 low halves (PRO_PRE_DMA)
        char* o = reinterpret_cast<char*>(out) + (pix * (C >> 4) + (c >> 4)) * 64 + (c & 15) * 2;
        *reinterpret_cast<uint2*>(o) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
        *reinterpret_cast<uint2*>(o + 32) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
    } else {            // one word per element: hi | lo << 16 (PRO_PRE)
        *reinterpret_cast<uint4*>(out + pix * C + c) = make_uint4(hb[0] | ((unsigned)lb[0] << 16), hb[1] | ((unsigned)lb[1] << 16),
                                                                 hb[2] | ((unsigned)lb[2] << 16), hb[3] | ((unsigned)lb[3] << 16));
    }
}

hipError_t preact_launch(const float* src0, int C0, const float* src1, int C1, const float* scale, const float* shift,
                         int silu, int planar, unsigned* out, int B, int HW, hipStream_t s) {
    if (C0 % 4 || C1 % 4 || (planar && (C0 + C1) % 16)) return hipErrorInvalidValue;
    const size_t total = (size_t)B * HW * ((C0 + C1) / 4);
    hipLaunchKernelGGL(preact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       src0, C0, src1, C1, scale, shift, silu, planar, out, B, HW);
    return hipGetLastError();
}

hipError_t gn_stats_launch(const GnArgs& a, hipStream_t s) {
    const int C = a.C0 + a.C1;
    if (C % 8 || C / 4 > GN_THREADS) return hipErrorInvalidValue;    // 8 groups, float4 loads
    if (a.C0 % 4 || a.C1 % 4) return hipErrorInvalidValue;
    const int ppi = GN_THREADS / (C / 4);
    const size_t lds = (size_t)ppi * C * 2 * sizeof(double);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(a.nsplit, a.B), dim3(GN_THREADS), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.B), dim3(GN_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace midd
