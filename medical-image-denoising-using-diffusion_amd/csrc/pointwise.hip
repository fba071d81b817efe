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
                                                  int replica, int tid) {
    if (active) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            scratch[(size_t)(cgrp * NV + e) * ppi + pl] = sum[e];
            scratch[(size_t)(C + cgrp * NV + e) * ppi + pl] = sq[e];
        }
    }
    auto fold = [&](int i) {
        double t = 0;
        for (int l = 0; l < ppi; ++l) t += (double)scratch[(size_t)i * ppi + l];
        return (float)t;
    };
    stat_publish(tot, b, C, bs, rep, replica, 0, C, fold, acc, tid, 256);
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
                    int ic, int H, int W, int Cout, int rows) {
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
                        const float v = plane[(size_t)gy * W + gx];
                        const float* wr = &wl[((dy * 3 + dx) * 2 * ic + ci) * Cout + cg * 16];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] += v * *reinterpret_cast<const f32x4*>(wr + k * 4);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<f32x4*>(out + ((size_t)b * HW + p) * Cout + cg * 16 + k * 4) = acc[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) { ssum[k * 4 + e] += acc[k][e]; ssq[k * 4 + e] += acc[k][e] * acc[k][e]; }
            }
        }
    }
    if (tot == nullptr) return;
    float* const scratch = wl + nw + Cout;
    stat_word* const acc_lds = reinterpret_cast<stat_word*>(scratch + 2 * (size_t)Cout * ppi + (((nw + Cout) & 1) ? 1 : 0));   // 8-byte aligned
    pointwise_publish<16>(ssum, ssq, active, pl, ppi, cg, Cout, scratch, acc_lds, tot, b, bs, rep, row % rep, tid);
}

hipError_t in_conv_launch(const float* x, const float* cond, const float* w, const float* bias, float* out,
                          stat_word* tot, int rep, int bs, int B, int ic, int H, int W, int Cout, hipStream_t s) {
    if (Cout % 16 || Cout / 16 > 256) return hipErrorInvalidValue;
    const int ppi = 256 / (Cout / 16);
    const int rows = pointwise_rows(H * W, ppi);
    const size_t lds = (size_t)(9 * 2 * ic * Cout + Cout + 2 * Cout * ppi + 2) * sizeof(float) + (size_t)(Cout + 2) * STAT_WORDS * sizeof(stat_word);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(in_conv_kernel, dim3(rows, B), dim3(256), lds, s, x, cond, w, bias, out, tot, rep, bs, ic, H, W, Cout, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ out_conv (+ sampler update)
// Workgroup = 16x16 output pixels.  The 18x18 halo is staged through LDS in 16-channel chunks
// with GroupNorm-apply + SiLU on the way in; pixel stride 20 floats keeps the float4 reads of
// 16 neighbouring lanes on distinct banks.
constexpr int OC_T = 16;
constexpr int OC_I = OC_T + 2;
constexpr int OC_PS = 20;         // padded pixel stride in floats

__global__ __launch_bounds__(256)
void out_conv_kernel(const OutConvArgs a) {
    __shared__ float tile[OC_I * OC_I * OC_PS];
    extern __shared__ float wl[];                 // [ic][9][C], then [2][C] GroupNorm scale / shift of this sample
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    float* const gnp = wl + a.ic * 9 * a.C;
    const int tiles_x = (a.W + OC_T - 1) / OC_T, tiles_y = (a.H + OC_T - 1) / OC_T;
    const int b = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - b * tiles_x * tiles_y;
    const int oy0 = (trem / tiles_x) * OC_T, ox0 = (trem % tiles_x) * OC_T;
    const int C = a.C;
    for (int i = tid; i < a.ic * 9 * C; i += 256) wl[i] = a.w[i];
    // (second source: C1 = 0, never read; a literal nullptr there crashes hipcc 7.2's inliner)
    gn_prologue_lds(a.gn_tot, C, a.gn_bs, a.gn_tot, 0, 1, a.stat_rep, a.gn_gamma, a.gn_beta, a.gn_eps, 1.0 / ((double)a.H * a.W * (C / GN_GROUPS_C)), b, 1.0f, gnp, tid, 256);

    float acc[4] = {0.f, 0.f, 0.f, 0.f};          // ic <= 4 output channels
    for (int c0 = 0; c0 < C; c0 += 16) {
        __syncthreads();
        for (int slot = tid; slot < OC_I * OC_I * 4; slot += 256) {
            const int pix = slot >> 2, q = slot & 3;
            const int iy = pix / OC_I, ix = pix - iy * OC_I;
            const int gy = oy0 + iy - 1, gx = ox0 + ix - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                v = *reinterpret_cast<const f32x4*>(a.src + ((size_t)(b * a.H + gy) * a.W + gx) * C + c0 + q * 4);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(gnp + c0 + q * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(gnp + C + c0 + q * 4);
                v = v * sc + sh;
                v.x = silu_pw(v.x); v.y = silu_pw(v.y); v.z = silu_pw(v.z); v.w = silu_pw(v.w);
            }
            *reinterpret_cast<f32x4*>(&tile[pix * OC_PS + q * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const float* px = &tile[((ty + dy) * OC_I + tx + dx) * OC_PS];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
                for (int oc = 0; oc < a.ic; ++oc) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[(oc * 9 + tap) * C + c0 + q * 4]);
                    acc[oc] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                }
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= a.H || ox >= a.W) return;
    for (int oc = 0; oc < a.ic; ++oc) {
        const size_t o = (((size_t)b * a.ic + oc) * a.H + oy) * a.W + ox;
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
    hipLaunchKernelGGL(out_conv_kernel, dim3(a.B * tiles), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ bilinear resize (NHWC)
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

hipError_t resize_bilinear_launch(const float* src, float* dst, stat_word* tot, int rep, int bs, int B, int H, int W, int C, int OH, int OW, hipStream_t s) {
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
                           float* __restrict__ dst, int B, int H, int W, int Cin, int Cout) {
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
            const float* ip = src + ((size_t)(b * H + (ny >> 1)) * W + (nx >> 1)) * Cin;
            const float* wp = w + ((size_t)(ky * 4 + kx) * Cin) * Cout + cq * 4;
            for (int ci = 0; ci < Cin; ++ci)
                acc += ip[ci] * *reinterpret_cast<const f32x4*>(wp + (size_t)ci * Cout);
        }
    }
    *reinterpret_cast<f32x4*>(dst + (size_t)pix * Cout + cq * 4) = acc;
}

hipError_t conv_transpose_launch(const float* src, const float* w, const float* bias, float* dst,
                                 int B, int H, int W, int Cin, int Cout, hipStream_t s) {
    if (Cout % 4) return hipErrorInvalidValue;
    const long total = (long)B * 4 * H * W * (Cout / 4);
    hipLaunchKernelGGL(conv_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       src, w, bias, dst, B, H, W, Cin, Cout);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ helpers
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C) {
    const long total = (long)B * H * W * C;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const long pix = gid / C;
    const long hw = pix % ((long)H * W);
    const int b = (int)(pix / ((long)H * W));
    dst[((size_t)b * C + c) * H * W + hw] = src[gid];
}

hipError_t nhwc_to_nchw_launch(const float* src, float* dst, int B, int H, int W, int C, hipStream_t s) {
    const long total = (long)B * H * W * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, B, H, W, C);
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
