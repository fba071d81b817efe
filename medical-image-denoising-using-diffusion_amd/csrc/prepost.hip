// Pre/post-processing either side of the sampler, on the device:
//   * Pillow's 8-bit bicubic resample (`transforms.Resize(..., BICUBIC)` / `Image.resize(size, Image.BICUBIC)` on an
//     'L' image: Backend/run.py:146,198; cddpmModels.py:488,502), bit for bit: separable, horizontal pass first,
//     coefficients normalised in double precision and rounded to 22-bit fixed point, int32 accumulation from 1 << 21,
//     arithmetic shift, clip, uint8 between the passes.  The coefficient tables are computed ON THE DEVICE in fp64
//     with FMA contraction off (same IEEE operations as Pillow's C), so the whole call is asynchronous.
//   * ToTensor (uint8 / 255 in fp32) and its inverse `(clamp(x, 0, 1) * 255).astype(uint8)` (run.py:107,145).
//   * PSNR / SSIM of `compute_metrics` (DDIMModel.py:290-300; skimage defaults: 7x7 uniform window, K1 = .01,
//     K2 = .03, sample covariance, mean over the interior), fp64, fixed summation order.
// Checker: oracle/resize_oracle.py (pinned against Pillow itself on the CPU); tests/test_gpu_prepost.py.
#include "midd_internal.h"
#include <cstdint>

namespace midd {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double bicubic_w(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
    if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
    return 0.0;
}

// one thread per output index: bounds[xx] = (first tap, taps), kk[xx][0..ksize) fixed-point coefficients
__global__ void resize_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
#pragma clang fp contract(off)
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_w((x + xmin - center + 0.5) * ss);
    int* k = kk + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
        int q = 0;
        if (x < xmax) {
            double w = bicubic_w((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            q = (w < 0.0) ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
        }
        k[x] = q;
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ uint8_t clip8(int acc) {
    const int v = acc >> PRECISION_BITS;                 // arithmetic shift, as Pillow's lookup of (in >> PRECISION_BITS)
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// rows = n * h rows of sw pixels -> rows of dw pixels
__global__ void resample_h_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long rows, int sw, int dw,
                                     const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * dw) return;
    const long row = idx / dw;
    const int xx = (int)(idx - row * dw);
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* s = src + row * sw + xmin;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < xmax; ++x) acc += (int)s[x] * k[x];
    dst[idx] = clip8(acc);
}

// n images of sh x w -> dh x w
__global__ void resample_v_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int sh, int dh, int w,
                                     const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * dh * w) return;
    const int x = (int)(idx % w);
    const long t = idx / w;
    const int yy = (int)(t % dh);
    const int img = (int)(t / dh);
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    const uint8_t* s = src + ((size_t)img * sh + ymin) * w + x;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < ymax; ++y) acc += (int)s[(size_t)y * w] * k[y];
    dst[idx] = clip8(acc);
}

__global__ void u8_to_unit_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = __fdiv_rn((float)src[i], 255.0f);
}

__global__ void unit_to_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float v = src[i];
    v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    dst[i] = (uint8_t)__fmul_rn(v, 255.0f);             // truncation, as ndarray.astype('uint8') on [0, 255]
}

static int resize_ksize(int in_size, int out_size) {
    const double scale = (double)in_size / (double)out_size;
    const double support = 2.0 * (scale < 1.0 ? 1.0 : scale);
    return (int)ceil(support) * 2 + 1;
}

// workspace: [h bounds 2*dw][h kk dw*ksize_h][v bounds 2*dh][v kk dh*ksize_v] ints, then the n*sh*dw intermediate
size_t resize_workspace_bytes(int n, int sw, int sh, int dw, int dh) {
    const size_t ints = 2 * (size_t)dw + (size_t)dw * resize_ksize(sw, dw) + 2 * (size_t)dh + (size_t)dh * resize_ksize(sh, dh);
    return ((ints * sizeof(int) + 255) & ~(size_t)255) + (size_t)n * sh * dw + 256;
}

hipError_t resize_bicubic_u8_launch(const uint8_t* src, int n, int sw, int sh, uint8_t* dst, int dw, int dh, void* ws, hipStream_t s) {
    if (n < 1 || sw < 1 || sh < 1 || dw < 1 || dh < 1) return hipErrorInvalidValue;
    const int kh = resize_ksize(sw, dw), kv = resize_ksize(sh, dh);
    int* hb = reinterpret_cast<int*>(ws);
    int* hk = hb + 2 * (size_t)dw;
    int* vb = hk + (size_t)dw * kh;
    int* vk = vb + 2 * (size_t)dh;
    const size_t ints = 2 * (size_t)dw + (size_t)dw * kh + 2 * (size_t)dh + (size_t)dh * kv;
    uint8_t* mid = reinterpret_cast<uint8_t*>(ws) + ((ints * sizeof(int) + 255) & ~(size_t)255);
    const uint8_t* cur = src;
    int cw = sw;
    const bool need_h = dw != sw, need_v = dh != sh;
    if (!need_h && !need_v) return hipMemcpyAsync(dst, src, (size_t)n * sh * sw, hipMemcpyDeviceToDevice, s);
    if (need_h) {
        hipLaunchKernelGGL(resize_coeffs_kernel, dim3((dw + 127) / 128), dim3(128), 0, s, sw, dw, kh, hb, hk);
        uint8_t* out = need_v ? mid : dst;
        const long total = (long)n * sh * dw;
        hipLaunchKernelGGL(resample_h_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cur, out, (long)n * sh, sw, dw, hb, hk, kh);
        cur = out; cw = dw;
    }
    if (need_v) {
        hipLaunchKernelGGL(resize_coeffs_kernel, dim3((dh + 127) / 128), dim3(128), 0, s, sh, dh, kv, vb, vk);
        const long total = (long)n * dh * cw;
        hipLaunchKernelGGL(resample_v_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cur, dst, n, sh, dh, cw, vb, vk, kv);
    }
    return hipGetLastError();
}

hipError_t u8_to_unit_launch(const uint8_t* src, float* dst, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(u8_to_unit_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, dst, count);
    return hipGetLastError();
}

hipError_t unit_to_u8_launch(const float* src, uint8_t* dst, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(unit_to_u8_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, dst, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ metrics
constexpr int MET_ROWS = 4;         // image rows per block
constexpr int MET_THREADS = 256;

__device__ __forceinline__ double clamp01(float v) { return (double)(v < 0.f ? 0.f : (v > 1.f ? 1.f : v)); }

// grid (ceil(h / MET_ROWS), n): per block the squared error of its rows and the SSIM sum of its interior pixels
__global__ __launch_bounds__(MET_THREADS)
void metrics_partial_kernel(const float* __restrict__ target, const float* __restrict__ pred, int h, int w, double* __restrict__ part) {
    __shared__ double red[MET_THREADS][2];
    const int img = blockIdx.y, y0 = blockIdx.x * MET_ROWS;
    const float* t = target + (size_t)img * h * w;
    const float* p = pred + (size_t)img * h * w;
    const double c1 = 0.01 * 0.01, c2 = 0.03 * 0.03, cov_norm = 49.0 / 48.0;
    double se = 0.0, ss = 0.0;
    const int rows = min(MET_ROWS, h - y0);
    for (int i = threadIdx.x; i < rows * w; i += MET_THREADS) {
        const int y = y0 + i / w, x = i % w;
        const double a = clamp01(t[(size_t)y * w + x]), b = clamp01(p[(size_t)y * w + x]);
        se += (a - b) * (a - b);
        if (y >= 3 && y < h - 3 && x >= 3 && x < w - 3) {
            double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx) {
                    const double u = clamp01(t[(size_t)(y + dy) * w + x + dx]), v = clamp01(p[(size_t)(y + dy) * w + x + dx]);
                    sx += u; sy += v; sxx += u * u; syy += v * v; sxy += u * v;
                }
            const double ux = sx / 49.0, uy = sy / 49.0;
            const double vx = cov_norm * (sxx / 49.0 - ux * ux), vy = cov_norm * (syy / 49.0 - uy * uy);
            const double vxy = cov_norm * (sxy / 49.0 - ux * uy);
            ss += ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
        }
    }
    red[threadIdx.x][0] = se; red[threadIdx.x][1] = ss;
    __syncthreads();
    for (int off = MET_THREADS / 2; off > 0; off >>= 1) {      // fixed tree: deterministic
        if (threadIdx.x < off) { red[threadIdx.x][0] += red[threadIdx.x + off][0]; red[threadIdx.x][1] += red[threadIdx.x + off][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)img * gridDim.x + blockIdx.x) * 2;
        o[0] = red[0][0]; o[1] = red[0][1];
    }
}

// one thread per image: out[img] = (psnr, ssim)
__global__ void metrics_final_kernel(const double* __restrict__ part, int blocks, int n, int h, int w, double* __restrict__ out) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= n) return;
    double se = 0.0, ss = 0.0;
    for (int i = 0; i < blocks; ++i) { se += part[((size_t)img * blocks + i) * 2]; ss += part[((size_t)img * blocks + i) * 2 + 1]; }
    const double mse = se / ((double)h * w);
    out[2 * img] = 10.0 * log10(1.0 / mse);
    out[2 * img + 1] = ss / ((double)(h - 6) * (w - 6));
}

size_t metrics_workspace_bytes(int n, int h) { return (size_t)n * ((h + MET_ROWS - 1) / MET_ROWS) * 2 * sizeof(double) + 256; }

hipError_t metrics_launch(const float* target, const float* pred, int n, int h, int w, double* out, void* ws, hipStream_t s) {
    if (n < 1 || h < 7 || w < 7) return hipErrorInvalidValue;       // the 7x7 SSIM window must fit (skimage raises too)
    const int blocks = (h + MET_ROWS - 1) / MET_ROWS;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(blocks, n), dim3(MET_THREADS), 0, s, target, pred, h, w, part);
    hipLaunchKernelGGL(metrics_final_kernel, dim3((n + 63) / 64), dim3(64), 0, s, part, blocks, n, h, w, out);
    return hipGetLastError();
}

}  // namespace midd
