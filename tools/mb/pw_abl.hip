// Round-3 ablation harness for the HBM-bound ends (run on the GPU box): which part of in_conv / out_conv / resize costs the time.
//   for n in 0 1 2 3 4 11 12 13; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPW_ABL=$n tools/mb/pw_abl.hip -o tools/mb/pw_abl_$n; done
// Each binary times the three kernels at B = 4, 256x256 (the sizes of a half-batch program), 20 repetitions.
#include "../../medical-image-denoising-using-diffusion_amd/csrc/pointwise.hip"
#include <cstdio>
#ifndef PW_BLOCKED
#define PW_BLOCKED 1      // activation layout of the split-fp16 plans (midd_internal.h)
#endif
#include <vector>
using namespace midd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int B = 4, H = 256, W = 256, C = 48;
    const size_t act = (size_t)B * H * W * C;
    float *x, *cond, *w, *bias, *out, *gam, *bet, *img, *lo;
    stat_word* tot;
    CK(hipMalloc(&x, (size_t)B * H * W * 4)); CK(hipMalloc(&cond, (size_t)B * H * W * 4));
    CK(hipMalloc(&w, 1 << 20)); CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&gam, 4096)); CK(hipMalloc(&bet, 4096));
    CK(hipMalloc(&out, act * 4)); CK(hipMalloc(&img, (size_t)B * H * W * 4)); CK(hipMalloc(&lo, act));
    CK(hipMalloc(&tot, 1 << 20)); CK(hipMemset(tot, 0, 1 << 20));
    std::vector<float> h(1 << 18, 0.01f);
    CK(hipMemcpy(w, h.data(), 1 << 20, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemcpy(gam, h.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(bet, h.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (size_t)B * H * W * 4)); CK(hipMemset(cond, 0, (size_t)B * H * W * 4)); CK(hipMemset(out, 0, act * 4)); CK(hipMemset(img, 0, (size_t)B * H * W * 4));
    CK(hipMemset(lo, 0, act));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& fn, double bytes) -> int {
        float total = 0; const int reps = 20;
        for (int r = -3; r < reps; ++r) {
            CK(hipEventRecord(e0)); CK(fn()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 0) total += ms;
        }
        printf("PW_ABL=%d blocked=%d %-10s %7.1f us  %5.2f TB/s\n", PW_ABL, PW_BLOCKED, name, total / reps * 1e3, bytes / (total / reps * 1e-3) * 1e-12);
        return 0;
    };
    if (timeit("in_conv", [&] { return in_conv_launch(x, cond, w, bias, out, tot, 4, 6, B, 1, H, W, C, PW_BLOCKED, 0); }, act * 4.0)) return 1;
    OutConvArgs a{};
    a.src = out; a.gn_tot = tot; a.stat_rep = 4; a.gn_bs = 6; a.gn_gamma = gam; a.gn_beta = bet; a.gn_eps = 1e-5f; a.w = w; a.bias = bias;
    a.blocked = PW_BLOCKED; a.B = B; a.H = H; a.W = W; a.C = C; a.ic = 1; a.eps_out = nullptr; a.x = img; a.noise = nullptr; a.c1 = 1.f; a.c2 = 0.1f; a.c3 = 0.f; a.clamp_eps = 0;
    if (timeit("out_conv", [&] { return out_conv_launch(a, 0); }, act * 4.0)) return 1;
    if (timeit("resize256", [&] { return resize_bilinear_launch(reinterpret_cast<const float*>(lo), out, tot, 4, 6, B, 128, 128, C, 256, 256, PW_BLOCKED, 0); }, act * 5.0)) return 1;
    if (timeit("resize128", [&] { return resize_bilinear_launch(reinterpret_cast<const float*>(lo), out, tot, 4, 12, B, 64, 64, 96, 128, 128, PW_BLOCKED, 0); }, (double)B * 128 * 128 * 96 * 5.0)) return 1;
    return 0;
}
