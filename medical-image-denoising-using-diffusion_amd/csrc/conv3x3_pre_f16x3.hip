// 3x3 stride-1 convolution in f16x3 arithmetic on PRE-ACTIVATED input (ConvArgs::prologue == PRO_PRE_DMA, opt-in).
//
// The general kernel (conv_mfma_f16x3.hip) lands raw fp32 activations in LDS and every workgroup normalises /
// activates / splits the halo tile it staged -- Cout/48 x 1.4 times per element, ~40 % of its non-MFMA
// instructions, plus a repack pass and a barrier per 16-channel chunk.  Here the input is the output of
// preact_planar_kernel (groupnorm.hip): per pixel and 16-channel block the 16 fp16 high halves (32 B) followed by
// the 16 low halves (32 B) of 2^s * act(GroupNorm(x)).  That is exactly the MFMA image layout
// [hi|lo][halo pixel][16 fp16], so the activation path is LDS-DMA only:
//   * chunk c+1 is DMA'd into the other of two image buffers while chunk c is multiplied -- no raw landing buffer,
//     no transform, no chunk-end barrier; out-of-image halo pixels read a zero buffer;
//   * the weight ring, the counted vmcnt waits, the step barrier, the MFMA step and the epilogue are those of the
//     general kernel (tile 16 x 4*MT pixels x 48 couts, 4 waves, persistent per-sample workgroups).
// Wait protocol (D = RING-1 <= 5 weight steps in flight, 5 steps per chunk): A(c+1) is issued in step 0 of chunk c
// right after that step's weight refill, so it is older than the refills of steps 1..4 and the first step of chunk
// c+1, which waits for all but the (D-1) youngest weight groups, has it landed; in steps 1..D of chunk c it is
// younger than the group waited for and is added to the count.
#include "f16x3_common.h"
#include <cstdlib>

namespace midd {

template <int MT>
struct PreGeom {
    static constexpr int NW = 4, NTHREADS = 256, TW = 16, NT = 3;
    static constexpr int BM = NW * MT * 16, TH = BM / TW, IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int PLANE = ((NPIX * 32 + 1023) / 1024) * 1024;     // one plane (hi or lo), padded to whole DMA pieces
    static constexpr int IPP = PLANE / 1024;                             // DMA instructions per plane
    static constexpr int NINSTR = 2 * IPP;
    static constexpr int APW = NINSTR / NW;                              // per wave and chunk
    static_assert(NINSTR % NW == 0, "activation pieces must divide over the waves (counted waits)");
    static constexpr int IMG_BYTES = 2 * PLANE;
    static constexpr int WPIECES = NT * 2, PPW = (WPIECES + NW - 1) / NW, WSLICE = WPIECES * 1024;
    static constexpr int STAT_FLOATS = NW * 2 * NT * 16, ADD_FLOATS = NT * 16;
    static constexpr int FIXED = 2 * IMG_BYTES + (STAT_FLOATS + ADD_FLOATS) * 4;
    static constexpr int ring_fit = (52 * 1024 - FIXED) / WSLICE;
    static constexpr int RING = ring_fit < 2 ? 2 : (ring_fit > 6 ? 6 : ring_fit);
    static constexpr int LDS_BYTES = FIXED + RING * WSLICE;
};

template <int MT>
__global__ __launch_bounds__(256, 3)
void conv3x3_pre_f16x3_kernel(const ConvArgs a) {
    using G = PreGeom<MT>;
    constexpr int NW = G::NW, NTHREADS = G::NTHREADS, TW = G::TW, TH = G::TH, IW = G::IW, NT = G::NT;
    constexpr int PLANE = G::PLANE, IPP = G::IPP, APW = G::APW, PPW = G::PPW, WSLICE = G::WSLICE, RING = G::RING;
    constexpr int D = RING - 1, HSTEPS = 5;
    static_assert(D <= 5, "first-step wait assumes the next chunk is older than the D-1 youngest weight groups");

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const img0 = lds;                                               // two image buffers
    char* const wring = lds + 2 * G::IMG_BYTES;
    float* const stat_lds = reinterpret_cast<float*>(wring + RING * WSLICE);
    float* const add_lds = stat_lds + G::STAT_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, kq = lane >> 4;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / a.wgs_per_img;
    int trem = blockIdx.x - b * a.wgs_per_img;
    int oy0 = (trem / a.tiles_x) * TH, ox0 = (trem % a.tiles_x) * TW;
    const int Cin = a.C0, nchunks = Cin >> 4;
    const int ntiles_total = a.Cout >> 4;
    const int ntile_wg = blockIdx.y * NT;
    const int total_steps = nchunks * HSTEPS;

    // ---- weights: LDS-DMA ring (as conv_mfma_f16x3.hip) -------------------------------------------
    const char* const wbase = reinterpret_cast<const char*>(a.wpack) + (size_t)ntile_wg * 2048;
    const size_t wstep_bytes = (size_t)ntiles_total * 2048;
    const int lane16 = lane * 16;
    int wr_step = 0, wr_slot = 0;
    const char* wr_src = wbase;
    auto issue_w = [&]() {
        char* slot = wring + wr_slot * WSLICE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int piece = wave + i * NW;
            if (piece >= G::WPIECES) piece -= G::WPIECES;                 // padding duplicate: same bytes, same place
            dma16(wr_src + piece * 1024 + lane16, slot + piece * 1024);
        }
        ++wr_step; wr_src += wstep_bytes;
        if (wr_step == total_steps) { wr_step = 0; wr_src = wbase; }
        wr_slot = (wr_slot + 1 == RING) ? 0 : wr_slot + 1;
    };

    // ---- activations: DMA piece j = wave + s*NW covers plane j / IPP, pixels (j % IPP)*32 .. +31, two lanes a pixel ----
    int g_off[APW];                       // pixel index into the tensor, or -1: zero source
    int lane_off[APW];                    // plane * 32 + (lane & 1) * 16
#pragma unroll
    for (int s = 0; s < APW; ++s) lane_off[s] = ((wave + s * NW) / IPP) * 32 + (lane & 1) * 16;
    auto set_tile = [&](int t) {
        const int iy0 = (t / a.tiles_x) * TH - 1, ix0 = (t % a.tiles_x) * TW - 1;
#pragma unroll
        for (int s = 0; s < APW; ++s) {
            const int j = wave + s * NW;
            const int p = (j % IPP) * 32 + (lane >> 1);
            int off = -1;
            if (p < G::NPIX) {
                const int iy = p / IW, ix = p - iy * IW;
                const int gy = iy0 + iy, gx = ix0 + ix;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) off = (b * a.H + gy) * a.W + gx;
            }
            g_off[s] = off;
        }
    };
    const char* const abase = reinterpret_cast<const char*>(a.src0);
    const char* const zsrc = reinterpret_cast<const char*>(a.zeros);
    const unsigned pix_bytes = (unsigned)Cin * 4u;
    auto issue_a = [&](int c, char* buf) {
#pragma unroll
        for (int s = 0; s < APW; ++s) {
            const char* src = (g_off[s] >= 0) ? abase + ((unsigned)g_off[s] * pix_bytes + (unsigned)(c * 64 + lane_off[s])) : zsrc;
            dma16(src, buf + (wave + s * NW) * 1024);
        }
    };

    // ---- fragment offsets (as the general kernel, 16-channel blocks, two taps per step) -------------
    int frag_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pp = (wave * MT + mt) * 16 + p16;
        const int py = pp / TW, px = pp - py * TW;
        frag_base[mt] = (py * IW + px) * 32 + (kq & 1) * 16;
    }
    const int wfrag_off = lane * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue --------------------------------------------------------------------------------
    set_tile(trem);
    issue_a(0, img0);
#pragma unroll
    for (int i = 0; i < D; ++i) issue_w();
    for (int i = tid; i < G::STAT_FLOATS; i += NTHREADS) stat_lds[i] = 0.f;
    {
        const int trow = (a.temb != nullptr) ? a.trow[b] : 0;
        for (int i = tid; i < G::ADD_FLOATS; i += NTHREADS) {
            const int co = ntile_wg * 16 + i;
            add_lds[i] = a.bias[co] + (a.temb != nullptr ? a.temb[(size_t)trow * a.temb_stride + co] : 0.f);
        }
    }
    wait_vm_and_barrier<0>();

    char* img = img0;                      // buffer of the current chunk
    int rd_slot = 0;
    half8 xh[MT], xl[MT];
    auto load_x = [&](const int (&xo)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xh[mt] = *reinterpret_cast<const half8*>(img + xo[mt]);
            xl[mt] = *reinterpret_cast<const half8*>(img + xo[mt] + PLANE);
        }
    };
    auto mfma_step = [&]() {
        const char* wslot = wring + rd_slot * WSLICE + wfrag_off;
        rd_slot = (rd_slot + 1 == RING) ? 0 : rd_slot + 1;
        half8 wh[NT], wl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wh[nt] = *reinterpret_cast<const half8*>(wslot + nt * 2048);
            wl[nt] = *reinterpret_cast<const half8*>(wslot + nt * 2048 + 1024);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], xh[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], xl[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], xh[mt], acc[mt][nt], 0, 0, 0);
    };

    // ---- epilogue (as the general kernel) ----------------------------------------------------------
    float* const my_stat = stat_lds + wave * (2 * NT * 16) + kq * 4;
    auto epilogue = [&]() {
        f32x4 tsum[NT], tsq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            tsum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tsq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int co = (ntile_wg + nt) * 16 + kq * 4;
            const f32x4 add = *reinterpret_cast<const f32x4*>(add_lds + nt * 16 + kq * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pp = (wave * MT + mt) * 16 + p16;
                const int py = pp / TW, px = pp - py * TW;
                const int oy = oy0 + py, ox = ox0 + px;
                if (oy < a.OH && ox < a.OW) {
                    const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.Cout + co;
                    f32x4 v = acc[mt][nt] * a.out_scale + add;
                    if (a.resid != nullptr) v += *reinterpret_cast<const f32x4*>(a.resid + o);
                    *reinterpret_cast<f32x4*>(a.out + o) = v;
                    tsum[nt] += v; tsq[nt] += v * v;
                }
                acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (a.stat_tot != nullptr) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { tsum[nt][e] = row16_sum(tsum[nt][e]); tsq[nt][e] = row16_sum(tsq[nt][e]); }
                if (p16 == 0) {
                    f32x4* ps = reinterpret_cast<f32x4*>(my_stat + nt * 16);
                    f32x4* pq = reinterpret_cast<f32x4*>(my_stat + NT * 16 + nt * 16);
                    *ps = *ps + tsum[nt];
                    *pq = *pq + tsq[nt];
                }
            }
        }
    };

    // ---- tile / chunk loop ---------------------------------------------------------------------------
    for (;;) {
        const int next_tile = trem + a.wgs_per_img;
        const bool has_next_tile = next_tile < tiles_per_img;
        for (int c = 0; c < nchunks; ++c) {
            const bool more_in_tile = (c + 1 < nchunks);
            const bool more = more_in_tile || has_next_tile;
            const int next_chunk = more_in_tile ? c + 1 : 0;
            char* const other = (img == img0) ? img0 + G::IMG_BYTES : img0;
#pragma unroll
            for (int hs = 0; hs < HSTEPS; ++hs) {
                const int t0 = 2 * hs, t1 = (2 * hs + 1 < 9) ? 2 * hs + 1 : 0;     // padded half has zero weights
                const int o0 = ((t0 / 3) * IW + (t0 % 3)) * 32, o1 = ((t1 / 3) * IW + (t1 % 3)) * 32;
                const int to = (kq >> 1) ? o1 : o0;
                int xo[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xo[mt] = frag_base[mt] + to;
                const bool with_a = more && hs >= 1 && hs <= D;
                if (hs != 0) {
                    load_x(xo);            // the chunk's image is published; overlap the fragment latency with the wait
                    if (with_a) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"((D - 1) * PPW + APW), "n"(2 * MT) : "memory");
                    else        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"((D - 1) * PPW), "n"(2 * MT) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * PPW) : "memory");   // W(s) and this chunk's image landed
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (hs == 0) load_x(xo);
                issue_w();
                if (hs == 0 && more) {
                    // every wave has passed this barrier, i.e. finished reading `other` (the previous chunk's image)
                    if (!more_in_tile) set_tile(next_tile);
                    issue_a(next_chunk, other);
                }
                mfma_step();
            }
            if (!more_in_tile) {
                epilogue();
                if (has_next_tile) { trem = next_tile; oy0 = (trem / a.tiles_x) * TH; ox0 = (trem % a.tiles_x) * TW; }
            }
            img = other;
        }
        if (!has_next_tile) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight refills issued past the last step

    // ---- the workgroup's per-channel sums -> the tensor's totals (stats_common.h) -----------------------------
    if (a.stat_tot != nullptr) {
        lds_barrier();
        constexpr int ROWF = 2 * NT * 16;
        for (int i = tid; i < ROWF; i += NTHREADS) {
            const int which = i / (NT * 16), c = i - which * (NT * 16);
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < NW; ++m) t += stat_lds[m * ROWF + i];
            stat_atomic_add(a.stat_tot + (((size_t)b * a.Cout + ntile_wg * 16 + c) * 2 + which) * STAT_LIMBS, t);
        }
    }
}

template <int MT>
static hipError_t launch_pre(const ConvArgs& a0, hipStream_t s) {
    using G = PreGeom<MT>;
    ConvArgs a = a0;
    a.tiles_x = (a.OW + G::TW - 1) / G::TW;
    a.tiles_y = (a.OH + G::TH - 1) / G::TH;
    const int ny = a.Cout / (G::NT * 16);
    a.wgs_per_img = conv16_wgs_per_img(a.tiles_x * a.tiles_y, a.B, ny, a.persist_wgs);
    if ((double)a.B * a.H * a.W * a.C0 * 4.0 >= 4294967296.0) return hipErrorInvalidValue;      // 32-bit DMA offsets
    hipLaunchKernelGGL((conv3x3_pre_f16x3_kernel<MT>), dim3(a.B * a.wgs_per_img, ny), dim3(G::NTHREADS), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

// tiles this kernel exists for: (16, mt, 3, 4, 1), mt = 1 | 2
bool conv3x3_pre_supports(const ConvTile& t) {
    return t.ks == 3 && t.stride == 1 && t.tw == 16 && t.nt == 3 && t.wm == 4 && t.wn == 1 && (t.mt == 1 || t.mt == 2);
}

hipError_t conv3x3_pre_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s) {
    if (!conv3x3_pre_supports(t) || a.C1 != 0 || a.C0 % 16 || a.Cout % 48 || a.H != a.OH || a.W != a.OW) return hipErrorInvalidValue;
    return t.mt == 2 ? launch_pre<2>(a, s) : launch_pre<1>(a, s);
}

}  // namespace midd
