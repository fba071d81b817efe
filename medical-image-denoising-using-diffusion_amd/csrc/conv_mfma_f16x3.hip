// Implicit-GEMM convolution with fp32 operands SPLIT into two fp16 halves and three
// v_mfma_f32_16x16x32_f16 products per K-step ("f16x3"), fp32 accumulation, for gfx950.
//
// Why: the fp32-input MFMA runs at 1/16 of the fp16 rate (MI355X_MICROARCH.md); fp16 x fp16
// products are exact in the fp32 accumulator, so with
//     x' = x * 2^s :  x' = xh + xl,   xh = fp16(x'),  xl = fp16(x' - xh)       (|err| <= 2^-22 |x'|)
//     w' = w * 2^k :  w' = wh + wl    (k per layer so that max|w'| ~ 2^14)
//     w'.x' ~= wh.xh + wh.xl + wl.xh                                          (wl.xl ~ 2^-22 dropped)
// three fp16 MFMAs into ONE fp32 accumulator reproduce the fp32 product to ~2^-21 relative —
// the same order as the fp32 accumulation error itself and far inside the 1e-3 parity gate —
// at 3/16 of the cost.  The power-of-two prescales (exact) keep the low halves in fp16's normal
// range; the epilogue multiplies by 2^-(k+s) (ConvArgs::out_scale), again exact.
//
// Same contract, fusion and launch geometry as conv_mfma_f32.hip (see that file's header):
// A = packed weights (rows = cout), B = input pixels staged through LDS with GroupNorm-apply
// (+SiLU) and the fp16 split done once per element while staging; epilogue bias / time
// embedding / residual, plus optional per-channel partial sums of the OUTPUT for the next
// GroupNorm (replaces a separate statistics pass over the tensor).
//
// K walk: 32 input channels (two 16-channel blocks) per step and tap; a trailing single block
// (Cin = 48, 144) pairs two TAPS per step instead, so only ceil(9/2)*... one half-step is padded.
// LDS image per chunk: [block 0/1][hi|lo][halo pixel][16 fp16] (32 B per pixel and plane), so
// a wave's fragment read is 2 x 512 contiguous bytes, conflict-free.
#include "midd_internal.h"
#include <cstdlib>

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float ACT_PRESCALE = 16.0f;             // 2^s, s = 4 (see header); must match midd_api.hip

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

template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64)
void conv_mfma_f16x3_kernel(const ConvArgs a) {
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int BM = WM * MT * 16;
    constexpr int TH = BM / TW;
    constexpr int PAD = (KS == 3) ? 1 : 0;
    constexpr int IH = (TH - 1) * STRIDE + KS;
    constexpr int IW = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = IH * IW;
    constexpr int NSLOT = NPIX * 8;                       // float4 (4-channel) slots per 32-channel chunk
    constexpr int SPT = (NSLOT + NTHREADS - 1) / NTHREADS;
    constexpr int TAPS = KS * KS;
    constexpr int HSTEPS = (TAPS + 1) / 2;                // steps of a trailing single-block chunk
    constexpr int PLANE = NPIX * 32;                      // bytes of one (block, hi|lo) plane
    constexpr int BUF = 4 * PLANE;
    static_assert(BM % TW == 0, "tile");

    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WN;
    const int wm = wave / WN;
    const int p16 = lane & 15;
    const int kq = lane >> 4;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles_per_img;
    const int trem = blockIdx.x - b * tiles_per_img;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

    const int Cin = a.C0 + a.C1;
    const int nblk = Cin >> 4;
    const int nchunks = (nblk + 1) >> 1;
    const int ntiles_total = a.Cout >> 4;
    const int ntile0 = blockIdx.y * (WN * NT) + wn * NT;

    // ---- staging geometry: thread -> (halo pixel, 4-channel quad q8 of the 32-channel chunk) ----
    const int q8 = tid & 7;
    const int sblk = q8 >> 2;                             // block within the chunk this thread stages
    int g_off[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const int slot = tid + s * NTHREADS;
        int off = -1;
        if (slot < NSLOT) {
            const int pix = slot >> 3;
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) off = (b * a.H + gy) * a.W + gx;
        }
        g_off[s] = off;
    }

    f32x4 stage[SPT];
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    bool stage_active = false;

    auto stage_load = [&](int c) {
        const int blk = 2 * c + sblk;
        stage_active = blk < nblk;
        if (!stage_active) return;
        const int ch = (blk << 4) + (q8 & 3) * 4;
        const float* src; int Cs, coff;
        if (ch < a.C0) { src = a.src0; Cs = a.C0; coff = ch; }
        else           { src = a.src1; Cs = a.C1; coff = ch - a.C0; }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g_off[s] >= 0) v = *reinterpret_cast<const f32x4*>(src + (size_t)g_off[s] * Cs + coff);
            stage[s] = v;
        }
        if (a.prologue != PRO_RAW) {
            sc = *reinterpret_cast<const f32x4*>(a.gn_scale + (size_t)b * Cin + ch);
            sh = *reinterpret_cast<const f32x4*>(a.gn_shift + (size_t)b * Cin + ch);
        }
    };
    auto stage_store = [&](int buf) {
        if (!stage_active) return;
        char* base = lds + buf * BUF + sblk * 2 * PLANE + (q8 & 3) * 8;
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const int slot = tid + s * NTHREADS;
            if (slot < NSLOT) {
                f32x4 v = stage[s];
                if (a.prologue != PRO_RAW && g_off[s] >= 0) {
                    v = v * sc + sh;
                    if (a.prologue == PRO_GN_SILU) {
                        v.x = silu16(v.x); v.y = silu16(v.y); v.z = silu16(v.z); v.w = silu16(v.w);
                    }
                }   // out-of-image pixels stay exactly zero: the conv pads its (normalised) input
                half4 hi, lo;
                split4(v, hi, lo);
                const int pix = slot >> 3;
                *reinterpret_cast<half4*>(base + pix * 32) = hi;
                *reinterpret_cast<half4*>(base + PLANE + pix * 32) = lo;
            }
        }
    };

    // ---- per-lane LDS byte offsets of the B fragments (tap (0,0), block 0, hi plane) ----
    int frag_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pp = (wm * MT + mt) * 16 + p16;
        const int py = pp / TW, px = pp - py * TW;
        frag_base[mt] = ((py * STRIDE) * IW + px * STRIDE) * 32 + (kq & 1) * 16;
    }
    const int kblk_off = (kq >> 1) * 2 * PLANE;           // full chunk: lanes kq>=2 read block 1

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weights: [step][cout tile][hi|lo][lane] x 16 B
    const half8* wp = reinterpret_cast<const half8*>(a.wpack) + (size_t)ntile0 * 128 + lane;
    const size_t wstep = (size_t)ntiles_total * 128;
    // weight fragments are fetched two steps ahead (L2 latency > one step of MFMAs)
    half8 wh[NT], wl[NT], w1h[NT], w1l[NT], w2h[NT], w2l[NT];
    const int total_steps = conv16_num_steps(Cin, TAPS);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        wh[nt] = wp[nt * 128]; wl[nt] = wp[nt * 128 + 64];
        const half8* w1 = wp + (size_t)(total_steps > 1 ? 1 : 0) * wstep;
        w1h[nt] = w1[nt * 128]; w1l[nt] = w1[nt * 128 + 64];
    }
    int step = 0;

    stage_load(0);
    stage_store(0);
    __syncthreads();

    auto do_step = [&](const char* buf, const int (&xo)[MT]) {
        {
            const int ns = (step + 2 < total_steps) ? step + 2 : 0;   // wraps harmlessly at the end
            const half8* w2 = wp + (size_t)ns * wstep;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { w2h[nt] = w2[nt * 128]; w2l[nt] = w2[nt * 128 + 64]; }
        }
        half8 xh[MT], xl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xh[mt] = *reinterpret_cast<const half8*>(buf + xo[mt]);
            xl[mt] = *reinterpret_cast<const half8*>(buf + xo[mt] + PLANE);
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
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { wh[nt] = w1h[nt]; wl[nt] = w1l[nt]; w1h[nt] = w2h[nt]; w1l[nt] = w2l[nt]; }
        ++step;
    };

    int cur = 0;
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1 < nchunks);
        if (more) stage_load(c + 1);
        const char* buf = lds + cur * BUF;
        if (2 * c + 1 < nblk) {
            // full chunk: one tap and 32 channels per step; lane group kq>>1 selects the block
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = tap / KS, dx = tap - dy * KS;
                int xo[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xo[mt] = frag_base[mt] + kblk_off + (dy * IW + dx) * 32;
                do_step(buf, xo);
            }
        } else {
            // trailing single block: two taps per step; lane group kq>>1 selects the tap
#pragma unroll
            for (int hs = 0; hs < HSTEPS; ++hs) {
                const int t0 = 2 * hs, t1 = (2 * hs + 1 < TAPS) ? 2 * hs + 1 : 0;   // padded half has zero weights
                const int o0 = ((t0 / KS) * IW + (t0 % KS)) * 32, o1 = ((t1 / KS) * IW + (t1 % KS)) * 32;
                const int to = (kq >> 1) ? o1 : o0;
                int xo[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xo[mt] = frag_base[mt] + to;
                do_step(buf, xo);
            }
        }
        if (more) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ---------------------------------------------------------------------------
    const int trow = (a.temb != nullptr) ? a.trow[b] : 0;
    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ssq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (ntile0 + nt) * 16 + kq * 4;
        f32x4 add = *reinterpret_cast<const f32x4*>(a.bias + co);
        if (a.temb != nullptr)
            add += *reinterpret_cast<const f32x4*>(a.temb + (size_t)trow * a.temb_stride + co);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pp = (wm * MT + mt) * 16 + p16;
            const int py = pp / TW, px = pp - py * TW;
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy < a.OH && ox < a.OW) {
                const size_t o = ((size_t)(b * a.OH + oy) * a.OW + ox) * a.Cout + co;
                f32x4 v = acc[mt][nt] * a.out_scale + add;
                if (a.resid != nullptr) v += *reinterpret_cast<const f32x4*>(a.resid + o);
                *reinterpret_cast<f32x4*>(a.out + o) = v;
                ssum[nt] += v; ssq[nt] += v * v;
            }
        }
    }
    if (a.stat_partial != nullptr) {
        // fold the 16 pixel lanes (fixed xor tree -> deterministic), lanes p16 == 0 publish 4 channels each
        const int row = trem * WM + wm;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[nt][e] = row16_sum(ssum[nt][e]); ssq[nt][e] = row16_sum(ssq[nt][e]); }
            if (p16 == 0) {
                const int co = (ntile0 + nt) * 16 + kq * 4;
                float* pr = a.stat_partial + ((size_t)(b * a.stat_rows + row) * 2) * a.Cout + co;
                *reinterpret_cast<f32x4*>(pr) = ssum[nt];
                *reinterpret_cast<f32x4*>(pr + a.Cout) = ssq[nt];
            }
        }
    }
}

// ------------------------------------------------------------------------------ dispatch
template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN>
static hipError_t launch16(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int BM = WM * MT * 16;
    constexpr int TH = BM / TW;
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + TH - 1) / TH;
    dim3 grid(a.B * a.tiles_x * a.tiles_y, a.Cout / (WN * NT * 16));
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    if constexpr (2 * 4 * IH * IW * 32 <= 160 * 1024) {
        hipLaunchKernelGGL((conv_mfma_f16x3_kernel<KS, STRIDE, TW, MT, NT, WM, WN>), grid, dim3(WM * WN * 64), 0, s, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;        // tile never picked (conv16_pick_tile), not instantiated
    }
}

int conv_stat_rows(const ConvTile& t, int OH, int OW) {
    const int bm = t.wm * t.mt * 16, th = bm / t.tw;
    return ((OW + t.tw - 1) / t.tw) * ((OH + th - 1) / th) * t.wm;
}

#define MIDD_CONV16_TILES(X)                  \
    /*  tw  mt nt wm wn */                    \
    X(16, 4, 3, 4, 1) X(16, 2, 3, 4, 1) X(16, 1, 3, 4, 1) X(8, 1, 3, 2, 1) \
    X(16, 4, 3, 2, 2) X(16, 2, 3, 2, 2) X(16, 1, 3, 2, 2) X(8, 1, 3, 1, 2) \
    X(16, 4, 3, 1, 3) X(16, 2, 3, 1, 3) X(8, 1, 3, 1, 3)                   \
    X(16, 4, 3, 1, 4) X(16, 2, 3, 1, 4) X(8, 2, 3, 1, 4) X(8, 1, 3, 1, 4)  \
    X(16, 4, 2, 4, 1) X(16, 2, 2, 4, 1) X(8, 1, 2, 2, 1)                   \
    X(16, 4, 2, 2, 2) X(16, 2, 2, 2, 2) X(8, 1, 2, 1, 2)                   \
    X(16, 4, 1, 4, 1) X(16, 2, 1, 4, 1) X(8, 1, 1, 2, 1)

struct Tile16 { int tw, mt, nt, wm, wn; };
static const Tile16 kTiles16[] = {
#define X(tw, mt, nt, wm, wn) {tw, mt, nt, wm, wn},
    MIDD_CONV16_TILES(X)
#undef X
};

bool conv16_pick_tile(int Cout, int B, int OH, int OW, int ks, int stride, ConvTile* t) {
    if (Cout % 16) return false;
    if (!((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1))) return false;
    const int nt = (Cout % 48 == 0) ? 3 : (Cout % 32 == 0) ? 2 : 1;
    const int nn = Cout / (16 * nt);
    int wn = 1;
    for (int cand = 4; cand >= 1; --cand)
        if (nn % cand == 0) { wn = cand; break; }
    const Tile16* best = nullptr;
    long best_score = -(1L << 60);
    static const int max_mt = getenv("MIDD_MAX_MT") ? atoi(getenv("MIDD_MAX_MT")) : 4;   // tuning knob
    for (const Tile16& d : kTiles16) {
        if (d.nt != nt || d.wn != wn) continue;
        if (d.mt > max_mt) continue;
        if (stride == 2 && d.mt > 2) continue;                 // 33x33 halo of a 16x16 s2 tile does not pay
        const int bm = d.wm * d.mt * 16, th = bm / d.tw;
        const long tiles = (long)((OW + d.tw - 1) / d.tw) * ((OH + th - 1) / th);
        const long wgs = (long)B * tiles * (Cout / (wn * nt * 16));
        const long covered = tiles * d.tw * th;
        const bool wasteful = covered * 4 > (long)OH * OW * 5;
        const long score = (wgs >= 512 ? 1000000 : wgs * 1000) + bm - (wasteful ? 500000 : 0);
        if (score > best_score) { best_score = score; best = &d; }
    }
    if (!best) return false;
    *t = ConvTile{ks, stride, best->tw, best->mt, best->nt, best->wm, best->wn};
    return true;
}

hipError_t conv16_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s) {
#define X(tw_, mt_, nt_, wm_, wn_)                                                            \
    if (t.tw == tw_ && t.mt == mt_ && t.nt == nt_ && t.wm == wm_ && t.wn == wn_) {           \
        if (t.ks == 3 && t.stride == 1) return launch16<3, 1, tw_, mt_, nt_, wm_, wn_>(a, s); \
        if (t.ks == 3 && t.stride == 2) return launch16<3, 2, tw_, mt_, nt_, wm_, wn_>(a, s); \
        if (t.ks == 1 && t.stride == 1) return launch16<1, 1, tw_, mt_, nt_, wm_, wn_>(a, s); \
    }
    MIDD_CONV16_TILES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace midd
