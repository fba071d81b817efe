// Implicit-GEMM convolution on the fp32-input MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Replaces the ATen conv2d calls of the reference's hot path
// (/root/reference/Backend/DDIM/DDIMModel.py:118,124,126 ResidualBlock convs, :140-141
// attention 1x1 convs, :195 stride-2 downsample, and the folded ConvTranspose+resample of
// :211/:241-242).  Fused into the same kernel:
//   prologue : GroupNorm-apply (+SiLU) on the input while it is staged into LDS
//              (DDIMModel.py:116-117,121-122,139), torch.cat as two base pointers (:243)
//   epilogue : bias, time-embedding broadcast add (:131), residual add (:133,:166)
//
// GEMM view:  D[cout][pixel] = sum_{tap,cin} W[cout][cin][tap] * X[pixel+tap][cin]
//   A operand = weights (rows = cout), B operand = input pixels (cols = pixel), so the 16x16
//   accumulator tile has the pixel on the lane (lane&15) and four consecutive couts in the
//   four registers -> one 16-byte NHWC store per lane.
//   K is walked in chunks of 16 input channels; within a chunk the four k-steps of the
//   16x16x4 MFMA take channel 4*kq+j in step j (kq = lane>>4), so one ds_read_b128 /
//   global dwordx4 feeds four MFMAs.  The weight tensor is pre-packed on the host in
//   exactly that fragment order (midd_api.hip: pack_conv_weights).
//
// Workgroup = WM x WN waves.  A TH x TW output tile (BM = WM*MT*16 pixels) with its halo is
// staged once per 16-channel chunk into LDS ([halo pixel][16 ch] = 64 B per pixel, so a
// fragment read is a contiguous 1 KiB, conflict-free), double buffered; the next chunk's
// global loads are in flight while the current chunk's 9*4*MT*NT MFMAs run.  Weights are
// L2-resident (<= 2.6 MB per layer) and go straight to registers, one tap ahead.
#include "midd_internal.h"

namespace midd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) {
    return v / (1.0f + expf(-v));
}

template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64)
void conv_mfma_f32_kernel(const ConvArgs a) {
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int BM = WM * MT * 16;
    constexpr int TH = BM / TW;
    constexpr int PAD = (KS == 3) ? 1 : 0;
    constexpr int IH = (TH - 1) * STRIDE + KS;
    constexpr int IW = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = IH * IW;
    constexpr int NSLOT = NPIX * 4;                       // float4 slots per staged chunk
    constexpr int SPT = (NSLOT + NTHREADS - 1) / NTHREADS;
    constexpr int TAPS = KS * KS;
    static_assert(BM % TW == 0, "tile");
    static_assert(NTHREADS % 4 == 0, "quad id must be thread-constant");

    __shared__ f32x4 lds[2][NSLOT];
    __shared__ float fold_scratch[WM * WN * 2 * NT * 16];  // statistics: one row per wave
    __shared__ stat_word stat_acc[(WN * NT * 16 + 2) * STAT_WORDS];      // ... and the publish step's block accumulators
    extern __shared__ float gnp[];                         // [2][Cin] GroupNorm scale, shift of this sample

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
    const int nchunks = Cin >> 4;
    const int ntiles_total = a.Cout >> 4;
    const int ntile0 = blockIdx.y * (WN * NT) + wn * NT;  // first cout tile of this wave

    // ---- per-thread staging geometry (constant over chunks) ------------------------
    const int q = tid & 3;
    int g_off[SPT];        // pixel offset (in pixels) into the image, or -1 when out of bounds / unused
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const int slot = tid + s * NTHREADS;
        int off = -1;
        if (slot < NSLOT) {
            const int pix = slot >> 2;
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) off = (b * a.H + gy) * a.W + gx;
        }
        g_off[s] = off;
    }

    f32x4 stage[SPT];
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};

    auto stage_load = [&](int c) {
        const int ch = c << 4;
        const float* src; int Cs, coff;
        if (ch < a.C0) { src = a.src0; Cs = a.C0; coff = ch; }
        else           { src = a.src1; Cs = a.C1; coff = ch - a.C0; }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g_off[s] >= 0)
                v = *reinterpret_cast<const f32x4*>(src + (size_t)g_off[s] * Cs + coff + q * 4);
            stage[s] = v;
        }
        if (a.prologue != PRO_RAW) {
            sc = *reinterpret_cast<const f32x4*>(gnp + ch + q * 4);
            sh = *reinterpret_cast<const f32x4*>(gnp + Cin + ch + q * 4);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const int slot = tid + s * NTHREADS;
            if (slot < NSLOT) {
                f32x4 v = stage[s];
                if (a.prologue != PRO_RAW) {
                    if (g_off[s] >= 0) {
                        v = v * sc + sh;
                        if (a.prologue == PRO_GN_SILU) {
                            v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
                        }
                    }   // zero padding applies to the conv INPUT, i.e. after norm+activation
                }
                lds[buf][slot] = v;
            }
        }
    };

    // ---- LDS fragment addresses (in float4 units) for tap (0,0) ----------------------
    int frag_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pp = (wm * MT + mt) * 16 + p16;
        const int py = pp / TW, px = pp - py * TW;
        frag_base[mt] = ((py * STRIDE) * IW + px * STRIDE) * 4 + kq;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack) + (size_t)ntile0 * 64 + lane;
    const size_t wtap_stride = (size_t)ntiles_total * 64;   // float4 per (chunk,tap)

    f32x4 wcur[NT], wnxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wcur[nt] = wp[nt * 64];

    if (a.prologue != PRO_RAW) {
        gn_prologue_lds(a.gn_tot0, a.C0, a.gn_bs0, a.gn_tot1, a.C1, a.gn_bs1, a.stat_rep, a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_inv_n, b, 1.0f, gnp, tid, NTHREADS);
        __syncthreads();
    }
    stage_load(0);
    stage_store(0);
    __syncthreads();

    int cur = 0;
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1 < nchunks);
        if (more) stage_load(c + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            // prefetch the next tap's weight fragments (next chunk's first tap at the end)
            {
                size_t nidx = (size_t)(c * TAPS + tap + 1);
                if (tap + 1 == TAPS && !more) nidx = 0;          // harmless re-read on the last step
                const f32x4* wn_ = wp + nidx * wtap_stride;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wnxt[nt] = wn_[nt * 64];
            }
            const int dy = tap / KS, dx = tap - dy * KS;
            f32x4 xf[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt] = lds[cur][frag_base[mt] + (dy * IW + dx) * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[nt][j], xf[mt][j], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wcur[nt] = wnxt[nt];
        }
        if (more) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: bias (+ time embedding) (+ residual), NHWC float4 stores --------------
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
                f32x4 v = acc[mt][nt] + add;
                if (a.resid != nullptr) v += *reinterpret_cast<const f32x4*>(a.resid + o);
                *reinterpret_cast<f32x4*>(a.out + o) = v;
                ssum[nt] += v; ssq[nt] += v * v;
            }
        }
    }
    if (a.stat_tot != nullptr) {
        // per-channel partial sums of the output for the next GroupNorm: the 16 pixel lanes are folded with
        // fixed-order shuffles, the WM wave rows through LDS, and the tile's sums go to the tensor's totals with exact
        // integer atomics (stats_common.h)
        float* const wrow = fold_scratch;                                    // [wave][2][NT*16]
        constexpr int ROWF = 2 * NT * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = ssum[nt][e], s2 = ssq[nt][e];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
                ssum[nt][e] = s1; ssq[nt][e] = s2;
            }
            if (p16 == 0) {
                *reinterpret_cast<f32x4*>(wrow + wave * ROWF + nt * 16 + kq * 4) = ssum[nt];
                *reinterpret_cast<f32x4*>(wrow + wave * ROWF + NT * 16 + nt * 16 + kq * 4) = ssq[nt];
            }
        }
        constexpr int NCOL = WN * NT * 16;
        auto fold = [&](int i) {             // fixed order over wm; stat_publish's first barrier publishes the rows
            const int which = i / NCOL, col = i - which * NCOL;
            const int wn_i = col / (NT * 16), c = col - wn_i * (NT * 16);
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) t += wrow[(m * WN + wn_i) * ROWF + which * (NT * 16) + c];
            return t;
        };
        stat_publish(a.stat_tot, b, a.Cout, a.stat_bs, a.stat_rep, trem % a.stat_rep, blockIdx.y * NCOL, NCOL, fold, stat_acc, tid, NTHREADS);
    }
}

// ------------------------------------------------------------------------------ dispatch
template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN>
static hipError_t launch_one(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int BM = WM * MT * 16;
    constexpr int TH = BM / TW;
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + TH - 1) / TH;
    dim3 grid(a.B * a.tiles_x * a.tiles_y, a.Cout / (WN * NT * 16));
    hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, STRIDE, TW, MT, NT, WM, WN>), grid, dim3(WM * WN * 64), 2 * (a.C0 + a.C1) * sizeof(float), s, a);
    return hipGetLastError();
}

// Tile menu.  NT*16*WN = couts per workgroup, WM*MT*16 = pixels per workgroup (TW wide).
#define MIDD_CONV_TILES(X)                    \
    /*  tw  mt nt wm wn */                    \
    X(16, 4, 3, 4, 1) X(16, 2, 3, 4, 1) X(16, 1, 3, 4, 1) X(8, 1, 3, 2, 1) \
    X(16, 4, 3, 2, 2) X(16, 2, 3, 2, 2) X(16, 1, 3, 2, 2) X(8, 1, 3, 1, 2) \
    X(16, 4, 3, 1, 3) X(16, 2, 3, 1, 3) X(8, 1, 3, 1, 3)                   \
    X(16, 4, 3, 1, 4) X(16, 2, 3, 1, 4) X(8, 2, 3, 1, 4) X(8, 1, 3, 1, 4)  \
    X(16, 4, 2, 4, 1) X(16, 2, 2, 4, 1) X(8, 1, 2, 2, 1)                   \
    X(16, 4, 2, 2, 2) X(16, 2, 2, 2, 2) X(8, 1, 2, 1, 2)                   \
    X(16, 4, 1, 4, 1) X(16, 2, 1, 4, 1) X(8, 1, 1, 2, 1)

struct TileDesc { int tw, mt, nt, wm, wn; };
static const TileDesc kTiles[] = {
#define X(tw, mt, nt, wm, wn) {tw, mt, nt, wm, wn},
    MIDD_CONV_TILES(X)
#undef X
};

bool conv_pick_tile(int Cout, int B, int OH, int OW, int ks, int stride, ConvTile* t) {
    if (Cout % 16) return false;
    if (!((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1))) return false;
    int nt = (Cout % 48 == 0) ? 3 : (Cout % 32 == 0) ? 2 : 1;
    const int nn = Cout / (16 * nt);                 // waves needed along cout
    // the widest instantiated wave layout along cout that divides nn (Cout = 128: nt 2, nn 4 -> wn 2), then the
    // largest pixel tile that still yields >= 2 workgroups per CU
    int wn = 0;
    for (const TileDesc& d : kTiles)
        if (d.nt == nt && nn % d.wn == 0 && d.wn > wn) wn = d.wn;
    if (!wn) return false;
    const TileDesc* best = nullptr;
    long best_score = -(1L << 60);
    for (const TileDesc& d : kTiles) {
        if (d.nt != nt || d.wn != wn) continue;
        if (stride == 2 && d.mt > 2) continue;
        const int bm = d.wm * d.mt * 16, th = bm / d.tw;
        const long wgs = (long)B * ((OW + d.tw - 1) / d.tw) * ((OH + th - 1) / th) * (Cout / (wn * nt * 16));
        // padding waste of ragged tiles counts against a tile
        const long covered = (long)((OW + d.tw - 1) / d.tw) * d.tw * ((OH + th - 1) / th) * th;
        const bool wasteful = covered * 4 > (long)OH * OW * 5;        // > 25 % padding
        long score = (wgs >= 512 ? 1000000 : wgs * 1000) + bm - (wasteful ? 500000 : 0);
        if (score > best_score) { best_score = score; best = &d; }
    }
    if (!best) return false;
    *t = ConvTile{ks, stride, best->tw, best->mt, best->nt, best->wm, best->wn};
    return true;
}

hipError_t conv_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s) {
#define X(tw_, mt_, nt_, wm_, wn_)                                                              \
    if (t.tw == tw_ && t.mt == mt_ && t.nt == nt_ && t.wm == wm_ && t.wn == wn_) {             \
        if (t.ks == 3 && t.stride == 1) return launch_one<3, 1, tw_, mt_, nt_, wm_, wn_>(a, s); \
        if (t.ks == 3 && t.stride == 2) return launch_one<3, 2, tw_, mt_, nt_, wm_, wn_>(a, s); \
        if (t.ks == 1 && t.stride == 1) return launch_one<1, 1, tw_, mt_, nt_, wm_, wn_>(a, s); \
    }
    MIDD_CONV_TILES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace midd
