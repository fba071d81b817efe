// Implicit-GEMM convolution with fp32 operands SPLIT into two fp16 halves and three
// v_mfma_f32_16x16x32_f16 products per K-step ("f16x3"), fp32 accumulation, for gfx950.
//
// Why: the fp32-input MFMA runs at 1/16 of the fp16 rate (MI355X_MICROARCH.md); fp16 x fp16
// products are exact in the fp32 accumulator, so with the exact power-of-two prescales
//     x' = x * 2^s :  x' = xh + xl,   xh = fp16(x'),  xl = fp16(x' - xh)       (|err| <= 2^-22 |x'|)
//     w' = w * 2^k :  w' = wh + wl    (k per layer so that max|w'| ~ 2^14)
//     w'.x' ~= wh.xh + wh.xl + wl.xh                                          (wl.xl ~ 2^-22 dropped)
// three fp16 MFMAs into ONE fp32 accumulator reproduce the fp32 product to ~2^-21 relative —
// the same order as the fp32 accumulation error itself and far inside the 1e-3 parity gate —
// at 3/16 of the cost.  The epilogue multiplies by 2^-(k+s) (ConvArgs::out_scale), exactly.
//
// Same contract and fusion as conv_mfma_f32.hip (A = packed weights, rows = cout; B = input
// pixels; GroupNorm-apply(+SiLU) prologue, virtual torch.cat, bias / time-embedding / residual
// epilogue) plus per-channel partial sums of the OUTPUT for the next GroupNorm.
//
// Data movement (what the fp16 rate makes necessary):
//   * every global->LDS transfer in the K loop is LDS-DMA (global_load_lds_dwordx4): no VGPRs
//     in flight, exact per-wave instruction counts, so counted s_waitcnt vmcnt(N) + raw
//     s_barrier keep two weight steps and the next activation chunk in flight across barriers;
//   * WEIGHTS go through a 3-slot LDS ring shared by all waves of the workgroup (one L2 read
//     per workgroup and step instead of one per wave: the per-wave register path was L2-bound);
//   * ACTIVATIONS of the next 32-channel chunk land raw (fp32) in LDS; each thread transforms
//     the slots it fetched itself (norm, SiLU, 2^s prescale, hi/lo split) into the MFMA image
//     [block 0/1][hi|lo][halo pixel][16 fp16] (32 B per pixel and plane: a fragment read is
//     2 x 512 contiguous bytes, conflict-free).
// K walk: 32 input channels (two 16-channel blocks) per step and tap; a trailing single block
// (Cin = 48, 144) pairs two TAPS per step instead.
#include "f16x3_common.h"
#include <cstdlib>
#include <type_traits>
#ifdef MIDD_CONV_TIMING
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#endif

namespace midd {

template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN, int CBT = 0>
struct Conv16Geom {
    static constexpr int NW = WM * WN;
    static constexpr int NTHREADS = NW * 64;
    static constexpr int BM = WM * MT * 16;
    static constexpr int TH = BM / TW;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int IW = (TW - 1) * STRIDE + KS;
    static constexpr int NPIX = IH * IW;
    static constexpr int CB = CBT ? CBT : conv16_cb(KS);            // 16-channel blocks per chunk (CBT = 2: the "wide" 3x3 variant)
    static constexpr bool WIDE = (KS == 3 && CB == 2);
    static constexpr int QPP = 4 * CB;                              // 16-byte slots per halo pixel and chunk
    static constexpr int NSLOT = NPIX * QPP;
    static constexpr int APW = (NSLOT + NTHREADS - 1) / NTHREADS;   // activation DMA pieces per wave and chunk
    static constexpr int RAW_BYTES = APW * NTHREADS * 16;
    static constexpr int PLANE = NPIX * 32;
    static constexpr int IMG_BYTES = 2 * CB * PLANE;
    static constexpr int WPIECES = WN * NT * 2;                     // 1 KiB weight pieces per step
    static constexpr int PPW = (WPIECES + NW - 1) / NW;             // pieces per wave and step (duplicates pad)
    static constexpr int WSLICE = WPIECES * 1024;
    // epilogue state kept in LDS instead of registers (the K loop is register-bound): GroupNorm partial
    // sums of the output, one row [2][NT*16] per wave, and the bias (+ time embedding) vector of the workgroup
    static constexpr int STAT_FLOATS = NW * 2 * NT * 16;
    static constexpr int ADD_FLOATS = WN * NT * 16;
    static constexpr int EPI_BYTES = (STAT_FLOATS + ADD_FLOATS) * 4;
    // GroupNorm scale/shift of the input, [2][Cin] floats, sized at launch (dynamic LDS); the ring is
    // dimensioned for up to NOMINAL_CIN input channels (more still runs, possibly one workgroup per CU fewer)
    static constexpr int NOMINAL_CIN = 384;
    static constexpr int FIXED_BYTES = RAW_BYTES + IMG_BYTES + EPI_BYTES + 2 * NOMINAL_CIN * 4 + 64;
    // weight steps resident in LDS (prefetch distance RING-1): L2->LDS latency is ~1-2k cycles under
    // load, a step is only 150-600 MFMA cycles, so take as many slots as fit in half the LDS (two
    // workgroups per CU), between 2 and 6.
#ifndef MIDD_LDS_TARGET_KB
#define MIDD_LDS_TARGET_KB 52
#endif
#ifndef MIDD_RING_MAX
#define MIDD_RING_MAX 6
#endif
    // 52 KB: three workgroups per CU
    // (stride-2 tiles stage a 33x17 halo and run one workgroup per CU whatever the ring: they take a deep ring -- with two
    // slots the counted wait for a step's weights was 18-33 % of a wave's time, in-kernel stamps of round 3)
    // wide 3x3 chunks (launches that leave at most ~2 workgroups per CU anyway): two workgroups per CU
#ifndef MIDD_LDS_WIDE_KB
#define MIDD_LDS_WIDE_KB 78
#endif
    static constexpr int LDS_TARGET = (STRIDE == 2 ? 120 : WIDE ? MIDD_LDS_WIDE_KB : MIDD_LDS_TARGET_KB) * 1024;
    static constexpr int ring_fit = (LDS_TARGET - FIXED_BYTES) / WSLICE;
    static constexpr int RING = ring_fit < 2 ? 2 : (ring_fit > MIDD_RING_MAX ? MIDD_RING_MAX : ring_fit);
    static constexpr int LDS_BYTES = FIXED_BYTES + RING * WSLICE;                 // at NOMINAL_CIN
    static constexpr int lds_bytes(int cin) { return LDS_BYTES + 2 * (cin - NOMINAL_CIN) * 4; }   // incl. the 16 mean/rstd floats
    static_assert(BM % TW == 0, "tile");
};

// Diagnostic build only (-DMIDD_CONV_TIMING, tools/conv_timing.py): s_memtime stamps of wave 0 of every
// workgroup, summed per launch shape.  Shares, not run times: the stamps drain the LDS queue.
#ifdef MIDD_CONV_TIMING
enum { TS_WAIT, TS_ISSUE, TS_MFMA, TS_CHUNK_WAIT, TS_TRANSFORM, TS_EPILOGUE, TS_PROLOGUE, TS_FIRSTWAIT, TS_DMAWAIT, TS_RES, TS_PUBLISH, TS_TOTAL, TS_REAL, TS_WGS, TS_N };
__device__ unsigned long long g_conv_timing[64][TS_N];
__device__ __forceinline__ unsigned long long ts_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define TS_DECL bool ts_after_epi = false; unsigned long long ts_acc[TS_N] = {}; unsigned long long ts_last = ts_stamp(); const unsigned long long ts_t0 = ts_last; const unsigned long long ts_r0 = __builtin_amdgcn_s_memrealtime();
#define TS(k) { const unsigned long long t_ = ts_stamp(); ts_acc[k] += t_ - ts_last; ts_last = t_; }
#else
#define TS_DECL
#define TS(k)
#endif

template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN, bool RES, int CBT>
#ifndef MIDD_CONV16_WAVES_PER_SIMD
#define MIDD_CONV16_WAVES_PER_SIMD 3
#endif
// the register budget is capped so that as many workgroups as the LDS target allows are resident
// (2 -> 3 workgroups per CU is worth ~25 %: the phases of one workgroup do not overlap themselves)
// (stride-2 tiles stage a 33x17 halo: their LDS allows one workgroup per CU anyway, so they get the whole register file)
// (wide chunks: two workgroups per CU by their LDS, so two waves per SIMD's worth of registers)
// The 128-pixel tile with the folded res_conv needs ~200 registers: capped at 168 it spilled 51 of them at every tile boundary --
// 34 MB of scratch writes and as many reads per 256x256 launch (WRITE_SIZE 84 MB for a 50 MB output, tools/traffic_per_op.sh).
// With two waves per SIMD's worth of registers nothing spills: same-box +4.4 % split, +0.6 % unsplit (round 3).
#ifndef MIDD_RES_MT2_WAVES
#define MIDD_RES_MT2_WAVES 2
#endif
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && STRIDE == 1) ? (CBT == 2 ? 2 : (RES && MT == 2) ? MIDD_RES_MT2_WAVES : MIDD_CONV16_WAVES_PER_SIMD) : 1)
void conv_mfma_f16x3_kernel(const ConvArgs a) {
    using G = Conv16Geom<KS, STRIDE, TW, MT, NT, WM, WN, CBT>;
    constexpr int NW = G::NW, NTHREADS = G::NTHREADS, TH = G::TH, IW = G::IW;
    constexpr int PAD = (KS == 3) ? 1 : 0;
    constexpr int NSLOT = G::NSLOT, APW = G::APW, PLANE = G::PLANE;
    constexpr int TAPS = KS * KS;
    constexpr int HSTEPS = (TAPS + 1) / 2;
    constexpr int PPW = G::PPW, WSLICE = G::WSLICE, RING = G::RING;

    extern __shared__ __attribute__((aligned(16))) char lds[];             // G::lds_bytes(Cin)
    char* const raw = lds;
    char* const img = lds + G::RAW_BYTES;
    char* const wring = img + G::IMG_BYTES;
    float* const stat_lds = reinterpret_cast<float*>(wring + RING * WSLICE);   // [wave][2][NT*16]
    float* const add_lds = stat_lds + G::STAT_FLOATS;                           // [WN*NT*16]
    float* const gnp = add_lds + G::ADD_FLOATS;                                 // [2][Cin] scale, shift

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN;
    const int wm = wave / WN;
    const int p16 = lane & 15;
    const int kq = lane >> 4;

    // Persistent workgroups: blockIdx.x = (sample, j); the workgroup walks tiles j, j+wgs_per_img, ...
    // of ITS sample, so the weight ring streams cyclically across tiles and the next tile's first
    // activation chunk is already in flight while this tile is finished and stored.
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / a.wgs_per_img;
    int trem = blockIdx.x - b * a.wgs_per_img;            // current tile of this sample
    int oy0 = (trem / a.tiles_x) * TH, ox0 = (trem % a.tiles_x) * TW;

    const int Cin = a.C0 + a.C1;
    const int nblk = Cin >> 4;
    constexpr int CB = G::CB, QPP = G::QPP;
    const int nchunks = (nblk + CB - 1) / CB;
    const int ntiles_total = a.Cout >> 4;
    const int ntile_wg = blockIdx.y * (WN * NT);          // first cout tile of this workgroup
    const int res_steps = RES ? a.res_steps : 0;          // folded res_conv: K steps after a tile's 3x3 steps (instantiations of their own: registers)
    const int total_steps = conv16_num_steps(Cin, TAPS, G::CB) + res_steps;

    // ---- weights: LDS-DMA ring ---------------------------------------------------------------
    // global layout [step][cout tile][hi|lo][lane] x 16 B; the workgroup's slice of one step is
    // contiguous.  Everything but the lane offset is wave-uniform, so the address arithmetic stays
    // on the scalar unit.
    const char* const wbase = reinterpret_cast<const char*>(a.wpack) + (size_t)ntile_wg * 2048;
    const size_t wstep_bytes = (size_t)ntiles_total * 2048;
    const int lane16 = lane * 16;
    int wr_step = 0, wr_slot = 0;                         // next step to fetch / the ring slot it goes to
    const char* wr_src = wbase;                           // = wbase + wr_step * wstep_bytes, kept incrementally
    // Diagnostic build (-DMIDD_DMA_CHECK, tools/dma_check.sh; never shipped): every destination of an asynchronous transfer -- ring
    // slot pieces, landing-buffer slots, the registers of untracked loads -- is filled with a NaN sentinel before the transfer is
    // requested, and every consumer checks what it reads: a counted wait that returns before its data has landed leaves the
    // sentinel in place and sets STATUS_DMA_EARLY.  Run over the whole GPU suite this checks the hand-counted vmcnt protocol
    // on the hardware, for every instantiation and schedule the tests reach.
#ifdef MIDD_DMA_CHECK
    constexpr unsigned SENT_W = 0x7FFF7FFFu;              // two fp16 NaNs: no packed weight
    constexpr unsigned SENT_A = 0x7FC0DEADu;              // an fp32 NaN: no finite activation
    unsigned dma_bad = 0;
    auto sent4 = [](unsigned v) { typedef unsigned u32x4_ __attribute__((ext_vector_type(4))); return __builtin_bit_cast(f32x4, (u32x4_){v, v, v, v}); };
    auto has_sent = [](const auto& q, unsigned v) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ u = __builtin_bit_cast(u32x4_, q);
        return (unsigned)((u[0] == v) | (u[1] == v) | (u[2] == v) | (u[3] == v));
    };
#endif
    auto issue_w = [&]() {
        char* slot = wring + wr_slot * WSLICE;
#ifdef MIDD_DMA_CHECK
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int piece = (WM == 1) ? wave * PPW + i : wave + i * NW;
            if (piece >= G::WPIECES) piece -= G::WPIECES;
            lds_store_raw(slot + piece * 1024 + lane16, sent4(SENT_W));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            // WM == 1: every wave owns a distinct cout slice, so it fetches exactly the pieces it
            // reads itself (no cross-wave hand-off, no barrier per step); otherwise round-robin.
            int piece = (WM == 1) ? wave * PPW + i : wave + i * NW;
            if (piece >= G::WPIECES) piece -= G::WPIECES;          // padding duplicate: same bytes, same place (issuing only the
                                                                   // WPIECES distinct pieces, with per-wave wait counts, measured -1 %: round 3)
            dma16(wr_src + piece * 1024 + lane16, slot + piece * 1024);
        }
        ++wr_step; wr_src += wstep_bytes;
        if (wr_step == total_steps) { wr_step = 0; wr_src = wbase; }   // cyclic: step 0 of the next tile follows the last
        wr_slot = (wr_slot + 1 == RING) ? 0 : wr_slot + 1;
    };

    // ---- activations: per-thread slots (halo pixel, 4-channel quad q8 of the 32-channel chunk) ----
    const int q8 = tid % QPP;                             // (NTHREADS % QPP == 0: constant per thread)
    const int sblk = q8 >> 2;
    int g_off[APW];            // pixel index inside the sample's image; -1: out of the image; -2: slot beyond the tile
    bool tile_pad = true;      // (uniform) the tile's halo leaves the image somewhere: only then a slot can be out of the image
    auto set_tile = [&](int t) {
        const int iy0 = (t / a.tiles_x) * TH * STRIDE - PAD, ix0 = (t % a.tiles_x) * TW * STRIDE - PAD;
        tile_pad = iy0 < 0 || ix0 < 0 || iy0 + G::IH > a.H || ix0 + IW > a.W;
#pragma unroll
        for (int s = 0; s < APW; ++s) {
            const int slot = tid + s * NTHREADS;
            int off = -2;
            if (slot < NSLOT) {
                const int pix = slot / QPP;
                const int iy = pix / IW, ix = pix - iy * IW;
                const int gy = iy0 + iy, gx = ix0 + ix;
                off = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? gy * a.W + gx : -1;
            }
            g_off[s] = off;
        }
    };
    set_tile(trem);
    // Lanes with nothing to fetch (padding, unused slots, missing second block) read a valid
    // dummy address; transform() writes zeros / nothing for them.
    auto issue_a = [&](int c) {
        // channel-blocked activations [B][C/16][H][W][16]: the chunk's 16-channel block of a halo row is one contiguous run
        const int blk = min(CB * c + sblk, nblk - 1);
        const float* src; int bsrc, nb;
        if ((blk << 4) < a.C0) { src = a.src0; bsrc = blk; nb = a.C0 >> 4; }
        else                   { src = a.src1; bsrc = blk - (a.C0 >> 4); nb = a.C1 >> 4; }
        const char* base = reinterpret_cast<const char*>(src) + ((size_t)(b * nb + bsrc) * (size_t)(a.H * a.W)) * 64 + (q8 & 3) * 16;
#ifdef MIDD_DMA_CHECK
#pragma unroll
        for (int s = 0; s < APW; ++s) lds_store_raw(raw + (wave + s * NW) * 1024 + lane16, sent4(SENT_A));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
        for (int s = 0; s < APW; ++s) {
            const unsigned byte_off = (unsigned)max(g_off[s], 0) * 64u;             // one block plane is < 4 GiB (host-checked)
            dma16(base + byte_off, raw + (wave + s * NW) * 1024);
        }
    };
    // The transform of a chunk, per slot: read the raw fp32 quad, GroupNorm-apply (+SiLU), 2^s prescale, hi/lo split,
    // write into the MFMA image.  (Measured in round 2: running the arithmetic of chunk c+1's transform under the MFMAs
    // of chunk c's steps 2..4 and only the LDS writes at the chunk boundary is 2-4 % SLOWER end to end -- the waves
    // reach the step barriers out of phase.)
    float rscale = RAW_PRESCALE;          // 2^a of a raw operand (set after the prologue's barrier)
    // The transform of a chunk, per slot: raw fp32 quad -> GroupNorm-apply (+SiLU) -> 2^s prescale -> hi/lo split -> MFMA image.
    // Instruction diet of round 3 (the vector ALU is what a chunk costs beside its MFMAs):
    //   * all raw quads of the chunk are requested before the first is used (one LDS round trip, not APW);
    //   * SiLU on the exponent's own argument: the prologue leaves sc' = -log2(e) rstd gamma, sh' = -log2(e) (beta - mean rstd gamma),
    //     so t = x sc' + sh' = -y log2(e) feeds v_exp directly, d = (1 + 2^t) / 16 is ONE fma, and the operand is
    //     u = t / d = -16 log2(e) silu(y); the constant -ln 2 that turns u back into 16 silu(y) sits in the packed
    //     weights (pack_conv_f16x3: SILU_WEIGHT_FACTOR) -- 5 instructions per element instead of 6, 2 of them transcendental;
    //   * hi = fp16(u) by v_cvt_pk_f16_f32, lo = fp16(u - hi) by ONE v_fma_mix per element (f16x3_common.h: split_pair);
    //   * no select for the conv's zero padding: out-of-image slots are zeroed ONCE per tile (first chunk) and otherwise
    //     simply not written (g_off < 0 also covers a thread's slot beyond the tile): the only conditional code is
    //     the pair of LDS stores.
    // (Measured in round 2: running the arithmetic of chunk c+1's transform under the MFMAs of chunk c's steps 2..4 and
    // only the LDS writes at the chunk boundary is 2-4 % SLOWER end to end -- the waves reach the step barriers out of phase.)
    auto transform = [&](int c) {
        const int blk = CB * c + sblk;
        if (blk >= nblk) return;
        const int ch = (blk << 4) + (q8 & 3) * 4;
        f32x4 rq[APW];
#pragma unroll
        for (int s = 0; s < APW; ++s) rq[s] = *reinterpret_cast<const f32x4*>(raw + (tid + s * NTHREADS) * 16);
#ifdef MIDD_DMA_CHECK
#pragma unroll
        for (int s = 0; s < APW; ++s) dma_bad |= has_sent(rq[s], SENT_A);
#endif
        f32x4 sc = {rscale, rscale, rscale, rscale}, sh = {0.f, 0.f, 0.f, 0.f};      // raw operand: per-sample 2^a (stats_common.h)
        if (a.prologue != PRO_RAW) {
            sc = *reinterpret_cast<const f32x4*>(gnp + ch);
            sh = *reinterpret_cast<const f32x4*>(gnp + Cin + ch);
        }
        char* base = img + sblk * 2 * PLANE + (q8 & 3) * 8;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if (c == 0 && tile_pad) {                  // (uniform) first chunk of a tile whose halo leaves the image: the conv's zero padding
#pragma unroll
            for (int s = 0; s < APW; ++s) {
                const int slot = tid + s * NTHREADS;
                if (g_off[s] == -1) {                  // (a thread's slots cover ITS 16-channel block: all blocks get zeroed between the threads)
                    *reinterpret_cast<u32x2*>(base + (slot / QPP) * 32) = (u32x2){0u, 0u};
                    *reinterpret_cast<u32x2*>(base + PLANE + (slot / QPP) * 32) = (u32x2){0u, 0u};
                }
            }
        }
#pragma unroll
        for (int s = 0; s < APW; ++s) {
            const int slot = tid + s * NTHREADS;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rq[s][e], sc[e], sh[e]);
#if defined(C16_ABL) && C16_ABL == 4     // ablation 4 (wrong results): no SiLU
            if (false) {
#else
            if (a.prologue == PRO_GN_SILU) {
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // v = t = -y log2(e):  u = t * 16 / (1 + 2^t)
                    const float d = __builtin_fmaf(__builtin_amdgcn_exp2f(v[e]), 1.0f / ACT_PRESCALE, 1.0f / ACT_PRESCALE);
                    v[e] = v[e] * __builtin_amdgcn_rcpf(d);
                }
            }
            unsigned h01, h23, l01, l23;
            split_pair(v[0], v[1], h01, l01);
            split_pair(v[2], v[3], h23, l23);
            if (g_off[s] >= 0) {                   // in the image and in the tile
                const int pix = slot / QPP;
                *reinterpret_cast<u32x2*>(base + pix * 32) = (u32x2){h01, h23};
                *reinterpret_cast<u32x2*>(base + PLANE + pix * 32) = (u32x2){l01, l23};
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
    int frag_full[MT];                                    // full chunk: lanes kq>=2 read block 1
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) frag_full[mt] = frag_base[mt] + (kq >> 1) * 2 * PLANE;
    const int wfrag_off = (wn * NT) * 2048 + lane * 16;   // this wave's cout tiles inside a ring slot

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- K loop -------------------------------------------------------------------------------
    // DMA protocol (D = RING-1 weight steps in flight).  Per wave, in program order:
    //   prologue : A(0) W(0)..W(D-1)                                   -> wait 0
    //   step s   : wait; barrier; issue W(s+D) [; issue A(c+1) in the first step of a chunk]; MFMAs
    //   chunk end: wait; barrier; transform(c+1)
    // When step s waits, the groups younger than W(s) are W(s+1..s+D-1), plus A(c+1) during steps
    // 1..D of the chunk (afterwards A(c+1) is older than W(s), i.e. already forced complete):
    //   N = (D-1)*PPW [+ APW].   At the chunk end the groups younger than A(c+1) are the
    // min(nsteps-1, D) weight groups issued after it.  The slot refilled after the barrier of step
    // s, (s+D)%RING == (s-1)%RING, was last read before that barrier by every wave.
    constexpr int D = RING - 1;
    const int ntile0 = ntile_wg + wn * NT;
    TS_DECL
    issue_a(0);
#pragma unroll
    for (int i = 0; i < D; ++i) issue_w();
    // (Measured in round 2: requesting the totals BEFORE the DMAs through loads the compiler does not track, with a counted
    // wait, so that the GroupNorm arithmetic overlaps the DMA latency instead of following it -- 0 % split, -0.7 % unsplit.)
#if !(defined(C16_ABL) && C16_ABL == 6)  // ablation 6 (wrong results): no GroupNorm prologue
    if (a.prologue == PRO_GN || a.prologue == PRO_GN_SILU)       // GroupNorm scale / shift of this sample (stats_common.h)
        gn_prologue_lds(a.gn_tot0, a.C0, a.gn_bs0, a.gn_tot1, a.C1, a.gn_bs1, a.stat_rep, a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_inv_n, b,
                        (a.prologue == PRO_GN_SILU) ? SILU_ARG_FACTOR : ACT_PRESCALE, gnp, tid, NTHREADS, a.status);
#endif
    // raw operand: its sum of squares from the producers' totals -> power-of-two prescale (stats_common.h); the scale /
    // shift area is free in this case
    stat_word* const raw_acc = reinterpret_cast<stat_word*>(gnp);
    if (a.prologue == PRO_RAW && a.gn_tot0 != nullptr && wave == 0)
        raw_sumsq_lds(a.gn_tot0, a.C0, a.gn_bs0, a.gn_tot1, a.C1, a.gn_bs1, a.stat_rep, b, raw_acc, lane);
    // folded res_conv: its operand is the (raw) block input; same prescale rule, from the block input's totals
    stat_word* const res_acc = reinterpret_cast<stat_word*>(gnp + 2 * Cin);          // the 64 spare bytes behind the scale / shift table
    if (res_steps > 0 && a.res_tot0 != nullptr && wave == NW - 1)
        raw_sumsq_lds(a.res_tot0, a.res_C0, a.res_bs0, a.res_tot1, a.res_C1, a.res_bs1, a.stat_rep, b, res_acc, lane);
    {
        const int trow = (a.temb != nullptr) ? a.trow[b] : 0;
        for (int i = tid; i < G::ADD_FLOATS; i += NTHREADS) {
            const int co = ntile_wg * 16 + i;
            add_lds[i] = a.bias[co] + (a.temb != nullptr ? a.temb[(size_t)trow * a.temb_stride + co] : 0.f);
        }
    }
    wait_vm_and_barrier<0>();               // everything above has landed / is visible (once per launch)
    float oscale = a.out_scale;             // epilogue factor: undoes the weight and the operand prescale (exact)
    if (a.prologue == PRO_RAW) {
        rscale = a.raw_scale_fixed;
        if (a.gn_tot0 != nullptr) {
            bool bad;
            const int ex = __builtin_amdgcn_readfirstlane(raw_prescale_exp(raw_acc, &bad));
            rscale = pow2f(ex);
            if (bad && tid == 0 && a.status != nullptr) atomicOr(a.status, (int)STATUS_NONFINITE);
        }
        oscale = a.out_scale / rscale;      // power of two: exact
    }
    float res_in = 1.0f, res_rescale = 1.0f;    // res phase: operand prescale 2^a; accumulator factor between the two products' units
    if (res_steps > 0) {
        if (a.res_tot0 != nullptr) {
            bool bad;
            res_in = pow2f(__builtin_amdgcn_readfirstlane(raw_prescale_exp(res_acc, &bad)));
            if (bad && tid == 0 && a.status != nullptr) atomicOr(a.status, (int)STATUS_NONFINITE);
        }
        // 3x3 product: true value = acc * out_scale;  res product: true value = acc * res_scale / res_in  (all powers of two)
        res_rescale = a.out_scale * res_in / a.res_scale;
        oscale = a.res_scale / res_in;
    }
    transform(0);
    if constexpr (WM == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    TS(TS_PROLOGUE)
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
#ifdef MIDD_DMA_CHECK
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dma_bad |= has_sent(wh[nt], SENT_W) | has_sent(wl[nt], SENT_W);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], xh[mt], acc[mt][nt], 0, 0, 0);
#if !(defined(C16_ABL) && C16_ABL == 8)  // ablation 8 (wrong results): one MFMA pass instead of three
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
#endif
    };

    // The activation fragments of a step only depend on the chunk's image (published at the chunk
    // boundary), not on the step's barrier: they are requested first, so their LDS latency overlaps the wait.
    // `first` = first step of a chunk: its barrier is also the one that publishes the freshly transformed
    // image, so the fragments are read after it (WM == 1 has a dedicated barrier after the transform).
    auto k_step = [&](auto with_a, bool first, bool first_with_more, int next_chunk, const int (&xo)[MT]) {
        constexpr bool WITH_A = decltype(with_a)::value;
#ifdef MIDD_DMA_CHECK_BREAK                // the checker's own test: a wait that is one weight step too permissive must be reported
        constexpr int N = D * PPW + (WITH_A ? APW : 0);
#else
        constexpr int N = (D - 1) * PPW + (WITH_A ? APW : 0);
#endif
        const bool early = (WM == 1) || !first;
        if (early) {
            load_x(xo);
            // LDS operations retire in order: "at most 2*MT outstanding" = everything older than the
            // fragment reads just issued (the previous step's weight reads) is done
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(N), "n"(2 * MT) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
        }
        TS(TS_DMAWAIT)                     // diagnostic build: the counted wait alone, then the barrier (TS_WAIT)
        if constexpr (WM != 1) {           // WM == 1: own weights only, no cross-wave hand-off per step
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#ifdef MIDD_CONV_TIMING
        if (ts_after_epi) { TS(TS_FIRSTWAIT) ts_after_epi = false; } else { TS(TS_WAIT) }
#endif
        if (!early) load_x(xo);
        issue_w();
        if (first_with_more) issue_a(next_chunk);  // each thread already consumed its own raw slots
        TS(TS_ISSUE)
        mfma_step();
        TS(TS_MFMA)
    };

    // ---- res_conv folded into the tile (ConvArgs::res_*; SURVEY 2.1, DDIMModel.py:126,133) -------------------------------
    // After a tile's 3x3 steps the accumulators are rescaled (power of two) and res_steps more K steps run over the BLOCK
    // INPUT's channels, 32 per step: the 1x1 res_conv.  No halo, no taps, every wave needs only ITS pixels: the B operand comes
    // straight from global memory into registers (lane = (pixel, 8 channels), as conv1x1_f16x3.hip), two steps ahead, through
    // loads the compiler does not track (a tracked load is awaited with vmcnt(0) while LDS-DMA is pending); the weights are
    // further steps of the same ring.  Replaces 15 launches per forward, their output tensors and conv2's residual read.
    // vmcnt bookkeeping per wave, program order:  A(0) A(1) | it 0: W A(2) | it 1: W A(3) | ...   (W = issue_w, PPW pieces;
    // A = RL loads).  Iteration r needs A(r): younger are the W of iteration r-1 (r >= 1) and A(r+1) (if any).  The ring slot
    // of step r was requested D >= 2 iterations (or 3x3 steps) earlier, i.e. before A(r): complete with it.
    constexpr int RL = 2 * MT;                            // untracked 16-byte loads per wave and res step
    constexpr int RG = (MT == 1) ? 2 : 1;                 // res steps per group: the loads of group g+1 fly under the MFMAs of group g
    // every hand-counted vmcnt immediate of this instantiation fits the 6-bit field (k_step, chunk end, res_mfma, res_wait)
    static_assert((D - 1) * PPW + APW <= 63 && D * PPW <= 63 && (D - 1) * PPW + RG * RL <= 63 && RG * PPW <= 63, "vmcnt immediate beyond 63");
    auto res_load = [&](int r, f32x4 (&dst)[MT][2]) {
        const int rc = a.res_C0 + a.res_C1;
        int ch = r * 32 + kq * 8;
        if (ch >= rc) ch = rc - 8;                        // trailing half step: valid dummy, meets zero weights
        const float* src; int nb, cc;               // 8 channels inside one 16-channel block of the blocked layout
        if (ch < a.res_C0) { src = a.res_src0; nb = a.res_C0 >> 4; cc = ch; } else { src = a.res_src1; nb = a.res_C1 >> 4; cc = ch - a.res_C0; }
        src += ((size_t)(b * nb + (cc >> 4)) * (size_t)(a.OH * a.OW)) * 16 + (cc & 15);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pp = (wm * MT + mt) * 16 + p16;
            const int py = pp / TW, px = pp - py * TW;
            const int oy = min(oy0 + py, a.OH - 1), ox = min(ox0 + px, a.OW - 1);
            const float* q = src + (size_t)(oy * a.OW + ox) * 16;
#ifdef MIDD_DMA_CHECK
            dst[mt][0] = sent4(SENT_A); dst[mt][1] = sent4(SENT_A);
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst[mt][0]) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "+v"(dst[mt][1]) : "v"(q) : "memory");
#else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst[mt][0]) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(dst[mt][1]) : "v"(q) : "memory");
#endif
        }
    };
    // The raw registers of a group are written by the load statements and named again ("+v") by ONE wait statement: no
    // use of them can be scheduled above the wait.  Loads and wait of a group sit in the SAME loop iteration, so the
    // compiler has no loop-carried copy of them to make before the data has landed (it did, with the raw registers
    // carried across iterations: copies of not-yet-loaded registers, NaN); what crosses iterations are the split
    // operands, ordinary values.
    auto res_wait = [&](f32x4 (&r)[RG][MT][2], auto n_t) {
        constexpr int N = decltype(n_t)::value;
        if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(%4) ; asm-loads-landed" : "+v"(r[0][0][0]), "+v"(r[0][0][1]), "+v"(r[0][1][0]), "+v"(r[0][1][1]) : "n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%4) ; asm-loads-landed" : "+v"(r[0][0][0]), "+v"(r[0][0][1]), "+v"(r[1][0][0]), "+v"(r[1][0][1]) : "n"(N) : "memory");
    };
    static_assert((MT == 2 && RG == 1) || (MT == 1 && RG == 2), "res_wait names exactly the registers of one group");
    half8 rxh[RG][MT], rxl[RG][MT];                       // split operands of the current group
    auto res_split = [&](int r, f32x4 (&ra)[MT][2], half8 (&oh)[MT], half8 (&ol)[MT]) {
        const bool valid = r * 32 + kq * 8 < a.res_C0 + a.res_C1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#ifdef MIDD_DMA_CHECK
            dma_bad |= has_sent(ra[mt][0], SENT_A) | has_sent(ra[mt][1], SENT_A);
#endif
            f32x4 v0 = ra[mt][0] * res_in, v1 = ra[mt][1] * res_in;
            if (!valid) { v0 = (f32x4){0.f, 0.f, 0.f, 0.f}; v1 = v0; }        // keep the dummy finite
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 hw, lw;
            unsigned hh, ll;
            split_pair(v0[0], v0[1], hh, ll); hw[0] = hh; lw[0] = ll;
            split_pair(v0[2], v0[3], hh, ll); hw[1] = hh; lw[1] = ll;
            split_pair(v1[0], v1[1], hh, ll); hw[2] = hh; lw[2] = ll;
            split_pair(v1[2], v1[3], hh, ll); hw[3] = hh; lw[3] = ll;
            oh[mt] = __builtin_bit_cast(half8, hw);
            ol[mt] = __builtin_bit_cast(half8, lw);
        }
    };
    // one res K step: the ring protocol of k_step (wait for W(step), barrier, refill), operands from registers.
    // in_flight: the next group's RG*RL loads were issued before this step (younger than W(step): they add to the count)
    auto res_mfma = [&](auto in_flight, half8 (&oh)[MT], half8 (&ol)[MT]) {
        constexpr int N = (D - 1) * PPW + (decltype(in_flight)::value ? RG * RL : 0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
        if constexpr (WM != 1) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        issue_w();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { xh[mt] = oh[mt]; xl[mt] = ol[mt]; }
        mfma_step();
    };
    // vmcnt bookkeeping per wave, program order (W = issue_w: PPW pieces; A(g) = RG*RL loads of group g):
    //   A(0) [wait 0] | group 0: A(1) W .. W [wait RG*PPW] | group 1: A(2) W .. W [wait RG*PPW] | ... | last group: W .. W
    // A step's wait for its ring slot W(s) (requested D steps earlier): younger are W(s+1 .. s+D-1) and the group's A if W(s)
    // was requested before them (the group's i-th step: i < D).
    auto res_phase = [&]() {
        if constexpr (RES) {
            if (res_steps == 0) return;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] *= res_rescale;
            {
                f32x4 ra[RG][MT][2];
#pragma unroll
                for (int i = 0; i < RG; ++i) res_load(min(i, res_steps - 1), ra[i]);
                res_wait(ra, std::integral_constant<int, 0>{});
#pragma unroll
                for (int i = 0; i < RG; ++i) res_split(i, ra[i], rxh[i], rxl[i]);
            }
            for (int r = 0; r < res_steps; r += RG) {
                const bool more = r + RG < res_steps;
                if (more) {
                    f32x4 ra[RG][MT][2];
#pragma unroll
                    for (int i = 0; i < RG; ++i) res_load(min(r + RG + i, res_steps - 1), ra[i]);     // (a group's missing last step: a duplicate, unused)
                    // the group's loads are younger than W(step) only while that slot was requested BEFORE them, i.e. for the
                    // group's steps i < D (round 4: with a two-slot ring, D = 1, the second step's slot is requested after the
                    // loads and nothing younger than it may stay outstanding -- found by tests/test_dma_protocol_cpu.py; the
                    // three two-slot tiles with MT = 1 are never picked for the default network)
                    res_mfma(std::true_type{}, rxh[0], rxl[0]);
                    if constexpr (RG == 2) {
#ifdef MIDD_DMA_CHECK_OLD_RES              // the checker's second self-test: round 3's count for this step (the loads counted although they are older)
                        if (r + 1 < res_steps) res_mfma(std::true_type{}, rxh[1], rxl[1]);
#else
                        if (r + 1 < res_steps) res_mfma(std::integral_constant<bool, (1 < D)>{}, rxh[1], rxl[1]);
#endif
                    }
                    res_wait(ra, std::integral_constant<int, RG * PPW>{});
#pragma unroll
                    for (int i = 0; i < RG; ++i) res_split(r + RG + i, ra[i], rxh[i], rxl[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < RG; ++i)
                        if (r + i < res_steps) res_mfma(std::false_type{}, rxh[i], rxl[i]);
                }
            }
        }
    };

    // ---- epilogue (per tile) ------------------------------------------------------------------
    // GroupNorm partial sums of the output run across ALL tiles of this (persistent) workgroup and are
    // published once at the end: one row per (workgroup, wave) instead of one per (tile, wave).  Per tile the
    // 16 pixel lanes are folded (fixed order -> deterministic) and lanes p16 == 0 add into the wave's LDS row.
    // GroupNorm partial sums of the output: each lane keeps the sums of ITS pixels (fixed (pixel lane, cout quad) of every
    // tile it walks) in registers across all tiles of this persistent workgroup; the 16 pixel lanes are folded (DPP row
    // sums, fixed order) and the waves combined through LDS ONCE, at the end.  (Doing the DPP fold and an LDS
    // read-modify-write per tile cost 9 % of the whole sampler: ablation with the statistics removed.)
    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ssq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto epilogue = [&]() {
        // The residual operand comes through loads the compiler does not track (a tracked load is awaited with vmcnt(0) while
        // LDS-DMA traffic is pending, stats_common.h), ALL of the tile's MT*NT at once and awaited once: vmcnt counts stores
        // too, so a wait per 16-pixel row (round 2) also waited for the previous row's output stores to retire.  The K loop's
        // fragment registers are dead here, which is what makes room for them.
        f32x4 rres[MT][NT];
        size_t obase[MT];
        bool rowok[MT];
        const size_t ohw = (size_t)a.OH * a.OW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pp = (wm * MT + mt) * 16 + p16;
            const int py = pp / TW, px = pp - py * TW;
            const int oy = oy0 + py, ox = ox0 + px;
            rowok[mt] = oy < a.OH && ox < a.OW;
            // blocked output [B][Cout/16][OH][OW][16]: the 16 pixel lanes x 4 cout quads of an MFMA tile write one contiguous KiB
            obase[mt] = (((size_t)b * (a.Cout >> 4) + ntile0) * ohw + (size_t)(min(oy, a.OH - 1) * a.OW + min(ox, a.OW - 1))) * 16 + kq * 4;
        }
        if (a.resid != nullptr) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {     // (rows beyond the image: a valid, clamped address; never stored)
#ifdef MIDD_DMA_CHECK
                    rres[mt][nt] = sent4(SENT_A);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(rres[mt][nt]) : "v"(a.resid + obase[mt] + nt * ohw * 16) : "memory");
#else
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rres[mt][nt]) : "v"(a.resid + obase[mt] + nt * ohw * 16) : "memory");
#endif
                }
            // one statement names every destination: nothing that uses (or copies) them can be scheduled above the wait
            if constexpr (MT == 2 && NT == 3) asm volatile("s_waitcnt vmcnt(0) ; asm-loads-landed" : "+v"(rres[0][0]), "+v"(rres[0][1]), "+v"(rres[0][2]), "+v"(rres[1][0]), "+v"(rres[1][1]), "+v"(rres[1][2]) :: "memory");
            else if constexpr (MT == 2 && NT == 2) asm volatile("s_waitcnt vmcnt(0) ; asm-loads-landed" : "+v"(rres[0][0]), "+v"(rres[0][1]), "+v"(rres[1][0]), "+v"(rres[1][1]) :: "memory");
            else if constexpr (MT == 2 && NT == 1) asm volatile("s_waitcnt vmcnt(0) ; asm-loads-landed" : "+v"(rres[0][0]), "+v"(rres[1][0]) :: "memory");
            else if constexpr (MT == 1 && NT == 3) asm volatile("s_waitcnt vmcnt(0) ; asm-loads-landed" : "+v"(rres[0][0]), "+v"(rres[0][1]), "+v"(rres[0][2]) :: "memory");
            else if constexpr (MT == 1 && NT == 2) asm volatile("s_waitcnt vmcnt(0) ; asm-loads-landed" : "+v"(rres[0][0]), "+v"(rres[0][1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0) ; asm-loads-landed" : "+v"(rres[0][0]) :: "memory");
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (rowok[mt]) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 add = *reinterpret_cast<const f32x4*>(add_lds + (wn * NT + nt) * 16 + kq * 4);
                    f32x4 v = acc[mt][nt] * oscale + add;
#ifdef MIDD_DMA_CHECK
                    if (a.resid != nullptr) dma_bad |= has_sent(rres[mt][nt], SENT_A);
#endif
                    if (a.resid != nullptr) v += rres[mt][nt];
                    *reinterpret_cast<f32x4*>(a.out + obase[mt] + nt * ohw * 16) = v;
#if !(defined(C16_ABL) && C16_ABL == 1)  // ablation 1 (wrong results): no statistics of the output
                    ssum[nt] += v; ssq[nt] += v * v;
#endif
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // the waves' LDS rows are folded over wm in a fixed order after a barrier; the workgroup's per-channel sums go to
    // the tensor's totals with exact integer atomics (stats_common.h)
    auto publish_stats = [&]() {
        if (a.stat_tot == nullptr) return;
#if defined(C16_ABL) && (C16_ABL == 1 || C16_ABL == 9)      // ablation 9 (wrong results): sums accumulated, never published
        return;
#endif
        float* const my_stat = stat_lds + wave * (2 * NT * 16) + kq * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[nt][e] = row16_sum(ssum[nt][e]); ssq[nt][e] = row16_sum(ssq[nt][e]); }
            if (p16 == 0) {                     // raw stores: see stat_publish
                lds_store_raw(my_stat + nt * 16, ssum[nt]);
                lds_store_raw(my_stat + NT * 16 + nt * 16, ssq[nt]);
            }
        }
        constexpr int ROWF = 2 * NT * 16;                              // floats of one wave's row
        constexpr int NCOL = WN * NT * 16;                             // channels of this workgroup's slice
        // the waves' rows are folded over wm in a fixed order inside stat_publish (its first barrier publishes them)
        auto fold = [&](int i) {
            const int which = i / NCOL, col = i - which * NCOL;
            const int wn_i = col / (NT * 16), c = col - wn_i * (NT * 16);
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) t += stat_lds[(m * WN + wn_i) * ROWF + which * (NT * 16) + c];
            return t;
        };
        static_assert(G::RAW_BYTES + G::IMG_BYTES + RING * WSLICE >= (NCOL + 2) * STAT_WORDS * 8, "block accumulators in the staging buffers (raw, image, ring: contiguous, idle here)");
        // the landing buffer has been idle for every wave since the last transform; the image / ring behind it may still be
        // read by a wave in its last steps, so accumulators that spill into them wait for everybody first
        if constexpr (G::RAW_BYTES < (NCOL + 2) * STAT_WORDS * 8) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        stat_publish(a.stat_tot, b, a.Cout, a.stat_bs, a.stat_rep, (blockIdx.x - b * a.wgs_per_img) % a.stat_rep, ntile_wg * 16, NCOL,
                     fold, reinterpret_cast<stat_word*>(raw), tid, NTHREADS);
    };

    // ---- tile / chunk loop -----------------------------------------------------------------------
    for (;;) {
        const int next_tile = trem + a.wgs_per_img;
        const bool has_next_tile = next_tile < tiles_per_img;
        for (int c = 0; c < nchunks; ++c) {
            const bool more_in_tile = (c + 1 < nchunks);
            const bool more = more_in_tile || has_next_tile;
            const int next_chunk = more_in_tile ? c + 1 : 0;
            const bool full = (CB == 2) && (2 * c + 1 < nblk);
            // the staging geometry switches to the next tile right before its first chunk is requested
            // (every transform of the current tile is done by then; the epilogue does not use it)
            if (!more_in_tile && has_next_tile) set_tile(next_tile);
            auto run_chunk = [&](auto more_t) {
                constexpr bool MORE = decltype(more_t)::value;
                if (full) {
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        const int dy = tap / KS, dx = tap - dy * KS;
                        int xo[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) xo[mt] = frag_full[mt] + (dy * IW + dx) * 32;
                        if (MORE && tap >= 1 && tap <= D) k_step(std::true_type{}, false, false, next_chunk, xo);
                        else                              k_step(std::false_type{}, tap == 0, MORE && tap == 0, next_chunk, xo);
                    }
                } else {
#pragma unroll
                    for (int hs = 0; hs < HSTEPS; ++hs) {
                        const int t0 = 2 * hs, t1 = (2 * hs + 1 < TAPS) ? 2 * hs + 1 : 0;   // padded half has zero weights
                        const int o0 = ((t0 / KS) * IW + (t0 % KS)) * 32, o1 = ((t1 / KS) * IW + (t1 % KS)) * 32;
                        const int to = (kq >> 1) ? o1 : o0;
                        int xo[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) xo[mt] = frag_base[mt] + to;
                        if (MORE && hs >= 1 && hs <= D) k_step(std::true_type{}, false, false, next_chunk, xo);
                        else                            k_step(std::false_type{}, hs == 0, MORE && hs == 0, next_chunk, xo);
                    }
                }
            };
            if (more) {
                run_chunk(std::true_type{});
                if (!more_in_tile) { res_phase(); TS(TS_RES) }         // the tile's 3x3 steps are done: the folded res_conv's steps
                // every wave is done reading the image, and A(next) (older than the last min(steps after it, D)
                // weight groups) has landed, before the image is rewritten
                const int after = (full ? TAPS : HSTEPS) - 1 + (more_in_tile ? 0 : res_steps);
                if (after >= D) wait_vm_and_barrier<D * PPW>();
                else if (after == 1) wait_vm_and_barrier<PPW>();
                else wait_vm_and_barrier<0>();
                TS(TS_CHUNK_WAIT)
                if (!more_in_tile) {                    // tile finished: store it, move to the next one
                    epilogue();
                    trem = next_tile;
                    oy0 = (trem / a.tiles_x) * TH; ox0 = (trem % a.tiles_x) * TW;
                    TS(TS_EPILOGUE)
#ifdef MIDD_CONV_TIMING
                    ts_after_epi = true;
#endif
                }
#if !(defined(C16_ABL) && C16_ABL == 5)  // ablation 5 (wrong results): no transform in the loop
                transform(next_chunk);
#endif
                TS(TS_TRANSFORM)
                if constexpr (WM == 1) {        // steps have no barrier of their own: publish the new image here
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            } else {
                run_chunk(std::false_type{});
                res_phase();
                TS(TS_RES)
            }
        }
        if (!has_next_tile) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight refills issued past the last step
    TS(TS_CHUNK_WAIT)
    epilogue();
    TS(TS_EPILOGUE)
    publish_stats();
#ifdef MIDD_DMA_CHECK
    if (dma_bad && a.status != nullptr) atomicOr(a.status, (int)STATUS_DMA_EARLY);
#endif
    TS(TS_PUBLISH)
#ifdef MIDD_CONV_TIMING
    if (tid == 0) {
        ts_acc[TS_TOTAL] = ts_last - ts_t0;
        ts_acc[TS_REAL] = __builtin_amdgcn_s_memrealtime() - ts_r0;
        ts_acc[TS_WGS] = 1;
        for (int k = 0; k < TS_N; ++k) atomicAdd(&g_conv_timing[a.dbg_slot][k], ts_acc[k]);
    }
#endif
}

// ------------------------------------------------------------------------------ dispatch
#ifdef MIDD_CONV_TIMING
static std::vector<std::string> g_timing_names;
static int conv_timing_slot(int ks, int st, int tw, int mt, int nt, int wm, int wn, int oh, int cin, int cout, int B, int ring, int wgs) {
    char buf[160];
    snprintf(buf, sizeof buf, "k%d s%d tile(%d,%d,%d,%d,%d) ring%d out%d^2 %d->%d B%d wgs%d", ks, st, tw, mt, nt, wm, wn, ring, oh, cin, cout, B, wgs);
    for (size_t i = 0; i < g_timing_names.size(); ++i) if (g_timing_names[i] == buf) return (int)i;
    if (g_timing_names.size() >= 63) return 63;
    g_timing_names.push_back(buf);
    return (int)g_timing_names.size() - 1;
}
extern "C" __attribute__((visibility("default"))) void mi_debug_conv_timing_dump(void) {
    static unsigned long long h[64][TS_N];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_timing), sizeof h);
    static const char* names[] = {"barrier", "dma-issue", "frag+mfma", "chunk-wait", "transform", "epilogue", "prologue", "wait-after-epi", "dma-wait", "res-phase", "publish"};
    for (size_t i = 0; i < g_timing_names.size(); ++i) {
        const double tot = (double)h[i][TS_TOTAL], wgs = (double)h[i][TS_WGS];
        if (wgs == 0) continue;
        printf("%-70s cyc/wg %9.0f clk %.2f GHz |", g_timing_names[i].c_str(), tot / wgs, tot / (double)h[i][TS_REAL] * 0.1);
        for (int k = 0; k < 11; ++k) printf(" %s %4.1f%%", names[k], 100.0 * (double)h[i][k] / tot);
        printf("\n");
    }
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_timing), h, sizeof h);
    fflush(stdout);
}
#endif
template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN, bool RES, int CBT = 0>
static hipError_t launch16(const ConvArgs& a0, hipStream_t s) {
    using G = Conv16Geom<KS, STRIDE, TW, MT, NT, WM, WN, CBT>;
    ConvArgs a = a0;
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + G::TH - 1) / G::TH;
    const int ny = a.Cout / (WN * NT * 16);
    a.wgs_per_img = conv16_wgs_per_img(a.tiles_x * a.tiles_y, a.B, ny, a.persist_wgs);
    dim3 grid(a.B * a.wgs_per_img, ny);
#ifdef MIDD_CONV_TIMING
    a.dbg_slot = conv_timing_slot(KS, STRIDE, TW, MT * 10 + G::CB, NT, WM, WN, a.OH, a.C0 + a.C1, a.Cout, a.B, G::RING, (int)grid.x * (int)grid.y);
#endif
    if constexpr (G::LDS_BYTES <= 160 * 1024 && (G::RING - 2) * G::PPW + G::APW <= 60) {
        const int lds_bytes = G::lds_bytes(a.C0 + a.C1);
        if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
        {
            static int raised[MIDD_MAX_DEVICES] = {};      // per instantiation and device
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_mfma_f16x3_kernel<KS, STRIDE, TW, MT, NT, WM, WN, RES, CBT>), lds_bytes, raised);
            if (e != hipSuccess) return e;
        }
        if ((double)a.H * a.W * 64.0 >= 4294967296.0) return hipErrorInvalidValue;  // 32-bit DMA offsets inside one block plane
        hipLaunchKernelGGL((conv_mfma_f16x3_kernel<KS, STRIDE, TW, MT, NT, WM, WN, RES, CBT>), grid, dim3(G::NTHREADS), lds_bytes, s, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;        // tile never picked (conv16_pick_tile), not instantiated
    }
}

// ~3 resident workgroups per CU; a sample's tiles are dealt evenly to its persistent workgroups
int conv16_wgs_per_img(int tiles, int B, int ny, int target) {
    const int target_wgs = target ? target : 768;
    int per_img = target_wgs / (B * ny);
    if (per_img < 1) per_img = 1;
    if (per_img > tiles) per_img = tiles;
    const int tiles_per_wg = (tiles + per_img - 1) / per_img;
    return (tiles + tiles_per_wg - 1) / tiles_per_wg;
}

// Tiles the picker can reach.  Measured and dropped (same-box A/B at B=8, 256x256, round 1 and again in round 2 on the
// atomics-statistics build): 16x16-pixel tiles (MT = 4: -2..-3 %), 96-cout tiles (NT = 6: -1.7 %), 4x2-wave and 8-wave
// workgroups (neutral to negative).
#define MIDD_CONV16_TILES(X)                  \
    /*  tw  mt nt wm wn */                    \
    X(16, 2, 3, 4, 1) X(16, 1, 3, 4, 1) X(8, 1, 3, 2, 1) \
    X(16, 2, 3, 2, 2) X(16, 1, 3, 2, 2) X(8, 1, 3, 1, 2) \
    X(16, 2, 3, 1, 3) X(8, 1, 3, 1, 3)                   \
    X(16, 2, 3, 1, 4) X(8, 2, 3, 1, 4) X(8, 1, 3, 1, 4)  \
    X(16, 2, 2, 4, 1) X(8, 1, 2, 2, 1)                   \
    X(16, 2, 2, 2, 2) X(8, 1, 2, 1, 2)                   \
    X(16, 2, 1, 4, 1) X(8, 1, 1, 2, 1)

struct Tile16 { int tw, mt, nt, wm, wn; };
static const Tile16 kTiles16[] = {
#define X(tw, mt, nt, wm, wn) {tw, mt, nt, wm, wn},
    MIDD_CONV16_TILES(X)
#undef X
};

// run-time mirror of Conv16Geom::LDS_BYTES / the vmcnt-encoding limit, for the tile picker
static bool tile16_fits(const Tile16& d, int ks, int stride, int cb = 0) {
    const int nw = d.wm * d.wn, nthreads = nw * 64;
    const int bm = d.wm * d.mt * 16, th = bm / d.tw;
    const int ih = (th - 1) * stride + ks, iw = (d.tw - 1) * stride + ks;
    if (cb == 0) cb = conv16_cb(ks);
    const bool wide = ks == 3 && cb == 2;
    const int npix = ih * iw, apw = (npix * 4 * cb + nthreads - 1) / nthreads;
    const int wpieces = d.wn * d.nt * 2, ppw = (wpieces + nw - 1) / nw;
    const long fixed = (long)apw * nthreads * 16 + 2L * cb * npix * 32 + (nw * 2 * d.nt * 16 + d.wn * d.nt * 16) * 4 + 2 * 384 * 4 + 64;
    long ring = ((long)(stride == 2 ? 120 : wide ? MIDD_LDS_WIDE_KB : MIDD_LDS_TARGET_KB) * 1024 - fixed) / (wpieces * 1024);
    ring = ring < 2 ? 2 : (ring > MIDD_RING_MAX ? MIDD_RING_MAX : ring);
    const long lds = fixed + ring * wpieces * 1024;
    return lds <= 160 * 1024 && (ring - 2) * ppw + apw <= 60;
}

// Launches whose grid is at most this many workgroups take the wide-chunk variant of the 4x1-wave tiles (they would
// leave the third workgroup slot of a CU empty anyway).
#ifndef MIDD_WIDE_MAX_WGS
#define MIDD_WIDE_MAX_WGS 512
#endif
bool conv16_pick_tile(int Cin, int Cout, int B, int OH, int OW, int ks, int stride, ConvTile* t, bool allow_wide) {
    if (Cout % 16) return false;
    if (!((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1))) return false;
    if (ks == 1 && conv1x1_pick_tile(Cin, Cout, B, OH, OW, t)) return true;      // dedicated 1x1 kernel (conv1x1_f16x3.hip)
    const int nt = (Cout % 48 == 0) ? 3 : (Cout % 32 == 0) ? 2 : 1;
    const Tile16* best = nullptr;
    long best_score = -(1L << 60), best_wgs = 0;
    // 192, not 256 workgroups: at B=4 (half-batches) the 32x32 layers with 144 couts would otherwise drop to 32-pixel
    // 2-wave tiles that stream the weights twice as often (same-box A/B: +1.7 %)
    constexpr long min_wgs = 192;
    for (const Tile16& d : kTiles16) {
        if (d.nt != nt) continue;
        const int nn_d = Cout / (16 * d.nt);              // cout slices of 16*nt; a workgroup takes wn of them
        if (nn_d % d.wn) continue;
        if (!tile16_fits(d, ks, stride)) continue;
        const int bm = d.wm * d.mt * 16, th = bm / d.tw;
        const long tiles = (long)((OW + d.tw - 1) / d.tw) * ((OH + th - 1) / th);
        const long wgs = (long)B * tiles * (nn_d / d.wn);
        const long covered = tiles * d.tw * th;
        const bool wasteful = covered * 4 > (long)OH * OW * 5;
        // enough workgroups first; then the pixels one weight fetch is shared over (the weight stream from L2 is what
        // starves small tiles), then the couts one activation staging is shared over
        const long share = (long)bm * 8 + d.wn;
        // (forcing the 2x2-wave 96-cout tile on the <= 32x32 / <= 64x64 maps -- the GroupNorm / SiLU / split transform shared by
        // two cout slices -- measured -6.5 % / -7 %: its 12 KB weight slices leave a two-slot ring, one step in flight)
        const long score = (wgs >= min_wgs ? 1000000 : wgs * (1000000 / min_wgs)) + share - (wasteful ? 500000 : 0);
        if (score > best_score) { best_score = score; best = &d; best_wgs = wgs; }
    }
    if (!best) return false;
    *t = ConvTile{ks, stride, best->tw, best->mt, best->nt, best->wm, best->wn};
    if (allow_wide && ks == 3 && stride == 1 && best->tw == 16 && best->nt == 3 && best->wm == 4 && best->wn == 1 && Cin >= 32 &&
        best_wgs <= MIDD_WIDE_MAX_WGS && tile16_fits(*best, ks, stride, 2))
        t->cb = 2;
    return true;
}

template <int KS, int STRIDE, int TW, int MT, int NT, int WM, int WN, int CBT = 0>
static bool info16(int Cin, int Cout, int B, int OH, int OW, int persist_wgs, ConvLaunchInfo* o) {
    using G = Conv16Geom<KS, STRIDE, TW, MT, NT, WM, WN, CBT>;
    o->tiles_x = (OW + TW - 1) / TW; o->tiles_y = (OH + G::TH - 1) / G::TH;
    o->grid_y = Cout / (WN * NT * 16);
    o->wgs_per_img = conv16_wgs_per_img(o->tiles_x * o->tiles_y, B, o->grid_y, persist_wgs);
    o->grid_x = B * o->wgs_per_img;
    o->ring = G::RING; o->ppw = G::PPW; o->apw = G::APW; o->lds_bytes = G::lds_bytes(Cin);
    return true;
}
bool conv16_launch_info(int Cin, int Cout, int B, int OH, int OW, const ConvTile& t, int persist_wgs, ConvLaunchInfo* out) {
    if (t.ks == 1 && t.tw == 0) return conv1x1_launch_info(Cin, Cout, B, OH, OW, t, persist_wgs, ATT_NONE, out);
    if (t.cb == 2) {
        if (t.mt == 2) return info16<3, 1, 16, 2, 3, 4, 1, 2>(Cin, Cout, B, OH, OW, persist_wgs, out);
        if (t.mt == 1) return info16<3, 1, 16, 1, 3, 4, 1, 2>(Cin, Cout, B, OH, OW, persist_wgs, out);
        return false;
    }
#define X(tw_, mt_, nt_, wm_, wn_)                                                            \
    if (t.tw == tw_ && t.mt == mt_ && t.nt == nt_ && t.wm == wm_ && t.wn == wn_) {           \
        if (t.ks == 3 && t.stride == 1) return info16<3, 1, tw_, mt_, nt_, wm_, wn_>(Cin, Cout, B, OH, OW, persist_wgs, out); \
        if (t.ks == 3 && t.stride == 2) return info16<3, 2, tw_, mt_, nt_, wm_, wn_>(Cin, Cout, B, OH, OW, persist_wgs, out); \
        if (t.ks == 1 && t.stride == 1) return info16<1, 1, tw_, mt_, nt_, wm_, wn_>(Cin, Cout, B, OH, OW, persist_wgs, out); \
    }
    MIDD_CONV16_TILES(X)
#undef X
    return false;
}

hipError_t conv16_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s) {
    if (t.ks == 1 && t.tw == 0) return conv1x1_launch(a, t, s);
    if (t.cb == 2) {          // wide chunks: the two 4x1-wave tiles the picker marks (conv16_pick_tile)
        if (!(t.ks == 3 && t.stride == 1 && t.tw == 16 && t.nt == 3 && t.wm == 4 && t.wn == 1)) return hipErrorInvalidValue;
        if (t.mt == 2) return a.res_steps > 0 ? launch16<3, 1, 16, 2, 3, 4, 1, true, 2>(a, s) : launch16<3, 1, 16, 2, 3, 4, 1, false, 2>(a, s);
        if (t.mt == 1) return a.res_steps > 0 ? launch16<3, 1, 16, 1, 3, 4, 1, true, 2>(a, s) : launch16<3, 1, 16, 1, 3, 4, 1, false, 2>(a, s);
        return hipErrorInvalidValue;
    }
#define X(tw_, mt_, nt_, wm_, wn_)                                                            \
    if (t.tw == tw_ && t.mt == mt_ && t.nt == nt_ && t.wm == wm_ && t.wn == wn_) {           \
        if (t.ks == 3 && t.stride == 1 && a.res_steps > 0) return launch16<3, 1, tw_, mt_, nt_, wm_, wn_, true>(a, s); \
        if (t.ks == 3 && t.stride == 1) return launch16<3, 1, tw_, mt_, nt_, wm_, wn_, false>(a, s); \
        if (t.ks == 3 && t.stride == 2) return launch16<3, 2, tw_, mt_, nt_, wm_, wn_, false>(a, s); \
        if (t.ks == 1 && t.stride == 1) return launch16<1, 1, tw_, mt_, nt_, wm_, wn_, false>(a, s); \
    }
    MIDD_CONV16_TILES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace midd
