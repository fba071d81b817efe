// 3x3 stride-1 convolution in f16x3 arithmetic, WAVE-SPECIALISED: the layers that dominate the sampler
// (/root/reference/Backend/DDIM/DDIMModel.py:118,124 -- both convolutions of every ResidualBlock; the folded
// ConvTranspose of :211+:241).  Same contract, number format, tile (16 x 8 pixels x 48 couts), weight pack, LDS image
// and epilogue as conv_mfma_f16x3.hip; what changes is who does what.
//
// In the general kernel every wave runs the whole pipeline in turn -- DMA issue, counted wait, barrier, transform
// (GroupNorm-apply / SiLU / hi-lo split of the staged raw chunk), fragment reads, MFMAs, epilogue -- and its in-kernel
// stamps show the phases of a workgroup ADDING UP: fragment reads + MFMA are 17-33 % of a workgroup's cycles, the
// transform 12-28 %, DMA issue 8-22 %, DMA waits 6-20 %; three co-resident workgroups hide only
// part of it (MFMA pipe 36 % busy).  Here a workgroup is 8 waves:
//   waves 0-3  CONSUMERS  fragment reads + MFMAs (32 pixels x 48 couts each), epilogue, statistics -- nothing else;
//   waves 4-7  PRODUCERS  weight ring refills, raw activation chunks TWO chunks ahead (double-buffered raw landing
//                         zone: ten K-steps, ~5 us, of HBM latency cover instead of one chunk), and the transform of
//                         chunk c+1 into the OTHER of two MFMA images while the consumers multiply chunk c.
// One s_barrier per K-step joins the two groups, and nobody waits for its own latest LDS traffic at it: a producer only
// guarantees that what it wrote in the step BEFORE the previous one is complete (LDS operations retire in order: a
// counted lgkmcnt that leaves the previous step's operations outstanding), so a step's ring copy (three steps ahead of
// its use, RING = 4) and image writes (finished two steps before the chunk boundary) are never on the critical path;
// a consumer reads the NEXT step's weight fragments right after the barrier and multiplies with the ones it read a
// step earlier (register double buffer), its activation fragments are requested ahead of the barrier.  Two such
// workgroups per CU (77 KB LDS each, <= 128 VGPRs): two MFMA-only waves per SIMD.
//
// Weights go global -> LDS by LDS-DMA into a 4-slot ring, three steps ahead.  (Prefetching them a whole chunk ahead
// into producer registers was tried: hipcc guards every use of a register filled by a global load with vmcnt(0) once
// opaque asm waits are around, which drains the raw-chunk DMA every step; and the ablations below say the weight
// latency is not what limits this kernel.)
//
// vmcnt accounting of a producer wave (PPW weight pieces per step, APW raw pieces per chunk, D = 3 steps ahead): step k of
// chunk q issues, after its barrier, W(s+D) and -- in step 0, behind it -- the raw chunk q+2.  The consumers read W(s+1)
// right after barrier s, so before that barrier the producer waits for W(s+1), issued two steps earlier: younger than it
// are W(s+2) and, when one of those two steps was a step 0 (k = 1, 2), the raw chunk: N = PPW (+ APW).  Raw chunk q+1,
// issued a whole chunk earlier, is older than anything waited for in chunk q: it has landed when its transform starts.
//
// Ablations (diagnostic builds, wrong results; whole sampler, B = 8, unsplit, ms in this kernel per denoise call):
// as is 150.9 | no SiLU 142.7 | ONE MFMA pass instead of three 143.2 | no transform in the loop 121.3 | no per-tile
// epilogue 101.5 -- the kernel is bound by its memory / synchronisation skeleton, the epilogue and the transform, not
// by the matrix pipe.
#include "f16x3_common.h"
#include <cstdio>
#include <cstdlib>

namespace midd {

// Diagnostic build only (-DMIDD_CONV_TIMING, `make timing`): s_memtime stamps of consumer wave 0 and producer wave 4 of
// every workgroup, summed per launch.  Never compiled into libmidd.so.
#ifdef MIDD_CONV_TIMING
enum { WT_C_BARRIER, WT_C_COMPUTE, WT_C_EPI, WT_C_TOTAL, WT_P_BARRIER, WT_P_WAITVM, WT_P_WORK, WT_P_TOTAL, WT_STEPS, WT_WGS, WT_N };
__device__ unsigned long long g_ws_timing[WT_N];
__device__ __forceinline__ unsigned long long wt_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define WT_DECL unsigned long long wt_acc[WT_N] = {}; unsigned long long wt_last = wt_stamp(); const unsigned long long wt_t0 = wt_last;
#define WT(k) { const unsigned long long t_ = wt_stamp(); wt_acc[k] += t_ - wt_last; wt_last = t_; }
#else
#define WT_DECL
#define WT(k)
#endif

struct WsGeom {
    static constexpr int TW = 16, TH = 8, MT = 2, NT = 3, NCONS = 4, NPROD = 4;
    static constexpr int NTHREADS = (NCONS + NPROD) * 64, PTHREADS = NPROD * 64;
    static constexpr int IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int NSLOT = NPIX * 4;                                     // 16-byte slots (pixel, 4-channel quad) per chunk
    static constexpr int APW = (NSLOT + PTHREADS - 1) / PTHREADS;              // raw DMA pieces per producer wave and chunk
    static constexpr int RAW_BYTES = APW * PTHREADS * 16;
    static constexpr int PLANE = NPIX * 32, IMG_BYTES = 2 * PLANE;
    static constexpr int WPIECES = NT * 2, PPW = (WPIECES + NPROD - 1) / NPROD, WSLICE = WPIECES * 1024;
    static constexpr int RING = 4, D = RING - 1, HSTEPS = 5;
    static constexpr int STAT_FLOATS = NCONS * 2 * NT * 16, ADD_FLOATS = NT * 16;
    static constexpr int FIXED = 2 * RAW_BYTES + 2 * IMG_BYTES + RING * WSLICE + (STAT_FLOATS + ADD_FLOATS) * 4 + 64;
    static int lds_bytes(int cin) { return FIXED + 2 * cin * 4; }
};

__global__ __launch_bounds__(WsGeom::NTHREADS, 4)      // two 8-wave workgroups per CU: <= 128 VGPRs
void conv3x3_ws_f16x3_kernel(const ConvArgs a) {
    using G = WsGeom;
    constexpr int TW = G::TW, TH = G::TH, IW = G::IW, MT = G::MT, NT = G::NT, APW = G::APW, PPW = G::PPW;
    constexpr int PLANE = G::PLANE, WSLICE = G::WSLICE, RING = G::RING, D = G::D, HSTEPS = G::HSTEPS;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const raw0 = lds;                                        // two raw landing zones
    char* const img0 = raw0 + 2 * G::RAW_BYTES;                    // two MFMA images [hi|lo][halo pixel][16 fp16]
    char* const wring = img0 + 2 * G::IMG_BYTES;
    float* const stat_lds = reinterpret_cast<float*>(wring + RING * WSLICE);      // [consumer wave][2][48]
    float* const add_lds = stat_lds + G::STAT_FLOATS;                              // [48] bias (+ time embedding)
    float* const gnp = add_lds + G::ADD_FLOATS;                                    // [2][Cin] scale, shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= G::NCONS;
    const int Cin = a.C0 + a.C1, nchunks = Cin >> 4;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / a.wgs_per_img;
    const int first_tile = blockIdx.x - b * a.wgs_per_img;
    const int my_tiles = (tiles_per_img - first_tile + a.wgs_per_img - 1) / a.wgs_per_img;
    const int total_chunks = my_tiles * nchunks;                  // the workgroup's chunk sequence q = 0 .. total_chunks-1
    const int ntiles_total = a.Cout >> 4, ntile_wg = blockIdx.y * NT;

    // ---- prologue, all 8 waves: GroupNorm scale / shift, bias + time embedding, zeroed statistics ----
    if (a.prologue != PRO_RAW)
        gn_prologue_lds(a.gn_tot0, a.C0, a.gn_tot1, a.C1, a.stat_rep, a.gn_gamma, a.gn_beta, a.gn_eps, a.gn_hw, b, ACT_PRESCALE, gnp, tid, G::NTHREADS);
    for (int i = tid; i < G::STAT_FLOATS; i += G::NTHREADS) stat_lds[i] = 0.f;
    {
        const int trow = (a.temb != nullptr) ? a.trow[b] : 0;
        for (int i = tid; i < G::ADD_FLOATS; i += G::NTHREADS) {
            const int co = ntile_wg * 16 + i;
            add_lds[i] = a.bias[co] + (a.temb != nullptr ? a.temb[(size_t)trow * a.temb_stride + co] : 0.f);
        }
    }

    if (producer) {
        // =========================================================================== PRODUCERS
        const int ptid = tid - G::NCONS * 64, pwave = wave - G::NCONS;
        const int q4 = ptid & 3;                                   // 4-channel quad of the 16-channel chunk
        // weights: the step sequence is the same for every tile, walked cyclically
        const char* const wbase = reinterpret_cast<const char*>(a.wpack) + (size_t)ntile_wg * 2048;
        const size_t wstep_bytes = (size_t)ntiles_total * 2048;
        const int steps_per_tile = nchunks * HSTEPS;
        const int lane16 = lane * 16;
        int piece_off[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int piece = pwave + i * G::NPROD;
            if (piece >= G::WPIECES) piece -= G::WPIECES;          // padding duplicate: same bytes, same place
            piece_off[i] = piece * 1024 + lane16;
        }
        int wr_slot = 0, wr_step = 0;                              // ring slot / tile-relative step of the next refill
        const char* wr_src = wbase;
        auto issue_w = [&]() {
            char* slot = wring + wr_slot * WSLICE;
#pragma unroll
            for (int i = 0; i < PPW; ++i) dma16(wr_src + piece_off[i], slot + piece_off[i] - lane16);
            ++wr_step; wr_src += wstep_bytes;
            if (wr_step == steps_per_tile) { wr_step = 0; wr_src = wbase; }
            wr_slot = (wr_slot + 1 == RING) ? 0 : wr_slot + 1;
        };
        // activations: slot = (halo pixel, quad); valid[parity] remembers, per landing zone, which of this thread's slots
        // lie inside the image (the conv pads its NORMALISED input with zeros, so the mask is applied after the transform)
        unsigned valid[2] = {0u, 0u};
        int s_iy[APW], s_ix[APW];                                  // this thread's halo pixels (the same in every tile)
#pragma unroll
        for (int s = 0; s < APW; ++s) {
            const int pix = min((ptid + s * G::PTHREADS) >> 2, G::NPIX - 1);
            s_iy[s] = pix / IW; s_ix[s] = pix - s_iy[s] * IW;
        }
        auto issue_a = [&](int q, int parity) {                    // chunk q of the sequence -> raw[parity]
            const int tile = first_tile + (q / nchunks) * a.wgs_per_img, c = q % nchunks;
            const int iy0 = (tile / a.tiles_x) * TH - 1, ix0 = (tile % a.tiles_x) * TW - 1;
            const int ch = (c << 4) + q4 * 4;
            const float* src; unsigned cs4, coff;
            if (ch < a.C0) { src = a.src0; cs4 = a.C0 * 4u; coff = ch * 4u; }
            else           { src = a.src1; cs4 = a.C1 * 4u; coff = (ch - a.C0) * 4u; }
            const char* base = reinterpret_cast<const char*>(src);
            char* dst = raw0 + parity * G::RAW_BYTES;
            unsigned v = 0u;
#pragma unroll
            for (int s = 0; s < APW; ++s) {
                const int gy = iy0 + s_iy[s], gx = ix0 + s_ix[s];
                const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                v |= (in ? 1u : 0u) << s;
                const unsigned off = in ? (unsigned)((b * a.H + gy) * a.W + gx) : 0u;      // tensors are < 4 GiB (host-checked)
                dma16(base + off * cs4 + coff, dst + (pwave + s * G::NPROD) * 1024);
            }
            valid[parity] = v;
        };
        auto transform_slot = [&](int q, int s) {                  // raw[q & 1] slot s -> img[q & 1]
            const int slot = ptid + s * G::PTHREADS;
            if (slot >= G::NSLOT) return;
            const int ch = ((q % nchunks) << 4) + q4 * 4;
            f32x4 sc = {RAW_PRESCALE, RAW_PRESCALE, RAW_PRESCALE, RAW_PRESCALE}, sh = {0.f, 0.f, 0.f, 0.f};
            if (a.prologue != PRO_RAW) {                           // 2^s folded into the affine by the prologue
                sc = *reinterpret_cast<const f32x4*>(gnp + ch);
                sh = *reinterpret_cast<const f32x4*>(gnp + Cin + ch);
            }
            f32x4 v = *reinterpret_cast<const f32x4*>(raw0 + (q & 1) * G::RAW_BYTES + slot * 16);
            v = v * sc + sh;
#if defined(WS_ABL) && WS_ABL == 1      // ablation (wrong results): no SiLU
            if (false) {
#else
            if (a.prologue == PRO_GN_SILU) {
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e)      // v = 16*y: silu -> v * 1/(1 + 2^(-y*log2 e))
                    v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[e] * (-1.4426950408889634f / ACT_PRESCALE)));
            }
            if (!((valid[q & 1] >> s) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            half4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 h = (_Float16)v[e];
                hi[e] = h;
                lo[e] = (_Float16)(v[e] - (float)h);
            }
            char* base = img0 + (q & 1) * G::IMG_BYTES + (slot >> 2) * 32 + q4 * 8;
            *reinterpret_cast<half4*>(base) = hi;
            *reinterpret_cast<half4*>(base + PLANE) = lo;
        };

        issue_a(0, 0);
        issue_a(min(1, total_chunks - 1), 1);
#pragma unroll
        for (int i = 0; i < D; ++i) issue_w();                     // steps 0..D-1 -> ring slots 0..D-1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // [P1] raw chunks 0 / 1, weights 0..D-1, gnp, add, stats visible
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < APW; ++s) transform_slot(0, s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // [P2] image of chunk 0 published
        asm volatile("" ::: "memory");

        WT_DECL
        for (int q = 0; q < total_chunks; ++q) {
            const bool more1 = q + 1 < total_chunks;
#pragma unroll
            for (int k = 0; k < HSTEPS; ++k) {
                // W(s+1) has landed (the consumers read it right after this barrier); of the LDS operations everything but
                // the previous step's transform (1 read + 2 writes, steps 0..APW-1) is complete -- they retire in order
#if defined(WS_ABL) && WS_ABL == 5      // ablation 5 (wrong results): never wait for DMA in the loop
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
#elif defined(WS_ABL) && WS_ABL == 6    // ablation 6 (wrong results): no weight refills in the loop
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
#else
                if (k == 1 || k == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(3)" ::"n"((D - 2) * PPW + APW) : "memory");
                else                  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(3)" ::"n"((D - 2) * PPW) : "memory");
#endif
                WT(WT_P_WORK)
                __builtin_amdgcn_s_barrier();                      // step barrier (consumers: done with the previous ring slot / image)
                asm volatile("" ::: "memory");
                WT(WT_P_BARRIER)
#if !(defined(WS_ABL) && WS_ABL == 6)
                issue_w();                                         // W(s + D) -> the slot the consumers finished two steps ago
#endif
                if (k == 0) issue_a(min(q + 2, total_chunks - 1), q & 1);   // raw[q & 1]: chunk q was transformed during chunk q-1 (past the end: a dummy)
                static_assert(APW <= HSTEPS - 2, "the next image must be complete two steps before the chunk boundary");
#if !(defined(WS_ABL) && WS_ABL == 3)   // ablation 3 (wrong results): no transform in the loop
                if (more1 && k < APW) transform_slot(q + 1, k);    // chunk q+1 -> the image the consumers are NOT reading: one slot per step
#endif
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the refills issued past the last step
#ifdef MIDD_CONV_TIMING
        if (tid == G::NCONS * 64) {
            wt_acc[WT_P_TOTAL] = wt_stamp() - wt_t0;
            for (int k = WT_P_BARRIER; k <= WT_P_TOTAL; ++k) atomicAdd(&g_ws_timing[k], wt_acc[k]);
        }
#endif
        __builtin_amdgcn_s_barrier();                              // [E1] consumers: statistics rows complete
        return;
    }

    // =============================================================================== CONSUMERS
    const int p16 = lane & 15, kq = lane >> 4;
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

    float* const my_stat = stat_lds + wave * (2 * NT * 16) + kq * 4;
    // Per tile: out = acc * 2^-(k+s) + bias (+ temb) (+ residual), NHWC 16-byte stores, per-channel sums for the next
    // GroupNorm.  Tiles inside the image (every tile of the 2^n-sized maps) take the branch-free path: the six residual
    // loads are requested together and the six stores issued back to back -- in the predicated form hipcc waits
    // vmcnt(0) in every block, i.e. for the previous STORE to complete (stores count in vmcnt): six write round trips
    // per tile, a third of the kernel in an ablation.
    auto epilogue = [&](int tile) {
        const int oy0 = (tile / a.tiles_x) * TH, ox0 = (tile % a.tiles_x) * TW;
        const bool full = oy0 + TH <= a.OH && ox0 + TW <= a.OW;
        f32x4 tsum[NT], tsq[NT];
        if (full) {
            size_t o[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pp = (wave * MT + mt) * 16 + p16;
                const int py = pp / TW, px = pp - py * TW;
                o[mt] = ((size_t)(b * a.OH + oy0 + py) * a.OW + ox0 + px) * a.Cout + ntile_wg * 16 + kq * 4;
            }
            f32x4 r[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) r[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (a.resid != nullptr) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) r[mt][nt] = *reinterpret_cast<const f32x4*>(a.resid + o[mt] + nt * 16);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                tsum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tsq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 add = *reinterpret_cast<const f32x4*>(add_lds + nt * 16 + kq * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 v = (acc[mt][nt] * a.out_scale + add) + r[mt][nt];
                    r[mt][nt] = v;
                    tsum[nt] += v; tsq[nt] += v * v;
                    acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(a.out + o[mt] + nt * 16) = r[mt][nt];
        } else {
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

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // [P1]
    asm volatile("" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // [P2] image of chunk 0 readable
    asm volatile("" ::: "memory");

    // weight fragments: the step being multiplied (wh_c / wl_c) and, read right after the step barrier, the next one
    half8 wh_c[NT], wl_c[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        wh_c[nt] = *reinterpret_cast<const half8*>(wring + wfrag_off + nt * 2048);
        wl_c[nt] = *reinterpret_cast<const half8*>(wring + wfrag_off + nt * 2048 + 1024);
    }
    int rd_slot = 1;                            // ring slot of the NEXT step
    WT_DECL
    for (int q = 0; q < total_chunks; ++q) {
        const char* img = img0 + (q & 1) * G::IMG_BYTES;
#pragma unroll
        for (int k = 0; k < HSTEPS; ++k) {
            const int t0 = 2 * k, t1 = (2 * k + 1 < 9) ? 2 * k + 1 : 0;          // padded half has zero weights
            const int o0 = ((t0 / 3) * IW + (t0 % 3)) * 32, o1 = ((t1 / 3) * IW + (t1 % 3)) * 32;
            const int to = (kq >> 1) ? o1 : o0;
            // the chunk's image is complete two steps before the chunk starts: its fragments are requested ahead of the barrier
            half8 xh[MT], xl[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                xh[mt] = *reinterpret_cast<const half8*>(img + frag_base[mt] + to);
                xl[mt] = *reinterpret_cast<const half8*>(img + frag_base[mt] + to + PLANE);
            }
            WT(WT_C_COMPUTE)
            // sched_barrier: hipcc otherwise sinks the next step's weight reads to their first use (one step later, with
            // the full LDS latency in front of the MFMAs) and splits the step's MFMAs around the s_barrier
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();       // the next step's weights (DMA'd two steps ago) have landed
            asm volatile("" ::: "memory");
            WT(WT_C_BARRIER)
            const char* wslot = wring + rd_slot * WSLICE + wfrag_off;
            rd_slot = (rd_slot + 1 == RING) ? 0 : rd_slot + 1;
            half8 wh_n[NT], wl_n[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                wh_n[nt] = *reinterpret_cast<const half8*>(wslot + nt * 2048);
                wl_n[nt] = *reinterpret_cast<const half8*>(wslot + nt * 2048 + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#if defined(WS_ABL) && WS_ABL == 2      // ablation (wrong results): one MFMA pass instead of three
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh_c[nt] + wl_c[nt], xh[mt] + xl[mt], acc[mt][nt], 0, 0, 0);
#else
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh_c[nt], xh[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh_c[nt], xl[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl_c[nt], xh[mt], acc[mt][nt], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { wh_c[nt] = wh_n[nt]; wl_c[nt] = wl_n[nt]; }
        }
#if !(defined(WS_ABL) && WS_ABL == 4)   // ablation 4 (wrong results): no per-tile epilogue
        if ((q + 1) % nchunks == 0) { WT(WT_C_COMPUTE) epilogue(first_tile + (q / nchunks) * a.wgs_per_img); WT(WT_C_EPI) }
#endif
    }
#ifdef MIDD_CONV_TIMING
    if (tid == 0) {
        wt_acc[WT_C_TOTAL] = wt_stamp() - wt_t0; wt_acc[WT_STEPS] = (unsigned long long)total_chunks * HSTEPS; wt_acc[WT_WGS] = 1;
        for (int k = 0; k <= WT_C_TOTAL; ++k) atomicAdd(&g_ws_timing[k], wt_acc[k]);
        atomicAdd(&g_ws_timing[WT_STEPS], wt_acc[WT_STEPS]); atomicAdd(&g_ws_timing[WT_WGS], wt_acc[WT_WGS]);
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // [E1] all consumer waves' statistics rows are in LDS
    asm volatile("" ::: "memory");
    if (a.stat_tot != nullptr) {
        constexpr int ROWF = 2 * NT * 16;
        for (int i = tid; i < ROWF; i += G::NCONS * 64) {
            const int which = i / (NT * 16), c = i - which * (NT * 16);
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < G::NCONS; ++m) t += stat_lds[m * ROWF + i];
            stat_atomic_add(stat_slot(a.stat_tot, b, a.Cout, ntile_wg * 16 + c, a.stat_rep, first_tile % a.stat_rep, which), t);
        }
    }
}

#ifdef MIDD_CONV_TIMING
extern "C" __attribute__((visibility("default"))) void mi_debug_ws_timing_dump(const char* tag) {
    unsigned long long h[WT_N];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ws_timing), sizeof h);
    const double steps = (double)h[WT_STEPS], wgs = (double)h[WT_WGS];
    if (wgs > 0)
        printf("%-40s wgs %6.0f steps/wg %5.1f | consumer cyc/step: barrier %6.0f compute %6.0f | epilogue/wg %7.0f total/wg %8.0f | "
               "producer cyc/step: barrier %6.0f waitvm %6.0f work %6.0f total/wg %8.0f\n", tag, wgs, steps / wgs,
               h[WT_C_BARRIER] / steps, h[WT_C_COMPUTE] / steps, h[WT_C_EPI] / wgs, h[WT_C_TOTAL] / wgs,
               h[WT_P_BARRIER] / steps, h[WT_P_WAITVM] / steps, h[WT_P_WORK] / steps, h[WT_P_TOTAL] / wgs);
    unsigned long long z[WT_N] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_timing), z, sizeof z);
    fflush(stdout);
}
#endif

bool conv3x3_ws_tile_ok(const ConvTile& t, int C0, int C1, int Cout) {
    static const bool on = !(getenv("MIDD_WS") && atoi(getenv("MIDD_WS")) == 0);
    return on && t.ks == 3 && t.stride == 1 && t.tw == 16 && t.mt == 2 && t.nt == 3 && t.wm == 4 && t.wn == 1 &&
           Cout % 48 == 0 && (C0 % 16) == 0 && (C1 % 16) == 0 && WsGeom::lds_bytes(C0 + C1) <= 80 * 1024;
}

hipError_t conv3x3_ws_launch(const ConvArgs& a0, hipStream_t s) {
    using G = WsGeom;
    ConvArgs a = a0;
    a.tiles_x = (a.OW + G::TW - 1) / G::TW;
    a.tiles_y = (a.OH + G::TH - 1) / G::TH;
    const int ny = a.Cout / (G::NT * 16);
    // two 8-wave workgroups per CU
    a.wgs_per_img = conv16_wgs_per_img(a.tiles_x * a.tiles_y, a.B, ny, a.persist_wgs ? (a.persist_wgs * 2) / 3 : 512);
    if ((double)a.B * a.H * a.W * (a.C0 > a.C1 ? a.C0 : a.C1) * 4.0 >= 4294967296.0) return hipErrorInvalidValue;      // 32-bit DMA offsets
    const int lds_bytes = G::lds_bytes(a.C0 + a.C1);
    static int raised = 0;
    if (lds_bytes > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws_f16x3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        raised = lds_bytes;
    }
    hipLaunchKernelGGL(conv3x3_ws_f16x3_kernel, dim3(a.B * a.wgs_per_img, ny), dim3(G::NTHREADS), lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace midd
