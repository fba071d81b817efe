// Internal declarations shared by the HIP kernels and the C-ABI host side of libmidd.so.
// Activations inside the library are fp32, channels innermost; only the boundary tensors ([B,in_channels,H,W],
// in_channels == 1 in every reference call site) are NCHW.  Two layouts:
//   fp32-MFMA plans   NHWC            [B][H][W][C]
//   split-fp16 plans  CHANNEL-BLOCKED [B][C/16][H][W][16]   (round 3)
// The f16x3 kernels stage a K chunk of 16 channels at a time.  In NHWC that is 64 bytes of every pixel's C*4-byte row while L2
// fetches whole 128-byte lines (a chunk pass alone moves twice its bytes: rocprofv3 FETCH_SIZE calibrated on this very
// pattern, tools/mb/fetch_calib.hip), the MFMA tiles of the epilogue leave as 64-byte pieces C*4 bytes apart, and so do the
// residual rows.  Blocked, the chunk of a halo row is one contiguous run, a 16-pixel x 16-cout MFMA tile leaves as ONE contiguous
// KiB, residual rows and the pointwise kernels move whole runs.  Per launch at batch 4, 256x256 (tools/traffic_per_op.sh, L2-side
// bytes): conv2's fetch 128 -> 115 MB (its residual rows), out_conv 100 -> 64 MB for a 50 MB tensor; the 3x3 kernel's own
// chunk reads were already close to halo-only (66.5 -> 64.3 MB: the L2 kept the other half of the lines for the next pass).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stats_common.h"

namespace midd {

// Launches that ask for more than 64 KB of dynamic LDS must raise the kernel's limit first, once per (kernel instantiation,
// device): `raised` is the instantiation's own static table, zero-initialised, indexed by device (the caches were per
// process).  Returns hipSuccess when `bytes` is already allowed.
constexpr int MIDD_MAX_DEVICES = 64;
inline hipError_t ensure_dynamic_lds(const void* kernel, int bytes, int (&raised)[MIDD_MAX_DEVICES]) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MIDD_MAX_DEVICES) return hipErrorInvalidDevice;
    if (bytes > raised[dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        raised[dev] = bytes;
    }
    return hipSuccess;
}

// ---------------------------------------------------------------- implicit-GEMM convolution
enum Prologue { PRO_RAW = 0, PRO_GN = 1, PRO_GN_SILU = 2 };

// element index of (sample b, pixel pix, channel ch) in an activation tensor of C channels and HW pixels per sample
__host__ __device__ inline size_t act_index(int blocked, int b, int C, int HW, int pix, int ch) {
    return blocked ? (((size_t)b * (C >> 4) + (ch >> 4)) * HW + pix) * 16 + (ch & 15) : ((size_t)b * HW + pix) * C + ch;
}

struct ConvArgs {          // (activation pointers: NHWC for the fp32-MFMA kernels, channel-blocked for the f16x3 kernels)
    const float* src0;      // C0 channels
    const float* src1;      // C1 channels (virtual torch.cat on dim=1); may be null when C1 == 0
    int C0, C1;
    int B, H, W;            // input spatial size
    int OH, OW;             // output spatial size
    const float* wpack;     // [Cin/16][taps][Cout/16][64 lanes][4]  (MFMA A-fragment order)
    const float* bias;      // [Cout]
    int Cout;
    // GroupNorm of the INPUT (prologue != RAW): per-channel fixed-point (sum, sum of squares) totals of the two
    // concatenated sources, [B][C0/bs0][rep][2][3] / [B][C1/bs1][rep][2][3] limbs, accumulated by the producers (stats_common.h); every
    // workgroup derives scale = rstd*gamma, shift = beta - mean*rstd*gamma of its sample in its prologue
    // prologue == RAW (f16x3 kernels): the same totals, when present, give the operand's power-of-two prescale
    // (raw_prescale_exp, stats_common.h); gn_tot0 == null: the fixed prescale raw_scale_fixed
    const stat_word* gn_tot0; const stat_word* gn_tot1;
    int gn_bs0, gn_bs1;     // channels per totals block of each source (whole blocks per GroupNorm group)
    float raw_scale_fixed;  // f16x3, prologue == RAW without totals: 2^s applied to the operand (out_scale undoes it)
    int* status;            // device word of the workspace: bit MI_STATUS_* set when a kernel meets a non-finite statistic / an operand beyond fp16 (may be null)
    const float* gn_gamma; const float* gn_beta; float gn_eps; double gn_inv_n;      // affine [Cin], eps, 1 / (pixels * channels per group)
    int prologue;
    const float* temb;      // time table [rows][temb_stride], already offset to this block's column
    int temb_stride;
    const int* trow;        // [B] table row per sample
    const float* resid;     // [B][OH][OW][Cout] (layout as above) added in the epilogue, or null
    float* out;             // [B][OH][OW][Cout] (layout as above)
    float out_scale;        // f16x3 only: 2^-(k+s) undoing the operand prescales (1 for fp32)
    // optional fused GroupNorm statistics of the OUTPUT: every workgroup adds the per-channel sum / sum of squares of
    // the pixels it produced to the totals [B][Cout/bs][rep][2][3] (exact integer atomics, stats_common.h); zeroed per forward
    stat_word* stat_tot;
    int stat_rep;           // copies of every totals block of this program (stats_common.h), also for gn_tot0/1
    int stat_bs;            // channels per totals block of the OUTPUT
    // res_conv folded into conv2 (f16x3 3x3 kernel; res_steps == 0 everywhere else): after a tile's 3x3 steps the kernel runs
    // res_steps more K steps of 32 channels each -- the 1x1 res_conv over the BLOCK INPUT (res_src0 [, res_src1]: the virtual
    // torch.cat), operands straight from global memory as in conv1x1_f16x3.hip -- into the same accumulators (rescaled by the
    // power of two between the two products' prescales).  wpack holds the 3x3 steps followed by the res steps; bias is the sum
    // of both biases; res_tot* are the block input's totals (per-sample operand prescale, stats_common.h)
    const float* res_src0; const float* res_src1;
    int res_C0, res_C1, res_steps;
    float res_scale;        // 2^-k of the res_conv weights
    const stat_word* res_tot0; const stat_word* res_tot1;
    int res_bs0, res_bs1;
    // attention hand-off of the f16x3 1x1 kernel (conv1x1_f16x3.hip; att_mode == ATT_NONE everywhere else)
    //   ATT_QKV_OUT  (the qkv projection): q goes to `out` as fp32 [B][N][C]; k and v are written as the split-fp16 images
    //                the attention kernel stages, att_k / att_v [B][heads][hi|lo][Npad][D] (x 2^4), keys >= N zeroed
    //   ATT_PART_IN  (the output projection): the operand is the attention kernel's key-split partials,
    //                src0 = part_o [ksplit][B][N][C] (unnormalised), att_ml [ksplit][B][heads][N][2] = (m, l); the splits
    //                are extra K steps, each element scaled by 2^(m_s - M) / (L 2^14) of its (pixel, head, split) on load
    int att_mode;
    _Float16* att_k; _Float16* att_v;
    const float* att_ml;
    int att_heads, att_D, att_npad, att_ksplit;
    int tiles_x, tiles_y;
    int wgs_per_img;        // f16x3: persistent workgroups per sample (each walks tiles j, j+wgs_per_img, ...)
    int persist_wgs;        // f16x3: persistent-workgroup target of the launch (0 = default)
#ifdef MIDD_CONV_TIMING
    int dbg_slot;           // diagnostic build: row of g_conv_timing
#endif
};

struct ConvTile {           // which template instance to launch
    int ks, stride, tw, mt, nt, wm, wn;
    int cb = 0;             // f16x3 3x3: 16-channel blocks per K chunk (0: conv16_cb(ks)); selects the weight packing too
};

enum ComputeMode { MODE_F32 = 0, MODE_F16X3 = 1 };
enum AttMode { ATT_NONE = 0, ATT_QKV_OUT = 1, ATT_PART_IN = 2 };
enum StatusBits { STATUS_NONFINITE = 1, STATUS_FP16_RANGE = 2,          // == MI_STATUS_* (include/midd.h)
                  STATUS_DMA_EARLY = 4 };      // diagnostic builds only (-DMIDD_DMA_CHECK, tools/dma_check.sh): data used before its counted wait had covered it

// Picks a tile for (Cout, output pixels, kernel size, stride); returns false if unsupported.
bool conv_pick_tile(int Cout, int B, int OH, int OW, int ks, int stride, ConvTile* t);
hipError_t conv_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s);

// split-fp16 variant (conv_mfma_f16x3.hip): same arguments, weights packed by pack_conv_f16x3
bool conv16_pick_tile(int Cin, int Cout, int B, int OH, int OW, int ks, int stride, ConvTile* t, bool allow_wide = false);
int conv16_wgs_per_img(int tiles, int B, int ny, int target = 0);   // target 0: 768
bool conv1x1_pick_tile(int Cin, int Cout, int B, int OH, int OW, ConvTile* t);   // ConvTile::tw == 0 marks it
hipError_t conv1x1_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s);      // persistent workgroups per sample (f16x3 kernels)
hipError_t conv16_launch(const ConvArgs& a, const ConvTile& t, hipStream_t s);
// launch geometry of an f16x3 convolution, host only (mi_debug_plan_dump; tests that pin which instantiations a shape reaches)
struct ConvLaunchInfo { int grid_x, grid_y, wgs_per_img, tiles_x, tiles_y, ring, ppw, apw, lds_bytes; };
bool conv16_launch_info(int Cin, int Cout, int B, int OH, int OW, const ConvTile& t, int persist_wgs, ConvLaunchInfo* out);
bool conv1x1_launch_info(int Cin, int Cout, int B, int OH, int OW, const ConvTile& t, int persist_wgs, int att_mode, ConvLaunchInfo* out);
// 16-channel blocks staged per K chunk of the f16x3 kernel (shared with the host packer).
//   3x3: 1 -> 16 channels per chunk, two taps per MFMA step (10 % padded MFMA slots, but half the
//             activation LDS of a 32-channel chunk, i.e. three resident workgroups per CU)
//   1x1: 2 -> 32 channels per chunk and step (no halo, so the image is small; halves the number of
//             chunk hand-overs, which dominate a 1x1)
//   3x3, "wide" (cb = 2, picked per launch where the grid leaves at most ~2 workgroups per CU anyway: the 32x32 / 64x64 maps
//             at small batches): 32 channels per chunk, one tap per step -- 9 full steps instead of 2 x 5 (no zero-weight
//             half step) and half the chunk hand-overs; weights packed in that K order as a second copy (pack_conv_f16x3)
__host__ __device__ constexpr int conv16_cb(int ks) { return ks == 1 ? 2 : 1; }
// number of 32-wide K steps the f16x3 kernel walks for (Cin, taps) with cb blocks per chunk (0: the default of the kernel size)
__host__ __device__ inline int conv16_num_steps(int Cin, int taps, int cb = 0) {
    const int nblk = Cin / 16;
    if (cb == 0) cb = conv16_cb(taps == 1 ? 1 : 3);
    if (cb == 1) return nblk * ((taps + 1) / 2);
    const int full = nblk / 2, half = nblk & 1;
    return full * taps + half * ((taps + 1) / 2);
}

// ---------------------------------------------------------------- GroupNorm statistics (stats_common.h)
constexpr int GN_GROUPS_ = 8;                  // nn.GroupNorm(8, C) everywhere in the reference (DDIMModel.py:116,121,139,214)
// per-channel totals of an activation tensor (either layout: `blocked`) no MFMA conv produced (in_conv output, bilinear 2x outputs): `rows` blocks per
// sample each add their partial sums to tot [B][C/bs][rep][2][3]
hipError_t chan_total_launch(const float* src, stat_word* tot, int rep, int bs, int B, int HW, int C, int rows, int blocked, hipStream_t s);
int chan_partial_rows(int HW, int C);

// ---------------------------------------------------------------- pre/post-processing (prepost.hip)
size_t resize_workspace_bytes(int n, int sw, int sh, int dw, int dh);
hipError_t resize_bicubic_u8_launch(const unsigned char* src, int n, int sw, int sh, unsigned char* dst, int dw, int dh, void* ws, hipStream_t s);
hipError_t u8_to_unit_launch(const unsigned char* src, float* dst, size_t count, hipStream_t s);
hipError_t unit_to_u8_launch(const float* src, unsigned char* dst, size_t count, hipStream_t s);
size_t metrics_workspace_bytes(int n, int h);
hipError_t metrics_launch(const float* target, const float* pred, int n, int h, int w, double* out, void* ws, hipStream_t s);

// ---------------------------------------------------------------- attention
// qkv: NHWC [B][N][3C], channel = s*C + head*D + d (s in q,k,v);  out: [B][N][C]
hipError_t attention_launch(const float* qkv, float* out, int B, int N, int C, int heads, hipStream_t s);
// split-fp16 variant (attention_f16x3.hip).  Three launches per attention block: the qkv projection's epilogue writes q (fp32
// [B][N][C]) and the split-fp16 images of k and v, the attention kernel leaves key-split partials, the output projection
// combines them while it loads its operand.  Scratch layout (attention16_layout): K image, V image, partial O, partial (m, l).
struct Att16Layout { size_t k_off, v_off, po_off, ml_off, bytes; int npad; };
Att16Layout attention16_layout(int B, int N, int C);
// split_B: the batch the key split is chosen for (B, or 1 for batch-invariant plans)
hipError_t attention16_launch(const float* q, const _Float16* Kp, const _Float16* Vp, float* part_o, float* part_ml,
                              int B, int ksplit, int tiles_per_split, int N, int C, int heads, hipStream_t s);
// key split of the f16x3 attention for N keys: splits and 32-key tiles per split (every split owns >= 1 tile)
void attention16_split(int N, int heads, int split_B, int* ksplit, int* tiles_per_split);
bool attention_supported(int head_dim);

// ---------------------------------------------------------------- small direct kernels
// in_conv: Conv3x3 on cat[x, cond] (NCHW [B,ic,H,W] each) -> [B][H][W][Cout] activations (NHWC, or channel-blocked when `blocked`)
// tot != nullptr: also leaves the GroupNorm totals of the output [B][Cout/bs][rep][2][3] (stats_common.h)
hipError_t in_conv_launch(const float* x, const float* cond, const float* w /*[9][2ic][Cout]*/, const float* bias,
                          float* out, stat_word* tot, int rep, int bs, int B, int ic, int H, int W, int Cout, int blocked, hipStream_t s);

struct OutConvArgs {
    const float* src;       // [B][H][W][C], or channel-blocked [B][C/16][H][W][16] when `blocked`
    int blocked;
    const stat_word* gn_tot; int stat_rep, gn_bs; const float* gn_gamma; const float* gn_beta; float gn_eps;   // GroupNorm of src: totals [B][C/bs][rep][2][3], affine [C]
    const float* w;         // [ic][9][C]
    const float* bias;      // [ic]
    int B, H, W, C, ic;
    float* eps_out;         // NCHW [B,ic,H,W] raw network output, or null
    // fused sampler update (DDIMModel.py:278-284) when x != null:  x <- clamp(c1*(x - c2*clamp(eps)) + c3*noise, 0, 1)
    float* x;               // NCHW [B,ic,H,W], updated in place
    const float* noise;     // NCHW or null
    float c1, c2, c3;
    int clamp_eps;
};
hipError_t out_conv_launch(const OutConvArgs& a, hipStream_t s);

// bilinear resize of an activation tensor, either layout (align_corners=False), any size ratio
hipError_t resize_bilinear_launch(const float* src, float* dst, stat_word* tot, int rep, int bs, int B, int H, int W, int C, int OH, int OW, int blocked, hipStream_t s);
// ConvTranspose2d(C,C,4,stride=2,padding=1) direct (only used by topologies where it cannot be folded)
hipError_t conv_transpose_launch(const float* src, const float* w /*[4][4][Cin][Cout]*/, const float* bias, float* dst,
                                 int B, int H, int W, int Cin, int Cout, int blocked, hipStream_t s);
// activation tensor (either layout) -> NCHW copy (debug fetch)
hipError_t nhwc_to_nchw_launch(const float* src, float* dst, int B, int H, int W, int C, int blocked, hipStream_t s);
hipError_t fill_i32_launch(int* dst, const int* host_vals, int n, hipStream_t s);

}  // namespace midd
