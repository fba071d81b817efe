/*
 * midd.h — C ABI of libmidd.so, the MI355X (gfx950) native implementation of the
 * reverse-diffusion sampler path.
 *
 * The reference has no FFI/operator interface for this path: its boundary is two Python
 * classes, UNetDiffusion / DiffusionDenoiser (/root/reference/Backend/DDIM/DDIMModel.py:169-289),
 * imported by name at /root/reference/Backend/run.py:13.  Every entry point below cites the
 * reference interface it replaces; the ctypes binding a maintainer adds on the reference
 * side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handle, pointers and sizes only; no torch / C++ types.
 *   - every function returns 0 on success or a negative MI_E* code; the message is
 *     available from mi_last_error() (thread-local).  Nothing throws across the ABI.
 *   - image / workspace pointers are DEVICE pointers on the current HIP device; weight,
 *     schedule and timestep-list pointers are HOST pointers.
 *   - forward/denoise allocate nothing: the caller supplies the workspace (size from
 *     mi_workspace_bytes) and a stream (void* = hipStream_t, NULL = default stream).
 *     They are asynchronous with respect to the host; the caller synchronises.
 *   - a plan is immutable after mi_unet_finalize and may be shared by threads; concurrent
 *     calls must use distinct workspaces (run.py:85-91 calls the sampler from a worker thread).
 *   - images are fp32, contiguous [B, in_channels, H, W] exactly as the reference passes
 *     them (DDIMModel.py:219, :269); H and W must be multiples of 2^(levels-1) (8).
 */
#ifndef MIDD_H
#define MIDD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK            0
#define MI_EINVAL       -1   /* bad argument / unsupported shape or topology */
#define MI_ESTATE       -2   /* call order (e.g. forward before finalize, missing weight) */
#define MI_EHIP         -3   /* a HIP runtime call or kernel launch failed */
#define MI_ENOMEM       -4   /* workspace too small */
#define MI_ERANGE       -5   /* mi_status: the last call met values outside what its arithmetic represents */

/* bits of the status word (mi_status) */
#define MI_STATUS_NONFINITE   1   /* NaN / Inf activations reached a GroupNorm statistic or a raw conv operand */
#define MI_STATUS_FP16_RANGE  2   /* split-fp16 mode: an attention operand (q, k, v) beyond +-4094 */

#define MI_MAX_LEVELS    8

#define MI_VARIANT_DDIM  0   /* Backend/DDIM/DDIMModel.py  */
#define MI_VARIANT_CDDPM 1   /* Backend/cddpm/cddpmModels.py */

/* arithmetic of the MFMA contractions (convolutions, attention) */
#define MI_COMPUTE_F32   0   /* fp32-input MFMA: bit-for-bit an fp32 fma chain */
#define MI_COMPUTE_F16X3 1   /* fp32 operands split into two fp16 halves, three fp16 MFMAs, fp32 accumulate
                                (~2^-21 relative per product; same parity gate) — about 5x the MFMA rate */
/* OR into compute_mode: results of a sample do not depend on the batch it is computed in, bit for bit
 * (denoise(x[:k]) == denoise(x)[:k]; SURVEY.md section 8e "sharded == single-GPU").  Tiles, persistent workgroups per
 * sample, chunk width and the attention key split are then chosen as the default plan of a 4-sample sub-batch chooses
 * them -- the grouping of the fp32 partial sums behind the GroupNorm statistics no longer varies with the batch.
 * Free at batch 8 (the plan is the default one), -6 % at batch 32, +12 % on the latency of a single image. */
#define MI_COMPUTE_BATCH_INVARIANT 0x100

/* sampler flags for mi_denoise */
#define MI_CLAMP_EPS     1   /* clamp(eps,-5,5) before the update: DDIMModel.py:278 (absent in cddpm) */
#define MI_NO_SPLIT      2   /* run the batch as ONE program on the caller's stream (default: two half-batches on two streams);
                                same results to rounding (per-program batch changes the tiles), used by bench.py's roofline leg */

typedef struct mi_plan mi_plan;

/* Constructor arguments of UNetDiffusion.__init__ (DDIMModel.py:169-170). */
typedef struct mi_unet_cfg {
    int32_t in_channels;                         /* 1 */
    int32_t model_channels;                      /* 48; must be a multiple of 16 */
    int32_t num_levels;                          /* len(channel_mult) = 4 */
    int32_t channel_mult[MI_MAX_LEVELS];         /* (1,2,3,4) */
    int32_t num_res_blocks;                      /* 2 */
    int32_t num_attention_levels;                /* len(attention_resolutions) = 1 */
    int32_t attention_levels[MI_MAX_LEVELS];     /* (3,) — level indices */
    int32_t time_emb_dim;                        /* 192 */
    int32_t variant;                             /* MI_VARIANT_* */
    int32_t compute_mode;                        /* MI_COMPUTE_* (not a reference argument) */
} mi_unet_cfg;

/* Replaces UNetDiffusion.__init__ (DDIMModel.py:169-217): derives the module lists
 * (downs / mid / ups) and the expected state-dict entries. */
int mi_unet_plan_create(const mi_unet_cfg* cfg, mi_plan** out);

/* Replaces model.load_state_dict(ckpt['model_state_dict']) (run.py:37-39): called once per
 * state-dict entry with the reference's key name (e.g. "downs.3.block1.2.weight",
 * "ups.6.weight" [Cin,Cout,4,4]).  `data` is a HOST pointer to contiguous fp32. */
int mi_unet_load_weights(mi_plan* plan, const char* key, const float* data,
                         const int64_t* shape, int ndim);

/* Number of state-dict entries the plan expects / the i-th expected key (for the host
 * shim's strict-load check). */
int mi_unet_num_weights(const mi_plan* plan);
const char* mi_unet_weight_name(const mi_plan* plan, int index);

/* Repacks all weights into kernel layouts on the device (MFMA fragment order, folded
 * ConvTranspose+resample 3x3 — DDIMModel.py:211,241-242), and precomputes the timestep
 * table: time_mlp (DDIMModel.py:99-106,173-178) followed by every ResidualBlock's
 * Linear(SiLU(.)) (DDIMModel.py:111-114,130) for t in [0, time_rows).
 * May be called again after further mi_unet_load_weights calls (re-uploads). */
int mi_unet_finalize(mi_plan* plan, int time_rows);

/* Bytes of device workspace one forward/denoise call needs at this shape. */
size_t mi_workspace_bytes(mi_plan* plan, int B, int H, int W);

/* Replaces UNetDiffusion.forward(x, condition, t) (DDIMModel.py:219-248).
 * x, condition: device fp32 [B,in_channels,H,W]; t: HOST int32[B] (the reference takes an
 * int64 tensor; the sampler always passes B equal values, DDIMModel.py:275);
 * eps: device fp32 [B,in_channels,H,W]. */
int mi_unet_forward(mi_plan* plan, const float* x, const float* condition, const int32_t* t,
                    float* eps, int B, int H, int W,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Replaces DiffusionDenoiser.denoise(noisy_img, inference_steps) (DDIMModel.py:268-289; cddpm:
 * cddpmModels.py:281-308).  The iteration list and the schedule tables are passed in by the
 * host shim, which computes them exactly as the reference does (DDIMModel.py:255-257,272-274).
 *   noisy      device fp32 [B,C,H,W]; never written (DDIMModel.py:271 clones it)
 *   x_out      device fp32 [B,C,H,W]; receives x after the last iteration
 *   t_list     HOST int32[n_iters] timesteps in execution order (each < noise_steps <= time_rows)
 *   beta/alpha/alpha_hat  HOST fp32[noise_steps]
 *   step_noise device fp32 [n_iters,B,C,H,W] or NULL: the already 0.5-scaled Gaussian noise of
 *              the cddpm variant (cddpmModels.py:297-302); entry i is ignored when t_list[i]==0
 *   flags      MI_CLAMP_EPS for the DDIM variant; MI_NO_SPLIT */
int mi_denoise(mi_plan* plan, const float* noisy, float* x_out, int B, int H, int W,
               const int32_t* t_list, int n_iters,
               const float* beta, const float* alpha, const float* alpha_hat, int noise_steps,
               const float* step_noise, int flags,
               void* workspace, size_t workspace_bytes, void* stream);

/* Status of the last mi_unet_forward / mi_denoise call that used `workspace` (its first word; the calls clear it when they
 * start).  SYNCHRONISES `stream` (one 4-byte device-to-host copy).  Returns MI_OK with *flags == 0, or MI_ERANGE with the
 * MI_STATUS_* bits in *flags: the kernels never turn a NaN / Inf activation or an operand beyond the split-fp16 range into
 * finite garbage silently -- the output is NaN where the reference's is, and this call says why.  (The reference itself
 * reports nothing: torch propagates NaN, DDIMModel.py:219-289.) */
int mi_status(const void* workspace, void* stream, int* flags);

/* Debug/test hook: after a forward call, copies the output of the named top-level module
 * (e.g. "downs.3", "mid_attn", "ups.6") from the workspace to `dst` (device fp32, NCHW
 * [B,C,h,w]); returns its C,h,w.  `dst` may be NULL to query the shape only. */
int mi_debug_fetch(mi_plan* plan, const char* module_name, int B, int H, int W,
                   const void* workspace, float* dst, int* C, int* h, int* w, void* stream);

/* Debug/test hook, host only (no GPU call): the key split the split-fp16 attention kernel uses for N keys (pixels of
 * the attention level) when the execution program holds B samples: number of splits and 32-key tiles per split.
 * Every split owns at least one tile for every N >= 1 (tests sweep it). */
int mi_debug_attention_split(int N, int B, int* ksplit, int* tiles_per_split);

/* Debug/test hook, host only (no GPU call, no finalize needed): the execution program the planner builds for (B, H, W) as
 * text, one line per kernel launch: kernel instantiation, tile, grid, persistent workgroups per sample, weight-ring depth and
 * DMA pieces per wave, folded res_conv steps, attention key split, LDS bytes.  side_by_side != 0: the sub-batch program the
 * two-stream mi_denoise runs (the reference has no counterpart: its graph is fixed, DDIMModel.py:219-248).  Writes at most
 * cap - 1 characters + NUL to buf (may be NULL) and returns the full length, or a negative MI_E* code. */
int mi_debug_plan_dump(mi_plan* plan, int B, int H, int W, int side_by_side, char* buf, size_t cap);

/* Debug/test hook, host only: the staging geometry of one instantiated tile of the split-fp16 3x3 / 1x1 kernel
 * (conv_mfma_f16x3.hip: Conv16Geom) -- weight-ring slots, weight DMA pieces per wave and step, activation DMA pieces per wave
 * and chunk, LDS bytes at 384 input channels.  tests/test_dma_protocol_cpu.py replays the kernel's issue / wait protocol with
 * these numbers for every instantiated tile and checks every hand-counted `s_waitcnt vmcnt(N)`.  MI_EINVAL: not instantiated. */
int mi_debug_conv16_geometry(int ks, int stride, int tw, int mt, int nt, int wm, int wn, int cb,
                             int* ring, int* ppw, int* apw, int* lds_bytes);

/* First 16 hex digits of the sha256 over the kernel sources (csrc/ *.h, *.hip) this library was BUILT from, embedded at
 * build time: what bench.py / tools/pmc_traffic.py compare profiles against (not the working tree). */
const char* mi_source_hash(void);

/* Per-kernel timing with HIP events on the caller's stream (bench.py's roofline leg; the
 * reference's only timer for this path is time.time() around denoise(), DDIMModel.py:495-498).
 * Between mi_profile_begin and mi_profile_end every kernel launched by mi_unet_forward /
 * mi_denoise on this plan is bracketed by an event pair.  mi_profile_end synchronises the
 * recorded events and returns one entry per distinct kernel symbol: launches, summed
 * duration, and the ALGORITHMIC work of those launches (flops = 2*MACs of the contraction the
 * launch performs; bytes = the tensors it must read + write once, fp32).  Calls made while
 * profiling must not be issued concurrently from several threads. */
typedef struct mi_profile_entry {
    char   name[128];      /* kernel symbol as rocprofv3 prints it (without "void " / argument list) */
    int64_t launches;
    double total_ms;
    double flops;          /* summed over the launches */
    double bytes;
} mi_profile_entry;
int mi_profile_begin(mi_plan* plan);
int mi_profile_end(mi_plan* plan, mi_profile_entry* out, int max_entries, int* n_entries);

void mi_plan_destroy(mi_plan* plan);

/* ---- pre/post-processing either side of the sampler, on the device (no plan needed) -----------------------
 * All pointers are device pointers; `stream` is a hipStream_t (NULL = default stream); calls are asynchronous.
 *
 * mi_resize_bicubic_u8: `n` 8-bit single-channel images [n][sh][sw] -> [n][dh][dw], bit-identical to Pillow's
 *   `Image.resize((dw, dh), Image.BICUBIC)` on an 'L' image, i.e. to `transforms.Resize((dh, dw), BICUBIC)` on a
 *   PIL input (Backend/run.py:146,198; Backend/cddpm/cddpmModels.py:488,502).  `workspace` needs
 *   mi_resize_workspace_bytes(n, sw, sh, dw, dh) bytes, 256-byte aligned.
 * mi_u8_to_unit_f32: `transforms.ToTensor()` on uint8 data: x / 255 in fp32 (run.py:199).
 * mi_unit_f32_to_u8: `(clamp(x, 0, 1) * 255).astype('uint8')` (run.py:107,145): fp32 multiply, truncation.
 * mi_image_metrics: `compute_metrics` (Backend/DDIM/DDIMModel.py:290-300): per image PSNR and SSIM of
 *   clip(pred, 0, 1) against clip(target, 0, 1) with data_range = 1 (skimage defaults: 7x7 uniform window,
 *   K1 = 0.01, K2 = 0.03, sample covariance, mean over the interior); fp32 images [n][h][w], h, w >= 7;
 *   out: device double [n][2] = (psnr, ssim); workspace: mi_metrics_workspace_bytes(n, h) bytes. */
size_t mi_resize_workspace_bytes(int n, int sw, int sh, int dw, int dh);
int mi_resize_bicubic_u8(const void* src_u8, int n, int sw, int sh, void* dst_u8, int dw, int dh,
                         void* workspace, size_t workspace_bytes, void* stream);
int mi_u8_to_unit_f32(const void* src_u8, void* dst_f32, size_t count, void* stream);
int mi_unit_f32_to_u8(const void* src_f32, void* dst_u8, size_t count, void* stream);
size_t mi_metrics_workspace_bytes(int n, int h);
int mi_image_metrics(const void* target_f32, const void* pred_f32, int n, int h, int w, void* out_f64,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Thread-local, never NULL. */
const char* mi_last_error(void);

/* Library version string, e.g. "midd 0.1 gfx950". */
const char* mi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIDD_H */
