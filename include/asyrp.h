/*
 * asyrp.h — C ABI of the MI355X-native Asyrp DDIM sampling engine (libasyrp_hip.so).
 *
 * The reference (kwonminki/Asyrp_official) has no FFI layer: its seam is two Python calls,
 *   B1  model(xt, t, index=..., t_edit=..., hs_coeff=..., delta_h=..., ...)  -> (et, et_modified, delta_h, middle_h)
 *       utils/diffusion_utils.py:46  ->  models/ddpm/diffusion.py:473 (DDPM.forward)
 *                                        models/improved_ddpm/unet.py:676 (UNetModel.forward)
 *   B2  denoising_step(xt, t, t_next, *, models, logvars, b, ...)            -> (xt_next, x0_t, delta_h, middle_h)
 *       utils/diffusion_utils.py:24-104, called from diffusion_latent.py:507 (Asyrp generation)
 *       and :1038 (DDIM inversion).
 * This header is what a binding for those two seams (and for the two loops around them,
 * diffusion_latent.py:1034-1045 and :503-520) binds to.  The Python host side that mirrors the
 * reference classes on top of it lives in asyrp_official_amd/ (ctypes).
 *
 * Conventions
 *   - plain C, no torch types.  Every `const float*` / `float*` named x, t, et, ... is a DEVICE
 *     pointer to fp32 data in the reference's own layout (NCHW, contiguous); pointers documented
 *     as "host" are host memory.
 *   - all work is enqueued on the caller's stream (`stream` is a hipStream_t passed as void*);
 *     no entry point synchronises the device unless documented.
 *   - return 0 on success, a negative ASYRP_E* code otherwise; asyrp_last_error() gives the text.
 *     Nothing throws across the ABI.
 *   - an engine is bound to one device and is NOT thread-safe; caller serialises.
 *   - the caller owns every I/O buffer; the engine owns packed weights + workspace until destroy.
 */
#ifndef ASYRP_H
#define ASYRP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASYRP_OK 0
#define ASYRP_EINVAL (-1)   /* bad argument / unsupported configuration */
#define ASYRP_EHIP (-2)     /* HIP runtime error */
#define ASYRP_ESTATE (-3)   /* call order violated (e.g. forward before finalize) */
#define ASYRP_EKEY (-4)     /* unknown / missing / mis-shaped parameter key */

#define ASYRP_MAX_LEVELS 8

/* UNet families of the reference (diffusion_latent.py:76-126 picks one by dataset). */
enum asyrp_family {
  ASYRP_FAMILY_DDPM = 0, /* models/ddpm/diffusion.py:327 DDPM(config)      — CelebA-HQ, LSUN */
  ASYRP_FAMILY_IDDPM = 1 /* models/improved_ddpm/unet.py:437 UNetModel (= models/guided_diffusion/unet.py:437) — AFHQ, FFHQ,
                            ImageNet, MetFaces, CelebA-HQ-P2; resblock_updown + scale-shift norm as in every arch dict */
};

/* Arithmetic of the implicit-GEMM convolutions (and of the fused attention kernel).
 * Parity modes - fp32-equivalent (same error class vs the reference's fp32 CPU path, ~1e-6 per UNet forward); they differ in
 * which matrix-core instruction carries the products:
 *   F16X3: operands split into two f16 terms, three f16 MFMAs per K-slice, fp32 accumulate (default)
 *   F32:   v_mfma_f32_32x32x2_f32, an exact fp32 fma chain (1/16 the f16 MFMA rate)
 * Fast mode - NOT fp32-equivalent, reported separately with its own measured error (bench.py --conv-math f16):
 *   F16:   ONE f16 MFMA per K-slice: activations and weights rounded to f16 (weights with the same power-of-two pre-scale),
 *          fp32 accumulate; GroupNorm statistics, softmax, timestep embedding and the DDIM update stay fp32/double as in the
 *          parity modes.  The reference's own reduced-precision hook is models/improved_ddpm/unet.py:660-674 (convert_to_fp16,
 *          unused by its scripts).  Inference only (the training step refuses it). */
enum asyrp_conv_math { ASYRP_MATH_F16X3 = 0, ASYRP_MATH_F32 = 1, ASYRP_MATH_F16 = 2 };

/* Hyper-parameters.  DDPM reads them from configs/<dataset>.yml `model:` (models/ddpm/diffusion.py:331-337);
 * iDDPM/ADM from the arch dicts (models/improved_ddpm/script_util.py:5-42). */
typedef struct asyrp_config {
  int32_t family;                 /* enum asyrp_family */
  int32_t resolution;             /* data.image_size / image_size */
  int32_t in_channels;            /* 3 */
  int32_t out_channels;           /* out_ch; 6 when learn_sigma */
  int32_t ch;                     /* model.ch / num_channels */
  int32_t n_levels;               /* len(ch_mult) */
  int32_t ch_mult[ASYRP_MAX_LEVELS];
  int32_t num_res_blocks;
  int32_t n_attn;                 /* number of entries in attn_resolutions */
  int32_t attn_resolutions[ASYRP_MAX_LEVELS]; /* spatial sizes at which attention runs (e.g. 16) */
  int32_t num_head_channels;      /* iDDPM: 64; DDPM: 0 = single head over all channels */
  int32_t n_delta;                /* number of DeltaBlocks layer_0..layer_{n-1} (setattr_layers) */
  int32_t conv_math;              /* enum asyrp_conv_math: how the conv / 1x1 GEMMs are evaluated */
  int32_t num_classes;            /* iDDPM class_cond: 1000 adds the (unused) label_emb.weight key; else 0 */
  int32_t nominal_batch;          /* batch class of the engine: the batch at which tile shapes and split-K factors are priced (0 = 32,
                                   * BASELINE config 2's per-GPU batch).  A property of the engine, never of a call: every call on
                                   * this engine, whatever its B, runs the same kernels on an image, so an image alone equals its row
                                   * of a batch bit for bit.  1 or 2 = the small class (single-image serving / the reference's
                                   * bs_train = 1): more, smaller workgroups and K split up to 8 ways.  Results of different classes
                                   * agree to fp32 rounding (as any two tile shapes do), not bitwise.  Accepted values: 0, 1, 2, 32
                                   * (the classes that are tested against the fixtures); anything else is ASYRP_EINVAL.
                                   * (ABI v8: was reserved[0] in v7.) */
  int32_t reserved[5];            /* must be zero (asyrp_create rejects anything else, so that a later field cannot be set by accident) */
} asyrp_config;

typedef struct asyrp_engine asyrp_engine;

/* The library is built with -fvisibility=hidden: the entry points declared in this header are its whole dynamic symbol table
 * (besides the HIP kernel stubs the runtime needs).  (The opaque handle type above stays outside the pragma.) */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif
/* Version of this ABI (bumped on any signature or semantic change); asyrp_abi_version() returns the library's.
 *   v8 (round 5): asyrp_config.nominal_batch (was reserved[0]) is validated (0, 1, 2, 32), the reserved words must be zero;
 *                 asyrp_run_edit accepts one hs_coeff tuple per image (n_coeff < 0).  The product library reads no environment variable. */
#define ASYRP_ABI_VERSION 8
int asyrp_abi_version(void);

/* Last error text of the calling thread ("" if none). */
const char* asyrp_last_error(void);

/* Create an engine on `device` able to run batches up to `max_batch` images.
 * Replaces: model construction, diffusion_latent.py:76-126 + model.setattr_layers(get_h_num) :586. */
int asyrp_create(asyrp_engine** out, const asyrp_config* cfg, int max_batch, int device);
void asyrp_destroy(asyrp_engine* e);

/* Hand one tensor of the reference state_dict to the engine (host pointer, fp32, contiguous,
 * PyTorch layout: conv [Cout,Cin,kh,kw], linear [out,in]).  `key` is the reference's own
 * state_dict name, e.g. "down.0.block.1.conv1.weight" or "layer_0.temb_proj.bias"
 * (so pretrained checkpoints and checkpoint/<name>.pth["0"] load unmodified).
 * Replaces: load_state_dict, diffusion_latent.py:124 and :663-676. */
int asyrp_load_param(asyrp_engine* e, const char* key, const float* host_data, const int64_t* shape, int ndim);

/* Diffusion schedule table, host pointer: alphas_cumprod[n] = cumprod(1-beta) computed by the caller
 * exactly as utils/diffusion_utils.py:67 does (fp32 cumprod of the fp32 betas). */
int asyrp_set_schedule(asyrp_engine* e, const float* alphas_cumprod_host, int n);

/* Sinusoidal-embedding frequencies, host pointer [ch/2], computed by the caller exactly as
 * models/ddpm/diffusion.py:52-54 (or improved_ddpm/nn.py:113-116) does, so they are bit-identical. */
int asyrp_set_temb_freqs(asyrp_engine* e, const float* freqs_host, int n);

/* Pack every loaded parameter into the kernel layout and upload.  Must be called after the last
 * asyrp_load_param and before any compute call; may be called again after re-loading some keys
 * (e.g. another Δh checkpoint).  Fails with ASYRP_EKEY if a required key is missing. */
int asyrp_finalize_params(asyrp_engine* e);

/* Number of parameter keys the engine expects, and the i-th key / its shape (for loaders). */
int asyrp_num_params(const asyrp_engine* e);
int asyrp_param_info(const asyrp_engine* e, int i, const char** key, int64_t shape[4], int* ndim);

/* B1 — one UNet evaluation (DDPM.forward, models/ddpm/diffusion.py:473-580).
 *   x [B,Cin,R,R], t [B] (float timesteps, device).
 *   index < 0  == index=None: single decoder; et_mod / delta_h_out are not written.
 *   index >= 0: DeltaBlocks layer_0..layer_index are summed (:513-516) when `apply_edit` != 0
 *               (the caller evaluates the reference's `t[0] >= t_edit` test, :510); with
 *               apply_edit == 0 the decoder input is h itself and et_mod == et bit-for-bit (:541-542).
 *   hs_coeff: host pointer, n_coeff = index+2 floats (c0, c1, ...).
 *   ignore_timestep: DeltaBlock gets temb=None (:514).
 *   delta_h_in (nullable, [B,Cb,Rb,Rb]): the reference's `delta_h=` tensor argument.  When given (and apply_edit), the
 *               DeltaBlocks are NOT evaluated; the decoder input is slerp(1-c0, h, |h|*delta_h/|delta_h|) per sample
 *               (:531-539) or, with use_mask != 0, slerp(1-c0, h*m, delta_h*m) + (1-m)*h with m = 1 on rows 4..Rb-2,
 *               columns 3..4 (:519-529); only hs_coeff[0] is read and delta_h_out is not written (the reference hands
 *               the caller's tensor back).  This is how diffusion_latent.py:516 applies the global mean delta-h.
 *   outputs: et [B,Cout,R,R]; et_mod [B,Cout,R,R] (nullable); delta_h_out [B,Cb,Rb,Rb] (nullable; the
 *   LAST DeltaBlock's raw output as the reference returns it); middle_h [B,Cb,Rb,Rb] (nullable). */
int asyrp_unet_forward(asyrp_engine* e, const float* x, const float* t, int B, int index, int apply_edit,
                       const float* hs_coeff_host, int n_coeff, int ignore_timestep, const float* delta_h_in, int use_mask,
                       float* et, float* et_mod, float* delta_h_out, float* middle_h, void* stream);

/* B2 — one fused DDIM step = UNet + update (denoising_step, utils/diffusion_utils.py:24-104,
 * sampling_type='ddim').  t / t_next are host ints shared by the batch (the reference builds
 * them as ones(B)*i, diffusion_latent.py:504-505); t_next = -1 means alpha_bar_next = 1 (:68-69).
 *   eta != 0 requires `noise` [B,3,R,R] (stands in for torch.randn_like, :97).
 *   learn_sigma: eps = first half of the output channels (:47-51).
 *   dt_lambda/dt_end: :99-100.  index/apply_edit/hs_coeff/ignore_timestep/delta_h_in/use_mask as above.
 *   outputs: xt_next, x0_t [B,3,R,R]; delta_h_out, middle_h nullable. */
int asyrp_ddim_step(asyrp_engine* e, const float* xt, int t, int t_next, int B, float eta, const float* noise,
                    int learn_sigma, int index, int apply_edit, const float* hs_coeff_host, int n_coeff,
                    int ignore_timestep, const float* delta_h_in, int use_mask, float dt_lambda, int dt_end,
                    float* xt_next, float* x0_t, float* delta_h_out, float* middle_h, void* stream);

/* The two hot loops back to back (diffusion_latent.py:1034-1045 then :503-520):
 *   x0 [B,3,R,R] --(n_inv-1 inversion steps over seq_inv)--> x_T --(n_gen Asyrp steps over seq_gen)--> x_edit.
 *   seq_inv / seq_gen: host int arrays, ascending timesteps exactly as diffusion_latent.py:955-957 builds them.
 *   n_inv == 0 skips inversion and starts generation from x0 interpreted as x_T.
 *   The edit is applied for steps with t >= t_edit; eta = 1 for steps with t < t_addnoise, consuming
 *   noise[k] ([n_noise,B,3,R,R], nullable when no such step) in step order.
 *   x_T (nullable) receives the inverted latent.
 *   hs_coeff_host / n_coeff: index + 2 coefficients for the whole batch (as asyrp_ddim_step), or -- n_coeff < 0, -n_coeff ==
 *   B * (index + 2) -- ONE TUPLE PER IMAGE, [B][index + 2]: the reference's editing-strength sweep (one generation pass per tuple,
 *   diffusion_latent.py:499-534, 726-755) as batch entries of one call.  Per image the arithmetic is that of a whole-batch call with
 *   its tuple (bit-identical; at most 128 images per call in this form). */
int asyrp_run_edit(asyrp_engine* e, const float* x0, int B, const int32_t* seq_inv_host, int n_inv,
                   const int32_t* seq_gen_host, int n_gen, int t_edit, int t_addnoise, int index,
                   const float* hs_coeff_host, int n_coeff, int learn_sigma, const float* noise, int n_noise,
                   float* x_T, float* x_edit, void* stream);

/* Loop A alone with a per-step read-out: DDIM inversion over seq_inv (n_inv - 1 steps, step k = 0..n_inv-2 takes
 * t = seq_inv[k] -> t_next = seq_inv[k+1]).  For the steps k in [tap_first, tap_first + tap_count) the step's outputs are
 * copied out: x_tap[k - tap_first] = x_{t_next} and x0t_tap[k - tap_first] = x0_t, each [B,3,R,R] (either buffer may be
 * null).  x_last (nullable) receives the final latent.  This is the engine half of the reference's LPIPS(t) table
 * builder, which feeds x and x0_t of every one of its (up to 1000) inversion steps to LPIPS
 * (diffusion_latent.py:1239-1276); callers walk a long inversion in windows by restarting from the last tap. */
int asyrp_run_inversion(asyrp_engine* e, const float* x0, int B, const int32_t* seq_inv_host, int n_inv, int learn_sigma,
                        int tap_first, int tap_count, float* x_tap, float* x0t_tap, float* x_last, void* stream);

/* ---- DeltaBlock training step (diffusion_latent.py:301-354), both UNet families, one DeltaBlock --------
 * The reference trains layer_0 with `loss(x0_t).backward(); optim.step()` where x0_t comes from
 * denoising_step(xt.detach(), ..., index=0, t_edit, hs_coeff) with gradients enabled on the DeltaBlock only
 * (diffusion_latent.py:282-290, 308-321, 349-350).  The loss (CLIP direction + L1, :337-347) stays in the caller's PyTorch;
 * the engine provides the step and the gradient of any such loss w.r.t. the DeltaBlock parameters:
 *   asyrp_train_forward   = asyrp_ddim_step(eta = 0, index = 0, apply_edit = 1) that keeps decoder #2's activations;
 *   asyrp_train_backward  takes dL/d(et_modified) [B,Cout,R,R] (the caller folds x0_t = (xt - et_mod*sqrt(1-a))/sqrt(a) and
 *                         xt_next = sqrt(a')*x0_t + ... into it), back-propagates through decoder #2 (transposed
 *                         convolutions on the same MFMA kernels; GroupNorm / SiLU / attention / upsample backward) to the
 *                         bottleneck and writes the gradients of the named DeltaBlock parameters ("layer_0.conv1.weight",
 *                         ... reference state_dict names and shapes) to the given DEVICE buffers.  Consumes the tape.
 *   asyrp_train_discard   drops an unconsumed tape (e.g. a step whose loss is not back-propagated).
 * The engine keeps ONE pending step.  asyrp_train_forward stamps it with a generation id (*tape_id); asyrp_train_backward must
 * present that id and fails with ASYRP_ESTATE when a later forward has replaced the step (forward A, forward B, backward(A)
 * would otherwise back-propagate B's activations); asyrp_train_discard(e, id) is a no-op for a stale id (id < 0: whatever is
 * pending).  Until the backward or a discard, the engine holds the tensors the backward reads (the skip tensors, the bottleneck,
 * decoder #2's and the DeltaBlock's activations) out of its workspace pool; the encoder's and decoder #1's intermediates are
 * recycled as in inference.  Every key handed to asyrp_train_backward must name a layer_0 parameter (ASYRP_EKEY otherwise).
 * Numerics note: the training forward evaluates decoder #2 without the skip-half sharing of the inference step and its attention
 * in the three-launch fp32-MFMA form (it keeps the softmax probabilities), so x0_t of asyrp_train_forward and of
 * asyrp_ddim_step on the same inputs agree to rounding (tested at rtol 1e-3 / atol 1e-4 x 1/sqrt(alpha_bar_t)), not bitwise. */
int asyrp_train_forward(asyrp_engine* e, const float* xt, int t, int t_next, int B, int learn_sigma,
                        const float* hs_coeff_host, int n_coeff, int ignore_timestep, float* xt_next, float* x0_t,
                        float* delta_h_out, float* middle_h, int64_t* tape_id, void* stream);
int asyrp_train_backward(asyrp_engine* e, int64_t tape_id, const float* d_et_mod, int n_grads, const char* const* keys,
                         float* const* grads_dev, void* stream);
void asyrp_train_discard(asyrp_engine* e, int64_t tape_id);

/* DDPM.get_temb (models/ddpm/diffusion.py:464-470): t [B] float timesteps (device) -> temb [B, 4*ch] (device).
 * For the iDDPM family: time_embed(timestep_embedding(t)) (models/improved_ddpm/unet.py:688). */
int asyrp_get_temb(asyrp_engine* e, const float* t, int B, float* temb_out, void* stream);

/* ---- vendored sampler signatures (never called by the reference) --------------------------------------------
 * The step arithmetic of GaussianDiffusion.p_sample / ddim_sample / ddim_reverse_sample / p_mean_variance
 * (models/guided_diffusion/gaussian_diffusion.py:232-321, 402-446, 544-630) in one elementwise launch.  Each of those updates is
 *     pred_xstart = clamp?(a*x - b*eps),   sample = p*pred_xstart + q*x + r*noise
 * with per-image scalars the caller derives from its float64 schedule tables; coef_host is [B][8] floats
 * {a, b, p, q, r, lo, hi, clip}.  model_out [B,out_channels,H,W]: eps = channels 0..C-1; when out_channels == 2C the channels
 * C..2C-1 are the learned variance interpolation v and the noise scale becomes r*exp(0.5*lv), lv = f*hi + (1-f)*lo,
 * f = (v+1)/2 (:254-262), written to log_variance [B,C,H,W] on request.  x, noise (nullable), sample, pred_xstart
 * (nullable each) are [B,C,H,W] NCHW device tensors.  Stateless; asynchronous on `stream`. */
int asyrp_sampler_update(int device, const float* x, const float* model_out, int out_channels, int B, int C, int HW,
                         const float* coef_host, const float* noise, float* sample, float* pred_xstart, float* log_variance,
                         void* stream);

/* Bytes of device memory held by the engine (weights + workspace). */
int64_t asyrp_device_bytes(const asyrp_engine* e);

/* ---- profiling hooks (used by bench.py for the roofline object) ------------------------------ */
/* Enable/disable per-kernel-family HIP-event timing on the launch stream.  While enabled every
 * launch of the dominant kernel family (implicit-GEMM conv) is bracketed by hipEvents. */
int asyrp_profile_enable(asyrp_engine* e, int on);
/* After a device sync: statistics of the implicit-GEMM launches recorded since the last read.
 * The DOMINANT variant (largest accumulated time) is reported in full:
 *   *variant = family*100000 + tile*1000 + ksize*100 + stride*10 + transposedB
 *     family 0 = igemm_f32 (tile: 1=128x128, 2=128x64, 3=64x64, 4=128x32),
 *     family 1 = igemm_f16x3 (tile: 1=256x128 4-wave pipelined, 2=128x128, 3=64x128, 4=64x64, 5=256x64,
 *                             6=256x128 8-wave on v_mfma_f32_32x32x16_f16, 7=the same tile on v_mfma_f32_16x16x32_f16
 *                             (igemm_f16x3_k32_kernel: the default for 3x3 stride-1 layers of 32x32 pixels and more whose
 *                             channel counts are multiples of 32), 8=its 128-pixel form (16x16-pixel layers), 9=its 8x8-patch form, 10=its stride-2 form,
 *                             11=the polyphase form of tile 7 for "nearest x2 then 3x3" (Upsample.conv / ResBlock(up=True)): four
 *                             phase-collapsed 2x2-tap convolutions on the source grid, 4/9 of the products; its FLOPs are
 *                             counted as issued (4 taps), 12=256x32, 14=the quad form of tile 7 for the 8x8-pixel layers: four images per
 *                             workgroup, split-K with a fixed-order reduce, 15 / 16 = the barrier-free 1x1 kernel of csrc/gemm1x1.hip
 *                             (256 / 128 pixels x 128 channels; weights in MFMA fragment order, no LDS staging)),
 *   its accumulated event time (ms), launch count, algorithmic FLOPs (2*M*N*K) and algorithmic bytes
 *   (input read once + output written once + weights once); all_ms / all_flops cover every variant.
 * Resets the record. */
int asyrp_profile_read(asyrp_engine* e, int* variant, double* ms, int64_t* launches, double* flops, double* bytes,
                       double* all_ms, double* all_flops);

/* Per-kernel-family rows of the same record (variant ids as above; 200000 + T = the fused attention kernel over T tokens
 * with flops = 4*T*T*C per image; 210000 + T = the split-plane attention kernel).  Does not reset the record.  Returns the number of rows written, negative on error. */
int asyrp_profile_table(asyrp_engine* e, int max_rows, int* variants, double* ms, int64_t* launches, double* flops,
                        double* bytes);

/* ---- op-level test hooks (tests/ only; same kernels the engine launches) ---------------------- */
/* y = conv2d(act(x)) [+bias] [+ per-image channel vector] [+ residual], NCHW fp32 in and out.
 *   x0 [B,C0,H,W] (+ optional x1 [B,C1,H,W] = channel concat), weight [Cout,C0+C1,k,k] (k = 1 or 3),
 *   stride 1 (pad k/2) or 2 (DDPM Downsample: pad right/bottom, models/ddpm/diffusion.py:103-107),
 *   upsample: nearest x2 before the conv (:84-85).
 *   gn_weight/gn_bias non-null: act = swish(GroupNorm32(x, eps)) (silu=1) or GroupNorm32 only (silu=0).
 *   conv_math: enum asyrp_conv_math; tile: 0 = the launcher's own choice, else force one tile shape of that
 *   kernel family so every compiled variant can be parity-tested (tile 7 falls back to tile 6 when Cin % 32 != 0);
 *   tile 11 = the polyphase form for upsample != 0 (3x3, Cin % 32 == 0, no residual);
 *   tiles 15 / 16 = the 1x1 kernel of csrc/gemm1x1.hip (1x1, Cin % 64 == 0, concat split % 32 == 0; ASYRP_EINVAL otherwise);
 *   tile 13 = the taps-in-N kernel of the UNet's last
 *   convolution (csrc/conv_out.hip: 3x3, stride 1, Cout*9 <= 32, GroupNorm + SiLU prologue required). */
int asyrp_op_conv2d(int device, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                    const float* weight, const float* bias, int Cout, int ksize, int stride, int upsample,
                    const float* gn_weight, const float* gn_bias, float gn_eps, int silu, const float* chan_add,
                    const float* residual, float* y, int conv_math, int tile, void* stream);
/* f16x3 conv whose epilogue also emits GroupNorm partial statistics of its OUTPUT, followed by the finalize kernel:
 * y [B,Cout,H,W] and scale/shift [B,Cout] such that y*scale+shift == GroupNorm32(y; gamma, beta, eps) (stride 1). */
int asyrp_op_conv2d_stats(int device, const float* x, int Cin, int B, int H, int W, const float* weight,
                          const float* bias, int Cout, int ksize, int tile, const float* gamma, const float* beta,
                          float eps, float* y, float* scale_out, float* shift_out, void* stream);
/* The tail of a ResnetBlock with a 1x1 shortcut as ONE launch (fused shortcut of the main f16x3 tile):
 *   y = conv3x3(swish(GroupNorm32(h; gn_weight, gn_bias, eps)), w3) + b3 + conv1x1(cat(x0, x1), w1) + b1
 * (models/ddpm/diffusion.py:159-170).  h [B,Ch,H,W], x0 [B,C0,H,W], x1 [B,C1,H,W] or null; channel counts multiples of 16. */
int asyrp_op_resblock_tail(int device, const float* h, int Ch, const float* x0, int C0, const float* x1, int C1, int B, int H,
                           int W, const float* w3, const float* b3, const float* w1, const float* b1, int Cout,
                           const float* gn_weight, const float* gn_bias, float gn_eps, float* y, void* stream);
/* AttnBlock core (models/ddpm/diffusion.py:205-221 / improved_ddpm/unet.py:379-396):
 * qkv [B,3C,T] as q|k|v (heads=1) or the "legacy" per-head [H,(q,k,v),Dh] order, out [B,C,T].
 * fused != 0: the one-launch f16x3 kernel (csrc/attention.hip; T <= 1024, head width <= 512, multiple of 16), the engine's
 * round-2 kernel (fused == 2: its single-product form, conv_math f16); fused == 3 / 4: the split-plane kernel attn_planes_kernel
 * (the engine's default since round 3 for T % 32 == 0, head width % 32 == 0; 4 = single product), fed here by a standalone
 * splitter instead of the q|k|v projection's epilogue; fused == 0: fp32-MFMA QK^T -> softmax pass -> fp32-MFMA PV (the conv_math="f32" engine and the fallback). */
int asyrp_op_attention(int device, const float* qkv, int B, int C, int T, int heads, int fused, float* out, void* stream);

#ifdef ASYRP_BENCH_HOOKS
/* ---- profiling library only: libasyrp_hip_bench.so = the same sources compiled with -DASYRP_BENCH_HOOKS ---------------
 * (python -m asyrp_official_amd.build --bench).  The product library neither exports this entry point nor contains the
 * ablation instantiations of the kernels.
 * Kernel micro-benchmark (scripts/conv_bench.py): times `iters` launches of one conv configuration on synthetic
 * NHWC buffers with HIP events and returns the average in *ms_out (host pointer).  `abl` != 0 selects the ablation
 * build of the main f16x3 tile with phases switched off (timing only; results are then wrong by construction). */
int asyrp_op_conv_bench(int device, int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample,
                        int prologue, int residual, int conv_math, int tile, int abl, int iters, float* ms_out,
                        void* stream);
/* The same launch, plus the phase stamps of the last one: stamps_host [stamps_cap >= workgroups][8] s_memrealtime ticks (100 MHz)
 * of igemm_f16x3_k32_kernel's ablation instantiation (abl != 0; abl bit 6 = also emit GroupNorm partial sums): 0 start, 1 first
 * tile staged, 2 K loop done, 3 epilogue stores issued, 4 end, 5 = XCC_ID << 32 | HW_ID, 6 stores drained (scripts/k32_phases.py). */
int asyrp_op_conv_stamps(int device, int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample,
                         int prologue, int residual, int conv_math, int tile, int abl, int iters, float* ms_out,
                         unsigned long long* stamps_host, int stamps_cap);
/* attn_planes_kernel on synthetic planes: average launch time over `iters` launches and the phase stamps of the last one,
 * stamps_host [B*heads*T/32][8] s_memrealtime ticks (100 MHz): 0 start, 1 Q staged, 2 S^T done, 3 P in LDS, 4 last PV item done,
 * 5 end (scripts/attn_phases.py). */
int asyrp_op_attention_phases(int device, int B, int C, int T, int heads, int np, int iters, float* ms_out,
                              unsigned long long* stamps_host, void* stream);
int asyrp_op_gemm1x1_phases(int device, int B, int H, int Cin, int Cout, int prologue, int np, int tile, int iters, float* ms_out,
                            unsigned long long* stamps_host, void* stream);
#endif

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ASYRP_H */
