// kernels.h — launch interface of the gfx950 kernels (internal; the public ABI is include/asyrp.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace asyrp {

// A/B switches (ASYRP_MAIN_TILE, ASYRP_POLYPHASE, ASYRP_ATTN, ...).  The PRODUCT library reads no environment variable: every switch
// is at its default there and the superseded kernels are reachable only as shape fallbacks.  The profiling library
// (libasyrp_hip_bench.so, -DASYRP_BENCH_HOOKS; loaded instead of the product by ASYRP_LIBRARY=bench, which bench.py records in its
// line) honours them, each read once per process.
inline const char* ab_env(const char* name) {
#ifdef ASYRP_BENCH_HOOKS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// records the calling thread's error string (what asyrp_last_error returns) and hands `code` back; defined in engine.hip
int set_last_error(int code, const char* msg);

// One implicit-GEMM launch: out[z][m][n] = alpha * sum_k A[z][m][k] * B[(z)][k][n] (+bias +chan_add +resid)
//   A = activations, NHWC, up to two channel-concatenated sources, optional per-(image,channel)
//       affine (GroupNorm apply / FiLM) + SiLU prologue, optional nearest-x2 upsample, 3x3 halo via LDS.
//   B = weights [ks*ks][Cin][ldb] (or, bT: B[k][n] = Bt[n][k], row stride ldb) optionally per-z.
//   z = zo*ZI + zi  (zo = image, zi = attention head); all z-offsets are in floats.
struct GemmArgs {
  const float* a0; const float* a1;
  int c0, c1, lda0, lda1;
  long long a0_zo, a0_zi, a1_zo, a1_zi;
  int Hin, Win, Hout, Wout;       // Hin/Win: source dims BEFORE upsample
  int Cin, Cout;
  int ks, stride, pad, ups;
  const float* pscale; const float* pshift;  // [zo][Cin] or null
  int ld_ps;                      // f16x3 only: row pitch of pscale/pshift when the launch covers a channel sub-range of a wider
                                  //   normalised tensor (0 = Cin)
  int silu;
  const float* w; int ldb; int bT; long long w_zo, w_zi;
  const float* bias;              // [Cout] or null
  const float* chan_add; int ld_chan_add;   // [zo][ld] or null
  const float* resid; int ldr; long long r_zo, r_zi;
  int rups;                       // residual is at half resolution: read pixel (oy>>1, ox>>1) of a (Hout/2 x Wout/2) tensor
                                  //   (nearest x2 of the skip path inside an iDDPM ResBlock(up=True), improved_ddpm/unet.py:281-283)
  float alpha;
  float cin_wmul;                 // conv_in_mfma_kernel: the power of two its weights are multiplied by before the f16 split (max|w| * cin_wmul in
                                  //   [1024, 2048), as for every other weight image); 0 = 1024
  float* out; int ldo; long long o_zo, o_zi;
  int ZI, Z;
  int tile;                       // 0 = auto, else TILE_* / XT_* id
  // f16x3 path (conv_f16x3.hip): weights pre-split/pre-packed by launch_pack_f16x3; alpha must carry
  // 1/(weight scale * f16x3_act_scale())
  int math;                       // MATH_F32 (v_mfma_f32_32x32x2_f32) or MATH_F16X3 (3 x v_mfma_f32_32x32x16_f16)
  const void* wpk; int cout_pad;
  // fused 1x1 shortcut (f16x3 main tile only): after the Cin/16 chunks x 9 taps of the 3x3 conv the K-loop runs Cin2/16 more
  // single-tap chunks over the raw (no prologue) virtual concat (s0|s1) with the 1x1 weights appended to `wpk`
  // (ResnetBlock.nin_shortcut, models/ddpm/diffusion.py:145-149,165-170; ResBlock.skip_connection, improved_ddpm/unet.py:264,298)
  const float* s0; const float* s1; int sc0, sc1, lds0, lds1; long long s0_zo, s1_zo; int Cin2;
  // split-K (f16x3 pipelined tiles): sk > 1 splits the Cin/16 chunks into sk equal ranges, one workgroup each
  // (gridDim.y = n_blocks * sk); every range writes alpha * acc to part[ks][z][pixel][Cout] and launch_splitk_reduce adds
  // them in a fixed order together with bias / chan_add / residual (and emits the GroupNorm partials).  Used for the 8x8
  // layers, whose M x N offers fewer tiles than the chip has CUs.
  int sk; float* part;
  double* stats;                  // f16x3 only, nullable: per-(image, M-block, out channel) {sum, sumsq} of the output,
                                  //   [Z][gemm_mblocks()][Cout][2]; consumed by launch_gn_finalize2
  int xmap;                       // set by the f16x3 launcher: XCD-aware block -> tile map (see igemm_f16x3_kernel)
  int abl;                        // ablation mask of the profiling build of the main tile (scripts/conv_bench.py); 0 in the product
  // polyphase launch of "nearest x2 then 3x3" (f16x3 family, K32Cfg<8, 2, 16, 1, 2>): Hin = Hout, Win = Wout are the SOURCE dimensions,
  // ups = 0, the output tensor is (2 Hout) x (2 Wout); wpk holds four images [phase py*2+px][chunk][4 taps][4 units][cout_pad][8]
  // of the phase-collapsed 2x2 weights, w_phase bytes apart; statistics rows: 4 * tiles per image (gemm_mblocks)
  int poly; long long w_phase;
  // split-plane output (32x32x16 family, 1x1 convolutions; the q|k|v projection of an attention block): instead of fp32 `out`, the
  // epilogue writes every output value as a two-term f16 split into planes the fused attention kernel (attention.hip,
  // attn_planes_kernel) consumes without any conversion work: o16h/o16l [zo][pixel][ld16] (same channel indexing as `out` would
  // have; the v channels are skipped) and the v channels TRANSPOSED, vth/vtl [zo][vc][HW] with vc = (n / v_mod) * v_dh +
  // (n % v_mod - v_off) for the channels with n % v_mod >= v_off (DDPM q|k|v blocks: v_mod = 3C, v_off = 2C, v_dh = C; the legacy
  // per-head [q|k|v] order of improved_ddpm/unet.py:389: v_mod = 3 Dh, v_off = 2 Dh, v_dh = Dh)
  _Float16* o16h; _Float16* o16l; _Float16* vth; _Float16* vtl; int ld16, v_mod, v_off, v_dh; long long o16_zo, vt_zo;
  int nz;                         // nominal batch of the engine's batch class (tile / split-K rules are priced at it, never at Z); 0 = 32
  int np;                         // f16x3 family: matrix products per term: 0 / 3 = two-term split (fp32-equivalent), 1 = single f16 product
  unsigned long long* dbg;        // profiling library only: phase stamps [workgroup][8] of the K32 ablation instantiation (null in the product)
  // profiling library only (K32 ablation instantiation): stag == 3 delays the second workgroup to arrive on a CU by stag_ticks
  // (100 MHz) once; the arrival counters sit behind the stamps in dbg.  0 in the product.
  int stag, stag_ticks;
  // gemm1x1.hip, XT_G1_128 on 8 x 8 maps (round 5; set by launch_gemm1x1): the workgroup's two wave rows take two IMAGES (64 pixels
  // each) instead of two 64-pixel halves of one image; gridDim.z = ceil(Z / 2).  Per image the instructions and their order are
  // those of the one-image form, so an image's bits do not depend on its partner.
  int g1_pair;
};

enum { MATH_F16X3 = 0, MATH_F32 = 1 };

enum { TILE_AUTO = 0, TILE_128x128 = 1, TILE_128x64 = 2, TILE_64x64 = 3, TILE_128x32 = 4 };

// tile ids of the f16x3 family: 1 = 256x128, 4 waves, software-pipelined loop (big 1x1 layers; 3x3 A/B reference);
// 6 = 256x128, 8 waves, per-tap loop (the big 3x3 layers and their fused shortcut); 12 = conv_out
enum { XT_AUTO = 0, XT_256x128 = 1, XT_128x128 = 2, XT_64x128 = 3, XT_64x64 = 4, XT_256x64 = 5, XT_256x128W8 = 6,
       XT_256x128K32 = 7 /* the 8-wave tile on v_mfma_f32_16x16x32_f16: 3x3 stride 1, Cin % 32 == 0 */,
       XT_128x128K32 = 8 /* its 128-pixel form (8 x 16 patch, 8 waves of 64 pixels x 32 channels) */,
       XT_64x128K32 = 9 /* 8 x 8 patch (one 8 x 8 image), 8 waves of 64 pixels x 16 channels */,
       XT_64x128K32S2 = 10 /* stride 2 (DDPM Downsample): 4 x 16 output patch, 8 waves of 64 pixels x 16 channels */,
       XT_256x128K32UP = 11 /* polyphase form of the main tile: nearest x2 + 3x3 as four 2x2-tap phases on the source grid */,
       XT_256x128K32Q = 14 /* quad form for the 8 x 8 layers: four images per workgroup, split-K (GemmArgs.sk / part) */,
       XT_G1_256 = 15 /* gemm1x1.hip: barrier-free 1x1 kernel, 256 pixels x 128 channels, 8 waves; wpk = its fragment-major image */,
       XT_G1_128 = 16 /* the same with 128 pixels x 128 channels, 4 waves, two workgroups per CU */,
       XT_CONV_IN = 17 /* conv_in.hip (the op-level test hooks only: the engine calls launch_conv_in itself) */,
       XT_256x32 = 12 /* Cout <= 32 */ };

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);            // dispatches on a.math
hipError_t launch_gemm_f32(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_f16x3(const GemmArgs& a, hipStream_t s);
int gemm_resolve_tile(const GemmArgs& a);     // TILE_* the fp32 launcher will pick
int gemm_resolve_tile_x(const GemmArgs& a);   // XT_* of the kernel the f16x3 launcher will run for `a`
// f16x3 weight image: [ceil(Cin/16)*ks*ks][4][roundup(Cout,128)][8] halfs
size_t f16x3_packed_halfs(int cout, int cin, int ks);
hipError_t launch_pack_f16x3(const float* w_dev /*[Cout][Cin][ks][ks]*/, void* dst, int cout, int cin, int ks,
                             float wscale, hipStream_t s);
float f16x3_act_scale();
// gemm1x1.hip: 1x1 convolutions without LDS staging; weight image [ceil(Cout/16)][Cin/32][hi|lo][64 lanes][8] halfs (MFMA fragment order)
size_t gemm1x1_packed_halfs(int cout, int cin);
hipError_t launch_gemm1x1_pack(const float* w_dev /*[Cout][Cin]*/, void* dst, int cout, int cin, float wscale, hipStream_t s);
bool xcd_map_enabled();                        // ASYRP_XCD_MAP != 0 (conv_f16x3.hip)
bool gemm1x1_ok(const GemmArgs& a);           // shape / alignment rules of the kernel (tile and wpk not looked at)
hipError_t launch_gemm1x1(const GemmArgs& a, hipStream_t s);   // a.tile = XT_G1_256 | XT_G1_128, a.wpk = the image above
// algorithmic work of one launch (2*M*N*K flops; A read once + out written once + weights once)
void gemm_work(const GemmArgs& a, double* flops, double* bytes);

// conv_out.hip: the UNet's last 3x3 convolution (Cout = 3 / 6) with the taps folded into N; `a.wpk` = f16x3 image of the equivalent
// 1x1 conv w1[tap*Cout + co][ci] (launch_pack_f16x3 with cout = 9*Cout, ks = 1), prologue scale/shift + SiLU mandatory
bool conv_out_two_tiles();                     // the 6-channel iDDPM head on two N tiles of conv_out.hip (default; ASYRP_CONV_OUT6=0 disables; Cin <= 128 only)
bool conv_out_supported(const GemmArgs& a);
hipError_t launch_conv_out(const GemmArgs& a, hipStream_t s);

// conv_in.hip: the UNet's first convolution (3 -> Cout, 3x3, pad 1, no prologue) as an fp32 stencil; a.w = [tap][3][Cout] fp32,
// statistics rows = 16 x 16 patches per image
bool conv_in_supported(const GemmArgs& a);
int conv_in_stat_blocks(const GemmArgs& a);
hipError_t launch_conv_in(const GemmArgs& a, hipStream_t s);

// GroupNorm(32) statistics of an NHWC tensor (two concatenated sources allowed) -> per-(image,channel)
// scale/shift so that y = x*scale + shift == GN(x)*gamma+beta; optional FiLM (scale,shift) folding:
// y = GN(x)*(1+fs)+fsh.  `partial` is scratch of gn_partial_floats() floats (holds doubles).
struct GnArgs {
  const float* a0; const float* a1;
  int c0, c1, lda0, lda1;
  long long a0_z, a1_z;
  int HW, N, C;
  const float* gamma; const float* beta; float eps;
  const float* film_scale; const float* film_shift; int ld_film;   // [N][ld] or null
  float* scale; float* shift;     // [N][C]
  double* partial;
};
size_t gn_partial_doubles(int N, int C, int HW);
hipError_t launch_gn(const GnArgs& a, hipStream_t s);
int gn_nblk_of(int HW);                       // M-blocks launch_gn_partial writes
// out = sum_ks part[ks] + bias + chan_add + resid; stats (nullable) = [Z][splitk_stat_blocks(HW)][Cout][2] doubles
hipError_t launch_splitk_reduce(const GemmArgs& a, hipStream_t s);
int splitk_stat_blocks(int HW);
int splitk_factor_shared(const GemmArgs& a);   // the same without the 16 x 16 rule (partial launches of a shared skip half)
int splitk_factor(const GemmArgs& a);   // 1 = none; a function of the LAYER SHAPE only (batch-invariant results)
bool splitk_quad(const GemmArgs& a);    // the split launch runs on the quad form (XT_256x128K32Q) instead of the 64x64 tile
bool splitk16(const GemmArgs& a);       // 16 x 16 maps: 2-way split on the 128-pixel K32 form (round 4)
bool splitk_unfused(const GemmArgs& a);  // a block's 1x1 shortcut must run as its own launch (quad form, small batch class)
int splitk_tile(const GemmArgs& a);     // XT_* a split launch of `a` runs on (quad form, a plain K32 form, or the 64x64 tile)
int gemm_main_tile();
bool gemm_can_fuse_shortcut(const GemmArgs& a);   // true when launch_gemm_f16x3 would run `a` (with s0/Cin2 set) on the fusing tile
int gemm_mblocks(const GemmArgs& a);          // M-blocks (gridDim.x) launch_gemm_f16x3 will use -> rows of GemmArgs.stats
// standalone partial statistics of one NHWC tensor -> partial [N][gn_nblk_of(HW)][C][2] doubles
hipError_t launch_gn_partial(const float* a, int lda, long long a_z, int HW, int N, int C, double* partial, hipStream_t s);
// GroupNorm(32) scale/shift from per-channel partial statistics of up to two channel-concatenated sources
struct GnFin2Args {
  const double* p0; int nblk0; int C0;      // [N][nblk0][C0][2]
  const double* p1; int nblk1; int C1;      // second source of a concat, or null
  int N, HW;
  const float* gamma; const float* beta; float eps;
  const float* film_scale; const float* film_shift; int ld_film;
  float* scale; float* shift;               // [N][C0+C1]
  float* mr;                                // nullable: [N][32][2] = (mean, rstd) of every group, kept for the backward pass
};
hipError_t launch_gn_finalize2(const GnFin2Args& a, hipStream_t s);

hipError_t launch_softmax_rows(float* x, long long rows, int T, hipStream_t s);

// Fragment-major layout of the attention planes: every 16-row x 32-column tile of a (rows x cols) f16 matrix is stored as the
// 1 KiB block one v_mfma_f32_16x16x32_f16 operand fetch reads -- [lane = (col % 32) / 8 * 16 + row % 16][col % 8] -- tiles
// row-major over (row / 16, col / 32).  A wave's operand load is then ONE fully coalesced 1-KiB access instead of 16 strided
// 64-byte segments (measured: the strided form held the attention kernel at ~23 GB/s of loads per CU, profiles/rd3h_*).
//   q|k planes: rows = tokens, cols = the 3C channels of the q|k|v rows (cols_per_row = ld16);  v^T planes: rows = v channels,
//   cols = tokens (cols_per_row = T).  rows % 16 == 0 and cols % 32 == 0.
__host__ __device__ inline long long frag_off(int row, int col, int cols_per_row) {
  return ((long long)((row >> 4) * (cols_per_row >> 5) + (col >> 5)) * 64 + ((col & 31) >> 3) * 16 + (row & 15)) * 8 + (col & 7);
}

// Fused attention (attention.hip): out[b][t][head*Dh + d] = sum_k softmax_k(scale * q_t.k_k) v_k[d], f16x3 matrix products.
// qkv rows are tokens: element (b, t, .) at qkv + b*img_stride + head*head_stride + {q,k,v}_off + t*ld (+ channel).
struct AttnArgs {
  const float* qkv; int ld; long long img_stride, head_stride; int q_off, k_off, v_off;
  int B, heads, T, Dh;
  float scale;
  float* out; int ldo; long long o_img_stride, o_head_stride;
  int np;                         // matrix products per term: 0 / 3 = two-term split, 1 = single f16 product
  // attn_planes_kernel: q|k as f16 hi/lo planes [B][T][ld16] (element (b, t, c) of q at b*T*ld16 + head*head_stride + q_off + t*ld16 + c,
  // all in halfs) and v transposed, [B][heads*Dh][T]; written by the q|k|v projection's epilogue (GemmArgs::o16h ...)
  const _Float16* qkh; const _Float16* qkl; const _Float16* vth; const _Float16* vtl; int ld16;
  unsigned long long* dbg;        // profiling library only: phase stamps [workgroup][8] (null in the product)
};
bool attn_fused_supported(int T, int Dh, int ld, int ldo);
hipError_t launch_attention_fused(const AttnArgs& a, hipStream_t s);
bool attn_planes_supported(int T, int Dh);
hipError_t launch_attention_planes(const AttnArgs& a, hipStream_t s);
// fp32 q|k|v [B][T][ld] -> the split planes above (test hook / standalone use; the engine gets them from the projection's epilogue)
hipError_t launch_qkv_to_planes(const float* qkv, int ld, int B, int T, int C3, int v_mod, int v_off, int v_dh, _Float16* o16h,
                                _Float16* o16l, _Float16* vth, _Float16* vtl, hipStream_t s);

// temb = dense1(swish(dense0(sinusoid(t)))) ; sin_first: DDPM [sin|cos] vs iDDPM [cos|sin]
hipError_t launch_temb_mlp(const float* t, const float* freqs, int half, int sin_first, const float* w0,
                           const float* b0, const float* w1, const float* b1, int ch, int temb_ch, float* temb,
                           float* temb_act /* swish(temb) */, int B, hipStream_t s);
// out[b][o] = sum_i W[o][i]*x[b][i] + bias[o]   (all per-block temb projections in one launch)
hipError_t launch_linear_rows(const float* x, int ldx, const float* W, const float* bias, int I, int O, float* out,
                              int ldo, int B, hipStream_t s);

// h2 = c0*h + sum_i c_{i+1} * d_i      (n_d <= 4), elementwise over `n` floats
hipError_t launch_mix(const float* h, const float* const* d, const float* coeff_host, int n_d, float* h2,
                      long long n, hipStream_t s);
// one coefficient tuple per image: coeff_host = [B][n_d + 1] (B <= MIX_MAX_IMAGES: the tuples travel as kernel arguments)
constexpr int MIX_MAX_IMAGES = 128;
hipError_t launch_mix_per_image(const float* h, const float* const* d, const float* coeff_host, int n_d, float* h2, int B,
                                long long per_image, hipStream_t s);
// h2 = slerp mix of h with an injected delta-h tensor (NHWC, one workgroup per sample); tt = 1 - hs_coeff[0]
hipError_t launch_slerp_mix(const float* h, const float* dh, float tt, int use_mask, int B, int H, int W, int C, float* h2,
                            hipStream_t s);

// iDDPM ResBlock(down=True) (models/improved_ddpm/unet.py:279-284): hp = avgpool2(silu(x*scale+shift)), xp = avgpool2(x),
// NHWC, H and W even; scale/shift [N][C] are the GroupNorm apply terms of in_layers.0
hipError_t launch_pool2(const float* x, int N, int H, int W, int C, const float* scale, const float* shift, float* hp,
                        float* xp, hipStream_t s);

// NCHW <-> NHWC
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, int lds, float* dst, int N, int C, int HW, hipStream_t s);

// DDIM update on NHWC(3) tensors, arithmetic order of utils/diffusion_utils.py:84-100
struct DdimArgs {
  const float* xt; const float* et; const float* et_mod; int ld_e;   // et: [N][HW][ld_e], first 3 channels used
  const float* noise;             // NHWC(3) or null
  float at, at_next, eta, dt_lambda; int apply_dt;
  float* xt_next; float* x0_t;    // x0_t nullable
  long long npix;                 // N*HW
};
hipError_t launch_ddim(const DdimArgs& a, hipStream_t s);

// per-image affine sampler update (kernels.hip: sampler_update_kernel); NCHW, all device pointers, `eps`/`var` are channel
// blocks of one model output whose images are eps_img floats apart
struct SamplerCoef { float a, b, p, q, r, lo, hi; int clip; };
struct SamplerArgs {
  enum { MAXB = 32 };
  const float* x; const float* eps; const float* var; const float* noise;   // var / noise nullable
  long long eps_img; int C, HW, nb;
  float* out; float* x0; float* logvar;                                     // each nullable
  SamplerCoef k[MAXB];
};
hipError_t launch_sampler_update(const SamplerArgs& a, hipStream_t s);

// ---- training step (backward.hip): data gradients through decoder #2, DeltaBlock parameter gradients ----
struct ActBwdArgs {
  const float* dA; int ldd;                 // gradient w.r.t. act(GN(x)), [N][HW][ldd]
  const float* x0; const float* x1; int c0, c1, ldx0, ldx1; long long x0_z, x1_z;   // forward input of the norm (virtual concat)
  const float* scale; const float* shift;   // [N][C]: the forward's GroupNorm apply terms
  int silu;                                 // act = SiLU (1) or identity (0: attention norm)
  float* dy;                                // [N][HW][C]: dA * act'(y)
  double* partial;                          // [N][act_bwd_nblk(HW)][C][2]: sum dy, sum dy*x
  int HW, N, C;
};
int act_bwd_nblk(int HW);
hipError_t launch_act_bwd_partial(const ActBwdArgs& a, hipStream_t s);
struct GnBwdFinArgs {
  const double* partial; int nblk;
  const float* gamma; const float* mr;      // mr [N][32][2] from the forward finalize
  const float* film_scale; int ld_film;     // iDDPM FiLM: the forward applied gamma*(1 + film_scale[n][c]); null otherwise
  int N, HW, C;
  float* coef;                              // [N][C][3]: dx = coef0*dy + coef1*x + coef2
};
hipError_t launch_gn_bwd_finalize(const GnBwdFinArgs& a, hipStream_t s);
hipError_t launch_gn_bwd_apply(const float* dy, int C, const float* x0, int ldx0, long long x0_z, const float* coef,
                               const float* add, float* dx, int Cd, int HW, int N, hipStream_t s);
hipError_t launch_sum2x2(const float* in, float* out, int N, int H, int W, int C, hipStream_t s);
hipError_t launch_transpose(const float* in, int ldi, long long in_z, float* out, int R, int Ccols, long long out_z, int Z,
                            hipStream_t s);
hipError_t launch_softmax_bwd(const float* P, float* dP, long long rows, int T, float scale, hipStream_t s);
hipError_t launch_colsum(const float* in, int ld, long long M, int C, float* out, hipStream_t s);
hipError_t launch_act_apply(const float* x, const float* scale, const float* shift, int silu, float* out, int N, int HW, int C,
                            hipStream_t s);
hipError_t launch_gn_param_grad(const double* partial, int nblk, const float* mr, int N, int C, float* dgamma, float* dbeta,
                                hipStream_t s);
hipError_t launch_scale(const float* a, float sc, float* out, long long n, hipStream_t s);

}  // namespace asyrp
