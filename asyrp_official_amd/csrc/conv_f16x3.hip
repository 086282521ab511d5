// conv_f16x3.hip — the dominant kernel of the engine: implicit-GEMM convolution on the f16 matrix cores with a
// two-term operand split ("f16x3"), fp32-equivalent results at 16/3 x the fp32-MFMA rate.
//
//   y = conv(act(x), w)  with  act(x) = x_hi + x_lo,  w = w_hi + w_lo   (each term an f16, exact two-term split)
//     ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi        (the dropped x_lo*w_lo term is 2^-22 relative)
// Every product of two f16 values is exact in fp32 and v_mfma_f32_32x32x16_f16 accumulates in fp32, so the
// result differs from an fp32 fma chain only by summation order: measured max |err| 1.5e-6 on the whole CelebA-HQ
// UNet forward (tests/experiments/split_precision_numerics.py), the same class as the fp32-MFMA kernel (3e-6).
// gfx950 has no xf32/TF32; its fp32-input MFMA runs at 157 TFLOP/s, the f16 MFMA at 2.5 PFLOP/s dense, so three
// f16 MFMAs per K-slice have a ceiling of 833 TFLOP/s of fp32-equivalent work.
//
// Data flow of one workgroup (BM output pixels x BN output channels of one image):
//   A (activations, fp32 NHWC in HBM, up to two channel-concatenated sources, optional nearest-x2 upsample):
//       global_load_dwordx4 -> registers -> GroupNorm-apply (per-(image,channel) scale/shift) + SiLU -> f16 hi/lo
//       split -> ds_write_b128 into the halo tile [unit u=4][pixel][8 x f16]; the tile of one 16-channel chunk is
//       staged ONCE and re-used by all 9 taps (LDS holds the (PH+2)x(PW+2) halo).
//   B (weights, pre-split and pre-packed at load time in exactly the LDS image order):
//       global_load_lds_dwordx4 (LDS-DMA, no VGPRs) one (chunk,tap) slice [u=4][BN][8 x f16] per K-step, double buffered.
//   (this header describes the 32x32x16 family; the 3x3 layers whose channel counts are multiples of 32 run on
//   igemm_f16x3_k32_kernel further down, same data flow, one v_mfma_f32_16x16x32_f16 per two (chunk, tap) slices)
//   MFMA: per K-step (16 channels of one tap) each wave issues TM*TN*3 v_mfma_f32_32x32x16_f16 from
//       (TM+TN)*2 ds_read_b128 fragments; "unit-major" LDS layout: B-fragment reads conflict-free, A-fragment reads
//       conflict-free with the row permutation below (4-wave tiles).
//   Two loop organisations: 8 waves per workgroup with a plain per-tap loop (4 waves per SIMD hide the latencies; the
//       default for the big 3x3 layers) and 4 waves with a software-pipelined loop (all other tiles) -- DESIGN.md 3.1.
//   Epilogue: acc * alpha (undoes the power-of-two operand scales) + bias + per-image channel vector (timestep
//       projection) + residual -> fp32 NHWC, 128-B contiguous per pixel row of a 32-channel MFMA tile.
// Reference ops replaced: models/ddpm/diffusion.py:151-170 (ResnetBlock convs + nin_shortcut), :72-110 (Up/Downsample),
// :179-198 (AttnBlock q,k,v,proj_out 1x1), :236-248 (DeltaBlock 1x1), :356-360/:426-430 (conv_in/conv_out).
#include <cstdlib>
#include "kernels.h"

namespace asyrp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int XKC = 16;                 // input channels per K-step (= K of one v_mfma_f32_32x32x16_f16)
constexpr float ACT_SCALE = 1.0f;       // optional power-of-two pre-scale of the activations (folded into alpha on the host).
                                        // 1: x_lo of |x| < 0.125 is an f16 subnormal, i.e. carries an absolute error
                                        // <= 2^-25 -- the matrix cores keep f16 subnormal inputs (tested), so a UNet's
                                        // O(1) activations keep fp32-class accuracy without the extra multiply
constexpr float H_MAX = 65504.0f;

template <int WM_, int WN_, int TM_, int TN_, int KS_, int STRIDE_, int RB_ = 2, int TS_ = 1>
struct XCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, KS = KS_, STRIDE = STRIDE_;
  static constexpr int RB = RB_;                       // weight-slice ring size: a slice is in flight for RB-1 barriers
  static constexpr int TS = TS_;                       // taps per barrier ("fat" K-step): one ring slot holds TS consecutive
                                                       // tap slices of a chunk; 3 on the small tiles whose single-tap steps
                                                       // (3-12 MFMAs per wave) are shorter than a barrier round trip
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int PW = (KS == 1) ? BM : (BM >= 128 ? 16 : 8);
  static constexpr int PH = BM / PW;
  static constexpr int TH = (PH - 1) * STRIDE + KS, TW = (PW - 1) * STRIDE + KS;
  static constexpr int NPIX = TH * TW;
  static constexpr int A_BYTES = NPIX * 64;            // [4 units][NPIX][16 B]
  static constexpr int B_BYTES = BN * 64;              // [4 units][BN][16 B]
  static constexpr int NU = NPIX * 2;                  // staging work items: (pixel, 8-channel half)
  static constexpr int NA = (NU + NT - 1) / NT;
  static constexpr int NTAPS = KS * KS;
  static constexpr int NPIECE = 4 * BN / 64;           // 1-KiB LDS-DMA pieces per B slice
  static constexpr size_t SMEM = 2 * (size_t)A_BYTES + RB * (size_t)B_BYTES * TS;
  static constexpr int NPW = (TS * NPIECE + NW - 1) / NW;   // LDS-DMA instructions each wave issues per ring slot
  // workgroups per CU the LDS admits x waves per workgroup / 4 SIMDs = waves per SIMD to budget registers for
  static constexpr int MINW = ((SMEM <= 76 * 1024) ? 2 : 1) * (NW / 4);
};

// MFMA row i (0..31) of a wave's 32-pixel group (two rows of a 16-wide tile, row pitch 18 pixels in the LDS halo) ->
// pixel index within the group.  `ds_read_b128` is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}
// (+32): with the identity mapping the second tile row starts 2 bank-quads after the first and every group has a 2-way
// conflict (A-fragment reads take 8 LDS cycles instead of 4: SQ_LDS_BANK_CONFLICT = 1/4..1/3 of SQ_LDS_IDX_ACTIVE).
// This bijection gives each lane group 16 distinct bank quads: {row0 x0-7, row1 x6-13} and {row0 x8-15, row1 x14,15,0-5}.
__device__ __forceinline__ int row_perm32(int i) {
  if (i < 4) return i;
  if (i < 12) return i + 4;
  if (i < 16) return i - 8;
  if (i < 18) return i + 14;
  if (i < 20) return i - 2;
  if (i < 28) return i + 2;
  return i - 10;
}

__device__ __forceinline__ float silu_fast(float v) {
  // x * sigmoid(x) (models/ddpm/diffusion.py:63-65) with v_exp_f32 / v_rcp_f32 (<= 2 ulp each)
  const float e = __expf(-v);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// two-term f16 split of 8 fp32 values: hi = rn_f16(x), lo = rn_f16(x - hi)  (x - hi is exact in fp32).
// Pairs go through v_cvt_pk_f16_f32; |x| is clamped to the f16 range first (saturates instead of producing inf).
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 s;
    s[0] = __builtin_amdgcn_fmed3f(v[j] * ACT_SCALE, -H_MAX, H_MAX);
    s[1] = __builtin_amdgcn_fmed3f(v[j + 1] * ACT_SCALE, -H_MAX, H_MAX);
    const h2 h = __builtin_convertvector(s, h2);
    f2 r;
    r[0] = s[0] - (float)h[0];
    r[1] = s[1] - (float)h[1];
    const h2 l = __builtin_convertvector(r, h2);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

// single-product mode: 8 fp32 values -> f16 (round to nearest even, |x| clamped to the f16 range)
__device__ __forceinline__ h8 round8(const float (&v)[8]) {
  h8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 s;
    s[0] = __builtin_amdgcn_fmed3f(v[j] * ACT_SCALE, -H_MAX, H_MAX);
    s[1] = __builtin_amdgcn_fmed3f(v[j + 1] * ACT_SCALE, -H_MAX, H_MAX);
    const h2 h = __builtin_convertvector(s, h2);
    r[j] = h[0]; r[j + 1] = h[1];
  }
  return r;
}

// VEC: Cin, c0, c1 and both row strides are multiples of 16 floats and the bases 16-B aligned (every layer but
// conv_in): a K-chunk is 16 contiguous floats of ONE source per pixel -> two global_load_dwordx4 per work item.
// ABL: profiling-only instantiation (compiled with -DASYRP_BENCH_HOOKS into libasyrp_hip_bench.so only) whose phases can be
//      switched off at run time through p.abl (timing ablations; results are then wrong by construction): 1 = no staging math, 2 = no weight LDS-DMA in the loop, 4 = no MFMA,
//      8 = no activation loads/writes in the loop, 16 = no per-step wait+barrier.
// PIPE: software-pipelined K-loop (4-wave tiles): the per-step barrier sits between MFMA pass 2 and pass 3, the next
//       step's x_lo / w_hi fragments are fetched right after it and their LDS latency is covered by pass 3, so every
//       step opens with matrix work already fed from registers.  Same products in the same order as the plain loop.
// SC:   fused 1x1 shortcut: Cin2/16 extra single-tap K-chunks over the raw tensor (s0|s1) after the 3x3 chunks (TS=1);
//       launched on the 8-wave tile (plain loop); the pipelined loop keeps its implementation for A/B
// NP:   matrix products per term: 3 = two-term split (fp32-equivalent), 1 = single f16 product (conv_math "f16": only the hi
//       planes are staged and multiplied; the weight slices are still DMA'd whole -- these tiles are not the dominant ones)
template <class T, bool VEC, bool ABL = false, bool PIPE = false, bool SC = false, int NP = 3>
__global__ void __launch_bounds__(T::NT, T::MINW) igemm_f16x3_kernel(const GemmArgs p) {
  const int abl = ABL ? p.abl : 0;
  constexpr int WN = T::WN, TM = T::TM, TN = T::TN, KS = T::KS, STRIDE = T::STRIDE, NW = T::NW;
  constexpr int NT = T::NT, BM = T::BM, BN = T::BN, PW = T::PW, TW = T::TW;
  constexpr int NPIX = T::NPIX, A_BYTES = T::A_BYTES, B_BYTES = T::B_BYTES, NA = T::NA, NTAPS = T::NTAPS;
  constexpr int NPIECE = T::NPIECE, NU = T::NU;
  // 8-wave workgroups run 4 waves per SIMD inside a 128-VGPR budget: activation loads are issued right before their
  // staging pass (not a chunk ahead) and fragments are fetched per MFMA pass; the other three waves hide the latency.
  constexpr bool LOWREG = (NW >= 8);
  // conflict-free A-fragment reads (row_perm32) on the 4-wave tiles: SQ_LDS_BANK_CONFLICT 23 % -> 0, +1.1...1.7 % on the
  // pipelined 256x128 tile.  The 8-wave tile measured 2 % SLOWER with it (interleaved A/B on one box,
  // profiles/r01_conv_microbench_perm1.txt / _scv1.txt) and keeps the identity map.
  constexpr bool RPERM = (KS == 3 && STRIDE == 1 && PW == 16) && !LOWREG;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // weight ring first: LDS-DMA destinations go through M0, kept below 64 KB; the halo tiles behind it are written by ds_write
  char* const Bs = smem;
  char* const As = smem + T::RB * T::TS * B_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int zo = blockIdx.z;
  const int sk = (PIPE && p.sk > 1) ? p.sk : 1;          // split-K: blockIdx.y = n_block * sk + k_range
  const int ks_id = blockIdx.y % sk;
  const int n0 = (blockIdx.y / sk) * BN;
  const int HWo = p.Hout * p.Wout;
  // M-tile of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin by linear block id, so with the identity map
  // the tiles resident on one XCD (one L2) are 8 apart and never share a halo row.  xmap: XCD k takes the contiguous band
  // [k*gx/8, (k+1)*gx/8) of tiles instead, neighbours meet in the same L2 (only the block -> tile assignment changes: every
  // tile is computed exactly as before and the statistics rows are indexed by the TILE, so results are bit-identical).
  int bx = blockIdx.x;
  if (p.xmap) bx = (bx & 7) * ((int)gridDim.x >> 3) + (bx >> 3);
  int m0 = 0, oy0 = 0, ox0 = 0;
  if (KS == 1) {
    m0 = bx * BM;
  } else {
    const int tiles_x = (p.Wout + PW - 1) / PW;
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    oy0 = ty * T::PH;
    ox0 = tx * PW;
  }
  const float* __restrict__ a0 = p.a0 + (long long)zo * p.a0_zo;
  const float* __restrict__ a1 = p.a1 ? p.a1 + (long long)zo * p.a1_zo : nullptr;
  const int ldps = p.ld_ps ? p.ld_ps : p.Cin;
  const float* __restrict__ ps = p.pscale ? p.pscale + (long long)zo * ldps : nullptr;
  const float* __restrict__ psh = p.pshift ? p.pshift + (long long)zo * ldps : nullptr;
  const char* __restrict__ wpk = reinterpret_cast<const char*>(p.wpk);
  const int Cin = p.Cin, Cout = p.Cout, c0 = p.c0;
  const int nch1 = (Cin + XKC - 1) / XKC;   // chunks of the conv proper (the fused shortcut's chunks follow)

  // ---- A staging map: work item u = (pixel, 8-channel half); NT is even so the half is per-thread constant ----
  const int hf = tid & 1;
  int aoff[NA];   // source pixel index, -1 = zero padding, -2 = no work item
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tid + i * NT;
    const int pix = u >> 1;
    int off = -2;
    if (u < NU) {
      if (KS == 1) {
        const int m = m0 + pix;
        off = (m < HWo) ? m : -1;
      } else {
        const int iy = pix / TW, ix = pix - iy * TW;
        const int gy = oy0 * STRIDE - p.pad + iy, gx = ox0 * STRIDE - p.pad + ix;
        const int Hu = p.Hin << p.ups, Wu = p.Win << p.ups;
        off = (gy >= 0 && gy < Hu && gx >= 0 && gx < Wu) ? ((gy >> p.ups) * p.Win + (gx >> p.ups)) : -1;
      }
    }
    aoff[i] = off;
  }

  float4 areg[NA][2];
  float4 sreg[4];   // scale[8], shift[8] of this thread's channels in the chunk being staged

  auto gload_A = [&](int chunk) {
    const int c = chunk * XKC + hf * 8;
    if (VEC) {
      const float* __restrict__ base;
      int ld;
      if (SC && chunk >= nch1) {   // shortcut phase: raw (s0|s1), no prologue
        const int cc = (chunk - nch1) * XKC + hf * 8;
        const bool second = ((chunk - nch1) * XKC >= p.sc0);
        base = second ? p.s1 + (long long)zo * p.s1_zo + (cc - p.sc0) : p.s0 + (long long)zo * p.s0_zo + cc;
        ld = second ? p.lds1 : p.lds0;
      } else {
        if (ps && !LOWREG) {   // (the 8-wave tiles fetch them at staging time: 16 registers less across the MFMA passes)
          sreg[0] = *reinterpret_cast<const float4*>(ps + c);
          sreg[1] = *reinterpret_cast<const float4*>(ps + c + 4);
          sreg[2] = *reinterpret_cast<const float4*>(psh + c);
          sreg[3] = *reinterpret_cast<const float4*>(psh + c + 4);
        }
        // the whole chunk lies in one source (c0 % 16 == 0): block-uniform select
        const bool second = (chunk * XKC >= c0);
        base = second ? a1 + (c - c0) : a0 + c;
        ld = second ? p.lda1 : p.lda0;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        const int sp = aoff[i];
        if (sp >= 0) {
          const float* src = base + (long long)sp * ld;
          v0 = *reinterpret_cast<const float4*>(src);
          v1 = *reinterpret_cast<const float4*>(src + 4);
        }
        areg[i][0] = v0;
        areg[i][1] = v1;
      }
    } else {
      if (ps) {
        float t[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          t[j] = (c + j < Cin) ? ps[c + j] : 1.f;
          t[8 + j] = (c + j < Cin) ? psh[c + j] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sreg[q] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float t[8];
        const int sp = aoff[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = c + j;
          t[j] = (sp >= 0 && cc < Cin) ? ((cc < c0) ? a0[(long long)sp * p.lda0 + cc] : a1[(long long)sp * p.lda1 + (cc - c0)])
                                      : 0.f;
        }
        areg[i][0] = make_float4(t[0], t[1], t[2], t[3]);
        areg[i][1] = make_float4(t[4], t[5], t[6], t[7]);
      }
    }
  };

  auto write_A = [&](int chunk, int buf) {
    const int c = chunk * XKC + hf * 8;
    if (LOWREG && VEC && ps && !(SC && chunk >= nch1)) {
      sreg[0] = *reinterpret_cast<const float4*>(ps + c);
      sreg[1] = *reinterpret_cast<const float4*>(ps + c + 4);
      sreg[2] = *reinterpret_cast<const float4*>(psh + c);
      sreg[3] = *reinterpret_cast<const float4*>(psh + c + 4);
    }
    const float sc[8] = {sreg[0].x, sreg[0].y, sreg[0].z, sreg[0].w, sreg[1].x, sreg[1].y, sreg[1].z, sreg[1].w};
    const float sh[8] = {sreg[2].x, sreg[2].y, sreg[2].z, sreg[2].w, sreg[3].x, sreg[3].y, sreg[3].z, sreg[3].w};
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (aoff[i] == -2) continue;
      float t[8] = {areg[i][0].x, areg[i][0].y, areg[i][0].z, areg[i][0].w,
                    areg[i][1].x, areg[i][1].y, areg[i][1].z, areg[i][1].w};
      if (aoff[i] >= 0 && !(abl & 1) && !(SC && chunk >= nch1)) {
        // block-uniform prologue mode hoisted out of the element loop (no per-element selects)
        if (ps) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = __builtin_fmaf(t[j], sc[j], sh[j]);
        }
        if (p.silu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = silu_fast(t[j]);
        }
        if (!VEC) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = (c + j < Cin) ? t[j] : 0.f;
        }
      }
      const int pix = (tid + i * NT) >> 1;
      char* dst = As + buf * A_BYTES + (hf * NPIX + pix) * 16;
      if (NP == 1) {
        *reinterpret_cast<h8*>(dst) = round8(t);
      } else {
        h8 hi, lo;
        split8(t, hi, lo);
        *reinterpret_cast<h8*>(dst) = hi;
        *reinterpret_cast<h8*>(dst + 2 * NPIX * 16) = lo;
      }
    }
  };

  // B slice of K-step `step` = [4 units][cout_pad][8 f16] in HBM -> [4][BN][16 B] in LDS, 1 KiB per wave-instruction
  auto issue_B = [&](int step, int buf) {
#pragma unroll
    for (int pc0 = 0; pc0 < NPIECE; pc0 += NW) {
      const int pc = pc0 + wave;
      if (pc < NPIECE) {
        constexpr int PPU = (BN >= 64) ? BN / 64 : 1;   // (the un-pipelined loop is not instantiated for BN = 32)
        const int u = pc / PPU, part = pc - u * PPU;
        const char* src = wpk + ((long long)(step * 4 + u) * p.cout_pad + n0 + part * 64 + lane) * 16;
        char* dst = Bs + buf * B_BYTES + (u * BN + part * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // ---- MFMA operand addressing (v_mfma_f32_32x32x16_f16: lane l holds row l&31, k = 8*(l>>5)..+7 of A and of B) ----
  const int kh = lane >> 5;
  int apix[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = (wm * TM + tm) * 32 + (RPERM ? row_perm32(lane & 31) : (lane & 31));
    if (KS == 1) {
      apix[tm] = m;
    } else {
      const int py = m / PW, px = m - py * PW;
      apix[tm] = (py * STRIDE) * TW + px * STRIDE;
    }
  }
  const int boff = (kh * BN + wn * TN * 32 + (lane & 31)) * 16;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int cb = ks_id * (nch1 / sk);                     // first chunk of this workgroup's K range (0 without split-K)
  const int nchunks = (sk > 1) ? cb + nch1 / sk : nch1 + (SC ? p.Cin2 / XKC : 0);   // one past the last chunk
  const int nsteps = (sk > 1) ? (nch1 / sk) * NTAPS : nch1 * NTAPS + (SC ? p.Cin2 / XKC : 0);

  if (PIPE) {
    constexpr int RB = T::RB, NPW = T::NPW, TS = T::TS;
    constexpr int SLOT_BYTES = TS * B_BYTES;
    // plane of the A fragments fetched one step ahead ("al"): x_lo of the three-product scheme, x_hi itself when NP == 1
    constexpr int A_PRE = (NP == 1) ? 0 : 2 * NPIX * 16;
    static_assert((TS * T::NPIECE) % NW == 0 || RB == 2,
                  "a ring deeper than 2 counts LDS-DMA instructions per wave: every wave must issue the same number per slot");
    static_assert(NTAPS % TS == 0, "a fat step must not straddle two channel chunks");
    static_assert(!SC || TS == 1, "the fused shortcut's single-tap chunks need single-tap steps");
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)Bs);
    // one ring slot = TS consecutive tap slices (they are consecutive in the packed weight image)
    auto issue_slot = [&](int fs, int slot) {
#pragma unroll
      for (int pc0 = 0; pc0 < TS * NPIECE; pc0 += NW) {
        const int pc = pc0 + wave;
        if ((TS * NPIECE) % NW != 0 && pc >= TS * NPIECE) break;   // wave-uniform
        const int tt = pc / NPIECE, q = pc - tt * NPIECE;
        int u, part, nl;
        constexpr int PPU = (BN >= 64) ? BN / 64 : 1;
        if (BN >= 64) { u = q / PPU; part = q - u * PPU; nl = part * 64 + lane; }
        else { u = q * 2 + (lane >> 5); part = 0; nl = lane & 31; }   // BN = 32: one 1-KiB piece carries two units
        const char* src = wpk + ((long long)((fs * TS + tt) * 4 + u) * p.cout_pad + n0 + nl) * 16;
        const unsigned dst = lds_base + slot * SLOT_BYTES + tt * B_BYTES +
                             ((BN >= 64) ? (u * BN + part * 64) * 16 : q * 1024);
        // inline asm: hipcc drains vmcnt(0) before every LDS-DMA it can see behind another one still in flight, which would
        // undo the counted waits below; M0 (the LDS destination base) is saved/restored inside the statement
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst))
                     : "memory");
      }
    };
    const int fs0 = cb * (NTAPS / TS);          // absolute index of this workgroup's first fat slice
    const int nfat = fs0 + nsteps / TS;          // one past its last
    // ---- prologue: first chunk + weight slot 0 staged, slots 1..RB-1 in flight, fragments of step 0 in registers ----
    issue_slot(fs0, 0);
    gload_A(cb);
    write_A(cb, cb & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 1; j < RB; ++j)
      if (fs0 + j < nfat) issue_slot(fs0 + j, j);
    if (NTAPS == 1 && cb + 1 < nchunks) gload_A(cb + 1);   // 1x1: the second chunk is written during step 0
    h8 ah[TM], al[TM], bh[TN], bl[TN];
    {
      const char* A = As + (cb & 1) * A_BYTES + kh * NPIX * 16;
      const char* B = Bs + boff;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) al[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16 + A_PRE);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bh[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16);
    }
    int chunk = cb, tap = 0, slot = 0, tt = 0, fs = fs0;   // slot = (fs - fs0) % RB, tt = step % TS
    for (int step = 0; step < nsteps; ++step) {
      const int ntaps_c = (SC && chunk >= nch1) ? 1 : NTAPS;   // shortcut chunks have the centre tap only
      const bool last_tap = (tap == ntaps_c - 1);
      const bool next_a = (chunk + 1 < nchunks);
      const bool endfat = (tt == TS - 1);
      const int tapA = (SC && chunk >= nch1) ? (NTAPS / 2) : tap;
      const int ky = tapA / KS, kx = tapA - ky * KS;
      const char* A = As + (chunk & 1) * A_BYTES + (kh * NPIX + ky * TW + kx) * 16;
      const char* B = Bs + slot * SLOT_BYTES + tt * B_BYTES + boff;
      // the step opens with matrix work on register-resident fragments; this step's x_hi / w_lo reads are issued behind
      // the first MFMA (pinned: the compiler's lgkmcnt wait for al/bh must not sit behind freshly issued reads)
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh[0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (NP == 3) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) ah[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bl[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16 + 2 * BN * 16);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          if (tm + tn > 0) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
      if (NP == 3) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
      }
      int nchunk = chunk, ntap = tap + 1;
      if (ntap == ntaps_c) { ntap = 0; ++nchunk; }
      int nslot = slot, ntt = tt + 1;
      if (endfat) {
        // next chunk's halo tile (loaded a step or more ago) -> LDS, before the barrier that publishes it
        if (last_tap && next_a) write_A(chunk + 1, (chunk + 1) & 1);
        // Slot fs+1 must have landed; the newer slots fs+2 .. fs+RB-1 (NPW LDS-DMA instructions each, fewer at the tail) stay
        // in flight across the barrier.  Activation loads issued after slot fs+1 only make the wait stricter.
        {
          const int lastq = (fs + RB - 1 < nfat - 1) ? fs + RB - 1 : nfat - 1;
          const int newer = lastq - (fs + 1);
          if (RB >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
          else if (RB >= 3 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes (halo tile) and reads of slot `fs` are complete
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (fs + RB < nfat) issue_slot(fs + RB, slot);       // slot of fat step `fs`: every wave's reads of it are complete
        nslot = (slot + 1 == RB) ? 0 : slot + 1;
        ntt = 0;
      }
      // activation loads for the chunk after step+1's: issued a full step (3x3: eight steps) before their staging pass
      {
        const int trig = (NTAPS > 1 && !(SC && nchunk >= nch1)) ? 1 : 0;
        if (ntap == trig && nchunk + 1 < nchunks) gload_A(nchunk + 1);
      }
      if (step + 1 < nsteps) {
        const int ntapA = (SC && nchunk >= nch1) ? (NTAPS / 2) : ntap;
        const int nky = ntapA / KS, nkx = ntapA - nky * KS;
        const char* An = As + (nchunk & 1) * A_BYTES + (kh * NPIX + nky * TW + nkx) * 16;
        const char* Bn = Bs + nslot * SLOT_BYTES + ntt * B_BYTES + boff;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) al[tm] = *reinterpret_cast<const h8*>(An + apix[tm] * 16 + A_PRE);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bh[tn] = *reinterpret_cast<const h8*>(Bn + tn * 32 * 16);
      }
      if (NP == 3) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
      }
      tap = ntap;
      chunk = nchunk;
      if (endfat) ++fs;
      slot = nslot;
      tt = ntt;
    }
  } else {
    // ---- prologue: stage chunk 0 and the first weight slice ----
    issue_B(0, 0);
    gload_A(0);
    write_A(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int chunk = 0, tap = 0;
    for (int step = 0; step < nsteps; ++step) {
      const bool more = (step + 1 < nsteps);
      const bool scph = SC && chunk >= nch1;            // fused-shortcut phase: one (centre-tap) step per chunk
      const int ntaps_c = scph ? 1 : NTAPS;
      const bool first_tap = (tap == 0), last_tap = (tap == ntaps_c - 1);
      const bool next_a = (chunk + 1 < nchunks);
      if (more && !(abl & 2)) issue_B(step + 1, (step + 1) & 1);
      if (!LOWREG && first_tap && next_a && !(abl & 8)) gload_A(chunk + 1);
      // 8-wave tile with the fused shortcut: the next chunk's loads go out ahead of this (last) tap's MFMA passes and are
      // staged behind them (16 live registers for one tap) -- in the shortcut phase every step is a chunk's last tap.
      // Measured as the ratio fused / plain launch time of one edit step (rocprofv3, same box): 0.880 this way, 0.891
      // with the loads right before the staging pass, 0.938 issued a whole step ahead, 0.906-0.910 on the 4-wave
      // pipelined tile (profiles/r01n_kernel_stats_*.csv)
      if (LOWREG && SC && last_tap && next_a && !(abl & 8)) gload_A(chunk + 1);

      {
        const int tapA = scph ? NTAPS / 2 : tap;
        const int ky = tapA / KS, kx = tapA - ky * KS;
        const char* A = As + (chunk & 1) * A_BYTES + (kh * NPIX + ky * TW + kx) * 16;
        const char* B = Bs + (step & 1) * B_BYTES + boff;
        // pass order (x_lo*w_hi, x_hi*w_hi, x_hi*w_lo) is the same in every variant: results are bit-identical across tiles
        if (NP == 1) {   // single product x_hi * w_hi
          h8 fa[TM], fb[TN];
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16);
  #pragma unroll
          for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16);
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
        } else if (LOWREG) {
          h8 fa[TM], fb[TN];
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16 + 2 * NPIX * 16);   // x_lo
  #pragma unroll
          for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16);                    // w_hi
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16);                   // x_hi
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
  #pragma unroll
          for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16 + 2 * BN * 16);      // w_lo
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
        } else {
          h8 ah[TM], al[TM], bh[TN], bl[TN];
  #pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            ah[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16);
            al[tm] = *reinterpret_cast<const h8*>(A + apix[tm] * 16 + 2 * NPIX * 16);
          }
  #pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            bh[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16);
            bl[tn] = *reinterpret_cast<const h8*>(B + tn * 32 * 16 + 2 * BN * 16);
          }
          if (ABL && (abl & 4)) {   // keep the fragment reads alive, drop the matrix work
  #pragma unroll
            for (int tm = 0; tm < TM; ++tm) asm volatile("" ::"v"(ah[tm]), "v"(al[tm]));
  #pragma unroll
            for (int tn = 0; tn < TN; ++tn) asm volatile("" ::"v"(bh[tn]), "v"(bl[tn]));
          } else {
  #pragma unroll
            for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
              for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
  #pragma unroll
            for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
              for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
  #pragma unroll
            for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
              for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
          }
        }
      }

      if (last_tap && next_a && !(abl & 8)) {
        if (LOWREG && !SC) gload_A(chunk + 1);
        write_A(chunk + 1, (chunk + 1) & 1);
      }
      if (!(abl & 16)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (++tap == ntaps_c) { tap = 0; ++chunk; }
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
  const float* __restrict__ rz = p.resid ? p.resid + (long long)zo * p.r_zo : nullptr;
  const float* __restrict__ cadd = p.chan_add ? p.chan_add + (long long)zo * p.ld_chan_add : nullptr;
  const bool has_b = (p.bias != nullptr), has_c = (cadd != nullptr);
  bool full = (n0 + BN <= Cout);
  if (KS == 1) full = full && (m0 + BM <= HWo);
  else full = full && (oy0 + T::PH <= p.Hout) && (ox0 + PW <= p.Wout);
  // GroupNorm statistics of the tensor being written (sum, sum of squares per output channel over this workgroup's
  // pixels), for the NEXT layer's normalisation: accumulated in double, reduced in a fixed order (deterministic,
  // batch-invariant), written as one partial per (image, M-block, channel).  LDS is free after the last K-step barrier.
  double* const red = reinterpret_cast<double*>(smem);   // [WM][BN][2]
  const bool want_stats = (p.stats != nullptr);
  auto stat_commit = [&](int tn, double s1, double s2) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (kh == 0) {
      double* d = red + ((size_t)wm * BN + (wn * TN + tn) * 32 + (lane & 31)) * 2;
      d[0] = s1;
      d[1] = s2;
    }
  };
  if (KS == 1 && p.o16h) {
    // split-plane output (the q|k|v projection of an attention block, GemmArgs::o16h): every value leaves as an exact two-term
    // f16 split; q and k rows into [pixel][ld16] planes, the v channels transposed into [channel][pixel] planes, both in the
    // fragment-major tile order of kernels.h frag_off (pixels % 16 == 0, ld16 % 32 == 0: the launcher of the planes path).  Accumulator
    // elements 4j .. 4j+3 of a lane are 4 consecutive pixels of one channel: one 8-byte store per plane in the transposed part.
    _Float16* __restrict__ oh = p.o16h + (long long)zo * p.o16_zo;
    // (the lo planes are absent in the single-product mode: no pointer arithmetic on null -- `if (ol)` below must see a real null
    //  for every image, not null + zo * stride)
    _Float16* __restrict__ ol = p.o16l ? p.o16l + (long long)zo * p.o16_zo : nullptr;
    _Float16* __restrict__ vh = p.vth + (long long)zo * p.vt_zo;
    _Float16* __restrict__ vl = p.vtl ? p.vtl + (long long)zo * p.vt_zo : nullptr;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + (lane & 31);
      if (n >= Cout) continue;
      const float add = (has_b ? p.bias[n] : 0.f) + (has_c ? cadd[n] : 0.f);
      const int nm = n % p.v_mod;
      const bool isv = nm >= p.v_off;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pix = m0 + (wm * TM + tm) * 32 + 8 * j + 4 * kh;
          typedef _Float16 h4v __attribute__((ext_vector_type(4)));
          h4v hi, lo;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = __builtin_amdgcn_fmed3f(acc[tm][tn][4 * j + q] * p.alpha + add, -H_MAX, H_MAX);
            const _Float16 hh = (_Float16)v;
            hi[q] = hh;
            lo[q] = (_Float16)(v - (float)hh);
          }
          if (isv) {
            // v^T plane, fragment-major (kernels.h frag_off): 4 consecutive tokens of one channel = 4 consecutive halfs
            const int vc = (n / p.v_mod) * p.v_dh + nm - p.v_off;
            if (pix + 3 < HWo) {
              const long long o = frag_off(vc, pix, HWo);
              *reinterpret_cast<h4v*>(vh + o) = hi;
              if (vl) *reinterpret_cast<h4v*>(vl + o) = lo;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (pix + q < HWo) { const long long o = frag_off(vc, pix + q, HWo); vh[o] = hi[q]; if (vl) vl[o] = lo[q]; }
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (pix + q < HWo) {
                const long long o = frag_off(pix + q, n, p.ld16);
                oh[o] = hi[q];
                if (ol) ol[o] = lo[q];
              }
          }
        }
      }
    }
    return;
  }
  if (sk > 1) {   // split-K: raw partial sums; bias / residual / statistics belong to launch_splitk_reduce
    float* __restrict__ part = p.part + ((size_t)(ks_id * (int)gridDim.z + zo) * HWo) * Cout;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + (lane & 31);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ri = (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int m = (wm * TM + tm) * 32 + (RPERM ? row_perm32(ri) : ri);
          int pixel;
          bool ok;
          if (KS == 1) {
            pixel = m0 + m;
            ok = pixel < HWo;
          } else {
            const int oy = oy0 + m / PW, ox = ox0 + (m % PW);
            ok = (oy < p.Hout) && (ox < p.Wout);
            pixel = oy * p.Wout + ox;
          }
          if (ok && n < Cout) part[pixel * Cout + n] = acc[tm][tn][r] * p.alpha;
        }
      }
    }
    return;
  }
  if (full) {
    // interior tile: no bounds checks
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + (lane & 31);
      const float add = (has_b ? p.bias[n] : 0.f) + (has_c ? cadd[n] : 0.f);
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        // 8 rows at a time (8 residual loads in flight, then 8 stores); 32-bit element offsets from the per-image
        // base (an image is < 2^31 elements): saddr + voffset addressing, one VGPR per address
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          int pixel[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = half * 8 + q;
            const int ri = (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int m = (wm * TM + tm) * 32 + (RPERM ? row_perm32(ri) : ri);
            pixel[q] = (KS == 1) ? (m0 + m) : ((oy0 + m / PW) * p.Wout + ox0 + (m % PW));
          }
          float rv[8];
          if (rz && p.rups) {   // residual lives at half resolution (nearest x2 of the skip path)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int oy = pixel[q] / p.Wout, ox = pixel[q] - oy * p.Wout;
              rv[q] = rz[((oy >> 1) * (p.Wout >> 1) + (ox >> 1)) * p.ldr + n];
            }
          } else if (rz) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rv[q] = rz[pixel[q] * p.ldr + n];
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) rv[q] = 0.f;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float v = (acc[tm][tn][half * 8 + q] * p.alpha + add) + rv[q];
            outz[pixel[q] * p.ldo + n] = v;
            if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
          }
        }
      }
      if (want_stats) stat_commit(tn, s1, s2);
    }
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + (lane & 31);
      const bool nok = (n < Cout);
      const float add = nok ? ((has_b ? p.bias[n] : 0.f) + (has_c ? cadd[n] : 0.f)) : 0.f;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ri = (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int m = (wm * TM + tm) * 32 + (RPERM ? row_perm32(ri) : ri);
          int pixel;
          bool ok;
          if (KS == 1) {
            pixel = m0 + m;
            ok = pixel < HWo;
          } else {
            const int oy = oy0 + m / PW, ox = ox0 + (m % PW);
            ok = (oy < p.Hout) && (ox < p.Wout);
            pixel = oy * p.Wout + ox;
          }
          if (ok && nok) {
            float rv = 0.f;
            if (rz) {
              const int oy = pixel / p.Wout, ox = pixel - oy * p.Wout;
              rv = p.rups ? rz[((oy >> 1) * (p.Wout >> 1) + (ox >> 1)) * p.ldr + n] : rz[pixel * p.ldr + n];
            }
            const float v = (acc[tm][tn][r] * p.alpha + add) + rv;
            outz[pixel * p.ldo + n] = v;
            if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
          }
        }
      }
      if (want_stats) stat_commit(tn, s1, s2);
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      if (n0 + c < Cout) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < T::WM; ++w) {
          s1 += red[((size_t)w * BN + c) * 2];
          s2 += red[((size_t)w * BN + c) * 2 + 1];
        }
        double* dst = p.stats + (((size_t)zo * gridDim.x + bx) * Cout + n0 + c) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// The 3x3 layers on v_mfma_f32_16x16x32_f16 ("K32"): the 256 x 128 main tile and its 128-pixel / 8x8-patch / stride-2 forms.
// The chip is power-limited under the f16 matrix instructions on non-zero data, so the sustained rate of an instruction stream
// is set by the ENERGY an instruction costs, not by its issue rate: scripts/calib/mfma_energy.hip measures 1.62 PFLOP/s for
// v_mfma_f32_32x32x16_f16 against 1.89 PFLOP/s for v_mfma_f32_16x16x32_f16 on the same UNet-like operands (the 16x16x32 form
// reduces 32 products into each accumulator element, i.e. moves half the fp32 accumulator bytes per flop), and
// scripts/calib/loop_shapes.hip 482 -> 546 TFLOP/s-equivalent for this tile's loop shape (profiles/r02s_*).
// Same workgroup shape, LDS images, weight image, staging and epilogue semantics as igemm_f16x3_kernel<XCfg<4,2,2,2,3,1>>; what
// changes is the K-step: ONE instruction reduces K = 32 = two consecutive (chunk, tap) slices of the flat K sequence
// (chunk-major, 9 taps per 16-channel chunk).  Lane l of a wave feeds row l & 15 and the 8-element k group l >> 4; groups
// 0,1 are the two channel halves of the step's first tap, groups 2,3 those of its second tap -- every lane computes its own
// LDS address, so the tap (and, across a chunk boundary, the halo buffer) is just a per-lane offset.  Cin % 32 == 0 makes the
// number of slices even.  A wave owns 64 pixels (4 tile rows of 16) x 64 channels = 4 x 4 accumulator blocks of 16 x 16
// (64 VGPRs): per step 16 ds_read_b128 feed 48 instructions.  The unit planes of the halo tile are pitched at 336 pixels
// (21 x 256 B) so that the two channel-half planes a ds_read_b128 lane group touches are bank-congruent.
// Summation order inside a K = 32 instruction differs from two K = 16 ones: results equal the other tiles' to fp32 rounding
// (tested at 1e-6 relative), not bitwise; the tile choice depends on the layer shape only, so batch invariance holds.
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NW waves as WM x WN; a wave owns 64 pixels (4 tile rows of 16) x 128/WN channels = 4 x TN accumulator blocks.
//   <8, 2>: 256 x 128, the main tile (16 x 16 patch; 74 KB of LDS, two workgroups per CU, 4 waves per SIMD)
//   <8, 4>: 128 x 128 for the 16 x 16-pixel layers (8 x 16 patch, wave = 64 pixels x 32 channels; 56 KB)
//   <8, 8, 8>: 64 x 128 for the 8 x 8-pixel layers (8 x 8 patch = one image, a 16-row fragment = two patch rows, wave = 64
//              pixels x 16 channels; 46 KB: three workgroups per CU)
//   <8, 8, 16, 2>: 64 x 128 at stride 2 (4 x 16 output patch, 9 x 33 halo, wave = 64 pixels x 16 channels; DDPM Downsample)
//   <8, 2, 16, 1, 2>: the POLYPHASE form of the main tile for "nearest x2, then 3x3" (Upsample.conv, models/ddpm/diffusion.py:84-87;
//              ResBlock(up=True).in_layers.2, improved_ddpm/unet.py:281-284).  Output pixel (2i+py, 2j+px) of that convolution
//              sees only the 2 x 2 source pixels (i+py-1 .. i+py, j+px-1 .. j+px): the nine taps collapse, per output phase
//              (py, px), into four taps whose weights are sums of the original ones (built on the host at parameter upload).
//              One workgroup computes a 16 x 16 patch of SOURCE positions for one phase (17 x 17 halo, 4 taps per chunk) and
//              scatters to the stride-2 output grid: 4/9 of the matrix work of the 3x3 form, no duplicated halo staging.
// Measured and NOT kept for the small forms (8 x 8 patches, the 128-pixel form, stride 2: 12-24 matrix instructions per wave and
// K-step, 0.69 us per step on the 8 x 8 layers for 0.08 us of matrix work), both interleaved on one box at B=32:
//   * a deeper LDS-DMA weight ring (3-4 slots, inline-asm DMA, hand-counted vmcnt): 8-10 % SLOWER (profiles/rd3d_*_NEGATIVE.txt);
//   * the weight slices through registers (global_load_dwordx4 -> ds_write_b128, two sets, two steps ahead): 30-45 % SLOWER
//     (profiles/rd3e_*_NEGATIVE.txt).
// Neither the latency nor the rate of the LDS-DMA fill is what these steps wait for; the two-slot LDS-DMA loop stays.
//   <8, 2, 16, 1, 3, true>: the QUAD form for the 8 x 8-pixel layers.  The 64-pixel form gives one workgroup 12 matrix
//              instructions per wave and K-step and re-streams a 2.4 MB weight slab per image: 0.69 us per step for 0.08 us of
//              matrix work, 81 TFLOP/s, 128 workgroups on 256 CUs at B = 32 (and neither a deeper weight ring nor weights
//              through registers helped, see above).  Here one workgroup takes FOUR images as a 2 x 2 arrangement of 8 x 8 patches
//              -- the main tile's shape: 256 pixels x 128 channels, 48 matrix instructions per wave and step, one weight slice
//              feeds four images -- each image with its own zero border in the 20 x 20 halo tile, and the K range is split over
//              gridDim.y (GemmArgs.sk): every workgroup writes alpha * acc of its range to `part`, launch_splitk_reduce adds the
//              ranges in a fixed order with bias / residual / statistics.  Per image the products and their order are those
//              of any other K32 form over the same K range, and an image's partial sums do not depend on which images share its
//              workgroup: batch invariance stays bitwise.
template <int NW_, int WN_, int PW_ = 16, int STRIDE_ = 1, int KS_ = 3, bool QUAD_ = false>
struct K32Cfg {
  static constexpr int NW = NW_, WN = WN_, WM = NW / WN, NT = NW * 64, TN = 8 / WN, STRIDE = STRIDE_;
  static constexpr int KS = KS_, NTAPS = KS * KS;              // 3 x 3 taps, or the 2 x 2 taps of one output phase
  static constexpr bool QUAD = QUAD_;
  static constexpr int PW = PW_, FR = 16 / PW;                 // FR patch rows per 16-row fragment
  static constexpr int BM = WM * 64, BN = 128, PH = BM / PW;
  static constexpr int TW = QUAD ? 20 : (PW - 1) * STRIDE + KS, TH = QUAD ? 20 : (PH - 1) * STRIDE + KS;   // QUAD: 2 x (8 + 2) per axis
  static constexpr int NPIX = TH * TW;                         // halo pixels (324 for the main tile)
  static constexpr int PLANE = (NPIX + 15) / 16 * 16;          // unit-plane pitch in pixels (multiple of 16: 256-B congruent)
  static constexpr int A_BYTES = 4 * PLANE * 16;               // [4 units][PLANE][16 B]
  static constexpr int B_BYTES = BN * 64;                      // one (chunk, tap) weight slice [4 units][BN][16 B]
  static constexpr int SLOT_BYTES = 2 * B_BYTES;               // the two slices of a K = 32 step
  static constexpr int NU = NPIX * 2, NA = (NU + NT - 1) / NT;
  static constexpr int NSC = (BM * 4 + NT - 1) / NT;           // shortcut-phase work items per thread
  static constexpr int NPW = 16 / NW;                          // LDS-DMA pieces per wave and step
  static constexpr size_t SMEM = 2 * (size_t)SLOT_BYTES + 2 * (size_t)A_BYTES;
  static constexpr int MINW = (2 * NW) / 4;                    // two workgroups per CU
  static_assert(8 * BM * 16 <= 2 * A_BYTES, "the shortcut phase's 32-channel centre tile lives in the two halo buffers");
  static_assert(NA <= 2 && NSC <= 2 && NSC <= NA, "staging registers");
  static_assert(PW == 16 || PW == 8, "a fragment is one or two patch rows");
  static_assert(KS == 3 || (KS == 2 && STRIDE == 1), "polyphase form: 2 x 2 taps at stride 1");
  static_assert(!QUAD || (PW == 16 && STRIDE == 1 && KS == 3 && WM == 4), "quad form: 16 x 16 virtual patch of four 8 x 8 images");
};

// SC: fused 1x1 shortcut.  After the 3x3 slices the flat K sequence continues with Cin2/16 single-tap slices over the raw
//     tensor (s0|s1) (weights appended to the image, as for igemm_f16x3_kernel); two of them make a K = 32 step.  Such a step
//     needs BOTH its 16-channel chunks in LDS at once, so the shortcut phase turns the two halo buffers into ONE centre-only tile
//     of 32 channels ([8 units][BM pixels][16 B], 32 KB on the main tile): raw loads for the next step travel in registers under the matrix
//     passes, the split + LDS write sits between two barriers (Cin2 % 32 == 0).
// ABL: profiling-only instantiation, compiled with -DASYRP_BENCH_HOOKS into libasyrp_hip_bench.so for scripts/conv_bench.py
//      and absent from the product library: p.abl switches phases off at run time -- 2 = no weight LDS-DMA in the loop, 4 = no
//      matrix instructions (fragment reads kept), 8 = no activation loads / staging in the loop (results are then wrong)
// NP:  matrix products per term.  3 = the fp32-equivalent two-term split (x_lo*w_hi + x_hi*w_hi + x_hi*w_lo); 1 = the single-
//      product f16 mode (conv_math "f16": x_hi*w_hi only, fp32 accumulate): the lo planes are neither computed, staged,
//      DMA'd nor read -- one third of the matrix work, half the weight bytes, a lighter staging pass.
// SPK: split-K on the plain forms (round 4): gridDim.y = N blocks * p.sk, every workgroup runs the chunks [cb, ce) of its range and
//      writes alpha * acc to p.part[range][image][pixel][Cout]; launch_splitk_reduce adds the ranges in a fixed order with bias /
//      timestep vector / residual and emits the GroupNorm partials.  For layers whose M x N offers fewer workgroups than the chip
//      has slots (16 x 16 maps at B = 32: 256 workgroups of the 128-pixel form, one per CU).  No fused shortcut, no polyphase.
// phase stamps (profiling library, ABL instantiation only): thread 0 of every workgroup records s_memrealtime (100 MHz) into
// GemmArgs.dbg [workgroup][8]: 0 start, 1 first tile staged, 2 K loop done, 3 epilogue stores issued, 4 end, 5 = XCC_ID << 32 | HW_ID
#define K32_STAMP(i) do { if (ABL && p.dbg && threadIdx.x == 0) p.dbg[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
template <class T, bool SC, bool ABL = false, int NP = 3, bool SPK = false>
__global__ void __launch_bounds__(T::NT, T::MINW) igemm_f16x3_k32_kernel(const GemmArgs p) {
  const int abl = ABL ? p.abl : 0;
  constexpr int NT = T::NT, BN = T::BN, PW = T::PW, TW = T::TW, PLANE = T::PLANE, WN = T::WN, WM = T::WM, TN = T::TN, BM = T::BM;
  constexpr int A_BYTES = T::A_BYTES, B_BYTES = T::B_BYTES, SLOT_BYTES = T::SLOT_BYTES, NA = T::NA, NU = T::NU, NSC = T::NSC;
  constexpr int WCH = BN / WN;                 // output channels per wave
  constexpr int FR = T::FR, STRIDE = T::STRIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- stagger experiment (profiling library only, GemmArgs.stag == 3; round 5): the second workgroup to arrive on a CU waits
  // stag_ticks once.  Measured (profiles/r05a_k32_stagger_stamps.txt): the two workgroups of a CU already run half a life apart
  // without it (start offset / life: p10 0.46, p50 0.50, p90 0.54 -- the first dispatch round starts in lockstep and the pair drifts
  // into anti-phase by itself), and forcing 30 / 45 / 60 us changes the launch time by +-1 %: there is nothing to collect here.
  if (ABL && p.stag == 3 && p.dbg) {
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int* flag = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) {   // arrival order on this CU (counters behind the stamps, zeroed by the bench hook before every launch)
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      const unsigned key = ((((xcc & 0xF) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)) & 2047;
      const size_t nwg = (size_t)gridDim.x * gridDim.y * gridDim.z;
      const unsigned long long arr = atomicAdd(p.dbg + nwg * 8 + key, 1ULL);
      p.dbg[(size_t)lin * 8 + 7] = arr;
      *flag = (arr == 1);
    }
    __syncthreads();
    const bool late = (*flag != 0);
    __syncthreads();
    if (late) {
      const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (unsigned long long)p.stag_ticks;
      while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(16);
    }
  }
  K32_STAMP(0);   // (after the stagger wait: the phase durations below are those of the tile itself)
  if (ABL && p.dbg && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.dbg[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
  }
  char* const Bs = smem;                       // LDS-DMA destinations first (M0 base below 64 KB)
  char* const As = smem + 2 * SLOT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  // polyphase form: blockIdx.z = image * 4 + output phase (py, px); Hout x Wout are the SOURCE dimensions (the M space)
  constexpr bool POLY = (T::KS == 2);
  constexpr bool QUAD = T::QUAD;
  constexpr int NTAPS = T::NTAPS, KW = T::KS;
  static_assert(!(POLY && SC), "the polyphase form has no fused shortcut");
  static_assert(!(QUAD && (SC || ABL)), "the quad form has no fused shortcut");
  static_assert(!(SPK && (SC || ABL || POLY || QUAD)), "split-K: plain 3x3 forms only");
  constexpr bool SPLIT = QUAD || SPK;
  // quad form: blockIdx.z = group of four images (zo = its first image), blockIdx.y = N block * sk + K range
  const int zo = POLY ? (int)blockIdx.z >> 2 : (QUAD ? (int)blockIdx.z * 4 : (int)blockIdx.z);
  const int phy = POLY ? ((int)blockIdx.z >> 1) & 1 : 0, phx = POLY ? (int)blockIdx.z & 1 : 0;
  const int sk = SPLIT ? p.sk : 1, ks_id = SPLIT ? (int)blockIdx.y % sk : 0;
  const int n0 = (SPLIT ? (int)blockIdx.y / sk : (int)blockIdx.y) * BN;
  int bx = blockIdx.x;
  if (p.xmap) bx = (bx & 7) * ((int)gridDim.x >> 3) + (bx >> 3);   // XCD-aware block -> tile map (see igemm_f16x3_kernel)
  const int tiles_x = (p.Wout + PW - 1) / PW;
  const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
  const int oy0 = ty * T::PH, ox0 = tx * PW;
  const float* __restrict__ a0 = p.a0 + (long long)zo * p.a0_zo;
  const float* __restrict__ a1 = p.a1 ? p.a1 + (long long)zo * p.a1_zo : nullptr;
  const int ldps = p.ld_ps ? p.ld_ps : p.Cin;
  const float* __restrict__ ps = p.pscale ? p.pscale + (long long)zo * ldps : nullptr;
  const float* __restrict__ psh = p.pshift ? p.pshift + (long long)zo * ldps : nullptr;
  const char* __restrict__ wpk = reinterpret_cast<const char*>(p.wpk) + (POLY ? (long long)(phy * 2 + phx) * p.w_phase : 0);
  const int Cout = p.Cout, c0s = p.c0;
  const int nch = p.Cin / XKC;               // even (launcher: Cin % 32 == 0)

  // ---- A staging map: work item = (halo pixel, 8-channel half), as in igemm_f16x3_kernel ----
  const int hf = tid & 1;
  int aoff[NA];   // source pixel index, -1 = zero padding, -2 = no work item
  int aimg[NA];   // quad form: the item's image within the group (its scale/shift row)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tid + i * NT;
    const int pix = u >> 1;
    int off = -2;
    aimg[i] = 0;
    if (QUAD) {
      if (u < NU) {
        // halo pixel (iy, ix) of the 20 x 20 tile = image (iy / 10, ix / 10) of the group, its pixel (iy % 10 - 1, ix % 10 - 1);
        // per-image tensors are dense (a_zo == 64 * lda: launcher), so image q's pixel l sits at row q * 64 + l of the group
        const int iy = pix / TW, ix = pix - iy * TW;
        const int qy = iy / 10, ly = iy - qy * 10 - 1, qx = ix / 10, lx = ix - qx * 10 - 1;
        const int q = qy * 2 + qx;
        aimg[i] = q;
        off = (ly >= 0 && ly < 8 && lx >= 0 && lx < 8 && zo + q < p.Z) ? q * 64 + ly * 8 + lx : -1;
      }
    } else if (u < NU) {
      const int iy = pix / TW, ix = pix - iy * TW;
      // polyphase: phase 0 reads source rows (i-1, i), phase 1 rows (i, i+1): the halo origin moves with the phase
      const int gy = oy0 * STRIDE - (POLY ? 1 - phy : p.pad) + iy, gx = ox0 * STRIDE - (POLY ? 1 - phx : p.pad) + ix;
      const int Hu = p.Hin << p.ups, Wu = p.Win << p.ups;
      off = (gy >= 0 && gy < Hu && gx >= 0 && gx < Wu) ? ((gy >> p.ups) * p.Win + (gx >> p.ups)) : -1;
    }
    aoff[i] = off;
  }
  float4 areg[NA][2];
  auto gload_A = [&](int chunk) {
    const int c = chunk * XKC + hf * 8;
    const bool second = (chunk * XKC >= c0s);                 // a chunk lies in one source (c0 % 16 == 0)
    const float* __restrict__ base = second ? a1 + (c - c0s) : a0 + c;
    const int ld = second ? p.lda1 : p.lda0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      int sp = aoff[i];
      asm volatile("" : "+v"(sp));   // (laundered: the 64-bit row offsets of both sources are otherwise hoisted out of the K loop and
                                     //  held across the matrix passes, where every register counts)
      if (sp >= 0) {
        const float* src = base + (long long)sp * ld;
        v0 = *reinterpret_cast<const float4*>(src);
        v1 = *reinterpret_cast<const float4*>(src + 4);
      }
      areg[i][0] = v0;
      areg[i][1] = v1;
    }
  };
  auto write_A = [&](int chunk, int buf) {
    const int c = chunk * XKC + hf * 8;
    float4 sreg[4];
    if (ps && !QUAD) {
      sreg[0] = *reinterpret_cast<const float4*>(ps + c);
      sreg[1] = *reinterpret_cast<const float4*>(ps + c + 4);
      sreg[2] = *reinterpret_cast<const float4*>(psh + c);
      sreg[3] = *reinterpret_cast<const float4*>(psh + c + 4);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (aoff[i] == -2) continue;
      if (ps && QUAD && aoff[i] >= 0) {   // every halo pixel belongs to one of the group's four images: its own scale / shift row
                                          // (border pixels and the images a ragged last group lacks are zeros: no row to read)
        const long long ro = (long long)aimg[i] * ldps + c;
        sreg[0] = *reinterpret_cast<const float4*>(ps + ro);
        sreg[1] = *reinterpret_cast<const float4*>(ps + ro + 4);
        sreg[2] = *reinterpret_cast<const float4*>(psh + ro);
        sreg[3] = *reinterpret_cast<const float4*>(psh + ro + 4);
      }
      const float sc[8] = {sreg[0].x, sreg[0].y, sreg[0].z, sreg[0].w, sreg[1].x, sreg[1].y, sreg[1].z, sreg[1].w};
      const float sh[8] = {sreg[2].x, sreg[2].y, sreg[2].z, sreg[2].w, sreg[3].x, sreg[3].y, sreg[3].z, sreg[3].w};
      float t[8] = {areg[i][0].x, areg[i][0].y, areg[i][0].z, areg[i][0].w,
                    areg[i][1].x, areg[i][1].y, areg[i][1].z, areg[i][1].w};
      if (aoff[i] >= 0) {
        if (ps) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = __builtin_fmaf(t[j], sc[j], sh[j]);
        }
        if (p.silu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = silu_fast(t[j]);
        }
      }
      int t_ = tid;
      asm volatile("" : "+v"(t_));   // (laundered, as above)
      const int pix = (t_ + i * NT) >> 1;
      char* dst = As + buf * A_BYTES + ((t_ & 1) * PLANE + pix) * 16;
      if (NP == 1) {
        *reinterpret_cast<h8*>(dst) = round8(t);
      } else {
        h8 hi, lo;
        split8(t, hi, lo);
        *reinterpret_cast<h8*>(dst) = hi;
        *reinterpret_cast<h8*>(dst + 2 * PLANE * 16) = lo;
      }
    }
  };
  // the two weight slices of K-step s are consecutive in the packed image: 16 1-KiB LDS-DMA pieces, 16 / NW per wave
  // (NP == 1: only the 8 pieces of the hi units, one per wave)
  auto issue_slot = [&](int s, int slot) {
    int l_ = lane;
    asm volatile("" : "+v"(l_));   // (laundered: the per-lane 64-bit base is otherwise hoisted and, in the fused-shortcut instantiation,
                                   //  spilled and reloaded from scratch in every K step)
#pragma unroll
    for (int k = 0; k < (NP == 1 ? (8 + T::NW - 1) / T::NW : T::NPW); ++k) {
      const int pc = wave + k * T::NW;
      if (NP == 1 && T::NW > 8 && pc >= 8) break;
      const int i = (NP == 1) ? pc >> 2 : pc >> 3, u = (NP == 1) ? (pc >> 1) & 1 : (pc & 7) >> 1, part = pc & 1;
      const char* src = wpk + ((long long)((2 * s + i) * 4 + u) * p.cout_pad + n0 + part * 64 + l_) * 16;
      char* dst = Bs + slot * SLOT_BYTES + i * B_BYTES + (u * BN + part * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // ---- operand addressing: lane = (row r16, k group kq = 2 * tap-of-the-step + channel half) ----
  const int r16 = lane & 15, kq = lane >> 4, tp = kq >> 1, kh = kq & 1;
  // row r16 of row block tm = patch pixel ((wm * 4 + tm) * FR + r16 / PW, r16 % PW)
  // (quad form: patch row py = 4 wm + tm lies in image row block py >> 3, column r16 in block r16 >> 3; each block has its own
  //  one-pixel border, i.e. two extra halo rows / columns in front of the second block)
  const int a_lane = QUAD ? (kh * PLANE + (wm * 4 + 2 * (wm >> 1)) * TW + r16 + 2 * (r16 >> 3)) * 16
                          : (kh * PLANE + ((wm * 4 * FR + r16 / PW) * STRIDE) * TW + (r16 % PW) * STRIDE) * 16;
  constexpr int A_TM = FR * STRIDE * TW * 16;  // byte pitch between row blocks (+ 2 * PLANE * 16 for x_lo, + the tap offset)
  const int b_lane = tp * B_BYTES + (kh * BN + wn * WCH + r16) * 16;      // + tn * 256 (+ 2 * BN * 16 for w_lo) + slot

  f32x4 acc[4][TN];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tm][tn][r] = 0.f;

  // split-K (quad form, SPK): this workgroup's chunks [cb, ce), an even number of them (launcher), i.e. whole K = 32 steps
  const int cb = SPLIT ? ks_id * (nch / sk) : 0, ce = SPLIT ? cb + nch / sk : nch;
  const int s_first = cb * NTAPS / 2;
  issue_slot(s_first, s_first & 1);
  gload_A(cb);
  write_A(cb, cb & 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  K32_STAMP(1);

  // one K = 32 step: 2 * (4 + TN) fragments, 12 * TN matrix instructions (16 and 48 on the main tile); pass order (x_lo*w_hi, x_hi*w_hi, x_hi*w_lo) as in every other tile.
  // A: lane address of row block 0 (x_hi); a_tm: byte pitch between row blocks; a_lo: byte offset of the x_lo planes
  auto mma_step = [&](const char* A, const int a_tm, const int a_lo, const char* B) {
    h8 fa[4], fb[TN];
    if (ABL && (abl & 4)) {   // all 16 fragment reads, no matrix work
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + tm * a_tm + hl * a_lo);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 256 + hl * 2 * BN * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]), "v"(fb[i % TN]));
      }
      return;
    }
    if (NP == 1) {   // single product: x_hi * w_hi
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + tm * a_tm);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 256);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + tm * a_tm + a_lo);                // x_lo
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 256);                         // w_hi
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + tm * a_tm);                       // x_hi
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 256 + 2 * BN * 16);           // w_lo
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
  };

  // ---- shortcut phase staging: work item = (centre pixel, 8-channel quarter q of the step's 32 raw channels) ----
  int sc_tid = tid;   // laundered where the phase starts (see the epilogue): its lane constants must not be hoisted above the K loop
  int scoff[NSC];
  auto sc_load = [&](int j) {
    const int q = sc_tid & 3;
    const int chunk = 2 * j + (q >> 1);
    const bool second = (chunk * XKC >= p.sc0);
    const int cc = chunk * XKC + (q & 1) * 8;
    const float* __restrict__ base = second ? p.s1 + (long long)zo * p.s1_zo + (cc - p.sc0) : p.s0 + (long long)zo * p.s0_zo + cc;
    const int ld = second ? p.lds1 : p.lds0;
#pragma unroll
    for (int i = 0; i < NSC; ++i) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      const int sp = scoff[i];
      if (sp >= 0) {   // (-1: outside the image, or no work item)
        const float* src = base + (long long)sp * ld;
        v0 = *reinterpret_cast<const float4*>(src);
        v1 = *reinterpret_cast<const float4*>(src + 4);
      }
      areg[i][0] = v0;
      areg[i][1] = v1;
    }
  };
  auto sc_write = [&]() {
    const int q = sc_tid & 3;
#pragma unroll
    for (int i = 0; i < NSC; ++i) {
      const float t[8] = {areg[i][0].x, areg[i][0].y, areg[i][0].z, areg[i][0].w,
                          areg[i][1].x, areg[i][1].y, areg[i][1].z, areg[i][1].w};
      const int pix = (sc_tid + i * NT) >> 2;
      if (pix >= BM) continue;
      char* dst = As + (q * BM + pix) * 16;
      if (NP == 1) {
        *reinterpret_cast<h8*>(dst) = round8(t);
      } else {
        h8 hi, lo;
        split8(t, hi, lo);
        *reinterpret_cast<h8*>(dst) = hi;
        *reinterpret_cast<h8*>(dst + 4 * BM * 16) = lo;
      }
    }
  };

  const int nsteps3 = ce * NTAPS / 2;            // one past this workgroup's last 3x3 step (absolute step index)
  const int nsc = SC ? p.Cin2 / (2 * XKC) : 0;
  const int nsteps = nsteps3 + nsc;
  int c0 = cb, t0 = 0, staged = cb;   // (c0, t0): chunk and tap of the step's first slice
  for (int s = s_first; s < nsteps3; ++s) {
    if (s + 1 < nsteps && !(abl & 2)) issue_slot(s + 1, (s + 1) & 1);
    int c1 = c0, t1 = t0 + 1;
    if (t1 == NTAPS) { t1 = 0; ++c1; }
    const int ky0 = (KW == 3) ? (t0 * 11) >> 5 : t0 >> 1, ky1 = (KW == 3) ? (t1 * 11) >> 5 : t1 >> 1;   // t / KW
    const int offA0 = (c0 & 1) * A_BYTES + (ky0 * TW + (t0 - KW * ky0)) * 16;
    const int offA1 = (c1 & 1) * A_BYTES + (ky1 * TW + (t1 - KW * ky1)) * 16;
    const char* A = As + a_lane + (tp ? offA1 : offA0);
    const char* B = Bs + (s & 1) * SLOT_BYTES + b_lane;
    mma_step(A + 0, A_TM, 2 * PLANE * 16, B);
    // two slices on
    t0 += 2;
    if (t0 >= NTAPS) { t0 -= NTAPS; ++c0; }
    // the halo tile of the chunk the NEXT step's second slice belongs to must be in LDS before the barrier below; its
    // buffer held chunk need-2, last read at least one barrier ago
    const int need = (t0 == NTAPS - 1) ? c0 + 1 : c0;
    if (need > staged && need < ce && !(abl & 8)) {
      gload_A(need);
      write_A(need, need & 1);
      staged = need;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (SC) {
    asm volatile("" : "+v"(sc_tid));
    const char* const Asc = As + (kq * BM + wm * 64 + r16) * 16;
#pragma unroll
    for (int i = 0; i < NSC; ++i) {
      const int pix = (sc_tid + i * NT) >> 2;
      const int gy = oy0 + pix / PW, gx = ox0 + pix % PW;
      scoff[i] = (pix < BM && gy < p.Hout && gx < p.Wout) ? gy * p.Wout + gx : -1;
    }
    sc_load(0);   // (the only exposed load of the phase: issuing it under the last 3x3 step costs the main loop 16 live registers)
    for (int j = 0; j < nsc; ++j) {
      const int s = nsteps3 + j;
      sc_write();                                           // loads landed before the previous step's closing barrier
      if (s + 1 < nsteps) issue_slot(s + 1, (s + 1) & 1);
      if (j + 1 < nsc) sc_load(j + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the tile is written; the loads above stay in flight
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      mma_step(Asc, 256, 4 * BM * 16, Bs + (s & 1) * SLOT_BYTES + b_lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  K32_STAMP(2);
  // ---- epilogue (round 4).  C/D layout of a 16 x 16 block: col = lane & 15 (channel), row = 4 * (lane >> 4) + r (pixel), i.e. a lane
  // holds ONE channel of four pixels: stored from there, every value is its own global_store_dword (64 per wave, four 64-B
  // segments each), and with the per-element bounds / residual branches hipcc put `s_waitcnt vmcnt(0)` in front of every one of
  // them -- 64 serialized store round trips, 17 us of a 105-us tile (37 us with a residual; phase stamps,
  // profiles/r04b_k32_phases.txt).  Now every row block goes through a wave-private LDS slab (16 pixels x WCH channels, all LDS
  // is free after the K loop's last barrier; DS operations of a wave execute in order, no barrier) and comes back as float4 =
  // four consecutive channels of one pixel: 16 global_store_dwordx4 per wave, 256 contiguous bytes per pixel, the residual read
  // the same way (unconditional, clamped address) before the slab round trip.  The arithmetic per value is unchanged
  // ((acc * alpha + (bias + chan_add)) + resid); the GroupNorm partials are summed in the new lane layout (double, fixed order:
  // in-lane over row blocks and items -> shfl_xor over the lanes that share a channel quad -> wave rows through LDS).
  constexpr int C4 = WCH / 4, EP = WCH + 4, NV = (16 * C4) / 64, RSTEP = 64 / C4;
  // the lane-derived constants of the epilogue are computed from a laundered thread id: hipcc otherwise hoists them above the K loop
  // and keeps them alive across it (the fused-shortcut instantiation spilled inside the loop for exactly that)
  int etid = tid;
  asm volatile("" : "+v"(etid));
  const int el = etid & 63, er16 = el & 15;
  static_assert(NV >= 1 && NV * 64 == 16 * C4, "slab items per el");
  static_assert((size_t)WM * BN * 16 + (size_t)T::NW * 16 * EP * 4 <= T::SMEM, "statistics rows + one slab per wave fit in the loop's LDS");
  float* const ep = reinterpret_cast<float*>(smem + WM * BN * 16) + wave * (16 * EP);   // behind `red`
  const int g = el >> 4, c4 = el % C4, prow = el / C4;
  const int nq = n0 + wn * WCH + c4 * 4;                 // this lane's channel quad after the slab round trip
  auto slab_write = [&](int tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(4 * g + r) * EP + tn * 16 + er16] = acc[tm][tn][r] * p.alpha;
    asm volatile("" ::: "memory");
  };
  auto slab_read = [&](int i) -> float4 {
    return *reinterpret_cast<const float4*>(ep + (prow + i * RSTEP) * EP + c4 * 4);
  };
  if (QUAD || SPK) {
    // raw partial sums of this K range: part[range][image][pixel][Cout]; bias / residual / statistics belong to launch_splitk_reduce
    const int HWo = QUAD ? 64 : p.Hout * p.Wout;
    {   // (launcher: Cout % 4 == 0, part 16-byte aligned -- k32_epilogue_ok)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        slab_write(tm);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int m = prow + i * RSTEP;
          int img, pixel;
          bool ok;
          if (QUAD) {
            const int py = wm * 4 + tm;
            img = zo + (py >> 3) * 2 + (m >> 3);
            pixel = (py & 7) * 8 + (m & 7);
            ok = (img < p.Z);
          } else {
            const int oy = oy0 + (wm * 4 + tm) * FR + m / PW, ox = ox0 + m % PW;
            img = zo;
            pixel = oy * p.Wout + ox;
            ok = (oy < p.Hout && ox < p.Wout);
          }
          const float4 a = slab_read(i);
          if (ok && nq < Cout) *reinterpret_cast<float4*>(p.part + (((size_t)ks_id * p.Z + img) * HWo + pixel) * Cout + nq) = a;
        }
        asm volatile("" ::: "memory");
      }
    }
    return;
  }
  float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
  const float* __restrict__ rz = p.resid ? p.resid + (long long)zo * p.r_zo : nullptr;
  const float* __restrict__ cadd = p.chan_add ? p.chan_add + (long long)zo * p.ld_chan_add : nullptr;
  const bool has_b = (p.bias != nullptr), has_c = (cadd != nullptr);
  const bool full = (n0 + BN <= Cout) && (oy0 + T::PH <= p.Hout) && (ox0 + PW <= p.Wout);
  double* const red = reinterpret_cast<double*>(smem);   // [WM wave rows][BN][2]; LDS is free after the last barrier
  const bool want_stats = (p.stats != nullptr);
  // (launcher: ldo, ldr, Cout multiples of 4, out / resid 16-byte aligned -- k32_epilogue_ok)
  {
    const bool nok = full || (nq < Cout);
    float add4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) add4[j] = nok ? ((has_b ? p.bias[nq + j] : 0.f) + (has_c ? cadd[nq + j] : 0.f)) : 0.f;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    // item i of row block tm: block row m = prow + i * RSTEP -> output pixel, residual pixel, inside the image?
    auto geom = [&](int tm, int i, int& pixel, int& rpix) -> bool {
      const int m = prow + i * RSTEP;
      const int oy = oy0 + (wm * 4 + tm) * FR + m / PW, ox = ox0 + m % PW;
      pixel = POLY ? (2 * oy + phy) * (2 * p.Wout) + 2 * ox + phx : oy * p.Wout + ox;
      rpix = p.rups ? ((oy >> 1) * (p.Wout >> 1) + (ox >> 1)) : pixel;
      return oy < p.Hout && ox < p.Wout;
    };
    auto finish = [&](const float4& a, const float4& r, float4& v) {
      v.x = (a.x + add4[0]) + r.x;
      v.y = (a.y + add4[1]) + r.y;
      v.z = (a.z + add4[2]) + r.z;
      v.w = (a.w + add4[3]) + r.w;
    };
    // (round 5, measured and NOT kept: the lane's 4 x NV values per channel summed in fp32 (packed adds / fmas) before joining the f64
    //  chain -- 118 instead of 126 VGPRs, but only +0.5 % on a launch that emits statistics (the cost of the statistics is the
    //  shuffles, the barrier and the LDS reduction, not the f64 arithmetic) and 5 % more mean error on a 39-step trajectory:
    //  profiles/r05d_k32_prod_fp32_partials_NEGATIVE.txt)
    auto stat4 = [&](const float4& v) {
      s1[0] += (double)v.x; s2[0] += (double)v.x * (double)v.x;
      s1[1] += (double)v.y; s2[1] += (double)v.y * (double)v.y;
      s1[2] += (double)v.z; s2[2] += (double)v.z * (double)v.z;
      s1[3] += (double)v.w; s2[3] += (double)v.w * (double)v.w;
    };
    if (full) {
      // straight-line code (no per-element predicate, so hipcc counts vmcnt instead of draining it): the residual rows of block
      // tm + 1 are requested right after block tm has left its accumulators for the slab, ahead of block tm's stores
      float4 rv[2][NV];
      auto load_rv = [&](int tm, float4 (&dst)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          int pixel, rpix;
          (void)geom(tm, i, pixel, rpix);
          dst[i] = *reinterpret_cast<const float4*>(rz + rpix * p.ldr + nq);
        }
      };
#pragma unroll
      for (int i = 0; i < NV; ++i) rv[0][i] = rv[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rz) load_rv(0, rv[0]);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        slab_write(tm);
        if (rz && tm + 1 < 4) load_rv(tm + 1, rv[(tm + 1) & 1]);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          int pixel, rpix;
          (void)geom(tm, i, pixel, rpix);
          const float4 a = slab_read(i);
          float4 v;
          finish(a, rv[tm & 1][i], v);
          *reinterpret_cast<float4*>(outz + pixel * p.ldo + nq) = v;
          if (want_stats) stat4(v);
        }
        asm volatile("" ::: "memory");
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        int po[NV];
        bool ok[NV];
        float4 rv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          int pixel, rpix;
          ok[i] = geom(tm, i, pixel, rpix) && nok;
          po[i] = pixel * p.ldo + nq;
          rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          // unconditional load from a clamped address: no per-element branch, the loads of a row block fly together
          if (rz) rv[i] = *reinterpret_cast<const float4*>(rz + (ok[i] ? rpix * p.ldr + nq : 0));
        }
        slab_write(tm);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float4 a = slab_read(i);
          float4 v;
          finish(a, rv[i], v);
          if (ok[i]) {
            *reinterpret_cast<float4*>(outz + po[i]) = v;
            if (want_stats) stat4(v);
          }
        }
        asm volatile("" ::: "memory");
      }
    }
    K32_STAMP(3);
    if (want_stats) {   // fixed-order reduction over the lanes that hold the same channel quad, then over the wave rows below
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = C4; off < 64; off <<= 1) {
          s1[j] += __shfl_xor(s1[j], off);
          s2[j] += __shfl_xor(s2[j], off);
        }
      }
      if (el < C4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double* d = red + ((size_t)wm * BN + wn * WCH + c4 * 4 + j) * 2;
          d[0] = s1[j];
          d[1] = s2[j];
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int c = etid; c < BN; c += NT) {
      if (n0 + c < Cout) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s1 += red[((size_t)w * BN + c) * 2];
          s2 += red[((size_t)w * BN + c) * 2 + 1];
        }
        // (polyphase: blockIdx.z = image * 4 + phase, i.e. 4 * gridDim.x statistics rows per image)
        double* dst = p.stats + (((size_t)blockIdx.z * gridDim.x + bx) * Cout + n0 + c) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
  K32_STAMP(4);
  if (ABL && p.dbg) {   // store drain time
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K32_STAMP(6);
  }
}

static bool is_vec(const GemmArgs& a) {
  return (((a.c0 | a.c1 | a.lda0 | a.lda1 | a.Cin) & 15) == 0) && ((((uintptr_t)a.a0) | ((uintptr_t)a.a1)) & 15) == 0 &&
         (!a.pscale || (((((uintptr_t)a.pscale) | ((uintptr_t)a.pshift)) & 15) == 0 && (a.ld_ps & 3) == 0));
}

// A/B switch of the XCD-aware tile map (ASYRP_XCD_MAP=0 disables it)
bool xcd_map_enabled() {
  static const bool on = [] { const char* e = ab_env("ASYRP_XCD_MAP"); return !(e && e[0] == '0'); }();
  return on;
}

template <class T, bool VEC, bool ABL = false, bool PIPE = false, bool SC = false, int NP = 3>
static hipError_t launch_x(const GemmArgs& a, hipStream_t s) {
  int gx;
  if (T::KS == 1) {
    gx = (a.Hout * a.Wout + T::BM - 1) / T::BM;
  } else {
    gx = ((a.Hout + T::PH - 1) / T::PH) * ((a.Wout + T::PW - 1) / T::PW);
  }
  int gy = (a.Cout + T::BN - 1) / T::BN;
  if (a.sk > 1) {
    const int nch = (a.Cin + XKC - 1) / XKC;
    if (!PIPE || SC || !a.part || a.s0 || nch % a.sk != 0 || (a.Cin % XKC) != 0) return hipErrorInvalidValue;
    gy *= a.sk;
  }
  dim3 grid(gx, gy, a.Z), block(T::NT);
  GemmArgs ax = a;
  ax.xmap = (xcd_map_enabled() && gx >= 16 && (gx & 7) == 0 && (gy * (long long)gx) % 8 == 0) ? 1 : 0;
  // per-device (the attribute lives in the device's code object); set once per process and device, under a lock-free
  // idempotent flag: two threads racing here both set the same value
  static bool attr_set[16] = {};
  if (T::SMEM > 64 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_f16x3_kernel<T, VEC, ABL, PIPE, SC, NP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::SMEM);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL((igemm_f16x3_kernel<T, VEC, ABL, PIPE, SC, NP>), grid, block, T::SMEM, s, ax);
  return hipGetLastError();
}

using K32Main = K32Cfg<8, 2>;
using K32Half = K32Cfg<8, 4>;
using K32Img8 = K32Cfg<8, 8, 8>;
using K32S2 = K32Cfg<8, 8, 16, 2>;
using K32Up = K32Cfg<8, 2, 16, 1, 2>;
using K32Quad = K32Cfg<8, 2, 16, 1, 3, true>;
template <class T, bool SC, bool ABL = false, int NP = 3, bool SPK = false>
static hipError_t launch_k32(const GemmArgs& a, hipStream_t s) {
  const int gx = T::QUAD ? 1 : ((a.Hout + T::PH - 1) / T::PH) * ((a.Wout + T::PW - 1) / T::PW);
  const int gy = ((a.Cout + T::BN - 1) / T::BN) * ((T::QUAD || SPK) ? a.sk : 1);
  // polyphase form: one z-slice per (image, output phase); quad form: one per group of four images, K ranges along y
  dim3 grid(gx, gy, T::QUAD ? (a.Z + 3) / 4 : a.Z * (T::KS == 2 ? 4 : 1)), block(T::NT);
  GemmArgs ax = a;
  ax.xmap = (xcd_map_enabled() && gx >= 16 && (gx & 7) == 0 && (gy * (long long)gx) % 8 == 0) ? 1 : 0;
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (T::SMEM > 64 * 1024 && (dev < 0 || dev >= 16 || !attr_set[dev])) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_f16x3_k32_kernel<T, SC, ABL, NP, SPK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::SMEM);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((igemm_f16x3_k32_kernel<T, SC, ABL, NP, SPK>), grid, block, T::SMEM, s, ax);
  return hipGetLastError();
}

// the float4 epilogue of the K32 kernel: output / residual rows and the channel count in whole, 16-byte aligned quads
static bool k32_epilogue_ok(const GemmArgs& a) {
  if ((a.Cout & 3) || (a.ldo & 3) || (a.o_zo & 3) || (((uintptr_t)a.out) & 15)) return false;
  if (a.resid && ((a.ldr & 3) || (a.r_zo & 3) || (((uintptr_t)a.resid) & 15))) return false;
  if (a.part && (((uintptr_t)a.part) & 15)) return false;
  return true;
}
// the layers the K32 kernel takes over from the 8-wave 32x32x16 tile (everything else about the tile is the same)
static bool k32_ok(const GemmArgs& a) {
  if (!k32_epilogue_ok(a)) return false;
  if (a.s0 && (a.ups || (a.Cin2 & 31) || a.Cin2 < 32)) return false;   // fused shortcut: an even number of raw slices
  return a.ks == 3 && a.stride == 1 && !a.abl && a.sk <= 1 && (a.Cin & 31) == 0 && a.Cin >= 32 && is_vec(a);
}
static bool k32s2_ok(const GemmArgs& a) {
  if (!k32_epilogue_ok(a)) return false;
  return a.ks == 3 && a.stride == 2 && !a.ups && !a.s0 && !a.abl && a.sk <= 1 && (a.Cin & 31) == 0 && a.Cin >= 32 && is_vec(a);
}
// split-K on the plain K32 forms (GemmArgs.sk, .part, .tile = the form): whole K = 32 steps per range
static bool k32spk_ok(const GemmArgs& a) {
  if (!k32_epilogue_ok(a)) return false;
  if (!(a.tile == XT_256x128K32 || a.tile == XT_128x128K32 || a.tile == XT_64x128K32)) return false;
  if (!(a.ks == 3 && a.stride == 1 && !a.ups && !a.s0 && !a.abl && !a.poly && !a.rups && a.sk >= 2 && a.part)) return false;
  return (a.Cin & 31) == 0 && ((a.Cin / XKC) % (2 * a.sk)) == 0 && is_vec(a);
}
// A/B switch: ASYRP_MAIN_TILE=6 keeps the 32x32x16 organisation for the automatic choice
static bool k32_preferred() {
  static const bool on = [] { const char* e = ab_env("ASYRP_MAIN_TILE"); return !(e && e[0] == '6'); }();
  return on;
}

// the 256x128 tile the big 3x3 layers run on: 8 waves (4 per SIMD with two workgroups per CU), plain per-tap loop.
// A 512x128 tile with 16 waves (one workgroup per CU, one weight slice and one barrier domain for all 16) measured +4 % on
// 128->128 @256 but -1 % on every other layer and -0.7 % on the whole edit (profiles/r02c_ab_*_tile16wave.txt): removed.
// Interleaved A/B against the 4-wave software-pipelined tile: +6...12 % per layer, +5 % on the whole edit
// (profiles/r01_conv_microbench_w8*.txt)
int gemm_main_tile() { return XT_256x128W8; }

// tile ids of the f16x3 family (GemmArgs.tile / profile variant).
// The choice is a function of the LAYER SHAPE only: the workgroup count is priced at a nominal batch, never at the
// actual one (a.Z).  The tile decides whether a 1x1 shortcut is fused (gemm_can_fuse_shortcut: one accumulator and
// pre-summed biases vs two launches) and how the GroupNorm partial sums are partitioned (gemm_mblocks), so a
// batch-dependent choice would make an image's bits depend on what it is batched or sharded with.
constexpr int NOMINAL_Z = 32;   // BASELINE.json configs[1]: 32 images per GPU
// Batch class (round 4).  The nominal batch is a property of the ENGINE (asyrp_config.nominal_batch, fixed at
// asyrp_create; GemmArgs.nz; 0 = 32), never of the call, so an image's bits still do not depend on what it is batched with on that
// engine.  The small class (nominal batch 1 or 2: single-image serving, the reference's bs_train = 1) keeps every 3x3 stride-1
// layer with Cin % 32 == 0 on the K32 family -- 256-pixel form from 64 x 64 up, 128-pixel form at 32 x 32 and 16 x 16, the 8 x 8
// patch form below -- and splits K until the launch offers >= 256 workgroups at the nominal batch (<= 8 ranges, whole K = 32
// steps per range); no quad grouping.  Results across classes agree to fp32 rounding, like any two tile shapes.
static int nominal_z(const GemmArgs& a) { return a.nz > 0 ? a.nz : NOMINAL_Z; }
static bool is_vec(const GemmArgs& a);
static bool k32_preferred();
// what k32_epilogue_ok will say about a launch of this layer once the split-K fields (sk, part) are filled in: the part buffer is
// the engine's own 256-byte-aligned workspace, so only the output / residual geometry decides
static bool k32_split_epilogue_ok(const GemmArgs& a) {
  if ((a.Cout & 3) || (a.ldo & 3) || (a.o_zo & 3) || (((uintptr_t)a.out) & 15)) return false;
  if (a.resid && ((a.ldr & 3) || (a.r_zo & 3) || (((uintptr_t)a.resid) & 15))) return false;
  return true;
}
int small_class_tile(const GemmArgs& a) {   // 0: the default rules apply
  if (nominal_z(a) > 2 || a.math != MATH_F16X3 || a.ks != 3 || a.stride != 1 || a.ups || a.abl || a.poly || a.rups) return 0;
  if ((a.Cin & 31) || a.Cin < 32 || !is_vec(a) || !k32_preferred() || !k32_split_epilogue_ok(a)) return 0;
  const long long M = (long long)a.Hout * a.Wout;
  if (M >= 4096) return XT_256x128K32;
  if (M >= 256) return XT_128x128K32;
  return (a.Hout == 8 && a.Wout == 8) ? XT_64x128K32 : 0;
}
int small_class_sk(const GemmArgs& a) {     // K ranges of a small-class launch of `a` WITHOUT a fused shortcut
  const int t = small_class_tile(a);
  if (!t) return 1;
  const int bm = t == XT_256x128K32 ? 256 : (t == XT_128x128K32 ? 128 : 64), pw = bm >= 128 ? 16 : 8, ph = bm / pw;
  const long long wgs = (long long)((a.Hout + ph - 1) / ph) * ((a.Wout + pw - 1) / pw) * ((a.Cout + 127) / 128) * nominal_z(a);
  int sk = 1;
  while (sk < 8 && wgs * sk < 256) sk *= 2;
  const int nch = a.Cin / XKC;
  while (sk > 1 && (nch % (2 * sk)) != 0) sk /= 2;
  return sk;
}
static int auto_tile_x(const GemmArgs& a) {
  if (const int t = small_class_tile(a)) return t;
  if (a.stride == 2) return XT_64x128;
  const long long M = (long long)a.Hout * a.Wout;
  auto blocks = [&](int bm, int bn) {
    long long mt;
    if (a.ks == 1) {
      mt = (M + bm - 1) / bm;
    } else {
      const int pw = bm >= 128 ? 16 : 8, ph = bm / pw;
      mt = (long long)((a.Hout + ph - 1) / ph) * ((a.Wout + pw - 1) / pw);
    }
    return mt * ((a.Cout + bn - 1) / bn) * nominal_z(a);
  };
  if (a.Cout <= 32 && a.ks == 3 && M >= 256 && blocks(256, 32) >= 256) return XT_256x32;
  if (a.Cout <= 64) return (M >= 256 && blocks(256, 64) >= 256) ? XT_256x64 : XT_64x64;
  if (M >= 256 && blocks(256, 128) >= 512) return a.ks == 3 ? gemm_main_tile() : XT_256x128;
  // 32x32 layers (one workgroup per CU at the nominal batch): the K32 form of the main tile still beats the 128x128 tile,
  // 337 vs 312-317 and 366 vs 321-324 TFLOP/s on 256->256 / 512->256 @32 (profiles/r02w_k32_ablations_tilechoice.txt), and
  // fuses the 1x1 shortcut; at 16x16 (half the CUs idle) it loses, 213 vs 249
  if (a.ks == 3 && M >= 1024 && blocks(256, 128) >= 256 && (a.Cin & 31) == 0) return gemm_main_tile();
  // 32x32 / 16x16 layers: 3 taps per barrier on the 128x128 tile beats the narrower tiles even at one workgroup per CU
  // (measured, profiles/r01_conv_microbench_*.txt); 8x8 layers (M = 64) fall through to 64-pixel tiles
  // (3x3 layers with channel counts that are multiples of 32 take the 128-pixel form of the K32 kernel: 16x16 layers 256 -> 284,
  // 266 -> 307 TFLOP/s, profiles/r02z_ab_k32_128px_tile.txt; eff_tile_x falls back to the 32x32x16 tile otherwise)
  if (M >= 128 && blocks(128, 128) >= 256) return (a.ks == 3 && k32_preferred()) ? XT_128x128K32 : XT_128x128;
  if (M >= 128 && blocks(64, 128) >= 256) return XT_64x128;
  // 8x8 layers that are not split over K (engine.hip sets the tile itself when it splits): the 8x8-patch form of the K32 kernel,
  // 99 -> 96 us on 512->512 (profiles/r03a_ab_k32_8x8_stride2.txt) -- these layers are bound by the fixed latencies of a
  // 64-pixel workgroup streaming a 1.2 MB weight slab, not by the instruction -- and it can fuse the block's 1x1 shortcut
  if (a.ks == 3 && M == 64 && a.Hout == 8 && k32_preferred()) return XT_64x128K32;
  return XT_64x64;
}

static int requested_tile_x(const GemmArgs& a) {
  if (a.stride == 2) return XT_64x128;
  return a.tile ? a.tile : auto_tile_x(a);
}

// the tile actually launched: ragged channel counts (conv_in: Cin = 3) use scalar-gather staging, compiled for two shapes
// polyphase launch (GemmArgs.poly): nearest x2 + 3x3 as four 2x2-tap convolutions on the source grid (K32Cfg<8, 2, 16, 1, 2>)
static bool k32up_ok(const GemmArgs& a) {
  if (!k32_epilogue_ok(a)) return false;
  return a.poly && a.ks == 3 && a.stride == 1 && !a.ups && !a.s0 && !a.abl && a.sk <= 1 && !a.resid && (a.Cin & 31) == 0 && a.Cin >= 32 &&
         a.Hin == a.Hout && a.Win == a.Wout && a.w_phase > 0 && is_vec(a);
}
// quad form (split-K over four-image groups) for the 8 x 8 layers: dense per-image tensors, an even number of chunks per K range
static bool k32quad_ok(const GemmArgs& a) {
  if (!k32_epilogue_ok(a)) return false;
  if (!(a.ks == 3 && a.stride == 1 && !a.ups && !a.s0 && !a.abl && !a.poly && a.Hout == 8 && a.Wout == 8 && a.Hin == 8 && a.Win == 8)) return false;
  if (a.sk < 2 || !a.part || (a.Cin & 31) || ((a.Cin / XKC) % (2 * a.sk)) != 0 || !is_vec(a)) return false;
  if (a.a0_zo != 64LL * a.lda0 || (a.a1 && a.a1_zo != 64LL * a.lda1)) return false;
  return true;
}
static int eff_tile_x(const GemmArgs& a) {
  if (a.tile == XT_G1_256 || a.tile == XT_G1_128) return a.tile;   // set by the engine only after gemm1x1_ok()
  if (a.poly) return XT_256x128K32UP;
  if (a.tile == XT_256x128K32Q) return k32quad_ok(a) ? XT_256x128K32Q : XT_64x64;
  if (a.sk > 1 && k32spk_ok(a)) return a.tile;                       // split-K on a plain K32 form (chosen by splitk_tile)
  if (a.stride == 2) return (k32s2_ok(a) && k32_preferred() && a.tile != XT_64x128) ? XT_64x128K32S2 : XT_64x128;
  int t = requested_tile_x(a);
  if (t == XT_256x128K32 && !k32_ok(a)) t = XT_256x128W8;
  if (t == XT_128x128K32 && !k32_ok(a)) t = XT_128x128;
  if (t == XT_64x128K32 && !k32_ok(a)) t = XT_64x64;
  if (t == XT_256x128W8 && !a.tile && k32_ok(a) && k32_preferred()) t = XT_256x128K32;
  if (is_vec(a)) return t;
  return (t == XT_256x128 || t == XT_128x128 || t == XT_256x64 || t == XT_256x128W8 || t == XT_256x128K32 || t == XT_128x128K32 ||
          t == XT_256x32) ? XT_256x128
                                                                                                               : XT_64x128;
}

int gemm_resolve_tile_x(const GemmArgs& a) { return eff_tile_x(a); }

bool gemm_can_fuse_shortcut(const GemmArgs& a) {
  if (a.poly || a.ks != 3 || a.stride != 1 || a.ups || a.abl || !a.s0 || a.Cin2 <= 0) return false;
  if (((a.sc0 | a.sc1 | a.lds0 | a.lds1 | a.Cin2) & 15) || ((((uintptr_t)a.s0) | ((uintptr_t)a.s1)) & 15)) return false;
  const int t = eff_tile_x(a);
  return is_vec(a) && (t == XT_256x128 || t == XT_256x128W8 || t == XT_256x128K32 || t == XT_128x128K32 || t == XT_64x128K32);
}

int gemm_mblocks(const GemmArgs& a) {
  int bm;
  if (a.poly) return 4 * ((a.Hout + K32Up::PH - 1) / K32Up::PH) * ((a.Wout + K32Up::PW - 1) / K32Up::PW);   // 4 phases per image
  if (eff_tile_x(a) == XT_64x128K32S2) return ((a.Hout + K32S2::PH - 1) / K32S2::PH) * ((a.Wout + K32S2::PW - 1) / K32S2::PW);
  switch (eff_tile_x(a)) {
    case XT_256x128: case XT_256x64: case XT_256x128W8: case XT_256x128K32: case XT_256x32: case XT_G1_256:
      bm = 256; break;
    case XT_128x128: case XT_128x128K32: case XT_G1_128: bm = 128; break;
    default: bm = 64;
  }
  if (a.ks == 1) return (a.Hout * a.Wout + bm - 1) / bm;
  const int pw = bm >= 128 ? 16 : 8, ph = bm / pw;
  return ((a.Hout + ph - 1) / ph) * ((a.Wout + pw - 1) / pw);
}

template <int NP>
static hipError_t launch_gemm_f16x3_np(const GemmArgs& a, hipStream_t s) {
  if (!a.wpk || a.bT || a.ZI > 1 || a.cout_pad < ((a.Cout + 127) / 128) * 128) return hipErrorInvalidValue;
  if (!(a.ks == 1 || a.ks == 3) || !(a.stride == 1 || a.stride == 2)) return hipErrorInvalidValue;
  if (a.ks == 1 && (a.stride != 1 || a.ups)) return hipErrorInvalidValue;
  const int tile = eff_tile_x(a);
  // last parameter: weight-slice ring size of the pipelined loop (slices stay in flight for RB-1 K-steps); the small
  // tiles have short K-steps (3-6 MFMAs per wave), so they need the deeper ring to cover the L2 latency of a slice
  using X256x128_3 = XCfg<4, 1, 2, 4, 3, 1, 2>;      // ring of 2: deeper rings measured equal (3) or slower (4) on this tile
  using X128x128_3 = XCfg<2, 2, 2, 2, 3, 1, 2, 3>;    // 11.5 KB x 2 halo + 2 x 24 KB weight slots = 71 KB: 2 workgroups per CU
  using X64x128_3 = XCfg<2, 2, 1, 2, 3, 1, 2, 3>;     // 61 KB
  using X64x64_3 = XCfg<2, 2, 1, 1, 3, 1, 3, 3>;      // 49 KB
  using X256x64_3 = XCfg<4, 1, 2, 2, 3, 1, 4>;
  using X256x128w8_3 = XCfg<4, 2, 2, 2, 3, 1>;
  // conv_out (Cout = 3 / 6): 32-wide N tile, 3 taps per barrier (18 MFMAs per wave between barriers).  Interleaved A/B at
  // B=32 (profiles/r02f_ab_convout.txt): 1 tap per barrier 564-638 us, 3 taps 536-554 us, 9 taps 602-608 us
  using X256x32_3 = XCfg<4, 1, 2, 1, 3, 1, 2, 3>;
  using X64x128_3s2 = XCfg<2, 2, 1, 2, 3, 2, 4>;
  using X256x128_1 = XCfg<4, 1, 2, 4, 1, 1, 2>;
  using X128x128_1 = XCfg<2, 2, 2, 2, 1, 1, 4>;
  using X64x128_1 = XCfg<2, 2, 1, 2, 1, 1, 4>;
  using X64x64_1 = XCfg<2, 2, 1, 1, 1, 1, 4>;
  using X256x64_1 = XCfg<4, 1, 2, 2, 1, 1, 4>;
  using X256x128_3plain = XCfg<4, 1, 2, 4, 3, 1>;   // un-pipelined loop (ragged-channel staging, ablation build)
  using X64x128_3plain = XCfg<2, 2, 1, 2, 3, 1>;
  using X64x128_3s2plain = XCfg<2, 2, 1, 2, 3, 2>;
  using X256x128_1plain = XCfg<4, 1, 2, 4, 1, 1>;
  using X64x128_1plain = XCfg<2, 2, 1, 2, 1, 1>;
  if (!is_vec(a)) {   // ragged channel counts (conv_in: Cin = 3): scalar-gather staging, two tile shapes per kernel size
    if (a.stride == 2) return launch_x<X64x128_3s2plain, false, false, false, false, NP>(a, s);
    const bool big = (tile == XT_256x128);
    if (a.ks == 3) return big ? launch_x<X256x128_3plain, false, false, false, false, NP>(a, s) : launch_x<X64x128_3plain, false, false, false, false, NP>(a, s);
    return big ? launch_x<X256x128_1plain, false, false, false, false, NP>(a, s) : launch_x<X64x128_1plain, false, false, false, false, NP>(a, s);
  }
  if (a.poly) return k32up_ok(a) ? launch_k32<K32Up, false, false, NP>(a, s) : hipErrorInvalidValue;
  if (tile == XT_256x128K32Q) return launch_k32<K32Quad, false, false, NP>(a, s);
  if (a.sk > 1 && k32spk_ok(a)) {
    if (tile == XT_256x128K32) return launch_k32<K32Main, false, false, NP, true>(a, s);
    if (tile == XT_128x128K32) return launch_k32<K32Half, false, false, NP, true>(a, s);
    return launch_k32<K32Img8, false, false, NP, true>(a, s);
  }
  if (a.abl && NP != 3) return hipErrorInvalidValue;
  if (a.abl) {   // timing ablations of the main tile: instantiated in the profiling library only (libasyrp_hip_bench.so)
#ifdef ASYRP_BENCH_HOOKS
    if (a.ks == 3 && a.stride == 1 && tile == XT_256x128) return launch_x<X256x128_3plain, true, true, false, false>(a, s);
    if (a.ks == 3 && a.stride == 1 && a.tile == XT_256x128K32 && !a.s0 && (a.Cin & 31) == 0) return launch_k32<K32Main, false, true>(a, s);
    if (a.ks == 3 && a.stride == 1 && tile == XT_256x128W8) return launch_x<XCfg<4, 2, 2, 2, 3, 1>, true, true, false, false>(a, s);
#endif
    return hipErrorInvalidValue;
  }
  if (a.s0) {   // fused 1x1 shortcut: main tile only (gemm_can_fuse_shortcut)
    if (!gemm_can_fuse_shortcut(a)) return hipErrorInvalidValue;
    if (tile == XT_256x128K32) return launch_k32<K32Main, true, false, NP>(a, s);
    if (tile == XT_128x128K32) return launch_k32<K32Half, true, false, NP>(a, s);
    if (tile == XT_64x128K32) return launch_k32<K32Img8, true, false, NP>(a, s);
    return launch_x<X256x128w8_3, true, false, false, true, NP>(a, s);
  }
  if (a.ks == 3) {
    if (a.stride == 2) return tile == XT_64x128K32S2 ? launch_k32<K32S2, false, false, NP>(a, s) : launch_x<X64x128_3s2, true, false, true, false, NP>(a, s);
    switch (tile) {
      case XT_256x128: return launch_x<X256x128_3, true, false, true, false, NP>(a, s);
      case XT_128x128: return launch_x<X128x128_3, true, false, true, false, NP>(a, s);
      case XT_64x128: return launch_x<X64x128_3, true, false, true, false, NP>(a, s);
      case XT_64x64: return launch_x<X64x64_3, true, false, true, false, NP>(a, s);
      case XT_256x64: return launch_x<X256x64_3, true, false, true, false, NP>(a, s);
      case XT_256x32: return launch_x<X256x32_3, true, false, true, false, NP>(a, s);
      case XT_256x128W8: return launch_x<X256x128w8_3, true, false, false, false, NP>(a, s);
      case XT_256x128K32: return launch_k32<K32Main, false, false, NP>(a, s);
      case XT_128x128K32: return launch_k32<K32Half, false, false, NP>(a, s);
      case XT_64x128K32: return launch_k32<K32Img8, false, false, NP>(a, s);
    }
  } else {
    switch (tile) {
      case XT_256x128: return launch_x<X256x128_1, true, false, true, false, NP>(a, s);
      case XT_128x128: return launch_x<X128x128_1, true, false, true, false, NP>(a, s);
      case XT_64x128: return launch_x<X64x128_1, true, false, true, false, NP>(a, s);
      case XT_64x64: return launch_x<X64x64_1, true, false, true, false, NP>(a, s);
      case XT_256x64: return launch_x<X256x64_1, true, false, true, false, NP>(a, s);
    }
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemm_f16x3(const GemmArgs& a, hipStream_t s) {
  if (a.np != 0 && a.np != 1 && a.np != 3) return hipErrorInvalidValue;
  if (a.tile == XT_G1_256 || a.tile == XT_G1_128) return launch_gemm1x1(a, s);   // gemm1x1.hip (its own weight image in a.wpk)
  // a.np: matrix products per term -- 3 (two-term split, fp32-equivalent; 0 means 3) or 1 (single f16 product, conv_math "f16")
  if (a.np == 1) return launch_gemm_f16x3_np<1>(a, s);
  if (a.np != 0 && a.np != 3) return hipErrorInvalidValue;
  return launch_gemm_f16x3_np<3>(a, s);
}

// ---------------------------------------------------------------------------------------------------
// split-K reduction: out = sum_ks part[ks] (fixed order) + bias + chan_add + resid, plus the GroupNorm partials of `out`
// one workgroup per (pixel block, image); a thread owns channels c, c + 256, ... and walks the block's pixels
// ---------------------------------------------------------------------------------------------------
static bool is_vec(const GemmArgs& a);
constexpr int SKR_PIX = 8;    // pixels per reduce workgroup == pixels per statistics row (8x8 image -> 8 workgroups per image)

int splitk_stat_blocks(int HW) { return (HW + SKR_PIX - 1) / SKR_PIX; }

int splitk16_ranges();
static int splitk_factor_impl(const GemmArgs& a, bool allow16) {
  if (a.math != MATH_F16X3 || !a.wpk || a.ks != 3 || a.stride != 1 || a.ups || a.s0 || a.rups || a.abl) return 1;
  // measured (profiles/r01_conv_microbench_kb8_splitk.txt, B=32): 1024->512 @8x8 179 -> 134 us; 512->512 @8x8 no gain (91 us
  // either way: with 32 chunks a workgroup's fixed prologue/epilogue latency equals its share of the loop)
  // round 3: with the quad form (four images per workgroup, K32Cfg<8, 2, 16, 1, 3, true>) every 8 x 8 layer with Cin % 256 == 0
  // splits, 512 -> 512 included; the 64 x 64 tile keeps the old rule (Cin >= 1024)
  // round 4: the 16 x 16 maps on the 128-pixel K32 form offer 2 x Cout/128 workgroups per image -- 256 at the nominal batch, one
  // per CU with two waves per SIMD (matrix pipe 32 % busy, half of the wave cycles parked: profiles/rd3_pmc_families_*) -- a 2-way K
  // split fills the second slot (ASYRP_SPLITK16=0: off).  Blocks with a fused 1x1 shortcut keep the single-pass form (a.s0 above).
  if (small_class_tile(a)) return small_class_sk(a);
  if (allow16 && splitk16(a)) return (a.Hout == 16) ? splitk16_ranges() : 2;
  if (a.Hout * a.Wout > 64 || a.Cin % 128 != 0 || !is_vec(a)) return 1;
  if (splitk_quad(a)) return 8;
  return a.Cin >= 1024 ? 8 : 1;
}
int splitk_factor(const GemmArgs& a) { return splitk_factor_impl(a, true); }
// for the partial launches of a dual step's shared skip half (engine.hip conv1_shared): sharing is worth more than the 16 x 16 split
int splitk_factor_shared(const GemmArgs& a) { return splitk_factor_impl(a, false); }
// a block's 1x1 shortcut runs as its own launch (and enters the reduce as the residual) instead of being fused: the quad form, and
// the small class whenever the unfused conv splits K
bool splitk_unfused(const GemmArgs& a) { return splitk_quad(a) || (small_class_tile(a) && small_class_sk(a) > 1); }
// K ranges of the 16 x 16 layers: 2 on the 128-pixel form (two tiles per image), or (ASYRP_SPLITK16=4, experiment) 4 on the
// 256-pixel main tile (one tile per image: twice the matrix work per weight slice and barrier)
int splitk16_ranges() {
  static const int n = [] { const char* e = ab_env("ASYRP_SPLITK16"); return (e && e[0] == '4') ? 4 : 2; }();
  return n;
}
bool splitk16(const GemmArgs& a) {
  static const bool on = [] { const char* e = ab_env("ASYRP_SPLITK16"); return !(e && e[0] == '0'); }();   // A/B switch, recorded by bench.py
  // ASYRP_SPLITK32=1 (experiment, off by default): the same for the 32 x 32 maps on the 256-pixel form (4 x Cout/128 workgroups per image)
  static const bool on32 = [] { const char* e = ab_env("ASYRP_SPLITK32"); return e && e[0] == '1'; }();
  const bool m16 = a.Hout == 16 && a.Wout == 16 && a.Hin == 16 && a.Win == 16, m32 = on32 && a.Hout == 32 && a.Wout == 32 && a.Hin == 32 && a.Win == 32;
  // (k32_split_epilogue_ok: the K32 float4 epilogue's rule -- Cout % 4, aligned rows -- so that the factor, the tile and the kernel
  //  that is launched are decided by ONE predicate; with Cout = 6 at 16 x 16 the old rule promised a K32 split form that
  //  k32spk_ok then refused)
  return on && nominal_z(a) > 2 && a.ks == 3 && a.stride == 1 && !a.ups && !a.poly && !a.s0 && !a.rups && (m16 || m32) &&
         a.Cin >= 256 && (a.Cin % (m16 ? 32 * splitk16_ranges() : 64)) == 0 && is_vec(a) && k32_preferred() && k32_split_epilogue_ok(a);
}
// the tile a split launch runs on (a function of the layer shape only, like the factor)
int splitk_tile(const GemmArgs& a) {
  if (const int t = small_class_tile(a)) return t;
  if (splitk_quad(a)) return XT_256x128K32Q;
  if (splitk16(a)) return (a.Hout == 32 || splitk16_ranges() == 4) ? XT_256x128K32 : XT_128x128K32;
  return XT_64x64;
}
bool splitk_quad(const GemmArgs& a) {
  static const bool on = [] { const char* e = ab_env("ASYRP_QUAD8"); return !(e && e[0] == '0'); }();   // A/B switch, recorded by bench.py
  return on && nominal_z(a) > 2 && a.ks == 3 && a.stride == 1 && !a.ups && !a.s0 && a.Hout == 8 && a.Wout == 8 && a.Hin == 8 && a.Win == 8 && a.Cin >= 512 &&
         (a.Cin % 256) == 0 && a.a0_zo == 64LL * a.lda0 && (!a.a1 || a.a1_zo == 64LL * a.lda1);
}

// SK > 0: compile-time split factor -- all SK x SKR_PIX partial values of a channel are loaded before the first add (64 loads in
// flight per thread instead of one dependent chain; the reduce of the quad form took as long as its GEMM: 36 us per 512 -> 512
// layer at B = 32); SK == 0: run-time factor.  Same fixed summation order either way (k ascending per pixel, pixels ascending).
template <int SK>
__global__ void splitk_reduce_kernel(const GemmArgs p, int HW) {
  const int blk = blockIdx.x, zo = blockIdx.y, Z = gridDim.y, Cout = p.Cout;
  const int p0 = blk * SKR_PIX, p1 = min(HW, p0 + SKR_PIX);
  const float* __restrict__ rz = p.resid ? p.resid + (long long)zo * p.r_zo : nullptr;
  const float* __restrict__ cadd = p.chan_add ? p.chan_add + (long long)zo * p.ld_chan_add : nullptr;
  float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
  for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
    const float add = (p.bias ? p.bias[c] : 0.f) + (cadd ? cadd[c] : 0.f);
    double s1 = 0.0, s2 = 0.0;
    if (SK > 0 && p1 - p0 == SKR_PIX) {
      float pv[SKR_PIX][SK > 0 ? SK : 1], rv[SKR_PIX];
#pragma unroll
      for (int q = 0; q < SKR_PIX; ++q) {
#pragma unroll
        for (int k = 0; k < SK; ++k) pv[q][k] = p.part[(((size_t)k * Z + zo) * HW + p0 + q) * Cout + c];
        rv[q] = rz ? rz[(size_t)(p0 + q) * p.ldr + c] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < SKR_PIX; ++q) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < SK; ++k) v += pv[q][k];
        v = (v + add) + rv[q];
        outz[(size_t)(p0 + q) * p.ldo + c] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    } else {
      for (int pix = p0; pix < p1; ++pix) {
        float v = 0.f;
        for (int k = 0; k < p.sk; ++k) v += p.part[(((size_t)k * Z + zo) * HW + pix) * Cout + c];
        v = (v + add) + (rz ? rz[(size_t)pix * p.ldr + c] : 0.f);
        outz[(size_t)pix * p.ldo + c] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    }
    if (p.stats) {
      double* dst = p.stats + (((size_t)zo * gridDim.x + blk) * Cout + c) * 2;
      dst[0] = s1;
      dst[1] = s2;
    }
  }
}

hipError_t launch_splitk_reduce(const GemmArgs& a, hipStream_t s) {
  if (a.sk < 2 || !a.part || a.rups) return hipErrorInvalidValue;
  const int HW = a.Hout * a.Wout;
  const dim3 grid(splitk_stat_blocks(HW), a.Z);
  if (a.sk == 8) hipLaunchKernelGGL(splitk_reduce_kernel<8>, grid, dim3(256), 0, s, a, HW);
  else if (a.sk == 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, grid, dim3(256), 0, s, a, HW);
  else if (a.sk == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(256), 0, s, a, HW);
  else hipLaunchKernelGGL(splitk_reduce_kernel<0>, grid, dim3(256), 0, s, a, HW);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// weight packing (device side, run once per parameter upload): PyTorch conv weight [Cout][Cin][k][k] fp32 ->
// [step = chunk*k*k + tap][unit u][cout_pad][8] f16, u = {hi k0-7, hi k8-15, lo k0-7, lo k8-15}, times `wscale`
// ---------------------------------------------------------------------------------------------------
__global__ void pack_f16x3_kernel(const float* w, _Float16* dst, int cout, int cin, int kk, int cout_pad, float wscale,
                                  long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    long long r = i >> 3;
    const int n = (int)(r % cout_pad);
    r /= cout_pad;
    const int u = (int)(r & 3);
    const int step = (int)(r >> 2);
    const int chunk = step / kk, tap = step - chunk * kk;
    const int k = chunk * XKC + (u & 1) * 8 + j;
    float v = 0.f;
    if (n < cout && k < cin) v = w[((long long)n * cin + k) * kk + tap] * wscale;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    dst[i] = (u < 2) ? h : l;
  }
}

size_t f16x3_packed_halfs(int cout, int cin, int ks) {
  const int nchunks = (cin + XKC - 1) / XKC, cout_pad = ((cout + 127) / 128) * 128;
  return (size_t)nchunks * ks * ks * 4 * cout_pad * 8;
}

hipError_t launch_pack_f16x3(const float* w_dev, void* dst, int cout, int cin, int ks, float wscale, hipStream_t s) {
  const int cout_pad = ((cout + 127) / 128) * 128;
  const long long total = (long long)f16x3_packed_halfs(cout, cin, ks);
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(pack_f16x3_kernel, dim3((unsigned)b), dim3(256), 0, s, w_dev, reinterpret_cast<_Float16*>(dst), cout,
                     cin, ks * ks, cout_pad, wscale, total);
  return hipGetLastError();
}

float f16x3_act_scale() { return ACT_SCALE; }

}  // namespace asyrp
