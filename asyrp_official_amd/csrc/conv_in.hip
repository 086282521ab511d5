// conv_in.hip -- the FIRST convolution of the UNets (conv_in 3 -> 128, models/ddpm/diffusion.py:346-350,484; iDDPM / ADM
// input_blocks.0.0 3 -> 128 / 256, models/improved_ddpm/unet.py:463-467): 3 x 3, stride 1, pad 1, no prologue, bias, and the
// GroupNorm partial statistics of its output (the first block's norm1).
//
// K = 27: there is no matrix work to speak of (0.06 GF per image against 33.5 MB of output); on the implicit-GEMM tile the layer
// ran its scalar-gather staging and a 32-deep K step per 16-channel chunk of zeros: 367 us per launch at B = 32, 2.9 TB/s of its
// output bytes (0.36 of the HBM roof).  Here it is what it is -- an HBM-write-bound stencil -- in plain fp32 FMAs:
//   one workgroup = a 16 x 16 output patch of one image; its 18 x 18 x 3 halo sits in LDS (4 KB);
//   a thread owns 4 consecutive output channels (its 27 x 4 weights live in registers) of one of 256 / (Cout / 4) pixel slots and
//   walks the patch in groups of four consecutive pixels (see the kernel); 108 FMAs (54 v_pk_fma_f32) per pixel and thread;
//   the Cout / 4 lanes of a pixel store one contiguous Cout x 4-byte row: every store instruction writes whole 512-B (1-KiB) rows.
// Results are plain fp32 (the reference's own arithmetic: the products are not split), in a fixed order: independent of the batch.
// Statistics: per-thread double sums over its pixels, then the pixel slots in a fixed order through LDS -> one row per patch
// ([Z][patches][Cout][2], the layout launch_gn_finalize2 reads).
#include "kernels.h"

namespace asyrp {

constexpr int CI_P = 16, CI_T = CI_P + 2, CI_NT = 256;
constexpr int CI_PITCH = 56;      // floats per halo row in LDS: 18 pixels x 3 channels = 54, padded so that every row starts 16-byte aligned
constexpr int CI_G = 4;           // consecutive pixels of a row a thread computes at once

// Round 4: the round-3 form read its 27 input values per pixel as 27 LDS broadcasts -- per CU as many LDS cycles as its four SIMDs
// had FMA cycles, and the kernel sat at 3.1 TB/s of its output bytes while this chip streams plain stores at 5.5 TB/s
// (scripts/calib/hbm_counters.hip, profiles/r04m_*).  A thread now takes FOUR consecutive pixels of a row: their 3 x 18 input values
// come as 12 ds_read_b128 + 3 ds_read_b64 (a quarter of the LDS cycles), feed 16 accumulators, and leave as four float4 stores.
// The FMA order per output value is unchanged (tap rows, then (kx, ci)), so results are bit-identical to the round-3 kernel.
__global__ void __launch_bounds__(CI_NT, 2) conv_in_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float halo[CI_T * CI_PITCH];
  __shared__ double red[CI_NT * 8];                       // [slot][Cout][2] doubles, Cout * slots = 4 * 256
  const int tid = threadIdx.x, zo = blockIdx.z;
  const int tiles_x = (p.Wout + CI_P - 1) / CI_P;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = ty * CI_P, ox0 = tx * CI_P;
  const int Cout = p.Cout, NQ = Cout >> 2, NS = CI_NT / NQ;          // lanes per pixel, pixel slots (launcher: 256 % NQ == 0)
  const float* __restrict__ a0 = p.a0 + (long long)zo * p.a0_zo;
  for (int i = tid; i < CI_T * CI_T * 3; i += CI_NT) {
    const int c = i % 3, pix = i / 3, iy = pix / CI_T, ix = pix - iy * CI_T;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    halo[iy * CI_PITCH + ix * 3 + c] = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? a0[((long long)gy * p.Win + gx) * p.lda0 + c] : 0.f;
  }
  const int cq = tid % NQ, slot = tid / NQ, n = cq * 4;
  float4 wv[27];                                          // w[k = tap * 3 + ci][n .. n + 3]   (GemmArgs.w = [tap][Cin][Cout] fp32)
#pragma unroll
  for (int k = 0; k < 27; ++k) wv[k] = *reinterpret_cast<const float4*>(p.w + (long long)k * p.ldb + n);
  const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
  const bool want_stats = (p.stats != nullptr);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  constexpr int NGRP = CI_P * CI_P / CI_G;                 // 64 groups of 4 pixels per patch
  for (int grp = slot; grp < NGRP; grp += NS) {
    const int py = grp / (CI_P / CI_G), px = (grp - py * (CI_P / CI_G)) * CI_G;
    float xr[3][20];                                      // per tap row: (4 + 2) pixels x 3 channels = 18 values (+ 2 read, unused)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = halo + (py + ky) * CI_PITCH + px * 3;        // 48 px bytes + 224 row bytes: 16-byte aligned
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
        xr[ky][4 * q] = v.x; xr[ky][4 * q + 1] = v.y; xr[ky][4 * q + 2] = v.z; xr[ky][4 * q + 3] = v.w;
      }
      const float2 v2 = *reinterpret_cast<const float2*>(row + 16);
      xr[ky][16] = v2.x; xr[ky][17] = v2.y;
    }
    float4 acc[CI_G];
#pragma unroll
    for (int g = 0; g < CI_G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const float4 w = wv[ky * 9 + j];
#pragma unroll
        for (int g = 0; g < CI_G; ++g) {
          const float xv = xr[ky][g * 3 + j];             // pixel px + g: its (kx, ci) = j value of this tap row
          acc[g].x = __builtin_fmaf(xv, w.x, acc[g].x);
          acc[g].y = __builtin_fmaf(xv, w.y, acc[g].y);
          acc[g].z = __builtin_fmaf(xv, w.z, acc[g].z);
          acc[g].w = __builtin_fmaf(xv, w.w, acc[g].w);
        }
      }
    }
    const int oy = oy0 + py;
#pragma unroll
    for (int g = 0; g < CI_G; ++g) {
      const int ox = ox0 + px + g;
      if (oy < p.Hout && ox < p.Wout) {
        const float4 v = make_float4(acc[g].x + bv.x, acc[g].y + bv.y, acc[g].z + bv.z, acc[g].w + bv.w);
        *reinterpret_cast<float4*>(outz + ((long long)oy * p.Wout + ox) * p.ldo + n) = v;
        if (want_stats) {
          s1[0] += (double)v.x; s2[0] += (double)v.x * (double)v.x;
          s1[1] += (double)v.y; s2[1] += (double)v.y * (double)v.y;
          s1[2] += (double)v.z; s2[2] += (double)v.z * (double)v.z;
          s1[3] += (double)v.w; s2[3] += (double)v.w * (double)v.w;
        }
      }
    }
  }
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[((size_t)slot * Cout + n + j) * 2] = s1[j];
      red[((size_t)slot * Cout + n + j) * 2 + 1] = s2[j];
    }
    __syncthreads();
    for (int c = tid; c < Cout; c += CI_NT) {
      double a = 0.0, b = 0.0;
      for (int s = 0; s < NS; ++s) {                      // fixed order over the pixel slots
        a += red[((size_t)s * Cout + c) * 2];
        b += red[((size_t)s * Cout + c) * 2 + 1];
      }
      double* dst = p.stats + (((size_t)zo * gridDim.x + blockIdx.x) * Cout + c) * 2;
      dst[0] = a;
      dst[1] = b;
    }
  }
}

// =====================================================================================================================
// conv_in_mfma_kernel (round 5).  The stencil above is VALU-bound, not HBM-bound: 27 x 128 FMAs per pixel are 123 us of packed
// FMAs per launch at B = 32 even at full issue rate, plus 73 us of half-rate f64 statistics, against 195 us for its 1.07 GB of
// output at the chip's measured store rate (5.5 TB/s, profiles/r04m_*): 318 us, 3.5 TB/s.  Here the 27 taps
// are the K dimension of ONE K = 32 matrix step (k = tap * 3 + ci, 27..31 zero) on v_mfma_f32_16x16x32_f16 with the engine's exact
// two-term operand split (x_lo w_hi + x_hi w_hi + x_hi w_lo, fp32 accumulate: 48 instructions per wave instead of 1 728 FMAs per
// lane), in the main tile's shape -- 8 waves x (64 pixels x 64 channels) on a 16 x 16 patch x 128 channels -- and leaves through
// the main tile's epilogue (wave-private LDS slabs -> float4 stores of 256 contiguous bytes per pixel, f64 statistics).  What is
// left is the store stream.
//   A: every lane gathers its own fragments from the fp32 halo in LDS (4 row blocks x 8 values) and splits them in registers.
//   B: the lane's 8 weights per column block come from the fp32 [tap][3][Cout] image (L2-hot), scaled by 2^10 before the split
//      (the split's lo term stays normal, as for every other weight image), undone by its inverse in the epilogue; the power of two comes
//      from max|w| at upload (GemmArgs.cin_wmul; 2^10 for PyTorch-default conv_in weights, |w| < 0.2).
// Results differ from the fp32 stencil by the dropped x_lo w_lo terms (2^-22 relative per product); batch-invariant bit for bit.
// =====================================================================================================================
typedef _Float16 cih8 __attribute__((ext_vector_type(8)));
typedef float cif4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ci_split8(const float (&v)[8], cih8& hi, cih8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float c = __builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
    const _Float16 h = (_Float16)c;
    hi[j] = h;
    lo[j] = (_Float16)(c - (float)h);
  }
}

constexpr int CIM_NT = 512, CIM_BN = 128, CIM_EP = 68;

// Persistent (second measurement of round 5: one tile per workgroup ran at 300 us against the stencil's 318 -- every workgroup paid
// the weight fragments (32 strided loads + splits per lane), the halo's load latency and the launch ramp again): a workgroup now
// keeps its B fragments in registers and walks tiles t = blockIdx.x, + gridDim.x, ... of its N block; the halo of tile t + 1 is
// requested before tile t's matrix step and written to LDS behind it, so that only the first tile of a workgroup waits for memory.
__global__ void __launch_bounds__(CIM_NT, 4) conv_in_mfma_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float halo[CI_T * CI_PITCH];
  __shared__ __attribute__((aligned(16))) float slabs[8 * 16 * CIM_EP];      // one 16-pixel x 64-channel slab per wave
  __shared__ double red[4 * CIM_BN * 2];                                     // [wave row][channel][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int tiles_x = (p.Wout + CI_P - 1) / CI_P, tiles_y = (p.Hout + CI_P - 1) / CI_P, tiles_img = tiles_x * tiles_y;
  const int ntiles = tiles_img * p.Z;
  const int n0 = blockIdx.y * CIM_BN;
  const int Cout = p.Cout;
  // ---- B fragments (once per workgroup, kept in LDS: 16 KB): lane = (column c16 of a 16-channel block, k group kq):
  // w[k = 8 kq + j][n] * 2^10, two-term split; the waves of wave row 0 build the two column halves, every wave reads its half per tile
  __shared__ __attribute__((aligned(16))) cih8 bfrag[2 * 4 * 2 * 64];         // [wn][tn][hi | lo][lane]
  const int r16 = lane & 15, kq = lane >> 4;
  // weight scale: a power of two chosen at upload from max|w| (engine.hip pack_x3's rule), so that no |w| saturates the f16 split
  const float wmul = p.cin_wmul > 0.f ? p.cin_wmul : 1024.f;
  if (wm == 0) {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wn * 64 + tn * 16 + r16;
      const int nn = n < Cout ? n : Cout - 1;
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // unconditional loads from clamped addresses, zeroed by a select (no divergent branches)
        const int k = 8 * kq + j;
        const float v = p.w[(long long)(k < 27 ? k : 26) * p.ldb + nn] * wmul;
        wv[j] = (k < 27 && n < Cout) ? v : 0.f;
      }
      cih8 h, l;
      ci_split8(wv, h, l);
      bfrag[((wn * 4 + tn) * 2 + 0) * 64 + lane] = h;
      bfrag[((wn * 4 + tn) * 2 + 1) * 64 + lane] = l;
    }
  }
  // element j of a lane's k group: halo[(py + ky) * PITCH + px * 3 + r] with k = 8 kq + j = ky * 9 + r; k >= 27 (k group 3, j >= 3)
  // reads slot ZERO_SLOT (the unused pad word 54 of halo row 0, kept 0)
  constexpr int ZERO_SLOT = 54;
  if (tid == 0) halo[ZERO_SLOT] = 0.f;
  // halo elements of this thread: i = tid, tid + 512 (< 18 * 18 * 3 = 972); the index arithmetic starts from a laundered thread id
  // each time (hipcc otherwise keeps every derived offset alive across the matrix step and the epilogue and spills)
  constexpr int NH = (CI_T * CI_T * 3 + CIM_NT - 1) / CIM_NT;   // 2
  float hv[NH];
  auto load_halo = [&](int t) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int zo = t / tiles_img, tt = t - zo * tiles_img, ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const float* __restrict__ a0 = p.a0 + (long long)zo * p.a0_zo;
#pragma unroll
    for (int e = 0; e < NH; ++e) {   // unconditional load from a clamped address, zeroed by a select
      const int i = t_ + e * CIM_NT;
      const int c = i % 3, pix = i / 3, iy = pix / CI_T, ix = pix - iy * CI_T;
      const int gy = ty * CI_P + iy - 1, gx = tx * CI_P + ix - 1;
      const bool ok = i < CI_T * CI_T * 3 && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
      const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
      const float v = a0[((long long)cy * p.Win + cx) * p.lda0 + c];
      hv[e] = ok ? v : 0.f;
    }
  };
  auto write_halo = [&]() {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int e = 0; e < NH; ++e) {
      const int i = t_ + e * CIM_NT;
      const int c = i % 3, pix = i / 3, iy = pix / CI_T, ix = pix - iy * CI_T;
      if (i < CI_T * CI_T * 3) halo[iy * CI_PITCH + ix * 3 + c] = hv[e];
    }
  };
  float* const ep = slabs + wave * (16 * CIM_EP);
  const bool want_stats = (p.stats != nullptr);
  const float ALPHA = 1.0f / wmul;      // exact: a power of two

  int t = blockIdx.x;
  if (t < ntiles) load_halo(t);
  for (; t < ntiles; t += gridDim.x) {
    const int zo = t / tiles_img, tt = t - zo * tiles_img, ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int oy0 = ty * CI_P, ox0 = tx * CI_P;
    write_halo();                                    // (the previous tile's fragment gathers finished before its barrier B)
    __syncthreads();                                 // barrier A: halo(t) visible; `red` of the previous tile consumed
    if (t + (int)gridDim.x < ntiles) load_halo(t + gridDim.x);   // in flight under the matrix step and the epilogue
    // ---- A fragments: row r16 of row block tm = patch pixel (py = wm * 4 + tm, px = r16); k = ky * 9 + (kx * 3 + ci) ----
    int aoff[8];
    {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int ar = l_ & 15, akq = l_ >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * akq + j, ky = k / 9, r = k - ky * 9;
        aoff[j] = (k < 27) ? ky * CI_PITCH + ar * 3 + r : -1;
      }
    }
    cif4 acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = cif4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const int py = wm * 4 + tm;
      float av[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = halo[aoff[j] >= 0 ? py * CI_PITCH + aoff[j] : ZERO_SLOT];
      cih8 ah, al;
      ci_split8(av, ah, al);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const cih8 bh = bfrag[((wn * 4 + tn) * 2 + 0) * 64 + lane], bl = bfrag[((wn * 4 + tn) * 2 + 1) * 64 + lane];
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[tm][tn], 0, 0, 0);
      }
    }
    __syncthreads();                                 // barrier B: every wave has read halo(t); the next write_halo may overwrite it
    // ---- epilogue: the K32 main tile's (conv_f16x3.hip): C/D layout col = lane & 15 (channel), rows 4 (lane >> 4) + r (pixels) ->
    // wave-private slab -> float4 = four consecutive channels of one pixel ----
    int el = lane;
    asm volatile("" : "+v"(el));
    const int g = el >> 4, er16 = el & 15, c4 = el & 15, prow = el >> 4;   // 16 channel quads x 4 pixel rows per pass, 4 passes per row block
    const int nq = n0 + wn * 64 + c4 * 4;
    const bool nok = nq < Cout;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && nok) bv = *reinterpret_cast<const float4*>(p.bias + nq);
    float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
    const bool full = (n0 + CIM_BN <= Cout) && (oy0 + CI_P <= p.Hout) && (ox0 + CI_P <= p.Wout);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    auto stat4 = [&](const float4& v) {
      s1[0] += (double)v.x; s2[0] += (double)v.x * (double)v.x;
      s1[1] += (double)v.y; s2[1] += (double)v.y * (double)v.y;
      s1[2] += (double)v.z; s2[2] += (double)v.z * (double)v.z;
      s1[3] += (double)v.w; s2[3] += (double)v.w * (double)v.w;
    };
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(4 * g + r) * CIM_EP + tn * 16 + er16] = acc[tm][tn][r] * ALPHA;
      asm volatile("" ::: "memory");
      const int oy = oy0 + wm * 4 + tm;
      if (full) {   // straight-line stores (no per-element predicate: hipcc counts vmcnt instead of draining it before every store)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = prow + 4 * i;
          const float4 a = *reinterpret_cast<const float4*>(ep + m * CIM_EP + c4 * 4);
          const float4 v = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
          *reinterpret_cast<float4*>(outz + ((long long)oy * p.Wout + ox0 + m) * p.ldo + nq) = v;
          if (want_stats) stat4(v);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = prow + 4 * i;
          const float4 a = *reinterpret_cast<const float4*>(ep + m * CIM_EP + c4 * 4);
          const float4 v = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
          if (nok && oy < p.Hout && ox0 + m < p.Wout) {
            *reinterpret_cast<float4*>(outz + ((long long)oy * p.Wout + ox0 + m) * p.ldo + nq) = v;
            if (want_stats) stat4(v);
          }
        }
      }
      asm volatile("" ::: "memory");
    }
    if (want_stats) {   // fixed order: in-lane -> the four lanes (g) that hold a channel quad -> the wave rows through LDS
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += __shfl_xor(s1[j], 16); s2[j] += __shfl_xor(s2[j], 16);
        s1[j] += __shfl_xor(s1[j], 32); s2[j] += __shfl_xor(s2[j], 32);
      }
      if (el < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double* d = red + ((size_t)wm * CIM_BN + wn * 64 + c4 * 4 + j) * 2;
          d[0] = s1[j];
          d[1] = s2[j];
        }
      }
      __syncthreads();
      for (int c = tid; c < CIM_BN; c += CIM_NT) {
        if (n0 + c < Cout) {
          double a = 0.0, b = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            a += red[((size_t)w * CIM_BN + c) * 2];
            b += red[((size_t)w * CIM_BN + c) * 2 + 1];
          }
          double* dst = p.stats + (((size_t)zo * tiles_img + tt) * Cout + n0 + c) * 2;   // rows = 16 x 16 patches per image
          dst[0] = a;
          dst[1] = b;
        }
      }
    }
  }
}

// A/B switch: ASYRP_CONV_IN_MFMA=0 keeps the fp32 stencil (exact fp32 products, VALU-bound)
static bool conv_in_mfma_enabled() {
  static const bool on = [] { const char* e = ab_env("ASYRP_CONV_IN_MFMA"); return !(e && e[0] == '0'); }();
  return on;
}

bool conv_in_supported(const GemmArgs& a) {
  if (!(a.ks == 3 && a.stride == 1 && !a.ups && a.pad == 1 && a.Cin == 3 && !a.a1 && !a.pscale && !a.silu && !a.resid && !a.chan_add)) return false;
  if (!a.w || a.bT || a.ZI > 1 || a.s0 || a.sk > 1 || a.poly || a.o16h) return false;
  if ((a.Cout & 3) || a.Cout < 4 || a.Cout > 256 || (CI_NT % (a.Cout >> 2)) != 0 || (a.ldb & 3) || (a.ldo & 3)) return false;
  if ((((uintptr_t)a.w) | ((uintptr_t)a.out) | ((uintptr_t)a.bias)) & 15) return false;
  return a.Hin == a.Hout && a.Win == a.Wout;
}

int conv_in_stat_blocks(const GemmArgs& a) { return ((a.Hout + CI_P - 1) / CI_P) * ((a.Wout + CI_P - 1) / CI_P); }

hipError_t launch_conv_in(const GemmArgs& a, hipStream_t s) {
  if (!conv_in_supported(a)) return hipErrorInvalidValue;
  if (conv_in_mfma_enabled()) {
    // persistent: two resident workgroups per CU and N block walk the (image, patch) tiles
    const long long tiles = (long long)conv_in_stat_blocks(a) * a.Z;
    const int ny = (a.Cout + CIM_BN - 1) / CIM_BN;
    const int gx = (int)(tiles < 512 / ny ? tiles : 512 / ny);
    hipLaunchKernelGGL(conv_in_mfma_kernel, dim3(gx, ny, 1), dim3(CIM_NT), 0, s, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(conv_in_kernel, dim3(conv_in_stat_blocks(a), 1, a.Z), dim3(CI_NT), 0, s, a);
  return hipGetLastError();
}

}  // namespace asyrp
