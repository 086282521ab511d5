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
  hipLaunchKernelGGL(conv_in_kernel, dim3(conv_in_stat_blocks(a), 1, a.Z), dim3(CI_NT), 0, s, a);
  return hipGetLastError();
}

}  // namespace asyrp
