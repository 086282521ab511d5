// conv_out.hip — the last convolution of the UNets (conv_out 128 -> 3, models/ddpm/diffusion.py:426-430,575-578; iDDPM out.2
// 128 -> 6, models/improved_ddpm/unet.py:654-658), fused with its GroupNorm-apply + SiLU prologue.
//
// On the implicit-GEMM tiles this layer pads its 3 (6) output channels to a 32-wide N tile and pays 9 taps x Cin of matrix work
// for them: 10x the useful products, which made an HBM-sized layer MFMA-bound (0.58-0.64 ms per launch at B = 32 against a
// 0.2 ms HBM floor).  Here the taps move into N instead:
//     Z[p][tap*Cout + co] = sum_ci act(x)[p][ci] * w[co][ci][tap]        one 1x1 GEMM, N = 9*Cout = 27 <= 32, K = Cin
//     y[q][co]            = bias[co] + sum_tap Z[q + offset(tap)][tap*Cout + co]        a 9-term stencil over the halo pixels
// so the matrix work drops 6.5x (f16x3 split products as everywhere else): 365 us instead of 587 us per launch at B = 32
// (rocprofv3, same box).  The 6-channel iDDPM head needs two N tiles and measured equal to the tile path, which it keeps.
// One workgroup = 4 waves on an output patch of one image; Z goes through LDS (aliasing the staging buffers) for the stencil.
//   * 14 x 14 patch (round 6, the 3-channel head): the 16 x 16 halo is exactly 256 pixels = 8 MFMA row tiles (two per wave) = 512
//     staging items (two per thread, none idle): 2.61 staged items per output pixel instead of the 8 x 16 patch's 4.0 (1.31 x halo
//     instead of 1.41 x, and no half-empty second staging round on 360 items); every lane fetches its B fragments of a chunk from the
//     L2-hot weight image beside the chunk's activation loads, so the weights need no LDS: 34 KB, four workgroups per CU.
//     323 -> 298 us per launch at B = 32, same box, interleaved (profiles/r06m_*); bit-identical to the 8 x 16 form (same products in
//     the same order per output; sha256 of the outputs on five shapes, gpurun_out/r06n/bits.txt).
//   * 8 x 16 patch (10 x 18 halo pixels = 6 row tiles), the whole weight image (<= 32 KB) in LDS for the launch: the 6-channel iDDPM
//     head on two N tiles, and maps smaller than a 14 x 14 patch.
// A 16 x 16 patch (halo overhead 1.27x) measured slower, 41 vs 45 TFLOP/s inside the edit: 58 KB of LDS leaves two workgroups per CU
// instead of four (gpurun_out/r4prep_e, round-4 prep).  Round 6 also measured a persistent form and a form fed straight from global
// memory without LDS staging: both slower (scripts/experiments/conv_out_*_NEGATIVE.patch).
#include <cstdlib>
#include <type_traits>
#include "kernels.h"

namespace asyrp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Patch geometry.  CoPatch<8, 16>: the round-2 patch (10 x 18 = 180 halo pixels: 6 row tiles on 4 waves, 360 staging items on 256
// threads).  CoPatch<14, 14> (round 6): the 16 x 16 halo is exactly 256 pixels = 8 row tiles (two per wave) = 512 staging items (two
// per thread, none idle): 2.61 staged items per output pixel instead of 4.0 (1.31 x halo instead of 1.41 x, and no half-empty
// second staging round) -- the kernel's time is its staging arithmetic.
// WLDS: the weight image lives in LDS for the launch (the 8 x 16 form), or -- false -- every lane fetches its two 16-byte B fragments
// of a chunk from the L2-hot image beside the chunk's activation loads (the 14 x 14 form: 34 KB of LDS instead of 49, four
// workgroups per CU instead of three).
template <int PH_, int PW_, bool WLDS_ = true>
struct CoPatch {
  static constexpr int PH = PH_, PW = PW_, TH = PH + 2, TW = PW + 2, NPIX = TH * TW, MT = (NPIX + 31) / 32;
  static constexpr bool WLDS = WLDS_;
};
using CoP816 = CoPatch<8, 16>;
using CoP1414 = CoPatch<14, 14, false>;
using CoP1414W = CoPatch<14, 14, true>;
constexpr float CO_HMAX = 65504.0f;

__device__ __forceinline__ float co_silu(float v) {
  const float e = __expf(-v);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ void co_split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 s;
    s[0] = __builtin_amdgcn_fmed3f(v[j], -CO_HMAX, CO_HMAX);
    s[1] = __builtin_amdgcn_fmed3f(v[j + 1], -CO_HMAX, CO_HMAX);
    const h2 h = __builtin_convertvector(s, h2);
    f2 r;
    r[0] = s[0] - (float)h[0];
    r[1] = s[1] - (float)h[1];
    const h2 l = __builtin_convertvector(r, h2);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

// TN = N tiles of 32 columns (9*Cout <= 32*TN).  Weight image = launch_pack_f16x3 of the equivalent 1x1 conv
// w1[n = tap*Cout + co][ci]: [chunk][unit 4][cout_pad][8 halfs].
// NP = matrix products per term: 3 (two-term split) or 1 (conv_math "f16": x_hi * w_hi only).
template <int TN, int NP = 3, class PT = CoP816>
__global__ void __launch_bounds__(256, 2) conv_out_kernel(const GemmArgs p) {
  constexpr int CO_PH = PT::PH, CO_PW = PT::PW, CO_TW = PT::TW, CO_NPIX = PT::NPIX, CO_MT = PT::MT;
  constexpr bool WLDS = PT::WLDS;
  constexpr int NT = 256, NW = 4, BN = 32 * TN;
  constexpr int A_BYTES = CO_NPIX * 64;                 // [4 units][NPIX][16 B]
  constexpr int ZLD = BN + 1;                           // padded row of the Z tile (floats)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = p.Cin >> 4;
  const int W_BYTES = WLDS ? nch * 4 * BN * 16 : 0;
  char* const Ws = smem;                                // [chunk][4][BN][16 B]
  char* const As = smem + W_BYTES;                      // 2 x A_BYTES
  float* const Zs = reinterpret_cast<float*>(smem);     // after the K loop: [CO_MT*32][ZLD] floats (aliases Ws / As)
  float* const Ps = reinterpret_cast<float*>(smem + W_BYTES + 2 * A_BYTES);   // scale[Cin] | shift[Cin] of this image (a global load
                                                        // inside the staging pass would sit behind the prefetched chunk in vmcnt order)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroups go to the 8 XCDs round-robin by linear id.  xmap: XCD k takes a contiguous range of (image, patch) pairs, so the
  // patches of an image -- which share their halo rows and columns -- meet in ONE L2 (with the identity map a 14 x 14 patch's
  // neighbours sit on other XCDs and every halo pixel is fetched from HBM again: 1.41 x the algorithmic bytes by counters)
  int lin = blockIdx.x + gridDim.x * blockIdx.z;
  if (p.xmap) lin = (lin & 7) * ((int)(gridDim.x * gridDim.z) >> 3) + (lin >> 3);
  const int zo = lin / (int)gridDim.x, patch = lin - zo * (int)gridDim.x;
  const int tiles_x = (p.Wout + CO_PW - 1) / CO_PW;
  const int ty = patch / tiles_x, tx = patch - ty * tiles_x;
  const int oy0 = ty * CO_PH, ox0 = tx * CO_PW;
  const float* __restrict__ a0 = p.a0 + (long long)zo * p.a0_zo;
  const float* __restrict__ ps = p.pscale + (long long)zo * p.Cin;
  const float* __restrict__ psh = p.pshift + (long long)zo * p.Cin;

  for (int i = tid; i < p.Cin; i += NT) {
    Ps[i] = ps[i];
    Ps[p.Cin + i] = psh[i];
  }
  // ---- weights -> LDS (once) ----
  if (WLDS) {
    const char* wpk = reinterpret_cast<const char*>(p.wpk);
    for (int i = tid; i < nch * 4 * BN; i += NT) {
      const int n = i % BN, cu = i / BN;                 // cu = chunk*4 + unit
      *reinterpret_cast<float4*>(Ws + (size_t)i * 16) = *reinterpret_cast<const float4*>(wpk + ((size_t)cu * p.cout_pad + n) * 16);
    }
  }
  // ---- staging map: work item = (halo pixel, 8-channel half) ----
  constexpr int NU = CO_NPIX * 2, NA = (NU + NT - 1) / NT;
  const int hf = tid & 1;
  int aoff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tid + i * NT, pix = u >> 1;
    int off = -2;
    if (u < NU) {
      const int iy = pix / CO_TW, ix = pix - iy * CO_TW;
      const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
      off = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? gy * p.Win + gx : -1;
    }
    aoff[i] = off;
  }
  // Two register sets: the loads of chunk c + 2 are issued while chunk c + 1 is staged and chunk c multiplied, so two chunks of the
  // tile are in flight per workgroup (round 3's counters: 62 % of this kernel's wave cycles were parked behind one chunk's loads).
  // Loads are unconditional (padding pixels and surplus work items read pixel 0 and are zeroed at staging time) and the set
  // index is a compile-time constant: hipcc then counts them and waits with vmcnt(2 * NA) for the older set only.
  float4 areg[2][NA][2];
  auto gload = [&](int chunk, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const float* base = a0 + chunk * 16 + hf * 8;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const float* src = base + (long long)max(aoff[i], 0) * p.lda0;
      areg[set][i][0] = *reinterpret_cast<const float4*>(src);
      areg[set][i][1] = *reinterpret_cast<const float4*>(src + 4);
    }
  };
  // !WLDS: this lane's B fragments (column lane & 31, k group lane >> 5) of a chunk, hi and lo planes, straight from the packed image;
  // chunk c + 2's are requested right AFTER chunk c's matrix instructions have consumed the register set they land in
  h8 breg[2][TN][2];
  auto gloadB = [&](int chunk, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    if (!WLDS) {
      const char* b = reinterpret_cast<const char*>(p.wpk) + (((size_t)chunk * 4 + (lane >> 5)) * p.cout_pad + (lane & 31)) * 16;
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        breg[set][n][0] = *reinterpret_cast<const h8*>(b + n * 32 * 16);
        if (NP == 3) breg[set][n][1] = *reinterpret_cast<const h8*>(b + n * 32 * 16 + 2 * (size_t)p.cout_pad * 16);
      }
    }
  };
  auto stage = [&](int chunk, int buf, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const int c = chunk * 16 + hf * 8;
    const float4 s0 = *reinterpret_cast<const float4*>(Ps + c), s1 = *reinterpret_cast<const float4*>(Ps + c + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(Ps + p.Cin + c), h1 = *reinterpret_cast<const float4*>(Ps + p.Cin + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (aoff[i] == -2) continue;
      float t[8] = {areg[set][i][0].x, areg[set][i][0].y, areg[set][i][0].z, areg[set][i][0].w,
                    areg[set][i][1].x, areg[set][i][1].y, areg[set][i][1].z, areg[set][i][1].w};
      if (aoff[i] >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = co_silu(__builtin_fmaf(t[j], sc[j], sh[j]));
      } else {   // zero padding of the 3x3 conv (the clamped load above fetched pixel 0)
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = 0.f;
      }
      h8 hi, lo;
      co_split8(t, hi, lo);
      const int pix = (tid + i * NT) >> 1;
      char* dst = As + buf * A_BYTES + (hf * CO_NPIX + pix) * 16;
      *reinterpret_cast<h8*>(dst) = hi;
      if (NP == 3) *reinterpret_cast<h8*>(dst + 2 * CO_NPIX * 16) = lo;
    }
  };

  // ---- MFMA rows: wave w owns row tiles {w, w + 4} (< CO_MT); A-fragment row = halo pixel (clamped for the padding rows) ----
  const int kh = lane >> 5;
  constexpr int TMW = (CO_MT + NW - 1) / NW;            // 2
  int arow[TMW];
#pragma unroll
  for (int t = 0; t < TMW; ++t) arow[t] = min((wave + NW * t) * 32 + (lane & 31), CO_NPIX - 1);
  f32x16 acc[TMW][TN];
#pragma unroll
  for (int t = 0; t < TMW; ++t)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  auto mma_chunk = [&](int chunk, auto set_c) {
    constexpr int set = decltype(set_c)::value;          // == chunk & 1
    const char* A = As + (chunk & 1) * A_BYTES + kh * CO_NPIX * 16;
    const char* B = Ws + ((size_t)chunk * 4 + kh) * BN * 16 + (lane & 31) * 16;
#pragma unroll
    for (int t = 0; t < TMW; ++t) {
      if (wave + NW * t >= CO_MT) continue;             // wave-uniform
      const h8 ah = *reinterpret_cast<const h8*>(A + arow[t] * 16);
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const h8 bh = WLDS ? *reinterpret_cast<const h8*>(B + n * 32 * 16) : breg[set][n][0];
        if (NP == 3) {
          const h8 al = *reinterpret_cast<const h8*>(A + arow[t] * 16 + 2 * CO_NPIX * 16);
          const h8 bl = WLDS ? *reinterpret_cast<const h8*>(B + n * 32 * 16 + 2 * BN * 16) : breg[set][n][1];
          acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t][n], 0, 0, 0);
          acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t][n], 0, 0, 0);
          acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t][n], 0, 0, 0);
        } else {
          acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t][n], 0, 0, 0);
        }
      }
    }
  };
  // chunk c lives in register set c & 1 until it is staged; loads past the last chunk re-read it (static load counts)
  gload(0, S0{});
  gloadB(0, S0{});
  gload(min(1, nch - 1), S1{});
  gloadB(min(1, nch - 1), S1{});
  __syncthreads();   // the scale / shift rows in Ps were written by other waves (without this barrier waves 1-3 staged chunk 0 from
                     // whatever the LDS held: a few-per-mille run-to-run difference in some tiles of some images, found in round 4
                     // by running the same batch twice -- scripts/batch_invariance_probe.py)
  stage(0, 0, S0{});
  __syncthreads();
  for (int chunk = 0; chunk < nch; chunk += 2) {   // nch is even (launcher: Cin % 32 == 0)
    // even chunk: set 0 is free (staged before the last barrier) -> chunk + 2; multiply chunk; stage chunk + 1 from set 1.
    // Every iteration issues the same loads and the same staging pass (past the end: the last chunk again, into the buffer nobody
    // reads any more), so the body is straight-line code with static counts.
    gload(min(chunk + 2, nch - 1), S0{});
    mma_chunk(chunk, S0{});
    gloadB(min(chunk + 2, nch - 1), S0{});
    stage(chunk + 1, 1, S1{});
    __syncthreads();
    gload(min(chunk + 3, nch - 1), S1{});
    mma_chunk(chunk + 1, S1{});
    gloadB(min(chunk + 3, nch - 1), S1{});
    stage(min(chunk + 2, nch - 1), 0, S0{});
    __syncthreads();
  }

  // ---- Z -> LDS (accumulator layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*kh), then the 9-term stencil ----
#pragma unroll
  for (int t = 0; t < TMW; ++t) {
    if (wave + NW * t >= CO_MT) continue;
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wave + NW * t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        Zs[row * ZLD + n * 32 + (lane & 31)] = acc[t][n][r];
      }
  }
  __syncthreads();
  const int Cout = p.Cout;
  float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
  for (int o = tid; o < CO_PH * CO_PW * Cout; o += NT) {
    const int pix = o / Cout, co = o - pix * Cout;
    const int py = pix / CO_PW, px = pix - py * CO_PW;
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy >= p.Hout || ox >= p.Wout) continue;
    float s = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      s += Zs[((py + ky) * CO_TW + px + kx) * ZLD + tap * Cout + co];
    }
    outz[((long long)oy * p.Wout + ox) * p.ldo + co] = s * p.alpha + (p.bias ? p.bias[co] : 0.f);
  }
}

template <class PT>
static size_t conv_out_smem_p(int Cin, int TN) {
  const size_t w = PT::WLDS ? (size_t)(Cin / 16) * 4 * 32 * TN * 16 : 0, a = 2 * (size_t)PT::NPIX * 64;
  const size_t z = (size_t)PT::MT * 32 * (32 * TN + 1) * sizeof(float), ps = 2 * (size_t)Cin * sizeof(float);
  return (w + a + ps > z) ? w + a + ps : z;
}
size_t conv_out_smem(int Cin, int TN) { return conv_out_smem_p<CoP816>(Cin, TN); }
// A/B switch (profiling build): ASYRP_CONV_OUT_PATCH=816 keeps the 8 x 16 patch for the 3-channel head
static bool conv_out_patch1414() {
  static const bool on = [] { const char* e = ab_env("ASYRP_CONV_OUT_PATCH"); return !(e && e[0] == '8'); }();
  return on;
}

bool conv_out_two_tiles() {
  // the 6-channel iDDPM head on two N tiles of this kernel (round 4: on by default -- with the loads two chunks ahead it measured
  // 55.5 vs 46.0 TFLOP/s on the AFHQ head (Cin = 128), +0.2 % on that edit, profiles/r04a_*); ASYRP_CONV_OUT6=0 puts the head back on
  // the implicit-GEMM tile
  static const bool on = [] { const char* e = ab_env("ASYRP_CONV_OUT6"); return !(e && e[0] == '0'); }();
  return on;
}
bool conv_out_supported(const GemmArgs& a) {
  if (!(a.ks == 3 && a.stride == 1 && !a.ups && !a.a1 && a.wpk && a.pscale && a.silu && !a.resid && !a.chan_add && !a.stats)) return false;
  if (!((a.Cin & 31) == 0 && a.Cin <= 256 && (a.lda0 & 3) == 0 && a.Hin == a.Hout && a.Win == a.Wout)) return false;
  if (a.Cout * 9 <= 32) return true;                                  // one N tile (Cout = 3)
  // two N tiles (Cout = 6): where two workgroups still fit a CU.  AFHQ head (Cin = 128): weights in LDS.  At Cin = 256 (ImageNet-ADM
  // head) the weights-in-LDS forms need ~90 KB, one workgroup per CU, and measured 39 TFLOP/s, below the implicit-GEMM tile
  // (ADVICE r04); since round 6 that head runs on the 14 x 14 patch with its B fragments from the L2-hot image (67 KB, two per CU)
  if (!(a.Cout * 9 <= 64 && conv_out_two_tiles())) return false;
  return conv_out_smem(a.Cin, 2) <= 64 * 1024 || (conv_out_patch1414() && a.Hout >= 14 && a.Wout >= 14);
}

template <int NP, int TN = 1, class PT = CoP816>
static hipError_t launch_conv_out_np(const GemmArgs& a, hipStream_t s) {
  const size_t smem = conv_out_smem_p<PT>(a.Cin, TN);
  dim3 grid(((a.Hout + PT::PH - 1) / PT::PH) * ((a.Wout + PT::PW - 1) / PT::PW), 1, a.Z), block(256);
  if (smem > 64 * 1024) {   // once per process and device (idempotent flag, as launch_k32)
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_out_kernel<TN, NP, PT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
  }
  GemmArgs ax = a;
  ax.xmap = (xcd_map_enabled() && ((long long)grid.x * grid.z) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((conv_out_kernel<TN, NP, PT>), grid, block, smem, s, ax);
  return hipGetLastError();
}

hipError_t launch_conv_out(const GemmArgs& a, hipStream_t s) {
  if (!conv_out_supported(a)) return hipErrorInvalidValue;
  if (a.Cout * 9 > 32) {   // two N tiles (the 6-channel iDDPM head): the 14 x 14 patch with the weights in LDS (67 KB: two workgroups per CU, as before)
    if (conv_out_patch1414() && a.Hout >= 14 && a.Wout >= 14) {
      if (conv_out_smem(a.Cin, 2) > 64 * 1024)   // Cin = 256: no room for the weights in LDS
        return a.np == 1 ? launch_conv_out_np<1, 2, CoP1414>(a, s) : launch_conv_out_np<3, 2, CoP1414>(a, s);
      return a.np == 1 ? launch_conv_out_np<1, 2, CoP1414W>(a, s) : launch_conv_out_np<3, 2, CoP1414W>(a, s);
    }
    return a.np == 1 ? launch_conv_out_np<1, 2>(a, s) : launch_conv_out_np<3, 2>(a, s);
  }
  // one N tile (the 3-channel head): the 14 x 14 patch where the layer has at least a patch of pixels, else the 8 x 16 one
  if (conv_out_patch1414() && a.Hout >= 14 && a.Wout >= 14)
    return a.np == 1 ? launch_conv_out_np<1, 1, CoP1414>(a, s) : launch_conv_out_np<3, 1, CoP1414>(a, s);
  return a.np == 1 ? launch_conv_out_np<1>(a, s) : launch_conv_out_np<3>(a, s);
}

}  // namespace asyrp
