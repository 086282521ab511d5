// gemm1x1.hip -- the 1x1 convolutions (q|k|v and proj_out of the attention blocks, nin_shortcut / skip_connection, the DeltaBlock)
// on v_mfma_f32_16x16x32_f16 with NO LDS staging and NO barrier in the K loop (round 3).
//
// Why a kernel of its own.  On the implicit-GEMM tiles a 1x1 layer is a K loop of Cin/16 single-tap steps, each of which needs
// fresh A and fresh B bytes: with K = 512 there is nothing to re-use across steps, every step waits for an LDS-DMA weight slice and
// a staged activation tile behind a barrier, and the launch ran at ~3 us per step whatever the tile shape
// (profiles/rd3o_1x1_gemm_tile_sweep.txt: 97 us for 512 -> 1536 @16x16 at B = 32, 107 TFLOP/s inside the edit).  Such a layer is
// bound by how many bytes a CU keeps in flight, so it gets the recipe that fixed the attention kernel (attention.hip):
//   * weights packed once in MFMA FRAGMENT ORDER ([16-channel tile][K = 32 step][hi | lo][lane][8 halfs], gemm1x1_pack_kernel): a
//     wave's B operand is one coalesced 1-KiB load straight into registers;
//   * activations are fp32 NHWC: lane (pixel r16, channel group g) loads its 8 consecutive channels (32 B; the four groups of a
//     pixel cover one 128-B line), applies the GroupNorm scale/shift (+ SiLU) and splits to f16 hi/lo IN REGISTERS -- which is
//     already the A-operand layout of the instruction (row = pixel, k group = g): no LDS, no barrier;
//   * a ring of D = 2 steps of raw A and of B in flight per wave, load counts static (unconditional, clamped) and pinned with
//     scheduling barriers so that hipcc emits counted waits (see attn_planes_kernel).
// One workgroup = 256 (or 128) consecutive pixels of one image x 128 output channels; 8 (4) independent waves (WM x 2), a wave = 64 pixels x 64
// channels = 4 x 4 accumulator blocks.  The two waves that share pixels convert them twice (VALU is idle here); L1 serves the re-reads.
// Products and their order per accumulator are those of the other K32 kernels (x_lo*w_hi, x_hi*w_hi, x_hi*w_lo per K = 32 step).
// Epilogue: bias + per-image channel vector + residual, GroupNorm partial statistics (double, fixed order), or the split f16 planes of
// the attention path (GemmArgs::o16h, fragment-major).
// Reference ops: models/ddpm/diffusion.py:179-198 (q, k, v, proj_out), :145-149 (nin_shortcut), :236-248 (DeltaBlock);
// models/improved_ddpm/unet.py:313-316, 264, 790-833.
#include <cstdint>
#include "kernels.h"

namespace asyrp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float G1_HMAX = 65504.0f;
constexpr int G1_MAXC = 2048;        // widest normalised input (scale/shift rows staged in LDS)

__device__ __forceinline__ float g1_silu(float v) {
  const float e = __expf(-v);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ void g1_split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 s;
    s[0] = __builtin_amdgcn_fmed3f(v[j], -G1_HMAX, G1_HMAX);
    s[1] = __builtin_amdgcn_fmed3f(v[j + 1], -G1_HMAX, G1_HMAX);
    const h2 h = __builtin_convertvector(s, h2);
    f2 r;
    r[0] = s[0] - (float)h[0];
    r[1] = s[1] - (float)h[1];
    const h2 l = __builtin_convertvector(r, h2);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

// fragment-major weight image: element (n, k) of W[Cout][Cin] (times wscale) at
//   ((((n / 16) * (Cin / 32) + k / 32) * 2 + plane) * 64 + (k % 32) / 8 * 16 + n % 16) * 8 + k % 8        (plane 0 = hi, 1 = lo)
// rows n >= Cout of the last 16-channel tile are zero
__global__ void gemm1x1_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int cout, int cin, float wscale, long long total) {
  const int nsteps = cin >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), plane = (int)((i >> 9) & 1);
    const long long ts = i >> 10;
    const int s = (int)(ts % nsteps), nt = (int)(ts / nsteps);
    const int n = nt * 16 + (lane & 15), k = s * 32 + (lane >> 4) * 8 + j;
    float v = 0.f;
    if (n < cout) v = w[(long long)n * cin + k] * wscale;
    const _Float16 h = (_Float16)v;
    dst[i] = plane ? (_Float16)(v - (float)h) : h;
  }
}

size_t gemm1x1_packed_halfs(int cout, int cin) { return (size_t)((cout + 15) / 16) * (cin / 32) * 2 * 64 * 8; }

hipError_t launch_gemm1x1_pack(const float* w_dev, void* dst, int cout, int cin, float wscale, hipStream_t s) {
  if (cin & 31) return hipErrorInvalidValue;
  const long long total = (long long)gemm1x1_packed_halfs(cout, cin);
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(gemm1x1_pack_kernel, dim3((unsigned)b), dim3(256), 0, s, w_dev, reinterpret_cast<_Float16*>(dst), cout, cin, wscale, total);
  return hipGetLastError();
}

#ifdef ASYRP_BENCH_HOOKS   // phase stamps of the profiling library (scripts/gemm1x1_phases.py): GemmArgs.part doubles as the stamp buffer
#define G1_STAMP(i) do { if (p.part && lane == 0) reinterpret_cast<unsigned long long*>(p.part)[((size_t)wg0 * (NT / 64) + wave) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G1_STAMP(i) do { } while (0)
#endif

template <int NP, int WM, bool PRO>
__global__ void __launch_bounds__(WM * 128, 4 / WM) gemm1x1_k32_kernel(const GemmArgs p) {
  constexpr int BN = 128, D = 2, BM = WM * 64, NT = WM * 128;
  __shared__ double red[WM * BN * 2];
  __shared__ float4 sps[PRO ? 2 * G1_MAXC / 4 : 1];     // the image's scale row, then its shift row (read per step with ds_read_b128)
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroups go to the 8 XCDs round-robin by linear id.  xmap: XCD k takes a contiguous range of (image, N block, M block), so the
  // workgroups that read one image's pixels meet in one L2 and the weights (<= 3 MB) stay resident in each (see attn_planes_kernel)
  int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int wg0 = wg;
  (void)wg0;
  G1_STAMP(0);
  if (p.xmap) wg = (wg & 7) * ((int)(gridDim.x * gridDim.y * gridDim.z) >> 3) + (wg >> 3);
  const int mb = wg % (int)gridDim.x, nbz = wg / (int)gridDim.x;
  const int zblk = nbz / (int)gridDim.y, n0 = (nbz % (int)gridDim.y) * BN, m0 = mb * BM;
  // pair form (GemmArgs.g1_pair, WM = 2, 8 x 8 maps): wave row wm = image 2 zblk + wm, rows 0..63 of that image; an absent second
  // image (odd batch) is computed on the last image's data and not stored
  const bool pair = (WM == 2) && p.g1_pair;
  const int zraw = pair ? zblk * 2 + wm : zblk;
  const bool zvalid = zraw < p.Z;
  const int zo = zvalid ? zraw : p.Z - 1;
  const int mw = pair ? 0 : m0 + wm * 64;              // first row of this wave within its image
  const int HWo = p.Hout * p.Wout, Cout = p.Cout, c0s = p.c0;
  const int nsteps = p.Cin >> 5, lda0 = p.lda0, lda1 = p.lda1;
  const float* __restrict__ a0 = p.a0 + (long long)zo * p.a0_zo;
  const float* __restrict__ a1 = p.a1 ? p.a1 + (long long)zo * p.a1_zo : nullptr;
  const int ldps = p.ld_ps ? p.ld_ps : p.Cin;
  const float* __restrict__ ps = PRO ? p.pscale + (long long)zo * ldps : nullptr;
  const float* __restrict__ psh = PRO ? p.pshift + (long long)zo * ldps : nullptr;
  const _Float16* __restrict__ wpk = reinterpret_cast<const _Float16*>(p.wpk);

  // this lane's pixel in each of the wave's four 16-row blocks (clamped: rows past the image are computed and discarded)
  int arow[4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) arow[tm] = min(mw + tm * 16 + r16, HWo - 1);
  // B: the wave's four 16-channel tiles; tile index clamped to the last existing one (its missing rows are zero in the image)
  const int ntiles = (Cout + 15) >> 4;
  long long boff[4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) boff[tn] = ((long long)min((n0 >> 4) + wn * 4 + tn, ntiles - 1) * nsteps * 2 * 64 + lane) * 8;

  float4 ar[D][4][2];        // raw activations: [ring][row block][8 floats]
  h8 bh[D][4], bl[D][4];
  auto load_step = [&](int s, int buf) {
    const int k = s * 32 + g * 8;
    const bool second = (s * 32 >= c0s);                              // wave-uniform: a 32-channel step lies in one source (c0 % 32 == 0)
    const float* __restrict__ base = second ? a1 + (k - c0s) : a0 + k;
    const int ld = second ? lda1 : lda0;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      const float* src = base + (long long)arow[tm] * ld;
      ar[buf][tm][0] = *reinterpret_cast<const float4*>(src);
      ar[buf][tm][1] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      bh[buf][tn] = *reinterpret_cast<const h8*>(wpk + boff[tn] + (long long)s * 1024);
      if (NP == 3) bl[buf][tn] = *reinterpret_cast<const h8*>(wpk + boff[tn] + (long long)s * 1024 + 512);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tm][tn][r] = 0.f;

#pragma unroll
  for (int u = 0; u < D; ++u) load_step(min(u, nsteps - 1), u);
  __builtin_amdgcn_sched_barrier(0);
  const int cq = p.Cin >> 2;
  const int spo = pair ? wm * 2 * cq : 0;             // this wave's scale | shift rows in LDS (pair form: one pair of rows per image)
  if (PRO) {   // the only barrier of the kernel: scale/shift rows of this image -> LDS while the first steps are in flight
    const int nq = p.Cin >> 2;
    if (pair) {   // both images' rows (launcher: Cin <= G1_MAXC / 2)
      for (int i = tid; i < 2 * nq; i += NT) {
        const int im = i / nq, q = i - im * nq;
        const long long zi = min(zblk * 2 + im, p.Z - 1);
        sps[im * 2 * nq + q] = *reinterpret_cast<const float4*>(p.pscale + zi * ldps + 4 * q);
        sps[im * 2 * nq + nq + q] = *reinterpret_cast<const float4*>(p.pshift + zi * ldps + 4 * q);
      }
    } else {
      for (int i = tid; i < nq; i += NT) {
        sps[i] = *reinterpret_cast<const float4*>(ps + 4 * i);
        sps[nq + i] = *reinterpret_cast<const float4*>(psh + 4 * i);
      }
    }
    __syncthreads();
  }
  G1_STAMP(1);
  for (int s0 = 0; s0 < nsteps; s0 += D) {           // nsteps % D == 0 (launcher: Cin % 64 == 0)
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int s = s0 + u;
      float sc[8], sh[8];
      if (PRO) {
        const int q = s * 8 + g * 2;
        const float4 c0 = sps[spo + q], c1 = sps[spo + q + 1], d0 = sps[spo + cq + q], d1 = sps[spo + cq + q + 1];
        sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
        sh[0] = d0.x; sh[1] = d0.y; sh[2] = d0.z; sh[3] = d0.w; sh[4] = d1.x; sh[5] = d1.y; sh[6] = d1.z; sh[7] = d1.w;
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        float t[8] = {ar[u][tm][0].x, ar[u][tm][0].y, ar[u][tm][0].z, ar[u][tm][0].w,
                      ar[u][tm][1].x, ar[u][tm][1].y, ar[u][tm][1].z, ar[u][tm][1].w};
        if (PRO) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = __builtin_fmaf(t[j], sc[j], sh[j]);
        }
        if (PRO && p.silu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = g1_silu(t[j]);
        }
        h8 xh, xl;
        g1_split8(t, xh, xl);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          if (NP == 3) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh[u][tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh[u][tn], acc[tm][tn], 0, 0, 0);
          if (NP == 3) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl[u][tn], acc[tm][tn], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_step(min(s + D, nsteps - 1), u);          // unconditional (the tail re-reads the last step): static load counts
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  G1_STAMP(2);
  // ---- epilogue: C/D layout of a 16 x 16 block: column = lane & 15 (channel), rows 4 g + r (pixels) ----
  const float* __restrict__ rz = p.resid ? p.resid + (long long)zo * p.r_zo : nullptr;
  const float* __restrict__ cadd = p.chan_add ? p.chan_add + (long long)zo * p.ld_chan_add : nullptr;
  if (p.o16h) {   // split f16 planes for the attention kernel (fragment-major, kernels.h frag_off); q|k rows, v transposed
    _Float16* __restrict__ oh = p.o16h + (long long)zo * p.o16_zo;
    _Float16* __restrict__ ol = p.o16l ? p.o16l + (long long)zo * p.o16_zo : nullptr;
    _Float16* __restrict__ vh = p.vth + (long long)zo * p.vt_zo;
    _Float16* __restrict__ vl = p.vtl ? p.vtl + (long long)zo * p.vt_zo : nullptr;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wn * 64 + tn * 16 + r16;
      if (n >= Cout || !zvalid) continue;
      const float add = (p.bias ? p.bias[n] : 0.f) + (cadd ? cadd[n] : 0.f);
      const int nm = n % p.v_mod;
      const bool isv = nm >= p.v_off;
      const int vc = (n / p.v_mod) * p.v_dh + nm - p.v_off;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const int pix = mw + tm * 16 + 4 * g;
        h4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_amdgcn_fmed3f(acc[tm][tn][r] * p.alpha + add, -G1_HMAX, G1_HMAX);
          const _Float16 hh = (_Float16)v;
          hi[r] = hh;
          lo[r] = (_Float16)(v - (float)hh);
        }
        if (isv) {
          if (pix + 3 < HWo) {
            const long long o = frag_off(vc, pix, HWo);
            *reinterpret_cast<h4*>(vh + o) = hi;
            if (vl) *reinterpret_cast<h4*>(vl + o) = lo;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (pix + r < HWo) { const long long o = frag_off(vc, pix + r, HWo); vh[o] = hi[r]; if (vl) vl[o] = lo[r]; }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (pix + r < HWo) { const long long o = frag_off(pix + r, n, p.ld16); oh[o] = hi[r]; if (ol) ol[o] = lo[r]; }
        }
      }
    }
    G1_STAMP(3);
    return;
  }
  float* __restrict__ outz = p.out + (long long)zo * p.o_zo;
  const bool want_stats = (p.stats != nullptr);
  const bool full_tile = (n0 + BN <= Cout) && (pair ? (zvalid && HWo == 64) : (m0 + BM <= HWo));   // (wave-uniform)
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int n = n0 + wn * 64 + tn * 16 + r16;
    const bool nok = n < Cout;
    const float add = nok ? ((p.bias ? p.bias[n] : 0.f) + (cadd ? cadd[n] : 0.f)) : 0.f;
    double s1 = 0.0, s2 = 0.0;
    if (full_tile) {
      // straight-line form for whole tiles (round 4): with the per-element bounds test hipcc put `s_waitcnt vmcnt(0)` in front of every
      // residual load and store -- 64 serialized round trips per wave (the K32 kernel's epilogue had the same disease,
      // profiles/r04b_*).  Here the sixteen residual values of a column block are requested together and the stores follow unwaited.
      float rv[4][4];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pix = mw + tm * 16 + 4 * g + r;
          rv[tm][r] = rz ? rz[(long long)pix * p.ldr + n] : 0.f;
        }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pix = mw + tm * 16 + 4 * g + r;
          const float v = (acc[tm][tn][r] * p.alpha + add) + rv[tm][r];
          outz[(long long)pix * p.ldo + n] = v;
          if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
        }
    } else {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pix = mw + tm * 16 + 4 * g + r;
          if (nok && pix < HWo && zvalid) {
            const float v = (acc[tm][tn][r] * p.alpha + add) + (rz ? rz[(long long)pix * p.ldr + n] : 0.f);
            outz[(long long)pix * p.ldo + n] = v;
            if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
          }
        }
      }
    }
    if (want_stats) {   // fixed order: the four row groups of a block, then the four wave rows below
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (g == 0) {
        double* d = red + ((size_t)wm * BN + wn * 64 + tn * 16 + r16) * 2;
        d[0] = s1;
        d[1] = s2;
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      if (n0 + c >= Cout) continue;
      if (pair) {   // a wave row is an image: its row of sums is that image's (single) statistics row
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          const int zi = zblk * 2 + w;
          if (zi < p.Z) {
            double* dst = p.stats + ((size_t)zi * Cout + n0 + c) * 2;
            dst[0] = red[((size_t)w * BN + c) * 2];
            dst[1] = red[((size_t)w * BN + c) * 2 + 1];
          }
        }
      } else {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s1 += red[((size_t)w * BN + c) * 2];
          s2 += red[((size_t)w * BN + c) * 2 + 1];
        }
        double* dst = p.stats + (((size_t)zblk * gridDim.x + mb) * Cout + n0 + c) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
  G1_STAMP(3);
}

// shapes the barrier-free 1x1 kernel covers (GemmArgs.g1 set by the caller, wpk = its fragment-major weight image)
bool gemm1x1_ok(const GemmArgs& a) {
  if (!(a.ks == 1 && a.stride == 1 && !a.ups && !a.s0 && !a.abl && !a.poly && a.sk <= 1 && !a.bT && a.ZI <= 1 && !a.rups)) return false;
  if ((a.Cin & 63) || (a.c0 & 31) || (a.a1 && ((a.c1 | a.lda1) & 3)) || (a.lda0 & 3)) return false;
  if (((uintptr_t)a.a0 | (uintptr_t)a.a1 | (uintptr_t)a.pscale | (uintptr_t)a.pshift) & 15) return false;
  if ((a.pscale && ((a.ld_ps & 3) || a.Cin > G1_MAXC)) || (a.silu && !a.pscale)) return false;
  return a.Hin == a.Hout && a.Win == a.Wout;
}

template <int NP, int WM>
static hipError_t launch_g1(const GemmArgs& a, hipStream_t s) {
  // 8 x 8 maps on the 4-wave form: two images per workgroup (GemmArgs.g1_pair); needs both images' scale/shift rows in LDS
  const bool pair = (WM == 2) && a.Hout * a.Wout == 64 && (!a.pscale || a.Cin <= G1_MAXC / 2);
  dim3 grid(pair ? 1 : (a.Hout * a.Wout + WM * 64 - 1) / (WM * 64), (a.Cout + 127) / 128, pair ? (a.Z + 1) / 2 : a.Z), block(WM * 128);
  GemmArgs ax = a;
  ax.g1_pair = pair ? 1 : 0;
  ax.xmap = (xcd_map_enabled() && ((long long)grid.x * grid.y * grid.z) % 8 == 0) ? 1 : 0;
  if (a.pscale) hipLaunchKernelGGL((gemm1x1_k32_kernel<NP, WM, true>), grid, block, 0, s, ax);
  else hipLaunchKernelGGL((gemm1x1_k32_kernel<NP, WM, false>), grid, block, 0, s, ax);
  return hipGetLastError();
}

// a.tile: XT_G1_256 (8 waves, 256 pixels x 128 channels) or XT_G1_128 (4 waves, 128 x 128, two workgroups per CU)
hipError_t launch_gemm1x1(const GemmArgs& a, hipStream_t s) {
  if (!gemm1x1_ok(a) || !a.wpk || (a.pscale && !a.pshift)) return hipErrorInvalidValue;
  const bool big = (a.tile == XT_G1_256);
  if (a.np == 1) return big ? launch_g1<1, 4>(a, s) : launch_g1<1, 2>(a, s);
  return big ? launch_g1<3, 4>(a, s) : launch_g1<3, 2>(a, s);
}

}  // namespace asyrp
