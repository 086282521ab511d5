// attention.hip — fused self-attention on the f16 matrix cores (gfx950), fp32-equivalent through the same two-term
// operand split as conv_f16x3.hip:  out = softmax(Q K^T * scale) V  per (image, head), no T x T tensor in HBM.
//
// Reference ops replaced: models/ddpm/diffusion.py:205-221 (AttnBlock: bmm(q,k)*C^-0.5, softmax, bmm(v,w^T)) and
// models/improved_ddpm/unet.py:379-396 (QKVAttentionLegacy: heads of 64 channels, softmax in fp32).
//
// One workgroup = 32 queries of one (image, head); its 4 waves split the KEYS (wave w owns keys [w*NKT*32, (w+1)*NKT*32)).
//   S^T = K Q^T   v_mfma_f32_32x32x16_f16, A = K rows (loaded global -> registers, 32 contiguous bytes per lane per K-step,
//                 split into f16 hi/lo in registers), B = Q (staged once per workgroup into LDS as hi/lo f16);
//                 three products per K-step: k_lo*q_hi + k_hi*q_hi + k_hi*q_lo (fp32 accumulate).
//                 The transposed product leaves, in the accumulator layout, each lane holding ITS query (lane&31) against
//                 16 keys per tile -- exactly the A-operand layout of the second product, so P never goes through LDS.
//   softmax       row max / row sum: in-register over the wave's keys, lane^32 shuffle, then a 4-wave exchange through
//                 LDS (2 barriers); p = exp(s - max) / sum as the reference's softmax-then-matmul order.
//   O = P V       A = P (registers, hi/lo split of p * 2^10: keeps p_lo a normal f16), B = V gathered global -> registers
//                 in the key order the accumulator layout dictates (8 dword loads per lane per 16 keys), same 3 products;
//                 every wave produces a partial O over its keys, the 4 partials are added in a fixed order through LDS
//                 (deterministic) and stored coalesced along the channel axis.
// Single pass over the keys (T <= 1024: the whole score row of a query lives in the 4 waves' registers), so no online
// rescaling is needed.  Larger T / head widths fall back to the unfused path in engine.hip.
#include "kernels.h"

namespace asyrp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float P_SCALE = 1024.0f;            // power of two: exact; undone on the output
constexpr float P_UNSCALE = 1.0f / 1024.0f;
constexpr float AH_MAX = 65504.0f;

__device__ __forceinline__ void asplit8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 s;
    s[0] = __builtin_amdgcn_fmed3f(v[j], -AH_MAX, AH_MAX);
    s[1] = __builtin_amdgcn_fmed3f(v[j + 1], -AH_MAX, AH_MAX);
    const h2 h = __builtin_convertvector(s, h2);
    f2 r;
    r[0] = s[0] - (float)h[0];
    r[1] = s[1] - (float)h[1];
    const h2 l = __builtin_convertvector(r, h2);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

// NKT = 32-key tiles per wave (T <= NKT*128); DCH = 32-channel output tiles accumulated per pass over the keys
// NP = matrix products per term: 3 (two-term split, fp32-equivalent) or 1 (conv_math "f16": hi terms only)
template <int NKT, int DCH, int NP = 3>
__global__ void __launch_bounds__(256, 1) attn_f16x3_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T, Dh = p.Dh, ld = p.ld;
  const int q0 = blockIdx.x * 32;
  const float* __restrict__ base = p.qkv + (long long)blockIdx.z * p.img_stride + (long long)blockIdx.y * p.head_stride;
  const float* __restrict__ Q = base + p.q_off;
  const float* __restrict__ K = base + p.k_off;
  const float* __restrict__ V = base + p.v_off;
  const int nks = Dh >> 4;                         // K-steps of 16 channels

  // LDS: Qs [nks][4 units: hi h0, hi h1, lo h0, lo h1][32 queries][8 halfs]  |  red [2][4 waves][32]  |  Os [4][32][DCH*32] f32
  char* const Qs = smem;
  float* const red = reinterpret_cast<float*>(smem + (size_t)nks * 4 * 32 * 16);
  float* const Os = red + 2 * 4 * 32;

  // ---- stage Q (hi/lo) ----
  for (int idx = tid; idx < 32 * (Dh >> 3); idx += 256) {
    const int q = idx & 31, oct = idx >> 5;
    const int row = min(q0 + q, T - 1);
    const float* src = Q + (long long)row * ld + oct * 8;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    h8 hi, lo;
    asplit8(v, hi, lo);
    const int ks = oct >> 1, hh = oct & 1;
    *reinterpret_cast<h8*>(Qs + ((size_t)(ks * 4 + hh) * 32 + q) * 16) = hi;
    *reinterpret_cast<h8*>(Qs + ((size_t)(ks * 4 + 2 + hh) * 32 + q) * 16) = lo;
  }
  __syncthreads();

  // ---- S^T = K Q^T over this wave's keys ----
  const int kbase = wave * NKT * 32;
  f32x16 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
  int koff[NKT];                                   // 32-bit element offsets: a (image, head) slab is far below 2^31 floats
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) koff[kt] = min(kbase + kt * 32 + c, T - 1) * ld + 8 * h;
  const bool wave_has_keys = (kbase < T);
  if (wave_has_keys) {
    // software pipeline over the K-steps: a ring of RK register sets keeps the K rows of the next RK steps in flight while a
    // step feeds the matrix cores (one wave per SIMD: nothing else hides the L2 latency)
    constexpr int RK = (NKT >= 8) ? 2 : (8 / NKT > 8 ? 8 : 8 / NKT);
    float4 ka[RK][NKT], kb[RK][NKT];
    auto loadK = [&](int ks, int buf) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        ka[buf][kt] = *reinterpret_cast<const float4*>(K + koff[kt] + ks * 16);
        kb[buf][kt] = *reinterpret_cast<const float4*>(K + koff[kt] + ks * 16 + 4);
      }
    };
    auto stepK = [&](int ks, int buf) {
      const h8 qh = *reinterpret_cast<const h8*>(Qs + ((size_t)(ks * 4 + h) * 32 + c) * 16);
      const h8 ql = *reinterpret_cast<const h8*>(Qs + ((size_t)(ks * 4 + 2 + h) * 32 + c) * 16);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const float v[8] = {ka[buf][kt].x, ka[buf][kt].y, ka[buf][kt].z, ka[buf][kt].w,
                            kb[buf][kt].x, kb[buf][kt].y, kb[buf][kt].z, kb[buf][kt].w};
        h8 khi, klo;
        asplit8(v, khi, klo);
        if (NP == 3) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo, qh, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qh, s[kt], 0, 0, 0);
        if (NP == 3) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, ql, s[kt], 0, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < RK; ++i)
      if (i < nks) loadK(i, i);
    for (int ks0 = 0; ks0 < nks; ks0 += RK) {
#pragma unroll
      for (int i = 0; i < RK; ++i) {
        const int ks = ks0 + i;
        if (ks < nks) {
          stepK(ks, i);
          if (ks + RK < nks) loadK(ks + RK, i);
        }
      }
    }
  }

  // ---- softmax over the keys of query (q0 + c): accumulator element r of tile kt is key kbase + kt*32 + (r&3) + 8*(r>>2) + 4*h ----
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = (key < T) ? s[kt][r] * p.scale : -INFINITY;
      s[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (h == 0) red[wave * 32 + c] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[c], red[32 + c]), fmaxf(red[64 + c], red[96 + c]));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(s[kt][r] - mx);       // exp(-inf) = 0 for masked keys
      s[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  if (h == 0) red[128 + wave * 32 + c] = sum;
  __syncthreads();
  sum = (red[128 + c] + red[160 + c]) + (red[192 + c] + red[224 + c]);
  const float inv_scale = P_SCALE;
  h8 phi[NKT][2], plo[NKT][2];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __fdiv_rn(s[kt][k2 * 8 + j], sum) * inv_scale;
      asplit8(v, phi[kt][k2], plo[kt][k2]);
    }

  // ---- O = P V, DCH output tiles per pass; partials of the 4 waves reduced through LDS ----
  const int ndt = (Dh + 31) >> 5;
  float* __restrict__ outz = p.out + (long long)blockIdx.z * p.o_img_stride + (long long)blockIdx.y * p.o_head_stride;
  for (int d0 = 0; d0 < ndt; d0 += DCH) {
    f32x16 o[DCH];
#pragma unroll
    for (int dt = 0; dt < DCH; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    if (wave_has_keys) {
      // the (key tile, 16-key step) loop is fully unrolled (P lives in registers); the V gather of step i+1 is issued
      // before the products of step i.  Addresses are clamped, never predicated: one exec mask for the whole kernel.
      int dcol[DCH];
      bool dok[DCH];
#pragma unroll
      for (int dt = 0; dt < DCH; ++dt) {
        const int d = (d0 + dt) * 32 + c;
        dok[dt] = d < Dh;
        dcol[dt] = min(d, Dh - 1);
      }
      constexpr int NSTEP = NKT * 2;
      constexpr int RV = (NSTEP < 4) ? NSTEP : ((NKT >= 8) ? 2 : 4);      // V gathers in flight ahead of the products
      float vb[RV][DCH][8];
      auto loadV = [&](int step, int buf) {
        const int kb0 = kbase + step * 16 + 4 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ro = min(kb0 + (j & 3) + 8 * (j >> 2), T - 1) * ld;
#pragma unroll
          for (int dt = 0; dt < DCH; ++dt) vb[buf][dt][j] = V[ro + dcol[dt]];
        }
      };
#pragma unroll
      for (int i = 0; i < RV - 1; ++i) loadV(i, i);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + RV - 1 < NSTEP) loadV(st + RV - 1, (st + RV - 1) % RV);
        const int kt = st >> 1, k2 = st & 1;
#pragma unroll
        for (int dt = 0; dt < DCH; ++dt) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = dok[dt] ? vb[st % RV][dt][j] : 0.f;
          h8 vhi, vlo;
          asplit8(v, vhi, vlo);
          if (NP == 3) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(plo[kt][k2], vhi, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(phi[kt][k2], vhi, o[dt], 0, 0, 0);
          if (NP == 3) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(phi[kt][k2], vlo, o[dt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // accumulator layout: column = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*h (query)
    float* ow = Os + (size_t)wave * 32 * DCH * 32;
#pragma unroll
    for (int dt = 0; dt < DCH; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * h;
        ow[(q * DCH + dt) * 32 + c] = o[dt][r];
      }
    __syncthreads();
    for (int e = tid; e < 32 * DCH * 32; e += 256) {
      const int q = e / (DCH * 32), dl = e - q * (DCH * 32);
      const int d = d0 * 32 + dl;
      const float v = ((Os[e] + Os[32 * DCH * 32 + e]) + (Os[2 * 32 * DCH * 32 + e] + Os[3 * 32 * DCH * 32 + e])) * P_UNSCALE;
      if (q0 + q < T && d < Dh) outz[(long long)(q0 + q) * p.ldo + d] = v;
    }
    __syncthreads();
  }
}

template <int NKT, int DCH, int NP>
static hipError_t launch_attn_np(const AttnArgs& a, hipStream_t s) {
  const size_t smem = (size_t)(a.Dh >> 4) * 4 * 32 * 16 + 2 * 4 * 32 * sizeof(float) + (size_t)4 * 32 * DCH * 32 * sizeof(float);
  static bool attr_set[16] = {};
  if (smem > 64 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_kernel<NKT, DCH, NP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
  }
  dim3 grid((a.T + 31) / 32, a.heads, a.B);
  hipLaunchKernelGGL((attn_f16x3_kernel<NKT, DCH, NP>), grid, dim3(256), smem, s, a);
  return hipGetLastError();
}
template <int NKT, int DCH>
static hipError_t launch_attn(const AttnArgs& a, hipStream_t s) {
  return a.np == 1 ? launch_attn_np<NKT, DCH, 1>(a, s) : launch_attn_np<NKT, DCH, 3>(a, s);
}

bool attn_fused_supported(int T, int Dh, int ld, int ldo) {
  // whole score row in registers: T <= 1024; K-steps of 16 channels; Q image (Dh * 128 B) + output exchange fit the 160 KB LDS;
  // rows are read as float4
  return T >= 1 && T <= 1024 && Dh >= 16 && Dh <= 512 && (Dh & 15) == 0 && (ld & 3) == 0 && ldo > 0;
}

hipError_t launch_attention_fused(const AttnArgs& a, hipStream_t s) {
  if (!attn_fused_supported(a.T, a.Dh, a.ld, a.ldo)) return hipErrorInvalidValue;
  if ((((uintptr_t)a.qkv) & 15) || ((a.q_off | a.k_off | a.v_off) & 3) || (a.head_stride & 3) || (a.img_stride & 3))
    return hipErrorInvalidValue;
  const int nkt = (a.T + 127) / 128;
  const bool wide = a.Dh > 64;     // 4 output tiles (128 channels) per pass for wide heads, 2 for the 64-channel heads
  if (nkt <= 1) return wide ? launch_attn<1, 4>(a, s) : launch_attn<1, 2>(a, s);
  if (nkt <= 2) return wide ? launch_attn<2, 4>(a, s) : launch_attn<2, 2>(a, s);
  if (nkt <= 4) return wide ? launch_attn<4, 4>(a, s) : launch_attn<4, 2>(a, s);
  return wide ? launch_attn<8, 4>(a, s) : launch_attn<8, 2>(a, s);
}

// =====================================================================================================================
// attn_planes_kernel -- the fused attention of the f16 engine since round 3 (the kernel above ran at
// 0.08 of its MFMA bound because every K row and every V column was fetched with exposed latency and split into f16 hi/lo in
// registers by each of the 8 workgroups that needed it).
//   * the q|k|v projection's epilogue already wrote q, k as f16 hi/lo PLANES and v TRANSPOSED ([channel][token]) hi/lo
//     (GemmArgs::o16h ...), so every matrix operand here is a plain 16-byte load: no conversion work, no gathers;
//   * v_mfma_f32_16x16x32_f16 (the instruction the convolutions moved to in round 2), three products per term (one when NP = 1);
//   * one workgroup = 32 queries of one (image, head), 8 waves (2 per SIMD).
//       phase 1  S^T = K Q^T: the waves split the KEYS (16-key tiles, wave w takes tiles w, w+8, ...); A = K rows straight from
//                global memory into registers (one 16-B load per plane, prefetched one 32-channel step ahead), B = Q staged once
//                into LDS; the transposed product leaves each lane holding ITS query against 4 keys per tile;
//       softmax  over the whole score row (T <= 1024 lives in the 8 waves' registers): max / sum through lane shuffles and a small
//                LDS exchange, p = exp(s - max) / sum in the reference's order; p * 2^10 is split hi/lo and written to LDS in
//                the B-operand layout of phase 2;
//       phase 2  O^T = V^T P^T: the waves split the OUTPUT CHANNELS (16-channel tile x 16-query tile items), each over ALL keys,
//                so there is no cross-wave reduction of partial outputs; A = V^T rows (16 B of consecutive keys per plane),
//                B = P from LDS; each lane ends with 4 consecutive channels of one query: a float4 store.
// Every K / V element is loaded once per workgroup.  T % 32 == 0, Dh % 32 == 0 (every reference configuration: T = 64 / 256 /
// 1024, Dh = 64 / 512); other shapes stay on attn_f16x3_kernel.
// =====================================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// phase stamps (profiling library only, -DASYRP_BENCH_HOOKS): wave 0 of every workgroup records s_memrealtime (100 MHz) at the
// phase boundaries into AttnArgs::dbg [workgroup][8]; scripts/attn_phases.py prints the breakdown
#ifdef ASYRP_BENCH_HOOKS
#define ATTN_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

// DK: channel steps in flight in phase 1 (divides Dh/32); JC: output items a wave runs side by side in phase 2 (= items per
// wave, Dh/64, at least 1).  Both are template parameters so that EVERY path through the loops issues the same number of
// loads: hipcc's waitcnt pass then emits exact counted waits.  With loads under run-time conditions it falls back to vmcnt(0)
// before every use and the prefetch rings collapse to one exposed round trip per step -- measured with the phase stamps of the
// profiling library: 15 us (phase 1) + 21 us (phase 2) of a 49 us workgroup life (profiles/rd3g_*).
template <int NKTW /* 16-key tiles per wave */, int NP, int DK, int JC>
__global__ void __launch_bounds__(512, 1) attn_planes_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ATTN_STAMP(0);
  const int T = p.T, Dh = p.Dh, ld = p.ld16;
  const int nks = Dh >> 5, nsteps = T >> 5;       // 32-channel steps of Q K^T, 32-key steps of P V
  // Workgroups go to the 8 XCDs round-robin by linear id: with the identity map the 8 query blocks of one (image, head) land on 8
  // different XCDs and each XCD's L2 fetches that image's K and V from HBM on its own (8x the compulsory traffic: the kernel ran
  // at the HBM rate of those re-reads, profiles/rd3d_*).  XCD k takes a contiguous range of (image, head, query block) instead,
  // so the query blocks that share K / V meet in one L2.
  int w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int nblk = gridDim.x * gridDim.y * gridDim.z;
  if ((nblk & 7) == 0) w = (w & 7) * (nblk >> 3) + (w >> 3);
  const int qb = w % (int)gridDim.x, hb = w / (int)gridDim.x;
  const int q0 = qb * 32, head = hb % (int)gridDim.y, b = hb / (int)gridDim.y;
  // fragment-major planes (kernels.h frag_off): the 1-KiB block of (16-token tile tt, 32-channel step cs) of the q|k planes sits at
  // ((tt * ld/32 + cs) * 64 + lane) * 8 halfs; q starts at channel step (head*head_stride + q_off)/32, k likewise
  const long long img = (long long)b * T * ld;
  const int ldb = ld >> 5;
  const int qcs = ((int)(head * p.head_stride) + p.q_off) >> 5, kcs = ((int)(head * p.head_stride) + p.k_off) >> 5;
  const _Float16* __restrict__ PH = p.qkh + img;
  const _Float16* __restrict__ PL = p.qkl + img;
  // v^T planes: rows = this image's v channels (head*Dh + d), cols = tokens: block (dt, st) at ((dt * T/32 + st) * 64 + lane) * 8
  const long long vimg = (long long)b * p.heads * Dh * T;
  const _Float16* __restrict__ VH = p.vth + vimg;
  const _Float16* __restrict__ VL = p.vtl + vimg;
  const int vdt0 = (head * Dh) >> 4;

  // LDS: Qs [plane][ks][query tile][lane][8 halfs] | Ps [plane][qt][step][kq][16 q][8 halfs] | red [2][8 waves][32]
  char* const Qs = smem;
  char* const Ps = smem + (size_t)2 * nks * 4 * 32 * 16;
  float* const red = reinterpret_cast<float*>(Ps + (size_t)2 * 2 * nsteps * 4 * 16 * 16);

  // ---- the first DK channel steps of this wave's K tiles go out before anything else: they fly while Q is staged ----
  long long koff[NKTW];
  bool kval[NKTW];
#pragma unroll
  for (int i = 0; i < NKTW; ++i) {
    const int kt = wave + 8 * i;
    kval[i] = (kt * 16 < T);                          // wave-uniform (T % 16 == 0); tiles past T read clamped rows, masked below
    koff[i] = ((long long)(min(kt, (T >> 4) - 1) * ldb + kcs) * 64 + lane) * 8;     // + ks * 512 halfs per channel step
  }
  // the K rows of the next DK channel steps are in flight while a step feeds the matrix cores: the kernel is bound by the
  // round trips of these loads (one workgroup per CU, two waves per SIMD), so the ring is as deep as the registers allow
  h8 kh[DK][NKTW], kl[DK][NKTW];
  auto loadK = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < NKTW; ++i) {
      kh[buf][i] = *reinterpret_cast<const h8*>(PH + koff[i] + ks * 512);
      if (NP == 3) kl[buf][i] = *reinterpret_cast<const h8*>(PL + koff[i] + ks * 512);
    }
  };
#pragma unroll
  for (int u = 0; u < DK; ++u) loadK(u, u);                      // nks % DK == 0 (launcher)
  // (scheduling barriers pin the loads where they are written: left alone, the machine scheduler sinks every load next to
  // its use to save registers, which is exactly the exposed round trip per step the ring exists to avoid)
  __builtin_amdgcn_sched_barrier(0);

  // ---- stage Q: [plane][ks][query tile] blocks of 1 KiB, copied as they are (lane-linear) by LDS-DMA (round 5): every block is one
  // global_load_lds_dwordx4 of a wave, all of a wave's blocks are in flight at once and no register is involved (the round-3 form
  // moved them through four 16-byte registers per thread in two dependent rounds: 4.3 us of a 21-us workgroup) ----
  {
    const int nblk = (NP == 1 ? 1 : 2) * nks * 2;      // 1-KiB blocks: (plane, ks, query tile); Qs sits at LDS offset 0, <= 64 KB
    const int qt0 = q0 >> 4;
    for (int blk = wave; blk < nblk; blk += 8) {
      const int qtile = blk & 1, ks = (blk >> 1) % nks, pl = (blk >> 1) / nks;
      const int tt = min(qt0 + qtile, (T >> 4) - 1);
      const _Float16* src = (pl ? PL : PH) + ((long long)(tt * ldb + qcs + ks) * 64 + lane) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Qs + (size_t)blk * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  ATTN_STAMP(1);

  // ---- phase 1: S^T = K Q^T, this wave's key tiles ----
  f32x4 s[NKTW][2];
#pragma unroll
  for (int i = 0; i < NKTW; ++i)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[i][qt][r] = 0.f;
  {
    for (int ks = 0; ks < nks; ks += DK) {
#pragma unroll
      for (int u = 0; u < DK; ++u) {
        const int k1 = ks + u;
        h8 qh[2], ql[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          qh[qt] = *reinterpret_cast<const h8*>(Qs + ((size_t)(k1 * 2 + qt) * 64 + lane) * 16);
          if (NP == 3) ql[qt] = *reinterpret_cast<const h8*>(Qs + ((size_t)((nks + k1) * 2 + qt) * 64 + lane) * 16);
        }
#pragma unroll
        for (int i = 0; i < NKTW; ++i) {
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            if (NP == 3) s[i][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[u][i], qh[qt], s[i][qt], 0, 0, 0);
            s[i][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[u][i], qh[qt], s[i][qt], 0, 0, 0);
            if (NP == 3) s[i][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[u][i], ql[qt], s[i][qt], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        loadK(min(k1 + DK, nks - 1), u);      // unconditional (the tail re-reads the last step): static load counts
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  ATTN_STAMP(2);
  // ---- phase 2 set-up, and its first two key steps of V^T go out now: they fly while the softmax runs ----
  const int nitems = (Dh >> 4) * 2;
  const int nj = (nitems - wave + 7) >> 3;           // items of this wave (wave-uniform): JC, or 0 for the waves beyond the items
  const int qt = wave & 1, dtb = wave >> 1;
  float* __restrict__ outz = p.out + (long long)b * p.o_img_stride + (long long)head * p.o_head_stride;
  const char* pb = Ps + ((size_t)(qt * nsteps * 4 + g) * 16 + r16) * 16;
  const size_t plo = (size_t)2 * nsteps * 4 * 16 * 16;
  int voff[JC];
#pragma unroll
  for (int jj = 0; jj < JC; ++jj)                    // (waves beyond the items read a clamped, valid tile and discard it)
    voff[jj] = ((vdt0 + min(dtb + 4 * jj, (Dh >> 4) - 1)) * nsteps * 64 + lane) * 8;   // + st * 512 halfs per key step
  h8 vh[2][JC], vl[2][JC];
  auto loadV = [&](int st, int buf) {
#pragma unroll
    for (int jj = 0; jj < JC; ++jj) {
      vh[buf][jj] = *reinterpret_cast<const h8*>(VH + voff[jj] + st * 512);
      if (NP == 3) vl[buf][jj] = *reinterpret_cast<const h8*>(VL + voff[jj] + st * 512);
    }
  };
  loadV(0, 0);
  loadV(min(1, nsteps - 1), 1);
  __builtin_amdgcn_sched_barrier(0);

  // ---- softmax over the keys of query (q0 + qt*16 + r16): accumulator element r of tile i is key (wave + 8i)*16 + 4g + r ----
  float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int i = 0; i < NKTW; ++i)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = kval[i] ? s[i][qt][r] * p.scale : -INFINITY;
        s[i][qt][r] = v;
        mx[qt] = fmaxf(mx[qt], v);
      }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16));
    mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32));
    if (g == 0) red[wave * 32 + qt * 16 + r16] = mx[qt];
  }
  __syncthreads();
  // (round 5) the probabilities go to phase 2 UNNORMALISED: e = exp(s - max) (in (0, 1], times 2^10, split hi / lo); the row sums
  // travel through `red` under the same barrier as the P tile and the output is divided once at the end, O = (V^T e) / sum.  The
  // round-3 form normalised every e first (one more barrier and a correctly rounded division per score); the two orders agree to
  // fp32 rounding.
  float sum[2] = {0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float m = red[qt * 16 + r16];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w * 32 + qt * 16 + r16]);
    mx[qt] = m;
#pragma unroll
    for (int i = 0; i < NKTW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[i][qt][r] - m);     // exp(-inf) = 0 for the tiles this wave does not own
        s[i][qt][r] = e;
        sum[qt] += e;
      }
    sum[qt] += __shfl_xor(sum[qt], 16);
    sum[qt] += __shfl_xor(sum[qt], 32);
    if (g == 0) red[256 + wave * 32 + qt * 16 + r16] = sum[qt];
  }
  // e * 2^10 as hi/lo -> LDS in the B-operand layout of phase 2: k group kq of a 32-key step = keys 8kq .. 8kq+7
#pragma unroll
  for (int i = 0; i < NKTW; ++i) {
    if (!kval[i]) continue;
    const int kt = wave + 8 * i;
    const int step = kt >> 1, kk = (kt & 1) * 16 + 4 * g;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      h4 hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = s[i][qt][r] * P_SCALE;
        const _Float16 hh = (_Float16)v;
        hi[r] = hh;
        lo[r] = (_Float16)(v - (float)hh);
      }
      char* dst = Ps + ((size_t)((qt * nsteps + step) * 4 + (kk >> 3)) * 16 + r16) * 16 + (kk & 7) * 2;
      *reinterpret_cast<h4*>(dst) = hi;
      if (NP == 3) *reinterpret_cast<h4*>(dst + (size_t)2 * nsteps * 4 * 16 * 16) = lo;
    }
  }
  __syncthreads();
  ATTN_STAMP(3);

  // ---- phase 2: O^T = V^T P^T; items (16-channel tile dt, query tile qt): wave w owns the items w, w+8, ... -- all of query
  // tile w & 1, channel tiles (w >> 1) + 4j -- and runs up to JC of them side by side: one P fragment per key step feeds JC
  // accumulators, and the V^T rows of the next step are in flight meanwhile ----
  if (nj > 0) {
    f32x4 o[JC];
#pragma unroll
    for (int jj = 0; jj < JC; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[jj][r] = 0.f;
    for (int st = 0; st < nsteps; st += 2) {          // nsteps is even or 1 (T % 64 == 0 or T == 32: launcher)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s1 = st + u;
        if (s1 < nsteps) {                            // (only false for u = 1 when nsteps == 1: wave-uniform, loads below unaffected)
          const h8 ph = *reinterpret_cast<const h8*>(pb + (size_t)s1 * 4 * 16 * 16);
          h8 pl;
          if (NP == 3) pl = *reinterpret_cast<const h8*>(pb + (size_t)s1 * 4 * 16 * 16 + plo);
#pragma unroll
          for (int jj = 0; jj < JC; ++jj) {
            if (NP == 3) o[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[u][jj], ph, o[jj], 0, 0, 0);
            o[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[u][jj], ph, o[jj], 0, 0, 0);
            if (NP == 3) o[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[u][jj], pl, o[jj], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        loadV(min(s1 + 2, nsteps - 1), u);            // two key steps ahead, unconditional
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ATTN_STAMP(4);
    // accumulator: column = query r16 of the tile, rows = channels 4g .. 4g+3 of the tile: 16 contiguous bytes of out[q][.]
    const int q = q0 + qt * 16 + r16;
    float tsum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tsum += red[256 + w * 32 + qt * 16 + r16];   // this lane's query: fixed order, deterministic
    const float inv = __fdiv_rn(P_UNSCALE, tsum);
    if (q < T) {
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        float4 v = make_float4(o[jj][0] * inv, o[jj][1] * inv, o[jj][2] * inv, o[jj][3] * inv);
        *reinterpret_cast<float4*>(outz + (long long)q * p.ldo + (dtb + 4 * jj) * 16 + 4 * g) = v;
      }
    }
  }
  ATTN_STAMP(5);
}

static size_t attn_planes_smem(int T, int Dh) {
  return (size_t)2 * (Dh >> 5) * 4 * 32 * 16 + (size_t)2 * 2 * (T >> 5) * 4 * 16 * 16 + 2 * 8 * 32 * sizeof(float);
}

template <int NKTW, int NP, int DK, int JC>
static hipError_t launch_attn_planes_t(const AttnArgs& a, hipStream_t s) {
  const size_t smem = attn_planes_smem(a.T, a.Dh);
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_planes_kernel<NKTW, NP, DK, JC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  dim3 grid(a.T / 32, a.heads, a.B);
  hipLaunchKernelGGL((attn_planes_kernel<NKTW, NP, DK, JC>), grid, dim3(512), smem, s, a);
  return hipGetLastError();
}

// the shapes the reference's configurations produce: single 512- / 64-wide head at T = 256 / 64 (DDPM), 64-wide heads at
// T = 1024 / 256 / 64 (iDDPM, ADM); other supported shapes fall back to attn_f16x3_kernel (attn_planes_supported)
template <int NP>
static hipError_t launch_attn_planes_np(const AttnArgs& a, hipStream_t s) {
  const int nkt = (a.T / 16 + 7) / 8;               // 16-key tiles per wave
  if (a.Dh == 512) {                                // nks = 16, 8 items per wave
    if (nkt <= 1) return launch_attn_planes_t<1, NP, 4, 8>(a, s);
    if (nkt <= 2) return launch_attn_planes_t<2, NP, 4, 8>(a, s);
    return hipErrorInvalidValue;
  }
  if (a.Dh == 64) {                                 // nks = 2, 1 item per wave
    if (nkt <= 1) return launch_attn_planes_t<1, NP, 2, 1>(a, s);
    if (nkt <= 2) return launch_attn_planes_t<2, NP, 2, 1>(a, s);
    if (nkt <= 4) return launch_attn_planes_t<4, NP, 2, 1>(a, s);
    return launch_attn_planes_t<8, NP, 2, 1>(a, s);
  }
  return hipErrorInvalidValue;
}

bool attn_planes_supported(int T, int Dh) {
  if (!(T >= 32 && T <= 1024 && (T == 32 || (T & 63) == 0) && attn_planes_smem(T, Dh) <= 160 * 1024)) return false;
  if (Dh == 512) return T <= 256;
  return Dh == 64;
}

hipError_t launch_attention_planes(const AttnArgs& a, hipStream_t s) {
  if (!attn_planes_supported(a.T, a.Dh) || !a.qkh || !a.vth || (a.np != 1 && (!a.qkl || !a.vtl))) return hipErrorInvalidValue;
  if ((a.ld16 & 31) || (a.q_off & 31) || (a.k_off & 31) || (a.head_stride & 31) || (a.ldo & 3) || (a.o_head_stride & 3) ||
      ((uintptr_t)a.qkh & 15) || ((uintptr_t)a.vth & 15) || ((uintptr_t)a.out & 15))
    return hipErrorInvalidValue;
  return a.np == 1 ? launch_attn_planes_np<1>(a, s) : launch_attn_planes_np<3>(a, s);
}

// fp32 q|k|v rows -> the split planes (what the projection's epilogue writes in the engine); one thread per element
__global__ void qkv_to_planes_kernel(const float* __restrict__ qkv, int ld, int T, int C3, int v_mod, int v_off, int v_dh,
                                     _Float16* o16h, _Float16* o16l, _Float16* vth, _Float16* vtl, long long total) {
  const int Cv = (C3 / v_mod) * v_dh;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % C3);
    const long long bt = i / C3;
    const int t = (int)(bt % T);
    const long long b = bt / T;
    const float v = __builtin_amdgcn_fmed3f(qkv[bt * ld + n], -AH_MAX, AH_MAX);
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const int nm = n % v_mod;
    if (nm >= v_off) {      // fragment-major (kernels.h frag_off): rows = v channels, cols = tokens
      const long long o = b * Cv * T + frag_off((n / v_mod) * v_dh + nm - v_off, t, T);
      vth[o] = h;
      vtl[o] = l;
    } else {                // rows = tokens, cols = the 3C channels
      const long long o = b * (long long)T * C3 + frag_off(t, n, C3);
      o16h[o] = h;
      o16l[o] = l;
    }
  }
}

hipError_t launch_qkv_to_planes(const float* qkv, int ld, int B, int T, int C3, int v_mod, int v_off, int v_dh, _Float16* o16h,
                                _Float16* o16l, _Float16* vth, _Float16* vtl, hipStream_t s) {
  const long long total = (long long)B * T * C3;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(qkv_to_planes_kernel, dim3((unsigned)nb), dim3(256), 0, s, qkv, ld, T, C3, v_mod, v_off, v_dh, o16h, o16l, vth,
                     vtl, total);
  return hipGetLastError();
}

}  // namespace asyrp
