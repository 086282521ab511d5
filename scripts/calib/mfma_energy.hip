// Calibration micro-benchmark (not part of the product): the chip is power-limited under v_mfma_f32_32x32x16_f16 on random
// data (scripts/calib/mfma_peak.hip: 2.46 PFLOP/s on zeros, 1.66 on random operands), so the sustained rate of a bare MFMA
// stream is a direct reading of the ENERGY one matrix instruction costs.  This benchmark reads it for
//   (1) the three operand classes of the f16x3 scheme on UNet-like data: x_hi*w_hi, x_lo*w_hi, x_hi*w_lo, and their 1:1:1 mix;
//   (2) lo terms truncated to 8 / 6 / 4 significant bits (would a shorter lo term buy rate?);
//   (3) accumulator order: the product loop's pass-major order (12 accumulators visited round-robin, three times) against
//       acc-major (the three products of one accumulator back to back: SrcC forwarded from the previous instruction);
//   (4) operand reuse between consecutive instructions (same A / same B / both change);
//   (5) v_mfma_f32_16x16x32_f16 for the same flops.
//   hipcc --offload-arch=gfx950 -O3 scripts/calib/mfma_energy.hip -o /tmp/mfma_energy && /tmp/mfma_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline float urand(unsigned k) { return (hashu(k) >> 8) * (1.0f / 16777216.0f); }
__device__ inline float nrand(unsigned k) {   // ~N(0,1): Box-Muller
  const float u1 = urand(2 * k) + 1e-7f, u2 = urand(2 * k + 1);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}
// keep `bits` significant bits of an f16 (round to nearest): bits = 11 keeps everything
__device__ inline _Float16 keep_bits(_Float16 v, int bits) {
  if (bits >= 11) return v;
  unsigned short u = __builtin_bit_cast(unsigned short, v);
  const int drop = 11 - bits;
  u = (unsigned short)((u + (1u << (drop - 1))) & ~((1u << drop) - 1));
  return __builtin_bit_cast(_Float16, u);
}

// operand classes
enum { CLS_GRID = 0, CLS_ZERO = 1, CLS_HI = 2, CLS_LO = 3 };
// x ~ SiLU(N(0,1)) (what the staging pass produces), w ~ N(0,1) * 2^10 / 4 (weights pre-scaled so that max|w| is in [1024,2048))
__device__ inline void make_operand(h8 (&v)[4], int cls, bool weight, int lo_bits, unsigned seed) {
  for (int s = 0; s < 4; ++s)
    for (int j = 0; j < 8; ++j) {
      const unsigned k = seed + (threadIdx.x + 977u * blockIdx.x) * 64u + s * 8u + j;
      _Float16 r;
      if (cls == CLS_ZERO) r = (_Float16)0.f;
      else if (cls == CLS_GRID) r = (_Float16)(((int)(hashu(k) & 2047) - 1024) * (1.0f / 1024.0f));
      else {
        float x = nrand(k);
        x = weight ? x * 256.f : x / (1.f + expf(-x));
        const _Float16 hi = (_Float16)x;
        r = (cls == CLS_HI) ? hi : keep_bits((_Float16)(x - (float)hi), lo_bits);
      }
      v[s][j] = r;
    }
}

// ORDER 0: pass-major over NACC accumulators (pass p uses operand set p);  ORDER 1: acc-major (3 products of one accumulator
// back to back).  Three operand sets: (a0,b0) (a1,b1) (a2,b2) = the three passes.
template <int NACC, int ORDER, int MINW>
__global__ void __launch_bounds__(256, MINW) mfma3_loop(float* out, int iters, int ca0, int cb0, int ca1, int cb1, int ca2, int cb2,
                                                       int lo_bits) {
  h8 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
  make_operand(a0, ca0, false, lo_bits, 1u << 20);
  make_operand(b0, cb0, true, lo_bits, 2u << 20);
  make_operand(a1, ca1, false, lo_bits, 3u << 20);
  make_operand(b1, cb1, true, lo_bits, 4u << 20);
  make_operand(a2, ca2, false, lo_bits, 5u << 20);
  make_operand(b2, cb2, true, lo_bits, 6u << 20);
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // the operand index pattern mirrors a TM x TN wave tile: accumulator i = (tm, tn) = (i / 2 & 1 ..): A changes every second
  // instruction, B alternates
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[(i >> 1) & 3], b0[i & 1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[(i >> 1) & 3], b1[i & 1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[(i >> 1) & 3], b2[i & 1], acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[(i >> 1) & 3], b0[i & 1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[(i >> 1) & 3], b1[i & 1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[(i >> 1) & 3], b2[i & 1], acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// operand reuse between consecutive instructions: REUSE 0 = both operands change every instruction, 1 = A fixed for 4 in a row,
// 2 = A and B both fixed for 4 in a row (different accumulators)
template <int REUSE>
__global__ void __launch_bounds__(256, 4) mfma_reuse_loop(float* out, int iters) {
  h8 a[4], b[4];
  make_operand(a, CLS_HI, false, 11, 1u << 20);
  make_operand(b, CLS_HI, true, 11, 2u << 20);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ia = (REUSE == 0) ? ((i + u) & 3) : u, ib = (REUSE == 2) ? u : ((i * 3 + u) & 3);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256, 4) mfma16_loop(float* out, int iters, int ca, int cb) {
  h8 a[4], b[4];
  make_operand(a, ca, false, 11, 1u << 20);
  make_operand(b, cb, true, 11, 2u << 20);
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i >> 2) & 3], b[i & 3], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + 1) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out;
template <class F>
static double timed(F launch, double flops_per_launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); launch();
  (void)hipDeviceSynchronize();
  double mean = 0.0;
  const int reps = 4;
  for (int rep = 0; rep < reps; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 4; ++k) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    mean += 4.0 * flops_per_launch / (ms * 1e-3) / 1e12 / reps;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return mean;
}

template <int NACC, int ORDER, int MINW>
static double run3(int wgs_per_cu, int ca0, int cb0, int ca1, int cb1, int ca2, int cb2, int lo_bits) {
  const int iters = 1500, grid = 256 * wgs_per_cu;
  return timed([&] { hipLaunchKernelGGL((mfma3_loop<NACC, ORDER, MINW>), dim3(grid), dim3(256), 0, 0, g_out, iters, ca0, cb0, ca1, cb1,
                                        ca2, cb2, lo_bits); },
               (double)grid * 4 * iters * 3 * NACC * 32768.0);
}

int main() {
  (void)hipMalloc(&g_out, (size_t)256 * 4 * 256 * sizeof(float));
  const int G = CLS_GRID, Z = CLS_ZERO, H = CLS_HI, L = CLS_LO;
  for (int round = 0; round < 2; ++round) {
    printf("== round %d: 4 waves/SIMD, 4 accumulators per wave (the main tile's shape), TFLOP/s of f16 MFMA ==\n", round);
    printf("  zero operands                                   %7.1f\n", run3<4, 0, 4>(4, Z, Z, Z, Z, Z, Z, 11));
    printf("  uniform grid n/1024 (mfma_peak.hip's data)      %7.1f\n", run3<4, 0, 4>(4, G, G, G, G, G, G, 11));
    printf("  x_hi * w_hi only                                %7.1f\n", run3<4, 0, 4>(4, H, H, H, H, H, H, 11));
    printf("  x_lo * w_hi only                                %7.1f\n", run3<4, 0, 4>(4, L, H, L, H, L, H, 11));
    printf("  x_hi * w_lo only                                %7.1f\n", run3<4, 0, 4>(4, H, L, H, L, H, L, 11));
    printf("  x_lo * w_lo only                                %7.1f\n", run3<4, 0, 4>(4, L, L, L, L, L, L, 11));
    printf("  x_hi * zero                                     %7.1f\n", run3<4, 0, 4>(4, H, Z, H, Z, H, Z, 11));
    printf("  f16x3 mix (lo*hi, hi*hi, hi*lo), pass-major     %7.1f\n", run3<4, 0, 4>(4, L, H, H, H, H, L, 11));
    printf("  f16x3 mix, acc-major (3 products back to back)  %7.1f\n", run3<4, 1, 4>(4, L, H, H, H, H, L, 11));
    printf("  f16x3 mix, lo terms kept to 8 bits              %7.1f\n", run3<4, 0, 4>(4, L, H, H, H, H, L, 8));
    printf("  f16x3 mix, lo terms kept to 6 bits              %7.1f\n", run3<4, 0, 4>(4, L, H, H, H, H, L, 6));
    printf("  f16x3 mix, lo terms kept to 4 bits              %7.1f\n", run3<4, 0, 4>(4, L, H, H, H, H, L, 4));
    printf("  f16x3 mix, 2 waves/SIMD x 8 acc, pass-major     %7.1f\n", run3<8, 0, 2>(2, L, H, H, H, H, L, 11));
    printf("  f16x3 mix, 2 waves/SIMD x 8 acc, acc-major      %7.1f\n", run3<8, 1, 2>(2, L, H, H, H, H, L, 11));
    const int iters = 1500, grid = 1024;
    printf("  hi*hi, both operands change every instruction   %7.1f\n",
           timed([&] { hipLaunchKernelGGL((mfma_reuse_loop<0>), dim3(grid), dim3(256), 0, 0, g_out, iters); }, (double)grid * 4 * iters * 16 * 32768.0));
    printf("  hi*hi, A fixed for 4 instructions               %7.1f\n",
           timed([&] { hipLaunchKernelGGL((mfma_reuse_loop<1>), dim3(grid), dim3(256), 0, 0, g_out, iters); }, (double)grid * 4 * iters * 16 * 32768.0));
    printf("  hi*hi, A and B fixed for 4 instructions         %7.1f\n",
           timed([&] { hipLaunchKernelGGL((mfma_reuse_loop<2>), dim3(grid), dim3(256), 0, 0, g_out, iters); }, (double)grid * 4 * iters * 16 * 32768.0));
    printf("  hi*hi on v_mfma_f32_16x16x32_f16                %7.1f\n",
           timed([&] { hipLaunchKernelGGL(mfma16_loop, dim3(grid), dim3(256), 0, 0, g_out, iters, H, H); }, (double)grid * 4 * iters * 32 * 16384.0));
    printf("  lo*hi on v_mfma_f32_16x16x32_f16                %7.1f\n",
           timed([&] { hipLaunchKernelGGL(mfma16_loop, dim3(grid), dim3(256), 0, 0, g_out, iters, L, H); }, (double)grid * 4 * iters * 32 * 16384.0));
    printf("  zeros on v_mfma_f32_16x16x32_f16                %7.1f\n",
           timed([&] { hipLaunchKernelGGL(mfma16_loop, dim3(grid), dim3(256), 0, 0, g_out, iters, Z, Z); }, (double)grid * 4 * iters * 32 * 16384.0));
  }
  (void)hipFree(g_out);
  return 0;
}
