// Calibration micro-benchmark (not part of the product): what the K-loop SHAPE of an f16x3 tile can sustain on this chip
// with random data when everything but the fragment reads, the MFMAs and the per-step barrier is removed.
// One "step" = one tap of one 16-channel chunk: each wave reads (TM+TN)*2 fragments (ds_read_b128, conflict-free) and
// issues TM*TN*3 v_mfma_f32_32x32x16_f16; optional s_barrier per step.  Compare with scripts/calib/mfma_peak.hip
// (no LDS at all) and with the ablation lines of scripts/conv_bench.py (the real kernels with their loads removed).
//   hipcc --offload-arch=gfx950 -O3 scripts/calib/loop_shapes.hip -o /tmp/loop_shapes && /tmp/loop_shapes
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// NW waves per workgroup, wave tile TM x TN MFMA blocks, BAR: barrier per step, MINW: waves per SIMD to budget registers for
// PERPASS: fragments are fetched per MFMA pass into TM + TN registers (the 128-VGPR organisation of the 8-wave tile)
// instead of all (TM+TN)*2 at the top of the step
template <int NW, int TM, int TN, bool BAR, int MINW, bool PERPASS = false>
__global__ void __launch_bounds__(NW * 64, MINW) loop_kernel(float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDS_BYTES = 32 * 1024;                 // fragment pool: 2048 x 16 B
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < LDS_BYTES / 2; i += NW * 64) {
    const unsigned h = hashu(i * 2654435761u + blockIdx.x);
    reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(((int)(h & 2047) - 1024) * (1.0f / 1024.0f));
  }
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int a = 0; a < TM; ++a)
    for (int b = 0; b < TN; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wave = tid >> 6;
  // a wave's fragment = 64 lanes x 16 B contiguous (conflict-free); base moves every step so the data differ
  for (int s = 0; s < steps; ++s) {
    const int base = ((s * 7 + wave * 3) & 15) * 1024;
    if (PERPASS) {
      h8 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const h8*>(smem + ((base + a * 2048) & (LDS_BYTES - 1)) + lane * 16);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = *reinterpret_cast<const h8*>(smem + ((base + 16384 + b * 2048) & (LDS_BYTES - 1)) + lane * 16);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const h8*>(smem + ((base + a * 2048 + 1024) & (LDS_BYTES - 1)) + lane * 16);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = *reinterpret_cast<const h8*>(smem + ((base + 16384 + b * 2048 + 1024) & (LDS_BYTES - 1)) + lane * 16);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    h8 al[TM], ah[TM], bh[TN], bl[TN];
  #pragma unroll
      for (int a = 0; a < TM; ++a) {
        al[a] = *reinterpret_cast<const h8*>(smem + ((base + a * 2048) & (LDS_BYTES - 1)) + lane * 16);
        ah[a] = *reinterpret_cast<const h8*>(smem + ((base + a * 2048 + 1024) & (LDS_BYTES - 1)) + lane * 16);
      }
  #pragma unroll
      for (int b = 0; b < TN; ++b) {
        bh[b] = *reinterpret_cast<const h8*>(smem + ((base + 16384 + b * 2048) & (LDS_BYTES - 1)) + lane * 16);
        bl[b] = *reinterpret_cast<const h8*>(smem + ((base + 16384 + b * 2048 + 1024) & (LDS_BYTES - 1)) + lane * 16);
      }
  #pragma unroll
      for (int a = 0; a < TM; ++a)
  #pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
  #pragma unroll
      for (int a = 0; a < TM; ++a)
  #pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
  #pragma unroll
      for (int a = 0; a < TM; ++a)
  #pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float sum = 0.f;
  for (int a = 0; a < TM; ++a)
    for (int b = 0; b < TN; ++b)
      for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
  out[blockIdx.x * (NW * 64) + tid] = sum;
}

template <int NW, int TM, int TN, bool BAR, int MINW, bool PERPASS = false>
static void run(const char* name, int wgs_per_cu) {
  const int steps = 4000, grid = 256 * wgs_per_cu;
  float* out;
  (void)hipMalloc(&out, (size_t)grid * NW * 64 * sizeof(float));
  auto k = loop_kernel<NW, TM, TN, BAR, MINW, PERPASS>;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 32 * 1024, 0, out, steps);
  (void)hipDeviceSynchronize();
  double mean = 0.0;
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 32 * 1024, 0, out, steps);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    mean += 3.0 * (double)grid * NW * steps * TM * TN * 3 * 32768.0 / (ms * 1e-3) / 1e12 / reps;
  }
  printf("%-72s %7.1f TFLOP/s f16 MFMA = %6.1f fp32-equivalent; %4.2f ds_read_b128 per MFMA\n", name, mean, mean / 3.0,
         (double)(TM + TN) * 2 / (TM * TN * 3));
  (void)hipFree(out);
}


// the same wave tile (64 x 64 outputs) on v_mfma_f32_16x16x32_f16: one step = K 32 (two taps of a 16-channel chunk), 4 x 4
// accumulator blocks of 16 x 16, (4 + 4) * 2 fragments of 1 KB (the same LDS bytes per flop as the 32x32x16 organisation)
// and 48 instructions of 16 cycles.  HALF: B fragments of all four column blocks stay live, the A fragments are fetched for
// two row blocks at a time (24 fragment registers instead of 32).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NW, bool BAR, int MINW, bool HALF>
__global__ void __launch_bounds__(NW * 64, MINW) loop16_kernel(float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDS_BYTES = 32 * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < LDS_BYTES / 2; i += NW * 64) {
    const unsigned h = hashu(i * 2654435761u + blockIdx.x);
    reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(((int)(h & 2047) - 1024) * (1.0f / 1024.0f));
  }
  __syncthreads();
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  const int wave = tid >> 6;
  auto frag = [&](int off) { return *reinterpret_cast<const h8*>(smem + (off & (LDS_BYTES - 1)) + lane * 16); };
  for (int s = 0; s < steps; ++s) {
    const int base = ((s * 7 + wave * 3) & 15) * 1024;
    h8 fb[4];
    // pass 1: x_lo * w_hi, pass 2: x_hi * w_hi (same B), pass 3: x_hi * w_lo (new B)
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      if (pass != 1) {
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = frag(base + 16384 + b * 2048 + (pass == 2 ? 1024 : 0));
      }
      if (HALF) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          h8 fa[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) fa[a] = frag(base + (hh * 2 + a) * 2048 + (pass == 0 ? 0 : 1024));
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[hh * 2 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[hh * 2 + a][b], 0, 0, 0);
        }
      } else {
        h8 fa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = frag(base + a * 2048 + (pass == 0 ? 0 : 1024));
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < 4; ++r) sum += acc[a][b][r];
  out[blockIdx.x * (NW * 64) + tid] = sum;
}

template <int NW, bool BAR, int MINW, bool HALF>
static void run16(const char* name, int wgs_per_cu) {
  const int steps = 2000, grid = 256 * wgs_per_cu;
  float* out;
  (void)hipMalloc(&out, (size_t)grid * NW * 64 * sizeof(float));
  auto k = loop16_kernel<NW, BAR, MINW, HALF>;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 32 * 1024, 0, out, steps);
  (void)hipDeviceSynchronize();
  double mean = 0.0;
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 32 * 1024, 0, out, steps);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    mean += 3.0 * (double)grid * NW * steps * 48 * 16384.0 / (ms * 1e-3) / 1e12 / reps;
  }
  printf("%-72s %7.1f TFLOP/s f16 MFMA = %6.1f fp32-equivalent; 0.33 ds_read_b128 per (half-size) MFMA\n", name, mean, mean / 3.0);
  (void)hipFree(out);
}

int main() {
  run<8, 2, 2, false, 4>("8 waves x (64x64), 2 WG/CU = 4 waves/SIMD, no barrier", 2);
  run<8, 2, 2, true, 4>("8 waves x (64x64), 2 WG/CU = 4 waves/SIMD, barrier per tap   [main tile]", 2);
  run<8, 2, 2, true, 4, true>("8 waves x (64x64), 2 WG/CU, barrier per tap, fragments fetched PER PASS", 2);
  run<4, 2, 4, false, 2>("4 waves x (64x128), 2 WG/CU = 2 waves/SIMD, no barrier", 2);
  run<4, 2, 4, true, 2>("4 waves x (64x128), 2 WG/CU = 2 waves/SIMD, barrier per tap  [pipelined tile]", 2);
  run<4, 4, 4, false, 1>("4 waves x (128x128), 1 WG/CU = 1 wave/SIMD, no barrier", 1);
  run<4, 4, 4, true, 1>("4 waves x (128x128), 1 WG/CU = 1 wave/SIMD, barrier per tap", 1);
  run<8, 4, 2, true, 2>("8 waves x (128x64), 1 WG/CU = 2 waves/SIMD, barrier per tap", 1);
  run<16, 2, 2, true, 4>("16 waves x (64x64), 1 WG/CU = 4 waves/SIMD, barrier per tap", 1);
  run16<8, true, 4, false>("16x16x32: 8 waves x (64x64), 2 WG/CU, barrier per K=32 step, 4+4 fragments per pass", 2);
  run16<8, true, 4, true>("16x16x32: 8 waves x (64x64), 2 WG/CU, barrier per K=32 step, 2+4 fragments per half pass", 2);
  run16<8, false, 4, true>("16x16x32: 8 waves x (64x64), 2 WG/CU, no barrier, 2+4 fragments per half pass", 2);
  run<8, 2, 2, true, 4, true>("32x32x16 again: 8 waves x (64x64), 2 WG/CU, barrier per tap, fragments PER PASS", 2);
  run16<8, true, 4, true>("16x16x32 again: 8 waves x (64x64), 2 WG/CU, barrier per step, half passes", 2);
  return 0;
}
