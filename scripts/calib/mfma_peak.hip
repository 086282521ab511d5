// Calibration micro-benchmark (not part of the product): sustained rate of v_mfma_f32_32x32x16_f16 on the whole chip with
// register-resident operands (no LDS, no global loads in the loop), random vs zero operands, 1 / 2 / 4 waves per SIMD.
// Gives the power-limited ceiling the f16x3 kernels can be priced against next to the 2.5 PFLOP/s spec figure.
//   hipcc --offload-arch=gfx950 -O3 scripts/calib/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int NACC, int MINW>
__global__ void __launch_bounds__(256, MINW) mfma_loop(float* out, int iters, int zero) {
  h8 a[4], b[4];
  for (int s = 0; s < 4; ++s)
    for (int j = 0; j < 8; ++j) {
      const unsigned h = hashu((threadIdx.x + 977u * blockIdx.x) * 64u + s * 8u + j);
      a[s][j] = zero ? (_Float16)0.f : (_Float16)(((int)(h & 2047) - 1024) * (1.0f / 1024.0f));
      b[s][j] = zero ? (_Float16)0.f : (_Float16)(((int)((h >> 11) & 2047) - 1024) * (1.0f / 1024.0f));
    }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[(i * 3 + u) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MINW>
static void run(const char* name, int wgs_per_cu, int zero) {
  const int iters = 2000, grid = 256 * wgs_per_cu;
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((mfma_loop<NACC, MINW>), dim3(grid), dim3(256), 0, 0, out, iters, zero);
  hipDeviceSynchronize();
  float best = 0.f, mean = 0.f;
  const int reps = 6;
  for (int rep = 0; rep < reps; ++rep) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((mfma_loop<NACC, MINW>), dim3(grid), dim3(256), 0, 0, out, iters, zero);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 4.0 * (double)grid * 4 * iters * 4 * NACC * 32768.0 / (ms * 1e-3) / 1e12;
    mean += tf / reps; if (tf > best) best = tf;
  }
  printf("%-44s %s operands: mean %7.1f  best %7.1f TFLOP/s (f16 MFMA)  = %6.1f fp32-equivalent with 3 products\n", name,
         zero ? "zero  " : "random", mean, best, mean / 3.0);
  hipFree(out);
}

int main() {
  for (int zero = 0; zero < 2; ++zero) {
    run<8, 1>("1 wave/SIMD, 8 accumulators (4 waves/CU)", 1, zero);
    run<8, 2>("2 waves/SIMD, 8 accumulators (8 waves/CU)", 2, zero);
    run<4, 4>("4 waves/SIMD, 4 accumulators (16 waves/CU)", 4, zero);
  }
  return 0;
}
