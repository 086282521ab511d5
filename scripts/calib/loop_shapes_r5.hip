// Calibration micro-benchmark (not part of the product), round 5: what the K-loop SHAPE of three candidate
// tilings of the big 3x3 layers sustains on this chip -- fragment reads, the per-step weight stream by LDS-DMA (global_load_lds, L2-hot
// slices as in the product), one barrier per K = 32 step, v_mfma_f32_16x16x32_f16 in the product's pass order on UNet-like operands
// (x = SiLU(N(0,1)) and w = N(0,1) * 2^8, both as exact two-term f16 splits) -- with no activation staging, no prologue, no epilogue:
//   A  the product's main tile: 8 waves x (64 px x 64 ch), 2 workgroups per CU, 16 KB of weights per step, 16 fragments / 48 MFMA
//   B  a 64 x 128 wave tile: 4 waves x (64 px x 128 ch) (128 accumulator VGPRs, 2 waves per SIMD), 24 fragments / 96 MFMA
//   C  1-D Winograd F(2,3) along x: 4 transform planes x 128 pixel pairs x 128 channels per workgroup = 16 waves x (64 rows x 64 ch),
//      1 workgroup per CU, FOUR weight slices (64 KB) per K = 32 step; 2/3 of the direct form's matrix work per output, so its
//      direct-equivalent rate is 1.5 x its matrix rate.  (Its real LDS budget does not close: 2 x 64 KB of weight slots + the
//      transformed halo of two 16-channel chunks, 2 x 36.8 KB = 201 KB > 160 KB; the A pool here is kept at the product's size so that
//      the number is an UPPER bound of the shape.)
// Shader clock and socket power are read from sysfs (hwmon) in the middle of each run.
//   hipcc --offload-arch=gfx950 -O3 scripts/calib/loop_shapes_r5.hip -o /tmp/loop_shapes_r5 && /tmp/loop_shapes_r5
#include <hip/hip_runtime.h>
#include <glob.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline float urand(unsigned k) { return (hashu(k) >> 8) * (1.0f / 16777216.0f); }
__device__ inline float nrand(unsigned k) {
  const float u1 = urand(2 * k) + 1e-7f, u2 = urand(2 * k + 1);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

// weight image in global memory, the product's slice order: [slice][4 units: hi k0-7, hi k8-15, lo k0-7, lo k8-15][128 couts][8 f16]
__global__ void fill_weights(_Float16* w, size_t n8, int zero) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int unit = (int)((i / 128) & 3);
    for (int j = 0; j < 8; ++j) {
      const float x = zero ? 0.f : nrand((unsigned)(i * 8 + j) * 2654435761u + 17u) * 256.f;
      const _Float16 hi = (_Float16)x;
      w[i * 8 + j] = (unit < 2) ? hi : (_Float16)(x - (float)hi);
    }
  }
}

constexpr int PLANE = 336, TW = 18;
constexpr int A_BYTES = 4 * PLANE * 16;      // one halo buffer of the product: [4 units][336 px][16 B]

// NW waves as WMW x WNW; a wave owns 64 rows x WCH channels; NSL weight slices of 8 KB per K = 32 step (2 for the direct forms, 8 for
// the Winograd form: 4 planes x 2); NABUF halo buffers
template <int NW, int WNW, int WCH, int NSL, int NABUF, int MINW, int NPLANE>
__global__ void __launch_bounds__(NW * 64, MINW) shape_kernel(const _Float16* __restrict__ wg, size_t wbytes, float* out, int steps, int zero) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 128, B_BYTES = BN * 64, SLOT = NSL * B_BYTES, TN = WCH / 16;
  constexpr int WPP = NW / NPLANE;             // waves per transform plane
  constexpr int WMW = WPP / WNW;
  char* const Bs = smem;
  char* const As = smem + 2 * SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = wave / WPP, wp = wave - plane * WPP, wm = wp / WNW, wn = wp - wm * WNW;
  // UNet-like activations: hi planes (units 0, 1) and lo planes (units 2, 3)
  for (int i = tid; i < NABUF * A_BYTES / 2; i += NW * 64) {
    const int unit = ((i * 2) % A_BYTES) / (PLANE * 16);
    float x = zero ? 0.f : nrand((unsigned)i * 40503u + blockIdx.x * 977u + 5u);
    x = x / (1.f + expf(-x));
    if (NPLANE > 1 && !zero) {   // Winograd input transform: sums / differences of two activated pixels
      float y = nrand((unsigned)i * 40503u + blockIdx.x * 977u + 77777u);
      y = y / (1.f + expf(-y));
      x = (i & 1) ? x + y : x - y;
    }
    const _Float16 hi = (_Float16)x;
    reinterpret_cast<_Float16*>(As)[i] = (unit < 2) ? hi : (_Float16)(x - (float)hi);
  }
  const char* wsrc = reinterpret_cast<const char*>(wg);
  auto issue_slot = [&](int s, int slot) {
    // SLOT / 1 KiB pieces, spread over the waves (the product: 16 pieces over 8 waves)
#pragma unroll
    for (int k = 0; k < (SLOT / 1024 + NW - 1) / NW; ++k) {
      const int pc = wave + k * NW;
      if ((SLOT / 1024) % NW != 0 && pc >= SLOT / 1024) break;
      const size_t off = ((size_t)s * SLOT + (size_t)pc * 1024) % wbytes;
      const char* src = wsrc + off + lane * 16;
      char* dst = Bs + slot * SLOT + pc * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  const int r16 = lane & 15, kq = lane >> 4, tp = kq >> 1, kh = kq & 1;
  const int a_lane = (kh * PLANE + (wm * 4) * TW + r16) * 16;
  constexpr int A_TM = TW * 16;
  // B: the wave's plane has its own pair of slices (Winograd), [tap-of-step][4 units][128][16 B]
  const int b_lane = plane * 2 * B_BYTES + tp * B_BYTES + (kh * BN + wn * WCH + r16) * 16;
  f32x4 acc[4][TN];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  issue_slot(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int t0 = 0;
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) issue_slot(s + 1, (s + 1) & 1);
    int t1 = t0 + 1;
    if (t1 == 9) t1 = 0;
    const int ky0 = (t0 * 11) >> 5, ky1 = (t1 * 11) >> 5;
    const int offA0 = ((NABUF > 1) ? (s & 1) * A_BYTES : 0) + (ky0 * TW + (t0 - 3 * ky0)) * 16;
    const int offA1 = ((NABUF > 1) ? (s & 1) * A_BYTES : 0) + (ky1 * TW + (t1 - 3 * ky1)) * 16;
    const char* A = As + a_lane + (tp ? offA1 : offA0);
    const char* B = Bs + (s & 1) * SLOT + b_lane;
    h8 fa[4], fb[TN];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + tm * A_TM + 2 * PLANE * 16);        // x_lo
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 256);                          // w_hi
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) fa[tm] = *reinterpret_cast<const h8*>(A + tm * A_TM);                          // x_hi
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) fb[tn] = *reinterpret_cast<const h8*>(B + tn * 256 + 2 * BN * 16);            // w_lo
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
    t0 += 2;
    if (t0 >= 9) t0 -= 9;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[a][b][r];
  out[(size_t)blockIdx.x * (NW * 64) + tid] = sum;
}

static std::string sysfs_first(const char* pat) {
  glob_t g;
  std::string r;
  if (glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc > 0) r = g.gl_pathv[0];
  globfree(&g);
  return r;
}
static double read_num(const std::string& p) {
  if (p.empty()) return -1;
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}
static void read_clk_power(double* mhz, double* watts) {
  static const std::string fq = sysfs_first("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input");
  static const std::string pa = sysfs_first("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average");
  static const std::string pi = sysfs_first("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input");
  const double f = read_num(fq);
  double p = read_num(pa);
  if (p < 0) p = read_num(pi);
  *mhz = f > 0 ? f / 1e6 : -1;
  *watts = p > 0 ? p / 1e6 : -1;
}

template <int NW, int WNW, int WCH, int NSL, int NABUF, int MINW, int NPLANE>
static void run(const char* name, int wgs_per_cu, double equiv, const _Float16* w, size_t wbytes, int zero) {
  constexpr int SLOT = NSL * 128 * 64;
  const size_t smem = 2 * (size_t)SLOT + NABUF * (size_t)A_BYTES;
  auto k = shape_kernel<NW, WNW, WCH, NSL, NABUF, MINW, NPLANE>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
    printf("%-60s: LDS %zu B refused\n", name, smem);
    return;
  }
  const int grid = 256 * wgs_per_cu;
  const int mfma_per_step = 12 * (WCH / 16);                    // per wave
  const int steps = (int)(1200.0 * 48 / mfma_per_step);         // the same matrix work per wave in every shape
  float* out;
  (void)hipMalloc(&out, (size_t)grid * NW * 64 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), smem, 0, w, wbytes, out, steps, zero);
  (void)hipDeviceSynchronize();
  const int launches = 60;                                       // ~1.5-2.5 s: long enough for the power controller to settle
  (void)hipEventRecord(e0, 0);
  for (int j = 0; j < launches; ++j) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), smem, 0, w, wbytes, out, steps, zero);
  (void)hipEventRecord(e1, 0);
  double mhz = 0, watts = 0; int n = 0;
  usleep(600 * 1000);
  while (hipEventQuery(e1) == hipErrorNotReady && n < 40) {
    double f, p;
    read_clk_power(&f, &p);
    mhz += f; watts += p; ++n;
    usleep(100 * 1000);
  }
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)launches * grid * NW * steps * mfma_per_step * 16384.0 / (ms * 1e-3) / 1e12;
  printf("%-60s %7.1f TFLOP/s f16 MFMA = %6.1f fp32-equivalent = %6.1f direct-conv-equivalent; LDS %3zu KB; sclk %5.0f MHz %5.0f W (n=%d)\n",
         name, tf, tf / 3.0, tf / 3.0 * equiv, smem / 1024, n ? mhz / n : -1.0, n ? watts / n : -1.0, n);
  fflush(stdout);
  (void)hipFree(out);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
  const size_t wbytes = 4u << 20;   // 4 MB of packed slices: L2-resident, like a layer's weight image
  _Float16* w;
  (void)hipMalloc(&w, wbytes);
  for (int zero = 0; zero < 2; ++zero) {
    hipLaunchKernelGGL(fill_weights, dim3(1024), dim3(256), 0, 0, w, wbytes / 16, zero);
    (void)hipDeviceSynchronize();
    printf("== operands: %s\n", zero ? "all zero (the clock-limited ceiling of each shape)" : "UNet-like (SiLU(N(0,1)) x N(0,1) * 2^8, two-term f16 splits)");
    for (int rep = 0; rep < 2; ++rep) {
      run<8, 2, 64, 2, 2, 4, 1>("A  main tile: 8 waves x (64x64), 2 WG/CU, 16 KB/step", 2, 1.0, w, wbytes, zero);
      run<4, 1, 128, 2, 2, 2, 1>("B  64x128 wave tile: 4 waves, 2 WG/CU, 16 KB/step", 2, 1.0, w, wbytes, zero);
      run<16, 2, 64, 8, 1, 4, 4>("C  Winograd F(2,3): 16 waves x (64x64), 1 WG/CU, 64 KB/step", 1, 1.5, w, wbytes, zero);
      if (zero) break;
    }
  }
  return 0;
}
