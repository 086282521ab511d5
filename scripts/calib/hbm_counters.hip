// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 with known byte counts, in the access patterns of the
// dominant conv kernel: 16-B-per-lane coalesced reads (its activation loads), 4-B-per-lane stores where each half-wave covers
// one 128-B row segment and rows are 512 B apart (its epilogue at Cout = 128), plus 16-B-per-lane stores for comparison.
// Every pass touches 2 GiB (8 x the 256 MiB Infinity Cache), so hits in it cannot hide traffic.
// build: hipcc --offload-arch=gfx950 -O3 hbm_counters.hip -o hbm_counters ; run under rocprofv3 --pmc (scripts/gpu_calib_hbm.sh)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_f4(const float4* __restrict__ in, float* out, size_t n4) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;   // never true for the fill pattern: keeps the loads alive, stores nothing
}

__global__ void write_f4(float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

// conv epilogue pattern: a wave stores 16 times; store r: lanes 0-31 -> row (8*(r>>2) + (r&3)), lanes 32-63 -> that row + 4,
// 32 consecutive floats each; a "tile" is 32 rows x 32 floats inside a [rows][128] fp32 image; every element written once
__global__ void write_row128(float* __restrict__ out, size_t rows) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t ntiles = (rows / 32) * 4;
  for (size_t t = wave; t < ntiles; t += nwaves) {
    const size_t r0 = (t >> 2) * 32;
    const int n0 = (int)(t & 3) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      out[(r0 + row) * 128 + n0 + c] = (float)r;
    }
  }
}

int main() {
  const size_t bytes = 2ull << 30;
  float* buf;
  float* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_f4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), sink, bytes / 16);
    hipLaunchKernelGGL(write_f4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4*>(buf), bytes / 16);
    hipLaunchKernelGGL(write_row128, dim3(4096), dim3(256), 0, 0, buf, bytes / 512);
  }
  hipDeviceSynchronize();
  printf("bytes per launch: %zu\n", bytes);
  // streaming rates with HIP events (round 4: what does this chip WRITE at? -- the yardstick for conv_in, whose launch is one
  // 1.07 GB fp32 output tensor at B = 32): 1 GiB and 2 GiB passes, best of 5
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (size_t nb : {(size_t)1 << 30, bytes}) {
    float best[3] = {1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep) {
      for (int k = 0; k < 3; ++k) {
        hipEventRecord(e0, 0);
        if (k == 0) hipLaunchKernelGGL(read_f4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), sink, nb / 16);
        if (k == 1) hipLaunchKernelGGL(write_f4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4*>(buf), nb / 16);
        if (k == 2) hipLaunchKernelGGL(write_row128, dim3(4096), dim3(256), 0, 0, buf, nb / 512);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best[k]) best[k] = ms;
      }
    }
    printf("stream %4zu MiB: read_f4 %.3f TB/s  write_f4 %.3f TB/s  write_row128 %.3f TB/s\n", nb >> 20, nb / (best[0] * 1e-3) / 1e12,
           nb / (best[1] * 1e-3) / 1e12, nb / (best[2] * 1e-3) / 1e12);
  }
  return 0;
}
