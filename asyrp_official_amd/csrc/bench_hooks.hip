// bench_hooks.hip -- the micro-benchmark and phase-stamp entry points of the PROFILING library (libasyrp_hip_bench.so): compiled and
// linked only by `python -m asyrp_official_amd.build --bench` (-DASYRP_BENCH_HOOKS); the product library does not contain this
// translation unit (round 6: split out of engine.hip).  Declared in include/asyrp.h under ASYRP_BENCH_HOOKS; driven by
// scripts/conv_bench.py, k32_phases.py, attn_phases.py, gemm1x1_phases.py.  This is the only file of the tree that calls getenv
// besides ab_env() (kernels.h, profiling build only).
#ifndef ASYRP_BENCH_HOOKS
#error "bench_hooks.hip belongs to the profiling library only (python -m asyrp_official_amd.build --bench)"
#endif
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/asyrp.h"
#include "kernels.h"

using namespace asyrp;

namespace {

int fail(int code, const std::string& msg) { return set_last_error(code, msg.c_str()); }

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t _e = (x);                                                                            \
    if (_e != hipSuccess) return fail(ASYRP_EHIP, std::string(#x) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define TRY(x)            \
  do {                    \
    int _r = (x);         \
    if (_r != 0) return _r; \
  } while (0)

__global__ void fill_hash_kernel(float* p, long long n, unsigned seed, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;     // uniform [-scale, scale)
  }
}

}  // namespace

extern "C" {

static int conv_bench_impl(int device, int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample,
                           int prologue, int residual, int conv_math, int tile, int abl, int iters, float* ms_out,
                           void* stream, unsigned long long* stamps_host, int stamps_cap);
int asyrp_op_conv_bench(int device, int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample,
                        int prologue, int residual, int conv_math, int tile, int abl, int iters, float* ms_out,
                        void* stream) {
  return conv_bench_impl(device, B, H, W, C0, C1, Cout, ksize, stride, upsample, prologue, residual, conv_math, tile, abl, iters,
                         ms_out, stream, nullptr, 0);
}
// the same, plus the phase stamps of the LAST launch ([workgroup][8] s_memrealtime ticks, K32_STAMP in conv_f16x3.hip; abl != 0)
int asyrp_op_conv_stamps(int device, int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample,
                         int prologue, int residual, int conv_math, int tile, int abl, int iters, float* ms_out,
                         unsigned long long* stamps_host, int stamps_cap) {
  return conv_bench_impl(device, B, H, W, C0, C1, Cout, ksize, stride, upsample, prologue, residual, conv_math, tile, abl, iters,
                         ms_out, nullptr, stamps_host, stamps_cap);
}
// abl bit 6 (64): the launch also emits GroupNorm partial sums (as conv1 of every block does)
static int conv_bench_impl(int device, int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample,
                           int prologue, int residual, int conv_math, int tile, int abl, int iters, float* ms_out,
                           void* stream, unsigned long long* stamps_host, int stamps_cap) {
  if (B < 1 || iters < 1 || !ms_out) return fail(ASYRP_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C0 + C1, HW = H * W;
  int Ho = H, Wo = W;
  if (upsample) { Ho *= 2; Wo *= 2; }
  if (stride == 2) { Ho /= 2; Wo /= 2; }
  std::vector<void*> tmp;
  // ASYRP_BENCH_ZERO=1: activations and weights all zero (what does the same instruction stream run at when the matrix pipe's
  // operands do not toggle?  profiling library only)
  static const bool zero_data = [] { const char* e = getenv("ASYRP_BENCH_ZERO"); return e && e[0] == '1'; }();
  auto dalloc = [&](size_t nfloats, float** p, float scale, unsigned seed) -> int {
    HIPCHK(hipMalloc(p, std::max<size_t>(nfloats, 1) * sizeof(float)));
    tmp.push_back(*p);
    hipLaunchKernelGGL(fill_hash_kernel, dim3(2048), dim3(256), 0, s, *p, (long long)nfloats, seed, zero_data ? 0.f : scale);
    return 0;
  };
  float *a0, *a1 = nullptr, *w, *wg, *bias, *yo, *rs = nullptr, *sc = nullptr, *sh = nullptr, *ca;
  TRY(dalloc((size_t)B * HW * C0, &a0, 2.0f, 1));
  if (C1) TRY(dalloc((size_t)B * HW * C1, &a1, 2.0f, 2));
  const float wb = 1.0f / std::sqrt((float)Cin * ksize * ksize);
  TRY(dalloc((size_t)Cout * Cin * ksize * ksize, &w, wb, 3));
  TRY(dalloc((size_t)Cout * Cin * ksize * ksize, &wg, wb, 3));
  TRY(dalloc(Cout, &bias, 0.1f, 4));
  TRY(dalloc((size_t)B * Cout, &ca, 0.5f, 5));
  TRY(dalloc((size_t)B * Ho * Wo * Cout, &yo, 0.f, 6));
  if (residual) TRY(dalloc((size_t)B * Ho * Wo * Cout, &rs, 1.0f, 7));
  if (prologue) {
    TRY(dalloc((size_t)B * Cin, &sc, 1.0f, 8));
    TRY(dalloc((size_t)B * Cin, &sh, 0.3f, 9));
  }
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = a0; g.c0 = C0; g.lda0 = C0; g.a0_zo = (long long)HW * C0;
  if (C1) { g.a1 = a1; g.c1 = C1; g.lda1 = C1; g.a1_zo = (long long)HW * C1; }
  g.Hin = H; g.Win = W; g.Hout = Ho; g.Wout = Wo; g.Cin = Cin; g.Cout = Cout;
  g.ks = ksize; g.stride = stride; g.ups = upsample; g.pad = (ksize == 3 && stride == 1) ? 1 : 0;
  g.pscale = sc; g.pshift = sh; g.silu = prologue ? 1 : 0;
  g.w = wg; g.ldb = Cout; g.bias = bias;
  g.chan_add = ca; g.ld_chan_add = Cout;
  if (rs) { g.resid = rs; g.ldr = Cout; g.r_zo = (long long)Ho * Wo * Cout; }
  g.alpha = 1.f; g.out = yo; g.ldo = Cout; g.o_zo = (long long)Ho * Wo * Cout; g.ZI = 1; g.Z = B;
  g.math = MATH_F32; g.tile = tile; g.abl = abl & 63;   // bit 6: GroupNorm partials, bit 7: stagger by arrival order, bits 8..: its delay in us
  if (abl & 64) {
    float* st;
    TRY(dalloc((size_t)B * 4 * ((Ho + 7) / 8) * ((Wo + 7) / 8) * Cout * 4, &st, 0.f, 14));   // >= [image][M block][Cout][2] doubles for any tile
    g.stats = reinterpret_cast<double*>(st);
  }
  unsigned long long* dbg = nullptr;
  size_t dbg_wgs = 0;
  if ((stamps_host && stamps_cap > 0) || (abl & 128)) {
    // [workgroup][8] stamps, then 2048 per-CU arrival counters (GemmArgs.stag == 3)
    dbg_wgs = std::max<size_t>((size_t)std::max(stamps_cap, 0), (size_t)B * ((Ho + 15) / 16) * ((Wo + 15) / 16) * ((Cout + 127) / 128));
    HIPCHK(hipMalloc(&dbg, dbg_wgs * 64 + 2048 * 8));
    tmp.push_back(dbg);
    HIPCHK(hipMemsetAsync(dbg, 0, dbg_wgs * 64 + 2048 * 8, s));
    g.dbg = dbg;
    if (abl & 128) { g.stag = 3; g.stag_ticks = ((abl >> 8) & 0xFFF) * 100; }
  }
  if (conv_math == ASYRP_MATH_F16X3 || conv_math == ASYRP_MATH_F16) {
    g.np = (conv_math == ASYRP_MATH_F16) ? 1 : 3;
    float* xp;
    TRY(dalloc((f16x3_packed_halfs(Cout, Cin, ksize) + 1) / 2, &xp, 0.f, 10));
    const float wscale = std::ldexp(1.0f, 10 - (int)std::floor(std::log2(wb)));
    HIPCHK(launch_pack_f16x3(w, xp, Cout, Cin, ksize, wscale, s));
    g.math = MATH_F16X3; g.wpk = xp; g.cout_pad = ((Cout + 127) / 128) * 128;
    g.alpha = 1.0f / (wscale * f16x3_act_scale());
    if (tile == XT_256x128K32UP) {   // polyphase form (timing only: four 2x2 images packed from the same synthetic weights)
      if (!upsample || ksize != 3 || stride != 1 || residual || (Cin & 31)) return fail(ASYRP_EINVAL, "polyphase tile: upsample 3x3 only");
      const size_t ph = f16x3_packed_halfs(Cout, Cin, 2);
      float* xpu;
      TRY(dalloc((4 * ph + 1) / 2, &xpu, 0.f, 12));
      for (int q = 0; q < 4; ++q)
        HIPCHK(launch_pack_f16x3(w + (size_t)q * Cout * Cin, reinterpret_cast<char*>(xpu) + (size_t)q * ph * 2, Cout, Cin, 2, wscale, s));
      g.poly = 1; g.ups = 0; g.Hout = H; g.Wout = W; g.tile = 0; g.wpk = xpu; g.w_phase = (long long)ph * 2;
    }
    if (tile == XT_G1_256 || tile == XT_G1_128) {   // gemm1x1.hip
      if (!gemm1x1_ok(g)) return fail(ASYRP_EINVAL, "shape not covered by the 1x1 kernel");
      float* xg;
      TRY(dalloc((gemm1x1_packed_halfs(Cout, Cin) + 1) / 2, &xg, 0.f, 13));
      HIPCHK(launch_gemm1x1_pack(w, xg, Cout, Cin, wscale, s));
      g.wpk = xg;
    }
  }
  const int sk = (tile == 0) ? splitk_factor(g) : 1;   // as the engine does when it picks the tile itself
  if (sk > 1) {
    g.sk = sk;
    g.tile = splitk_tile(g);
    TRY(dalloc((size_t)sk * B * Ho * Wo * Cout, &g.part, 0.f, 11));
  }
  auto once = [&]() -> hipError_t {
    if (g.stag == 3 && hipMemsetAsync(dbg + dbg_wgs * 8, 0, 2048 * 8, s) != hipSuccess) return hipErrorUnknown;
    hipError_t e = launch_gemm(g, s);
    if (e == hipSuccess && sk > 1) e = launch_splitk_reduce(g, s);
    return e;
  };
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  hipError_t le = hipSuccess;
  for (int i = 0; i < 2 && le == hipSuccess; ++i) le = once();
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && le == hipSuccess; ++i) le = once();
  (void)hipEventRecord(e1, s);
  hipError_t se = hipStreamSynchronize(s);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (dbg && stamps_host && stamps_cap > 0 && se == hipSuccess) (void)hipMemcpy(stamps_host, dbg, (size_t)stamps_cap * 64, hipMemcpyDeviceToHost);
  for (void* p : tmp) (void)hipFree(p);
  if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("conv bench launch: ") + hipGetErrorString(le));
  if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("conv bench sync: ") + hipGetErrorString(se));
  *ms_out = ms / iters;
  return 0;
}

// profiling library only: run attn_planes_kernel `iters` times on synthetic planes and return (a) the average launch time (HIP
// events) and (b) the phase stamps of the last launch, [B*heads*T/32 workgroups][8] s_memrealtime ticks (100 MHz)
int asyrp_op_attention_phases(int device, int B, int C, int T, int heads, int np, int iters, float* ms_out,
                              unsigned long long* stamps_host, void* stream) {
  const int Dh = C / heads;
  if (!attn_planes_supported(T, Dh) || iters < 1) return fail(ASYRP_EINVAL, "shape not covered");
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const size_t nqk = (size_t)B * T * 3 * C, nv = (size_t)B * C * T, nwg = (size_t)B * heads * (T / 32);
  float* qkv = nullptr; _Float16 *h, *l, *vh, *vl; float* out; unsigned long long* dbg;
  HIPCHK(hipMalloc(&qkv, nqk * 4)); HIPCHK(hipMalloc(&h, nqk * 2)); HIPCHK(hipMalloc(&l, nqk * 2));
  HIPCHK(hipMalloc(&vh, nv * 2)); HIPCHK(hipMalloc(&vl, nv * 2)); HIPCHK(hipMalloc(&out, nv * 4)); HIPCHK(hipMalloc(&dbg, nwg * 64));
  hipLaunchKernelGGL(fill_hash_kernel, dim3(2048), dim3(256), 0, s, qkv, (long long)nqk, 7u, 2.0f);
  HIPCHK(launch_qkv_to_planes(qkv, 3 * C, B, T, 3 * C, heads == 1 ? 3 * C : 3 * Dh, heads == 1 ? 2 * C : 2 * Dh, heads == 1 ? C : Dh, h, l, vh, vl, s));
  AttnArgs a;
  memset(&a, 0, sizeof a);
  a.qkh = h; a.qkl = l; a.vth = vh; a.vtl = vl; a.ld16 = 3 * C;
  a.head_stride = (heads == 1) ? 0 : 3 * Dh; a.q_off = 0; a.k_off = (heads == 1) ? C : Dh;
  a.B = B; a.heads = heads; a.T = T; a.Dh = Dh; a.scale = 1.0f / std::sqrt((float)Dh);
  a.out = out; a.ldo = C; a.o_img_stride = (long long)T * C; a.o_head_stride = Dh; a.np = np; a.dbg = dbg;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  hipError_t le = launch_attention_planes(a, s);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && le == hipSuccess; ++i) le = launch_attention_planes(a, s);
  (void)hipEventRecord(e1, s);
  hipError_t se = hipStreamSynchronize(s);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (stamps_host) (void)hipMemcpy(stamps_host, dbg, nwg * 64, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(qkv); (void)hipFree(h); (void)hipFree(l); (void)hipFree(vh); (void)hipFree(vl); (void)hipFree(out); (void)hipFree(dbg);
  if (le != hipSuccess || se != hipSuccess) return fail(ASYRP_EHIP, "attention phases run failed");
  *ms_out = ms / iters;
  return 0;
}

// phase stamps of gemm1x1_k32_kernel (per wave: start, prologue done, K loop done, end; 100 MHz s_memrealtime)
int asyrp_op_gemm1x1_phases(int device, int B, int H, int Cin, int Cout, int prologue, int np, int tile, int iters, float* ms_out,
                            unsigned long long* stamps_host, void* stream) {
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * H, bm = (tile == XT_G1_256) ? 256 : 128;
  const size_t nwave = (size_t)B * ((HW + bm - 1) / bm) * ((Cout + 127) / 128) * (bm / 32);
  float *a, *w, *o, *sc, *sh; void* xg; unsigned long long* dbg;
  HIPCHK(hipMalloc(&a, (size_t)B * HW * Cin * 4)); HIPCHK(hipMalloc(&w, (size_t)Cout * Cin * 4)); HIPCHK(hipMalloc(&o, (size_t)B * HW * Cout * 4));
  HIPCHK(hipMalloc(&sc, (size_t)B * Cin * 4)); HIPCHK(hipMalloc(&sh, (size_t)B * Cin * 4)); HIPCHK(hipMalloc(&xg, gemm1x1_packed_halfs(Cout, Cin) * 2));
  HIPCHK(hipMalloc(&dbg, nwave * 32));
  hipLaunchKernelGGL(fill_hash_kernel, dim3(2048), dim3(256), 0, s, a, (long long)B * HW * Cin, 3u, 2.0f);
  hipLaunchKernelGGL(fill_hash_kernel, dim3(2048), dim3(256), 0, s, w, (long long)Cout * Cin, 4u, 0.1f);
  hipLaunchKernelGGL(fill_hash_kernel, dim3(64), dim3(256), 0, s, sc, (long long)B * Cin, 5u, 1.0f);
  hipLaunchKernelGGL(fill_hash_kernel, dim3(64), dim3(256), 0, s, sh, (long long)B * Cin, 6u, 1.0f);
  HIPCHK(launch_gemm1x1_pack(w, xg, Cout, Cin, 8192.f, s));
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = a; g.c0 = Cin; g.lda0 = Cin; g.a0_zo = (long long)HW * Cin;
  g.Hin = H; g.Win = H; g.Hout = H; g.Wout = H; g.Cin = Cin; g.Cout = Cout; g.ks = 1; g.stride = 1;
  if (prologue) { g.pscale = sc; g.pshift = sh; g.silu = prologue > 1; }
  g.alpha = 1.f / 8192.f; g.out = o; g.ldo = Cout; g.o_zo = (long long)HW * Cout; g.ZI = 1; g.Z = B;
  g.math = MATH_F16X3; g.np = np; g.tile = tile; g.wpk = xg; g.cout_pad = ((Cout + 127) / 128) * 128;
  g.part = reinterpret_cast<float*>(dbg);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  hipError_t le = launch_gemm1x1(g, s);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && le == hipSuccess; ++i) le = launch_gemm1x1(g, s);
  (void)hipEventRecord(e1, s);
  hipError_t se = hipStreamSynchronize(s);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (stamps_host) (void)hipMemcpy(stamps_host, dbg, nwave * 32, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(w); (void)hipFree(o); (void)hipFree(sc); (void)hipFree(sh); (void)hipFree(xg); (void)hipFree(dbg);
  if (le != hipSuccess || se != hipSuccess) return fail(ASYRP_EHIP, "gemm1x1 phases run failed");
  *ms_out = ms / iters;
  return 0;
}

}  // extern "C"
