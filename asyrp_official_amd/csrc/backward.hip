// backward.hip — kernels of the DeltaBlock training step (diffusion_latent.py:301-354): the data-gradient
// path through decoder #2 and the parameter gradients of the DeltaBlock.  The heavy work (transposed 3x3 / 1x1 convolutions)
// reuses the implicit-GEMM kernels with transposed, 180-degree-rotated weight images; this file holds the HBM-bound glue:
//   silu_gn_bwd_partial   dy = dA * act'(GN(x)) and the per-channel partial sums the GroupNorm backward needs
//   gn_bwd_finalize       per-(image, group) reductions -> per-(image, channel) coefficients of dx = A*dy + C*x + D
//   gn_bwd_apply          dx (+ another gradient branch) for the leading channels of a (virtually concatenated) input
//   sum2x2                backward of nearest x2 upsampling
//   transpose / softmax_bwd   attention backward glue (the four GEMMs run on igemm_f32)
//   colsum, act_apply, gn_param_grad   DeltaBlock parameter gradients
// All reductions are fixed-order (double accumulators): gradients are deterministic and batch-order independent.
#include "kernels.h"

#include <math.h>

namespace asyrp {

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- dy = dA * act'(y), y = x*scale + shift; partial[n][blk][c] = {sum_pix dy, sum_pix dy*x} ----
__global__ void silu_gn_bwd_partial_kernel(const ActBwdArgs p, int ppb, int nblk, int Q, int PL) {
  extern __shared__ __attribute__((aligned(16))) double bsm[];   // [PL][C][2]
  const int n = blockIdx.y, blk = blockIdx.x;
  const int tid = threadIdx.x, q = tid % Q, pl = tid / Q;
  const int c = q * 4;
  const float* xs;
  int ldx;
  if (c < p.c0) { xs = p.x0 + (long long)n * p.x0_z + c; ldx = p.ldx0; }
  else { xs = p.x1 + (long long)n * p.x1_z + (c - p.c0); ldx = p.ldx1; }
  const float4 sc = *reinterpret_cast<const float4*>(p.scale + (size_t)n * p.C + c);
  const float4 sh = *reinterpret_cast<const float4*>(p.shift + (size_t)n * p.C + c);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  const int pend = min(p.HW, (blk + 1) * ppb);
  for (int pix = blk * ppb + pl; pix < pend; pix += PL) {
    const float4 xv = *reinterpret_cast<const float4*>(xs + (long long)pix * ldx);
    const float4 dv = *reinterpret_cast<const float4*>(p.dA + ((long long)n * p.HW + pix) * p.ldd + c);
    const float x[4] = {xv.x, xv.y, xv.z, xv.w}, d[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = d[j];
      if (p.silu) {
        const float y = x[j] * scv[j] + shv[j];
        const float s = sigmoid_f(y);
        g = g * (s * (1.0f + y * (1.0f - s)));      // d/dy [y * sigmoid(y)]
      }
      o[j] = g;
      s1[j] += (double)g;
      s2[j] += (double)g * (double)x[j];
    }
    *reinterpret_cast<float4*>(p.dy + ((long long)n * p.HW + pix) * p.C + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bsm[((size_t)pl * p.C + c + j) * 2 + 0] = s1[j];
    bsm[((size_t)pl * p.C + c + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0, b = 0;
      for (int l = 0; l < PL; ++l) {
        a += bsm[((size_t)l * p.C + c + j) * 2 + 0];
        b += bsm[((size_t)l * p.C + c + j) * 2 + 1];
      }
      double* dst = p.partial + (((size_t)n * nblk + blk) * p.C + c + j) * 2;
      dst[0] = a;
      dst[1] = b;
    }
  }
}

static inline int bwd_ppb(int HW) {
  int ppb = HW / 64;
  if (ppb < 64) ppb = 64;
  if (ppb > HW) ppb = HW;
  return ppb;
}
int act_bwd_nblk(int HW) { const int ppb = bwd_ppb(HW); return (HW + ppb - 1) / ppb; }

hipError_t launch_act_bwd_partial(const ActBwdArgs& a, hipStream_t s) {
  if ((a.C & 3) || (a.c0 & 3) || (a.ldx0 & 3) || (a.x1 && (a.ldx1 & 3)) || (a.ldd & 3)) return hipErrorInvalidValue;
  const int Q = a.C / 4;
  const int PL = Q >= 256 ? 1 : 256 / Q;
  if (Q * PL > 1024) return hipErrorInvalidValue;
  const int ppb = bwd_ppb(a.HW), nblk = act_bwd_nblk(a.HW);
  const size_t sm = (size_t)PL * a.C * 2 * sizeof(double);
  hipLaunchKernelGGL(silu_gn_bwd_partial_kernel, dim3(nblk, a.N), dim3(Q * PL), sm, s, a, ppb, nblk, Q, PL);
  return hipGetLastError();
}

// ---- GroupNorm backward coefficients: dx = A_c*dy + Cg*x + Dg  (y = gamma*(x-mean)*rstd + beta, statistics over a group) ----
//   S1 = sum_group gamma*dy, S2 = sum_group gamma*dy*xhat, m = cg*HW
//   dx = rstd*(gamma*dy - S1/m - xhat*S2/m)
__global__ void gn_bwd_finalize_kernel(const GnBwdFinArgs p) {
  __shared__ double sm[256][2];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int cg = p.C / 32;
  const double mean = (double)p.mr[((size_t)n * 32 + g) * 2], rstd = (double)p.mr[((size_t)n * 32 + g) * 2 + 1];
  double a = 0.0, b = 0.0;
  for (int j = 0; j < cg; ++j) {
    const int c = g * cg + j;
    const double gam = (double)p.gamma[c] * (p.film_scale ? 1.0 + (double)p.film_scale[(size_t)n * p.ld_film + c] : 1.0);
    double p1 = 0.0, p2 = 0.0;
    for (int k = tid; k < p.nblk; k += 256) {
      const double* q = p.partial + (((size_t)n * p.nblk + k) * p.C + c) * 2;
      p1 += q[0];
      p2 += q[1];
    }
    a += gam * p1;
    b += gam * rstd * (p2 - mean * p1);
  }
  sm[tid][0] = a;
  sm[tid][1] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { sm[tid][0] += sm[tid + o][0]; sm[tid][1] += sm[tid + o][1]; }
    __syncthreads();
  }
  const double m = (double)cg * p.HW;
  const double S1 = sm[0][0], S2 = sm[0][1];
  if (tid < cg) {
    const int c = g * cg + tid;
    float* dst = p.coef + ((size_t)n * p.C + c) * 3;
    dst[0] = (float)(rstd * (double)p.gamma[c] * (p.film_scale ? 1.0 + (double)p.film_scale[(size_t)n * p.ld_film + c] : 1.0));
    dst[1] = (float)(-rstd * rstd * S2 / m);
    dst[2] = (float)(-rstd * S1 / m + mean * rstd * rstd * S2 / m);
  }
}

hipError_t launch_gn_bwd_finalize(const GnBwdFinArgs& a, hipStream_t s) {
  if (a.C % 32 != 0 || a.C / 32 > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(32, a.N), dim3(256), 0, s, a);
  return hipGetLastError();
}

// dx[n][pix][c] = A*dy + Cg*x + Dg (+ add) for c < Cd (the leading channels, all inside source x0)
__global__ void gn_bwd_apply_kernel(const float* dy, int C, const float* x0, int ldx0, long long x0_z, const float* coef,
                                    const float* add, float* dx, int Cd, int HW, long long total4) {
  const int C4 = Cd >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long long np = i / C4;
    const int n = (int)(np / HW);
    const long long pix = np - (long long)n * HW;
    const float4 d = *reinterpret_cast<const float4*>(dy + np * C + c);
    const float4 x = *reinterpret_cast<const float4*>(x0 + (long long)n * x0_z + pix * ldx0 + c);
    const float* cf = coef + ((size_t)n * C + c) * 3;
    float4 o;
    o.x = cf[0] * d.x + (cf[1] * x.x + cf[2]);
    o.y = cf[3] * d.y + (cf[4] * x.y + cf[5]);
    o.z = cf[6] * d.z + (cf[7] * x.z + cf[8]);
    o.w = cf[9] * d.w + (cf[10] * x.w + cf[11]);
    if (add) {
      const float4 a = *reinterpret_cast<const float4*>(add + np * Cd + c);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *reinterpret_cast<float4*>(dx + np * Cd + c) = o;
  }
}

static inline int bw_blocks(long long n) { long long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

hipError_t launch_gn_bwd_apply(const float* dy, int C, const float* x0, int ldx0, long long x0_z, const float* coef,
                               const float* add, float* dx, int Cd, int HW, int N, hipStream_t s) {
  if ((C & 3) || (Cd & 3) || (ldx0 & 3) || Cd > C) return hipErrorInvalidValue;
  const long long total4 = (long long)N * HW * (Cd / 4);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(bw_blocks(total4)), dim3(256), 0, s, dy, C, x0, ldx0, x0_z, coef, add, dx, Cd,
                     HW, total4);
  return hipGetLastError();
}

// ---- nearest x2 backward: out[n][y][x][c] = sum of the 2x2 block of in (NHWC, in is 2H x 2W) ----
__global__ void sum2x2_kernel(const float* in, float* out, int H, int W, int C, long long total4) {
  const int C4 = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const long long n = r / H;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const float4 v = *reinterpret_cast<const float4*>(in + ((n * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx) * C + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    *reinterpret_cast<float4*>(out + ((n * H + y) * W + x) * C + c) = acc;
  }
}

hipError_t launch_sum2x2(const float* in, float* out, int N, int H, int W, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  const long long total4 = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(sum2x2_kernel, dim3(bw_blocks(total4)), dim3(256), 0, s, in, out, H, W, C, total4);
  return hipGetLastError();
}

// ---- batched transpose: out[z][c][r] = in[z][r][c]  (in row stride ldi) ----
__global__ void transpose_kernel(const float* in, int ldi, long long in_z, float* out, int R, int Ccols, long long out_z) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < Ccols) ? in[(long long)z * in_z + (long long)r * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < Ccols && r < R) out[(long long)z * out_z + (long long)c * R + r] = tile[tx][k];
  }
}

hipError_t launch_transpose(const float* in, int ldi, long long in_z, float* out, int R, int Ccols, long long out_z, int Z,
                            hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((Ccols + 31) / 32, (R + 31) / 32, Z), dim3(256), 0, s, in, ldi, in_z, out, R, Ccols,
                     out_z);
  return hipGetLastError();
}

// ---- softmax backward, one wave per row: dS = P * (dP - sum_k dP*P) * scale, in place over dP ----
__global__ void softmax_bwd_kernel(const float* P, float* dP, long long rows, int T, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = P + row * T;
  float* dr = dP + row * T;
  double acc = 0.0;
  for (int i = lane; i < T; i += 64) acc += (double)dr[i] * (double)pr[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  const float r = (float)acc;
  for (int i = lane; i < T; i += 64) dr[i] = pr[i] * (dr[i] - r) * scale;
}

hipError_t launch_softmax_bwd(const float* P, float* dP, long long rows, int T, float scale, hipStream_t s) {
  const int wpb = 4;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + wpb - 1) / wpb)), dim3(wpb * 64), 0, s, P, dP, rows, T, scale);
  return hipGetLastError();
}

// ---- out[c] = sum_m in[m][c] (double accumulate, fixed order); one thread per column, 64 rows-lanes reduced through LDS ----
__global__ void colsum_kernel(const float* in, int ld, long long M, int C, float* out) {
  __shared__ double sm[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;     // 64 columns x 4 row lanes
  const int c = blockIdx.x * 64 + cl;
  double a = 0.0;
  if (c < C)
    for (long long m = rl; m < M; m += 4) a += (double)in[m * ld + c];
  sm[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < C) out[c] = (float)((sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]));
}

hipError_t launch_colsum(const float* in, int ld, long long M, int C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, s, in, ld, M, C, out);
  return hipGetLastError();
}

// ---- a = act(x*scale + shift) materialised (the forward fuses it into the consuming conv's staging pass) ----
__global__ void act_apply_kernel(const float* x, const float* scale, const float* shift, int silu, float* out, int C, int HW,
                                 long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long n = i / ((long long)C * HW);
    float y = x[i] * scale[n * C + c] + shift[n * C + c];
    if (silu) y = y * sigmoid_f(y);
    out[i] = y;
  }
}

hipError_t launch_act_apply(const float* x, const float* scale, const float* shift, int silu, float* out, int N, int HW, int C,
                            hipStream_t s) {
  const long long total = (long long)N * HW * C;
  hipLaunchKernelGGL(act_apply_kernel, dim3(bw_blocks(total)), dim3(256), 0, s, x, scale, shift, silu, out, C, HW, total);
  return hipGetLastError();
}

// ---- GroupNorm parameter gradients from the backward partials: dgamma_c = sum_n rstd*(P2 - mean*P1), dbeta_c = sum_n P1 ----
__global__ void gn_param_grad_kernel(const double* partial, int nblk, const float* mr, int N, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int g = c / (C / 32);
  double dg = 0.0, db = 0.0;
  for (int n = 0; n < N; ++n) {
    double p1 = 0.0, p2 = 0.0;
    for (int k = 0; k < nblk; ++k) {
      const double* q = partial + (((size_t)n * nblk + k) * C + c) * 2;
      p1 += q[0];
      p2 += q[1];
    }
    const double mean = (double)mr[((size_t)n * 32 + g) * 2], rstd = (double)mr[((size_t)n * 32 + g) * 2 + 1];
    dg += rstd * (p2 - mean * p1);
    db += p1;
  }
  dgamma[c] = (float)dg;
  dbeta[c] = (float)db;
}

hipError_t launch_gn_param_grad(const double* partial, int nblk, const float* mr, int N, int C, float* dgamma, float* dbeta,
                                hipStream_t s) {
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3((C + 127) / 128), dim3(128), 0, s, partial, nblk, mr, N, C, dgamma, dbeta);
  return hipGetLastError();
}

// out = a * s  (elementwise)
__global__ void scale_kernel(const float* a, float sc, float* out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = a[i] * sc;
}
hipError_t launch_scale(const float* a, float sc, float* out, long long n, hipStream_t s) {
  hipLaunchKernelGGL(scale_kernel, dim3(bw_blocks(n)), dim3(256), 0, s, a, sc, out, n);
  return hipGetLastError();
}

}  // namespace asyrp
