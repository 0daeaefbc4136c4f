// zuko_amd — backward (vector-Jacobian product) kernels of the univariate transforms, the base
// density and the activations: SURVEY 8(f) rank 1.  The reference has no backward code of its own —
// `loss.backward()` runs PyTorch autograd through every ATen op of the forward
// (zuko/transforms.py:480-490, 554-567 for the spline; :436-446 affine) — so these kernels are the
// fused adjoints of those op sequences.
//
// Spline: given gy[N, D] and gl (the gradient w.r.t. log|det J|, per row when the forward reduced over
// features), produce gx[N, D] and g_phi[N, D, 3K-1] for the PACKED parameter layout.  Per element the
// forward is recomputed from phi (same device functions as the forward kernels), the local map
//   (x, x0, x1, y0, y1, d0, d1) -> (y, ladj)
// is differentiated with 7-wide forward-mode dual numbers (no hand-derived formulas to get wrong), and
// the adjoint is pushed back through bin gather -> cumsum -> softmax -> softclip / exp by hand (those
// are sparse: only the two knots of the active bin receive gradient).
// Data movement mirrors the forward kernel: a wave owns 64 consecutive elements, phi comes in and
// g_phi goes out through a wave-private LDS image with 16-byte global accesses.
#include "zk_univariate_bwd.h"

namespace zk {

struct BwdArgs {
  int64_t N, D;
  const float* x;
  const float* phi;   // [N, D, total] packed
  const float* gy;    // [N, D]
  const float* gl;    // [N] (reduced) or [N, D]
  int gl_reduced;
  float* gx;          // [N, D]
  float* gphi;        // [N, D, total]
  float bound, ls;
  int64_t iters;
};

extern __shared__ __attribute__((aligned(16))) float bw_lds[];

// KIND 0: affine (total 2), 1: RQS with K bins.  Requires 16-byte aligned phi / gphi and D*total*... any D.
template <int KIND, int K> __global__ __launch_bounds__(256) void uni_backward_kernel(BwdArgs a) {
  constexpr int TOTAL = KIND == 0 ? 2 : 3 * K - 1;
  constexpr int PER_WAVE = 64 * TOTAL + 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wsh = bw_lds + wave * PER_WAVE;
  const int64_t total_elems = a.N * a.D;
  const int64_t wt0 = ((int64_t)blockIdx.x * 4 + wave) * a.iters;
  for (int64_t it = 0; it < a.iters; ++it) {
    const int64_t e0 = (wt0 + it) * 64;
    if (e0 >= total_elems) break;  // wave-uniform
    const int64_t cnt = (total_elems - e0) < 64 ? (total_elems - e0) : 64;
    const float* src = a.phi + e0 * TOTAL;
    const int shift = (int)(((uintptr_t)src / 4) % 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // stage phi (alignment-preserving image, as the forward kernel)
    {
      const int64_t n = cnt * TOTAL;
      int head = shift ? 4 - shift : 0;
      if (head > n) head = (int)n;
      if (lane < head) wsh[shift + lane] = src[lane];
      const int64_t body = (n - head) / 4;
      const float4* vs = reinterpret_cast<const float4*>(src + head);
      float4* vd = reinterpret_cast<float4*>(wsh + shift + head);
      for (int64_t i = lane; i < body; i += 64) vd[i] = vs[i];
      const int64_t done = head + body * 4;
      if (lane < n - done) wsh[shift + done + lane] = src[done + lane];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const bool valid = lane < cnt;
    float g[TOTAL];
    float gxv = 0.f;
    if (valid) {
      const int64_t e = e0 + lane;
      float p[TOTAL];
#pragma unroll
      for (int i = 0; i < TOTAL; ++i) p[i] = wsh[shift + lane * TOTAL + i];
      const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
      const float gyv = a.gy ? a.gy[e] : 0.f;
      if (KIND == 0) affine_backward_element(p, a.x[e], gyv, glv, a.ls, gxv, g);
      else rqs_backward_element<K>(p, a.x[e], gyv, glv, a.bound, a.ls, gxv, g);
      a.gx[e] = gxv;
    }
    // g_phi out through the same LDS image (same shift: gphi has the alignment of phi by contract)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (valid) {
#pragma unroll
      for (int i = 0; i < TOTAL; ++i) wsh[shift + lane * TOTAL + i] = g[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
      float* dst = a.gphi + e0 * TOTAL;
      const int64_t n = cnt * TOTAL;
      int head = shift ? 4 - shift : 0;
      if (head > n) head = (int)n;
      if (lane < head) dst[lane] = wsh[shift + lane];
      const int64_t body = (n - head) / 4;
      float4* vd = reinterpret_cast<float4*>(dst + head);
      const float4* vs = reinterpret_cast<const float4*>(wsh + shift + head);
      for (int64_t i = lane; i < body; i += 64) vd[i] = vs[i];
      const int64_t done = head + body * 4;
      if (lane < n - done) dst[done + lane] = wsh[shift + done + lane];
    }
  }
}

// Any bin count up to 64 (zuko/transforms.py:469-477 accepts any `bins`): one lane per element, knots and softmax probabilities
// in run-time indexed local arrays, parameters read from / gradients written to global memory directly.  A slow path next to the
// K = 4 / 8 / 16 instantiations above, with the same arithmetic (max-subtracted softmax, strict-count bin search, the same
// local vector-Jacobian product).
#define ZK_BWD_MAXK 64
__global__ __launch_bounds__(256) void rqs_backward_generic_kernel(BwdArgs a, int K) {
  const int total = 3 * K - 1;
  const int64_t total_elems = a.N * a.D;
  const float bound = a.bound, ls = a.ls;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_elems; e += (int64_t)gridDim.x * 256) {
    const float* p = a.phi + e * total;
    float* g = a.gphi + e * total;
    float kn[2][ZK_BWD_MAXK + 1], pr[2][ZK_BWD_MAXK], kd[ZK_BWD_MAXK + 1];
    for (int ax = 0; ax < 2; ++ax) {
      float m = softclip2<float>(p[ax * K], ls);
      for (int j = 1; j < K; ++j) m = fmaxf(m, softclip2<float>(p[ax * K + j], ls));
      float ssum = 0.f;
      for (int j = 0; j < K; ++j) { pr[ax][j] = expf(softclip2<float>(p[ax * K + j], ls) - m); ssum += pr[ax][j]; }
      float cum = 0.f;
      kn[ax][0] = -bound;
      for (int j = 0; j < K; ++j) { pr[ax][j] /= ssum; cum += pr[ax][j]; kn[ax][j + 1] = bound * (2.f * cum - 1.f); }
    }
    kd[0] = kd[K] = 1.f;
    for (int j = 1; j < K; ++j) kd[j] = expf(softclip<float>(p[2 * K + j - 1], ls));
    const float x = a.x[e];
    const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
    const float gyv = a.gy ? a.gy[e] : 0.f;
    int cnt = 0;
    for (int j = 0; j <= K; ++j) cnt += (kn[0][j] < x) ? 1 : 0;
    const int k = cnt - 1;
    for (int i = 0; i < total; ++i) g[i] = 0.f;
    if (!(k >= 0 && k < K)) { a.gx[e] = gyv; continue; }  // identity outside [-B, B] (and for NaN: no knot compares below it)
    float gv[7];
    rqs_local_vjp(x, kn[0][k], kn[0][k + 1], kn[1][k], kn[1][k + 1], kd[k], kd[k + 1], gyv, glv, gv);
    a.gx[e] = gv[0];
    const float twoB = 2.f * bound;
    float dotw = 0.f, doth = 0.f;
    for (int i = 0; i < K; ++i) {
      dotw += pr[0][i] * twoB * ((i < k ? gv[1] : 0.f) + (i <= k ? gv[2] : 0.f));
      doth += pr[1][i] * twoB * ((i < k ? gv[3] : 0.f) + (i <= k ? gv[4] : 0.f));
    }
    const float cw = fabsf(ls) * 0.5f, cd = fabsf(ls);
    for (int i = 0; i < K; ++i) {
      const float gw = twoB * ((i < k ? gv[1] : 0.f) + (i <= k ? gv[2] : 0.f));
      const float gh = twoB * ((i < k ? gv[3] : 0.f) + (i <= k ? gv[4] : 0.f));
      g[i] = pr[0][i] * (gw - dotw) * softclip_grad(p[i], cw);
      g[K + i] = pr[1][i] * (gh - doth) * softclip_grad(p[K + i], cw);
    }
    if (k >= 1) g[2 * K + k - 1] = gv[5] * kd[k] * softclip_grad(p[2 * K + k - 1], cd);
    if (k + 1 <= K - 1) g[2 * K + k] = gv[6] * kd[k + 1] * softclip_grad(p[2 * K + k], cd);
  }
}

// gz[n, d] = -g[n] (z - loc) / scale^2 ; the ladj gradient is g itself (done by the caller)
__global__ __launch_bounds__(256) void normal_backward_kernel(int64_t N, int64_t D, const float* z, const float* loc, const float* scale, const float* g, float* gz) {
  const int64_t total = N * D;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t d = e % D;
    const float s = scale[d];
    gz[e] = -g[e / D] * (z[e] - loc[d]) / (s * s);
  }
}

// gin = gout * act'(.) expressed through the activation's OUTPUT y (relu, elu, tanh, sigmoid, leaky relu)
__global__ __launch_bounds__(256) void act_backward_kernel(int64_t n, const float* y, const float* gout, int act, float* gin) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = y[e], g = gout[e];
    float d = 1.f;
    switch (act) {
      case 1: d = v > 0.f ? 1.f : 0.f; break;
      case 2: d = v > 0.f ? 1.f : v + 1.f; break;
      case 3: d = 1.f - v * v; break;
      case 6: d = v * (1.f - v); break;
      case 7: d = v > 0.f ? 1.f : 0.01f; break;
      default: break;
    }
    gin[e] = g * d;
  }
}

// seeds of the inverse map's adjoint: gy = gx * exp(-ladj) (= gx / f'(x)), seed = -gy
__global__ __launch_bounds__(256) void inverse_seed_kernel(int64_t n, const float* gx, const float* ladj, float* gy, float* seed) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = gx[e] * expf(-ladj[e]);
    gy[e] = v;
    seed[e] = -v;
  }
}

}  // namespace zk

using namespace zk;

extern "C" {

// kind 0 = affine (phi [N, D, 2] = [shift, scale]), 1 = RQS (phi [N, D, 3K-1]; K in {4, 8, 16} LDS-staged, any other K <= 64 generic); fp32, packed phi/gphi.
int zk_univariate_backward(int kind, int64_t N, int64_t D, int K, double bound, double slope, const void* x, const void* phi, const void* gy, const void* gl,
                           int gl_reduced, void* gx, void* gphi, void* stream) {
  if (N <= 0 || D <= 0) return 0;
  BwdArgs a{};
  a.N = N; a.D = D; a.x = (const float*)x; a.phi = (const float*)phi; a.gy = (const float*)gy; a.gl = (const float*)gl; a.gl_reduced = gl_reduced;
  a.gx = (float*)gx; a.gphi = (float*)gphi; a.bound = (float)bound; a.ls = (float)log(slope);
  if ((((uintptr_t)phi ^ (uintptr_t)gphi) % 16) != 0) return ZK_EINVAL;
  const int total = kind == 0 ? 2 : 3 * K - 1;
  const int64_t tiles = (N * D + 63) / 64;
  const int64_t blocks = (tiles + 3) / 4;
  const size_t lds = 4 * (size_t)(64 * total + 8) * sizeof(float);
  int64_t per_cu = (160 * 1024) / (int64_t)lds;
  per_cu = per_cu > 8 ? 8 : per_cu;
  const int64_t cap = 256 * per_cu;
  const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
  a.iters = (tiles + (int64_t)grid * 4 - 1) / ((int64_t)grid * 4);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL((uni_backward_kernel<0, 1>), dim3(grid), dim3(256), lds, st, a);
  else if (kind == 1 && K == 4) hipLaunchKernelGGL((uni_backward_kernel<1, 4>), dim3(grid), dim3(256), lds, st, a);
  else if (kind == 1 && K == 8) hipLaunchKernelGGL((uni_backward_kernel<1, 8>), dim3(grid), dim3(256), lds, st, a);
  else if (kind == 1 && K == 16) hipLaunchKernelGGL((uni_backward_kernel<1, 16>), dim3(grid), dim3(256), lds, st, a);
  else if (kind == 1 && K >= 1 && K <= ZK_BWD_MAXK) {
    const int64_t nb = (N * D + 255) / 256;
    hipLaunchKernelGGL(rqs_backward_generic_kernel, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, st, a, K);
  } else return ZK_EINVAL;
  return ZK_LAUNCH_CHECK();
}

int zk_diag_normal_backward(int64_t N, int64_t D, const void* z, const void* loc, const void* scale, const void* g, void* gz, void* stream) {
  if (N <= 0 || D <= 0) return 0;
  const int64_t nb = (N * D + 255) / 256;
  hipLaunchKernelGGL(normal_backward_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, N, D, (const float*)z, (const float*)loc,
                     (const float*)scale, (const float*)g, (float*)gz);
  return ZK_LAUNCH_CHECK();
}

int zk_inverse_seed(int64_t n, const void* gx, const void* ladj, void* gy, void* seed, void* stream) {
  if (n <= 0) return 0;
  const int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(inverse_seed_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, n, (const float*)gx, (const float*)ladj, (float*)gy, (float*)seed);
  return ZK_LAUNCH_CHECK();
}

int zk_act_backward(int64_t n, const void* y, const void* gout, int act, void* gin, void* stream) {
  if (n <= 0) return 0;
  if (!(act == 0 || act == 1 || act == 2 || act == 3 || act == 6 || act == 7)) return ZK_EINVAL;
  const int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(act_backward_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, n, (const float*)y, (const float*)gout, act, (float*)gin);
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
