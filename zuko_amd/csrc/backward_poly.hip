// zuko_amd — backward (vector-Jacobian product) of the polynomial univariate maps: SOS polynomial
// (zuko/transforms.py:927-963) and (bounded) Bernstein polynomial (:640-831).
//
// The reference gets these gradients from autograd through its own Python: GaussLegendre.backward
// (zuko/utils.py:297-326: d area / d x = integrand at x, parameters through the quadrature), plain autograd for
// log g(x), and autograd through the softmax / softplus / cumsum constraints and the Beta-pdf basis of the Bernstein
// polynomial.  Here ONE kernel per map evaluates the very same device functions the forward kernels use
// (zk_univariate.h: sos_f / sos_g, bern_theta_* / bern_fwd) on forward-mode dual numbers that carry d/dx and
// d/d(every unconstrained parameter) — no hand-derived adjoints to get wrong — and contracts them with (gy, gl):
//     gx = gy dy/dx + gl dladj/dx,     gparam_i = gy dy/dp_i + gl dladj/dp_i.
// These maps are not on the benchmark path (SOSPF / BPF): one thread per element, parameters read from global memory.
#include "zk_univariate.h"

namespace zk {

template <int NV> struct DualN {
  float v;
  float d[NV];
  __device__ __forceinline__ DualN() {}
  __device__ __forceinline__ DualN(float c) : v(c) {
#pragma unroll
    for (int i = 0; i < NV; ++i) d[i] = 0.f;
  }
  __device__ __forceinline__ DualN(double c) : DualN((float)c) {}
  __device__ __forceinline__ DualN(int c) : DualN((float)c) {}
  static __device__ __forceinline__ DualN var(float c, int i) { DualN r(c); r.d[i] = 1.f; return r; }
};
#define ZK_DUAL_BIN(OP, VAL, DER)                                                                          \
  template <int NV> __device__ __forceinline__ DualN<NV> operator OP(const DualN<NV>& a, const DualN<NV>& b) { \
    DualN<NV> r; r.v = VAL;                                                                                \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) r.d[i] = DER;                                           \
    return r;                                                                                              \
  }
ZK_DUAL_BIN(+, a.v + b.v, a.d[i] + b.d[i])
ZK_DUAL_BIN(-, a.v - b.v, a.d[i] - b.d[i])
ZK_DUAL_BIN(*, a.v * b.v, a.d[i] * b.v + a.v * b.d[i])
template <int NV> __device__ __forceinline__ DualN<NV> operator/(const DualN<NV>& a, const DualN<NV>& b) {
  DualN<NV> r; const float ib = 1.f / b.v; r.v = a.v * ib;
#pragma unroll
  for (int i = 0; i < NV; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
template <int NV> __device__ __forceinline__ DualN<NV> operator-(const DualN<NV>& a) { return DualN<NV>(0.f) - a; }
template <int NV> __device__ __forceinline__ DualN<NV>& operator+=(DualN<NV>& a, const DualN<NV>& b) { a = a + b; return a; }
template <int NV> __device__ __forceinline__ DualN<NV>& operator*=(DualN<NV>& a, const DualN<NV>& b) { a = a * b; return a; }
template <int NV> __device__ __forceinline__ bool operator<(const DualN<NV>& a, const DualN<NV>& b) { return a.v < b.v; }
template <int NV> __device__ __forceinline__ bool operator>(const DualN<NV>& a, const DualN<NV>& b) { return a.v > b.v; }
template <int NV> __device__ __forceinline__ bool operator<=(const DualN<NV>& a, const DualN<NV>& b) { return a.v <= b.v; }
template <int NV> __device__ __forceinline__ bool operator>=(const DualN<NV>& a, const DualN<NV>& b) { return a.v >= b.v; }

template <int NV> __device__ __forceinline__ DualN<NV> dual_scale(const DualN<NV>& a, float value, float deriv) {  // f(a) with f(a.v) = value, f'(a.v) = deriv
  DualN<NV> r; r.v = value;
#pragma unroll
  for (int i = 0; i < NV; ++i) r.d[i] = a.d[i] * deriv;
  return r;
}
#define ZK_DUAL_MATH(NV)                                                                                                              \
  template <> __device__ __forceinline__ DualN<NV> t_exp<DualN<NV>>(DualN<NV> a) { const float e = expf(a.v); return dual_scale(a, e, e); } \
  template <> __device__ __forceinline__ DualN<NV> t_log<DualN<NV>>(DualN<NV> a) { return dual_scale(a, logf(a.v), 1.f / a.v); }            \
  template <> __device__ __forceinline__ DualN<NV> t_log1p<DualN<NV>>(DualN<NV> a) { return dual_scale(a, log1pf(a.v), 1.f / (1.f + a.v)); }
ZK_DUAL_MATH(17)
ZK_DUAL_MATH(18)

struct PolyBwdArgs {
  int64_t N, D;
  const float* x;
  const float* p;      // [N, D, total] contiguous unconstrained parameters
  const float* gy;     // [N, D] or null
  const float* gl;     // [N] (reduced) / [N, D] or null
  int gl_reduced;
  float* gx;           // [N, D]
  float* gp;           // [N, D, total]
  int total;
  float bound, slope, eps;
  int P, L1, has_const, bounded;
  double node[ZK_SOS_MAX_NODES], weight[ZK_SOS_MAX_NODES];
};

// SOS: variables 0 = x, 1 .. P*L1 = a, (P*L1 + 1 = constant, whose derivative is gy itself)
template <int NV> __global__ __launch_bounds__(128) void sos_backward_kernel(PolyBwdArgs a) {
  typedef DualN<NV> T;
  const int64_t total_e = a.N * a.D;
  const int na = a.P * a.L1;
  for (int64_t e = (int64_t)blockIdx.x * 128 + threadIdx.x; e < total_e; e += (int64_t)gridDim.x * 128) {
    const float* pe = a.p + e * a.total;
    SosConst<T> c;
    c.bound = T(10.0f); c.slope = T(a.slope); c.P = a.P; c.L1 = a.L1;
    for (int i = 0; i < a.L1; ++i) { c.node[i] = T((float)a.node[i]); c.weight[i] = T((float)a.weight[i]); }
    auto ld = [&](int j) { return T::var(pe[j], 1 + j); };
    const T X = T::var(a.x[e], 0);
    const T y = sos_f<T>(c, ld, X);
    const T l = t_log<T>(sos_g<T>(c, ld, X));
    const float gyv = a.gy ? a.gy[e] : 0.f;
    const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
    a.gx[e] = gyv * y.d[0] + glv * l.d[0];
    float* ge = a.gp + e * a.total;
    for (int j = 0; j < na; ++j) ge[j] = gyv * y.d[1 + j] + glv * l.d[1 + j];
    if (a.has_const) ge[na] = gyv;  // y = f(x) + constant (zuko/flows/polynomial.py:23-29)
  }
}

// Bernstein: variables 0 = x, 1 .. M = theta (unconstrained); NC = constrained coefficients
template <int NC, int M, bool BOUNDED> __global__ __launch_bounds__(64) void bern_backward_kernel(PolyBwdArgs a) {
  typedef DualN<M + 1> T;
  const int64_t total_e = a.N * a.D;
  for (int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x; e < total_e; e += (int64_t)gridDim.x * 64) {
    const float* pe = a.p + e * M;
    auto ld = [&](int j) { return T::var(pe[j], 1 + j); };
    T th[NC];
    const T bound = T(a.bound);
    if (BOUNDED) bern_theta_bounded<T, NC>(ld, bound, th);
    else bern_theta_unbounded<T, NC>(ld, th);
    const T eps = T(a.eps);
    const BernTails<T> tails = bern_tails<T, NC>(th, BOUNDED, bound, eps);
    T y, dydx;
    bern_fwd<T, NC>(th, tails, bound, T::var(a.x[e], 0), y, dydx, eps);
    const T l = t_log<T>(dydx);
    const float gyv = a.gy ? a.gy[e] : 0.f;
    const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
    a.gx[e] = gyv * y.d[0] + glv * l.d[0];
    float* ge = a.gp + e * M;
#pragma unroll
    for (int j = 0; j < M; ++j) ge[j] = gyv * y.d[1 + j] + glv * l.d[1 + j];
  }
}

}  // namespace zk

using namespace zk;

extern "C" {

// SOS polynomial adjoint (fp32): p = [a (P x L1) | constant?] packed per element, P * L1 == 15 (the SOSPF default: 3 polynomials of degree 4).
int zk_sos_backward(int64_t N, int64_t D, int P, int L1, double slope, const double* gl_nodes01, const double* gl_weights01, int has_const, const void* x,
                    const void* params, const void* gy, const void* gl, int gl_reduced, void* gx, void* gparams, void* stream) {
  if (N <= 0 || D <= 0) return 0;
  if (P * L1 != 15 || L1 > ZK_SOS_MAX_NODES) return ZK_EINVAL;
  PolyBwdArgs a{};
  a.N = N; a.D = D; a.x = (const float*)x; a.p = (const float*)params; a.gy = (const float*)gy; a.gl = (const float*)gl; a.gl_reduced = gl_reduced;
  a.gx = (float*)gx; a.gp = (float*)gparams; a.total = P * L1 + (has_const ? 1 : 0); a.slope = (float)slope; a.P = P; a.L1 = L1; a.has_const = has_const;
  for (int i = 0; i < L1; ++i) { a.node[i] = gl_nodes01[i]; a.weight[i] = gl_weights01[i]; }
  const int64_t nb = (N * D + 127) / 128;
  hipLaunchKernelGGL((sos_backward_kernel<17>), dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(128), 0, (hipStream_t)stream, a);
  return ZK_LAUNCH_CHECK();
}

// Bernstein adjoint (fp32): theta [N, D, M] unconstrained; built for the BPF defaults — bounded with M = 17 (22 coefficients)
// and unbounded with M = 16 (18 coefficients).
int zk_bernstein_backward(int64_t N, int64_t D, int M, int bounded, double bound, double eps, const void* x, const void* theta, const void* gy, const void* gl,
                          int gl_reduced, void* gx, void* gtheta, void* stream) {
  if (N <= 0 || D <= 0) return 0;
  if (!(eps > 0.0 && eps < 0.5)) return ZK_EINVAL;
  PolyBwdArgs a{};
  a.N = N; a.D = D; a.x = (const float*)x; a.p = (const float*)theta; a.gy = (const float*)gy; a.gl = (const float*)gl; a.gl_reduced = gl_reduced;
  a.gx = (float*)gx; a.gp = (float*)gtheta; a.total = M; a.bound = (float)bound; a.bounded = bounded; a.eps = (float)eps;
  const int64_t nb = (N * D + 63) / 64;
  const unsigned grid = (unsigned)(nb > 16384 ? 16384 : nb);
  if (bounded && M == 17) hipLaunchKernelGGL((bern_backward_kernel<22, 17, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
  else if (!bounded && M == 16) hipLaunchKernelGGL((bern_backward_kernel<18, 16, false>), dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
  else return ZK_EINVAL;
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
