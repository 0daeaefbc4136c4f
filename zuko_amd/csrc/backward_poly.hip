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
#include "zk_dual.h"
#include <stdlib.h>

namespace zk {

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

// ---- the SOS adjoint written out (round 6) ---------------------------------------------------------------------------------------------
// The dual-number kernel carries 17 derivative components through the quadrature: 7 ms per SOSPF layer at 2^16 x 64 elements, 80 % of a SOSPF training
// step.  With u = x / B, q_p(u) = 1 + sum_j a_pj u^j, g = mean_p q_p^2 + slope, f(x) = x sum_i w_i g(x_i) (x_i = the quadrature points, zuko/utils.py:349-363):
//     d y / d a_pj = x sum_i w_i (2 / P) q_p(u_i) u_i^j        d y / dx = g(x)   (GaussLegendre.backward, utils.py:297-326: the integrand at x)
//     d ladj / d a_pj = (2 / P) q_p(u) u^j / g(x)              d ladj / dx = (2 / (P B)) sum_p q_p(u) q_p'(u) / g(x)
template <int P, int L1> __global__ __launch_bounds__(128) void sos_adjoint_kernel(PolyBwdArgs a) {
  const int64_t total_e = a.N * a.D;
  const float B = 10.0f;  // (MonotonicTransform's bound, as the dual-number kernel and zk_sos_forward's default)
  for (int64_t e = (int64_t)blockIdx.x * 128 + threadIdx.x; e < total_e; e += (int64_t)gridDim.x * 128) {
    const float* pe = a.p + e * a.total;
    float c[P][L1];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < L1; ++j) c[p][j] = pe[p * L1 + j];
    const float x = a.x[e];
    const float gyv = a.gy ? a.gy[e] : 0.f;
    const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
    const float u = x / B;
    float q[P], gsum = 0.f, gq = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float pw = 1.f, dot = 0.f, ddot = 0.f;
#pragma unroll
      for (int j = 0; j < L1; ++j) {
        dot += c[p][j] * pw;
        if (j + 1 < L1) ddot += (float)(j + 1) * c[p][j + 1] * pw;
        pw *= u;
      }
      q[p] = 1.f + dot;
      gsum += q[p] * q[p];
      gq += q[p] * ddot;
    }
    const float g = gsum / (float)P + a.slope;
    a.gx[e] = gyv * g + glv * (2.f / ((float)P * B)) * gq / g;
    float acc[P][L1];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < L1; ++j) acc[p][j] = 0.f;
#pragma unroll
    for (int i = 0; i < L1; ++i) {
      const float w = (float)a.node[i];
      const float pt = (w < 0.5f) ? w * x : x - x * (1.f - w);  // torch.lerp(0, x, w), as sos_f
      const float ui = pt / B, wi = (float)a.weight[i];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float pw = 1.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < L1; ++j) { dot += c[p][j] * pw; pw *= ui; }
        const float wq = wi * (1.f + dot);
        pw = 1.f;
#pragma unroll
        for (int j = 0; j < L1; ++j) { acc[p][j] += wq * pw; pw *= ui; }
      }
    }
    float* ge = a.gp + e * a.total;
    const float ky = gyv * x * (2.f / (float)P), kl = glv * (2.f / (float)P) / g;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float pw = 1.f;
#pragma unroll
      for (int j = 0; j < L1; ++j) {
        ge[p * L1 + j] = ky * acc[p][j] + kl * q[p] * pw;
        pw *= u;
      }
    }
    if (a.has_const) ge[P * L1] = gyv;  // y = f(x) + constant (zuko/flows/polynomial.py:23-29)
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
  // (ZUKO_AMD_POLY_ADJOINT=dual, or a layout other than 3 polynomials of degree 4: the forward-mode dual-number kernel the hand adjoint is checked against)
  const char* adj_env = getenv("ZUKO_AMD_POLY_ADJOINT");  // (read per call: the tests switch it)
  const bool dual = adj_env && adj_env[0] == 'd';
  if (!dual && P == 3 && L1 == 5) hipLaunchKernelGGL((sos_adjoint_kernel<3, 5>), dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(128), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((sos_backward_kernel<17>), dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(128), 0, (hipStream_t)stream, a);
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
