// zuko_amd — the Bernstein adjoint kernel as a template (backward_bern.hip instantiates the bounded map, backward_bern_u.hip the unbounded
// one: each instantiation differentiates 17-18 dual components through the whole forward map and takes ~90 s to compile).
#pragma once
#include "zk_dual.h"

namespace zk {

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

// ---- the adjoint written out (round 6) ------------------------------------------------------------------------------------------------
// The dual-number kernel above carries 17-18 derivative components through ~700 operations per element: 38 ms per BPF layer at 2^16 x 64 elements,
// 96 % of a BPF training step.  By hand, with b^n_i(u) the Bernstein basis of degree n and D = theta_{i+1} - theta_i:
//     y = sum_i theta_i b^M_i(u),   dB/du = M sum_i D_i b^{M-1}_i(u),   d2B/du2 = M (M - 1) sum_i (D_{i+1} - D_i) b^{M-2}_i(u),   ladj = log(dB/du / 2B)
//     d y / d theta_i = b^M_i,      d ladj / d theta_i = M (b^{M-1}_{i-1} - b^{M-1}_i) / (dB/du),      d y / dx = dB/du / 2B,   d ladj / dx = d2B/du2 / (dB/du 2B)
// the basis of degree M - 2 by its two-term recurrence from the end nearer to u (ratios u / v or v / u <= 1), the other two by degree elevation
// b^{n+1}_i = v b^n_i + u b^n_{i-1} (convex); in the linear tails (transforms.py:742-760) the bounded map does not depend on theta, the unbounded one through
// its offset B(eps) and slope B'(eps).  theta -> unconstrained parameters: suffix sums (theta is a cumulative sum), then the softmax (bounded, transforms.py:797-818)
// or softplus (unbounded, :703-727) Jacobians.  Checked against the dual-number kernel (ZUKO_AMD_POLY_ADJOINT=dual) and float64 autograd in tests/test_gpu_backward.py.
template <int NC, int M, bool BOUNDED> __global__ __launch_bounds__(128) void bern_adjoint_kernel(PolyBwdArgs a) {
  constexpr int MM = NC - 1;  // degree
  const int64_t total_e = a.N * a.D;
  const float B = a.bound, eps = a.eps;
  for (int64_t e = (int64_t)blockIdx.x * 128 + threadIdx.x; e < total_e; e += (int64_t)gridDim.x * 128) {
    const float* pe = a.p + e * M;
    float t[M];
#pragma unroll
    for (int j = 0; j < M; ++j) t[j] = pe[j];
    float th[NC];
    if (BOUNDED) bern_theta_bounded<float, NC>([&](int j) { return t[j]; }, B, th);
    else bern_theta_unbounded<float, NC>([&](int j) { return t[j]; }, th);
    const float x = a.x[e];
    const float gyv = a.gy ? a.gy[e] : 0.f;
    const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
    const float u = (x + B) / (2.f * B);
    const bool lo = u <= eps, hi = u >= 1.f - eps, tail = lo || hi;
    const float ue = lo ? eps : (hi ? 1.f - eps : u), ve = 1.f - ue;
    // basis of degree MM - 2 at ue
    float b2[MM - 1];
    {
      constexpr int n2 = MM - 2;
      const float ra = ue / ve, rb = ve / ue;
      float ba[n2 + 1], bb[n2 + 1];
      ba[0] = zk_ipow<float>(ve, n2);
      bb[n2] = zk_ipow<float>(ue, n2);
#pragma unroll
      for (int i = 0; i < n2; ++i) {
        ba[i + 1] = ba[i] * (float)((double)(n2 - i) / (double)(i + 1)) * ra;
        bb[n2 - 1 - i] = bb[n2 - i] * (float)((double)(n2 - i) / (double)(i + 1)) * rb;
      }
#pragma unroll
      for (int i = 0; i <= n2; ++i) b2[i] = ue <= 0.5f ? ba[i] : bb[i];
    }
    float b1[MM], b0[NC];
#pragma unroll
    for (int i = 0; i < MM; ++i) b1[i] = (i < MM - 1 ? ve * b2[i] : 0.f) + (i > 0 ? ue * b2[i - 1] : 0.f);
#pragma unroll
    for (int i = 0; i < NC; ++i) b0[i] = (i < MM ? ve * b1[i] : 0.f) + (i > 0 ? ue * b1[i - 1] : 0.f);
    float dval = 0.f, ddval = 0.f;
#pragma unroll
    for (int i = 0; i < MM; ++i) dval += (th[i + 1] - th[i]) * b1[i];
#pragma unroll
    for (int i = 0; i < MM - 1; ++i) ddval += ((th[i + 2] - th[i + 1]) - (th[i + 1] - th[i])) * b2[i];
    dval *= (float)MM;
    ddval *= (float)(MM * (MM - 1));
    // d loss / d theta_i and d loss / dx
    float gth[NC];
    float gx;
    const float inv2B = 1.f / (2.f * B);
    if (!tail) {
      const float c = glv * (float)MM / dval;
#pragma unroll
      for (int i = 0; i < NC; ++i) gth[i] = gyv * b0[i] + c * ((i > 0 ? b1[i - 1] : 0.f) - (i < MM ? b1[i] : 0.f));
      gx = (gyv * dval + glv * ddval / dval) * inv2B;
    } else if (BOUNDED) {
#pragma unroll
      for (int i = 0; i < NC; ++i) gth[i] = 0.f;
      gx = gyv;  // slope 2B / 2B; ladj = 0
    } else {
      const float du = lo ? u - eps : (u - 1.f) + eps;
      const float c = (gyv * du + glv / dval) * (float)MM;
#pragma unroll
      for (int i = 0; i < NC; ++i) gth[i] = gyv * b0[i] + c * ((i > 0 ? b1[i - 1] : 0.f) - (i < MM ? b1[i] : 0.f));
      gx = gyv * dval * inv2B;
    }
    a.gx[e] = gx;
    // theta = cumulative sum of increments: d loss / d increment_m = sum_{k >= m} gth_k
#pragma unroll
    for (int i = NC - 2; i >= 0; --i) gth[i] += gth[i + 1];
    float* ge = a.gp + e * M;
    if (BOUNDED) {  // increments 3 .. M + 2 = softmax(t) * span
      const float edge = (2.f * B) / (float)(M + 4), span = 2.f * B - 4.f * edge;
      float mxv = t[0];
#pragma unroll
      for (int j = 1; j < M; ++j) mxv = t[j] > mxv ? t[j] : mxv;
      float sm[M], ssum = 0.f;
#pragma unroll
      for (int j = 0; j < M; ++j) { sm[j] = expf(t[j] - mxv); ssum += sm[j]; }
      const float r = 1.f / ssum;
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < M; ++j) { sm[j] *= r; dot += sm[j] * gth[3 + j]; }
#pragma unroll
      for (int j = 0; j < M; ++j) ge[j] = span * sm[j] * (gth[3 + j] - dot);
    } else {  // increments: t0 | sp(t1), sp(t1), sp(t2), .., sp(t_{M-1}), sp(t_{M-1})   (NC = M + 2)
      auto sig = [](float v) { return v > 20.f ? 1.f : 1.f / (1.f + expf(-v)); };  // derivative of torch's softplus (threshold 20)
      ge[0] = gth[0];
      ge[1] = sig(t[1]) * (gth[1] + gth[2]);
#pragma unroll
      for (int j = 2; j < M - 1; ++j) ge[j] = sig(t[j]) * gth[j + 1];
      ge[M - 1] = sig(t[M - 1]) * (gth[M] + gth[M + 1]);
    }
  }
}

// host-side launchers of the two instantiations (one per translation unit)
void bern_bwd_launch_bounded(unsigned grid, void* stream, const PolyBwdArgs& a);
void bern_bwd_launch_unbounded(unsigned grid, void* stream, const PolyBwdArgs& a);
void bern_adj_launch(bool bounded, void* stream, const PolyBwdArgs& a);

}  // namespace zk
