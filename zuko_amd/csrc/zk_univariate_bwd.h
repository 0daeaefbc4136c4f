// zuko_amd — per-element adjoints (vector-Jacobian products) of the univariate maps: shared by the stand-alone backward kernels
// (backward.hip) and the fused backward of an autoregressive transform (fused_ar_split_impl.h: arxb_kernel).  The reference has no
// backward code of its own: PyTorch autograd runs through every ATen op of zuko/transforms.py:480-490, 554-567 (spline) and :436-446
// (affine); these are the fused adjoints of those op sequences.
//
// Spline: the forward is recomputed from phi (same device functions as the forward kernels), the local map
//   (x, x0, x1, y0, y1, d0, d1) -> (y, ladj)
// is differentiated in reverse by hand (rqs_local_vjp), and the adjoint is pushed back through bin gather -> cumsum -> softmax ->
// softclip / exp (those are sparse: only the two knots of the active bin receive gradient).
#pragma once
#include "zk_univariate.h"

namespace zk {

// (reciprocals, exponentials and logarithms of the adjoint use the hardware approximations — v_rcp / v_exp / v_log, ~1 ulp — like the
//  forward's rqs_lean: with IEEE division and ocml expf the kernel was VALU-bound at 2.6 TB/s; gradients are compared at 2e-4)
__device__ __forceinline__ float softclip_grad(float v, float c_abs) {  // d/dv [ v / (1 + |v| / c) ]
  const float t = 1.f + fabsf(v) * __builtin_amdgcn_rcpf(c_abs);
  return __builtin_amdgcn_rcpf(t * t);
}

// gv[0..6] = gy * dy/d(.) + gl * dladj/d(.) for (.) = x, x0, x1, y0, y1, d0, d1 of the active bin (zuko/transforms.py:554-567): the
// reverse sweep of
//   w = x1 - x0, h = y1 - y0, s = h / w, z = (x - x0) / w, zz = z (1 - z), q = d0 + d1 - 2 s, den = s + q zz, num = s z^2 + d0 zz,
//   y = y0 + h num / den,   ladj = 2 log s + log P - 2 log den  with  P = 2 s zz + d0 (1 - z)^2 + d1 z^2
// written out by hand (checked against autograd in float64; until round 4 this was 7-wide forward-mode dual arithmetic, five times the
// instructions — the adjoint is what bounds the fused backward kernel, csrc/fused_ar_split_impl.h: arxb_kernel).
__device__ __forceinline__ void rqs_local_vjp(float x, float x0, float x1, float y0, float y1, float d0, float d1, float gy, float gl, float (&gv)[7]) {
  const float w = x1 - x0, h = y1 - y0;
  const float iw = __builtin_amdgcn_rcpf(w);
  const float s = h * iw, z = (x - x0) * iw, omz = 1.f - z, zz = z * omz;
  const float q = d0 + d1 - 2.f * s;
  const float den = s + q * zz, num = s * z * z + d0 * zz;
  const float iden = __builtin_amdgcn_rcpf(den);
  const float r = num * iden;
  const float P = 2.f * s * zz + d0 * omz * omz + d1 * z * z;
  const float Pb = gl * __builtin_amdgcn_rcpf(P);
  const float rb = gy * h;
  const float numb = rb * iden;
  const float denb = -(2.f * gl + rb * r) * iden;
  float sb = 2.f * gl * __builtin_amdgcn_rcpf(s) + Pb * 2.f * zz + denb * (1.f - 2.f * zz) + numb * z * z;
  const float d0b = Pb * omz * omz + (denb + numb) * zz;
  const float d1b = Pb * z * z + denb * zz;
  const float zzb = Pb * 2.f * s + denb * q + numb * d0;
  const float omzb = Pb * 2.f * d0 * omz + zzb * z;
  const float zb = Pb * 2.f * d1 * z + numb * 2.f * s * z + zzb * omz - omzb;
  const float xb = zb * iw;
  const float hb = gy * r + sb * iw;
  const float wb = -(zb * z + sb * s) * iw;
  gv[0] = xb;
  gv[1] = -xb - wb;
  gv[2] = wb;
  gv[3] = gy - hb;
  gv[4] = hb;
  gv[5] = d0b;
  gv[6] = d1b;
}

template <int K> __device__ __forceinline__ void rqs_backward_element(const float* p, float x, float gyv, float glv, float bound, float ls, float& gxv, float* g) {
  typedef MathFast M;
  constexpr int TOTAL = 3 * K - 1;
  float kx[K + 1], ky[K + 1], kd[K + 1], pw[K], ph[K];
  // forward recompute, keeping the softmax probabilities
  // u = 1 / (1 + |p| / c) is both the soft clip (p u) and the square root of its derivative (u^2): one reciprocal per parameter
  float sg[TOTAL];
  const float r2ls = 2.f * __builtin_amdgcn_rcpf(ls), rls = __builtin_amdgcn_rcpf(ls);
  auto axis = [&](int off, float (&knot)[K + 1], float (&prob)[K]) {
    float v[K], m;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float u = __builtin_amdgcn_rcpf(1.f + fabsf(p[off + j] * r2ls));
      sg[off + j] = u * u;
      v[j] = p[off + j] * u;
      m = (j == 0) ? v[0] : fmaxf(m, v[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { v[j] = __expf(v[j] - m); s += v[j]; }
    const float r = __builtin_amdgcn_rcpf(s);
    float cum = 0.f;
    knot[0] = -bound;
#pragma unroll
    for (int j = 0; j < K; ++j) { prob[j] = v[j] * r; cum += prob[j]; knot[j + 1] = bound * (2.f * cum - 1.f); }
  };
  axis(0, kx, pw);
  axis(K, ky, ph);
  kd[0] = 1.f;
  kd[K] = 1.f;
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const float u = __builtin_amdgcn_rcpf(1.f + fabsf(p[2 * K + j - 1] * rls));
    sg[2 * K + j - 1] = u * u;
    kd[j] = __expf(p[2 * K + j - 1] * u);
  }
  bool inside;
  float x0, x1, y0, y1, d0, d1;
  const int k = rqs_locate<float, K>(kx, kx, ky, kd, x, inside, x0, x1, y0, y1, d0, d1);
#pragma unroll
  for (int i = 0; i < TOTAL; ++i) g[i] = 0.f;
  if (!inside) { gxv = gyv; return; }  // identity outside [-B, B]: y = x, ladj = 0, no parameter gradient
  float gv[7];
  rqs_local_vjp(x, x0, x1, y0, y1, d0, d1, gyv, glv, gv);
  gxv = gv[0];
  // knots -> softmax probabilities: kx_j = B (2 sum_{i<j} p_i - 1): only knots k and k+1 carry gradient
  const float twoB = 2.f * bound;
  float gpw[K], gph[K], dotw = 0.f, doth = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    gpw[i] = twoB * ((i < k ? gv[1] : 0.f) + (i <= k ? gv[2] : 0.f));
    gph[i] = twoB * ((i < k ? gv[3] : 0.f) + (i <= k ? gv[4] : 0.f));
    dotw += pw[i] * gpw[i];
    doth += ph[i] * gph[i];
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    g[i] = pw[i] * (gpw[i] - dotw) * sg[i];
    g[K + i] = ph[i] * (gph[i] - doth) * sg[K + i];
  }
  // slopes: kd_j = exp(softclip(ud_{j-1})), j = 1..K-1 (ends are the constant 1)
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const float gk = (j == k ? gv[5] : 0.f) + (j == k + 1 ? gv[6] : 0.f);
    g[2 * K + j - 1] = gk * kd[j] * sg[2 * K + j - 1];
  }
}

__device__ __forceinline__ void affine_backward_element(const float* p, float x, float gyv, float glv, float ls, float& gxv, float* g) {
  const float u = __builtin_amdgcn_rcpf(1.f + fabsf(p[1] * __builtin_amdgcn_rcpf(ls)));  // soft clip p u, its derivative u^2
  const float e = __expf(p[1] * u);
  gxv = gyv * e;
  g[0] = gyv;                                  // shift
  g[1] = (gyv * x * e + glv) * (u * u);        // unconstrained log-scale
}

}  // namespace zk
