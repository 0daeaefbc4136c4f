// zuko_amd — forward-mode dual numbers and the argument block shared by the polynomial adjoints (backward_poly.hip: SOS,
// backward_bern.hip: Bernstein — two translation units so that the two heavy instantiation sets compile side by side).
#pragma once
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

}  // namespace zk
