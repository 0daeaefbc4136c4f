// zuko_amd — per-element device math of the univariate monotone transforms.
//
// Everything here works on ONE (sample, feature) element held in registers: the callers
// (the standalone HBM-streaming kernels in elementwise.hip and the fused conditioner+spline
// kernel in fused_ar.hip) decide where the parameters come from (LDS-staged packed phi,
// strided global memory, or MFMA accumulators).
//
// The fp64 paths (and fp32 affine / SOS / Bernstein) follow the ORDER of the reference expression
// trees so that results track the PyTorch-CPU path to the last few ulps (translation units including
// this header are built with -ffp-contract=off); the fp32 spline has one hardware-shaped
// implementation, rqs_lean, shared by every kernel that evaluates it:
//   RQS        zuko/transforms.py:469-567     affine  zuko/transforms.py:412-446
//   SOS        zuko/transforms.py:927-963 + zuko/utils.py:349-363 (Gauss-Legendre), :170-180 (bisection)
//   Bernstein  zuko/transforms.py:640-831
#pragma once

#include "zk_common.h"

namespace zk {

// ---------------------------------------------------------------------------------------------
// math policies.  MathIEEE: correctly rounded division, ocml exp/log (<= 1 ulp) — the fp64 kernels and
// the generic-K fp32 spline.  MathFast (fp32): v_rcp/v_exp/v_log based (a few ulp) — the affine
// epilogue of the fused conditioner kernel, whose parameters already carry ~1e-6 relative GEMM
// summation-order noise.
// ---------------------------------------------------------------------------------------------
template <typename T> struct MathIEEE {
  static __device__ __forceinline__ T div(T a, T b) { return a / b; }
  static __device__ __forceinline__ T div_safe(T a, T b) { return a / b; }
  static __device__ __forceinline__ T exp(T v) { return t_exp(v); }
  static __device__ __forceinline__ T log(T v) { return t_log(v); }
  static __device__ __forceinline__ T max(T a, T b) { return b > a ? b : a; }              // NaN in `a` sticks, as torch.max
  static __device__ __forceinline__ T knot(T cum, T bound) { return bound * (T(2) * cum - T(1)); }  // transforms.py:488-489, literally
};
// MathTight (fp32): within ~1.5 ulp of MathIEEE at a third of the instruction count — division by
// v_rcp_f32 + one Newton step on the quotient (correctly rounded except in rare ties; operands here
// are far from the exponent extremes v_div_scale/v_div_fixup exist for), exp by v_exp_f32 on a
// compensated x*log2(e) product.  Used by the standalone fp32 affine kernel; log stays ocml.
struct MathTight {
  // a/b for FINITE operands (NaN propagates; an infinite operand yields NaN instead of inf/0 — every
  // use in the spline/softclip math is one where the reference itself produces NaN for such inputs)
  static __device__ __forceinline__ float div(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
  }
  // same, but keeps IEEE results for infinite operands (affine inverse of y = +-inf)
  static __device__ __forceinline__ float div_safe(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    const float e = fmaf(-b, q, a);
    return (e - e == 0.f) ? fmaf(e, r, q) : q;
  }
  static __device__ __forceinline__ float exp(float v) {
    const float l2e = 1.44269504088896340736f, ln2 = 0.69314718055994530942f;
    const float hi = v * l2e;
    const float lo = fmaf(v, l2e, -hi);  // rounding error of the product (exact); l2e's own 1.3e-8 relative error is dropped
    const float e = __builtin_amdgcn_exp2f(hi);
    return fmaf(e, lo * ln2, e);
  }
  static __device__ __forceinline__ float log(float v) { return logf(v); }
  static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
  static __device__ __forceinline__ float knot(float cum, float bound) { return fmaf(cum, 2.f * bound, -bound); }
};
template <typename T> struct MathStd { typedef MathIEEE<T> type; };
template <> struct MathStd<float> { typedef MathTight type; };  // policy of the standalone kernels

struct MathFast {
  static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  static __device__ __forceinline__ float div_safe(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
  static __device__ __forceinline__ float knot(float cum, float bound) { return fmaf(cum, 2.f * bound, -bound); }
  static __device__ __forceinline__ float exp(float v) { return __expf(v); }
  static __device__ __forceinline__ float log(float v) { return __logf(v); }
};

// ---------------------------------------------------------------------------------------------
// soft clipping of unconstrained parameters (transforms.py:436, :480-482)
// ---------------------------------------------------------------------------------------------
template <typename T, class M = MathIEEE<T>> __device__ __forceinline__ T softclip(T v, T ls) { return M::div(v, T(1) + t_abs(M::div(v, ls))); }
template <typename T, class M = MathIEEE<T>> __device__ __forceinline__ T softclip2(T v, T ls) { return M::div(v, T(1) + t_abs(M::div(T(2) * v, ls))); }

// ---------------------------------------------------------------------------------------------
// monotonic affine (transforms.py:436-446)
// ---------------------------------------------------------------------------------------------
template <typename T, class M = MathIEEE<T>> __device__ __forceinline__ void affine_fwd(T shift, T scale, T ls, T x, T& y, T& ladj) {
  T lsc = softclip<T, M>(scale, ls);
  y = x * M::exp(lsc) + shift;
  ladj = lsc;
}
template <typename T, class M = MathIEEE<T>> __device__ __forceinline__ T affine_inv(T shift, T scale, T ls, T y) {
  T lsc = softclip<T, M>(scale, ls);
  return M::div_safe(y - shift, M::exp(lsc));
}

// ---------------------------------------------------------------------------------------------
// rational-quadratic spline
// ---------------------------------------------------------------------------------------------

// softmax over K soft-clipped values followed by the padded cumulative sum mapped to [-B, B]
// (transforms.py:480-481, 484-485, 488-489).  `ld(j)` returns the j-th unconstrained value.
template <typename T, int K, class M = MathIEEE<T>, typename Ld> __device__ __forceinline__ void rqs_axis_knots(Ld ld, T bound, T ls, T (&knot)[K + 1]) {
  T v[K];
  T m;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    v[j] = softclip2<T, M>(ld(j), ls);
    m = (j == 0) ? v[0] : M::max(m, v[j]);
  }
  T s = T(0);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    v[j] = M::exp(v[j] - m);
    s += v[j];
  }
  T r = M::div(T(1), s);
  T cum = T(0);
  knot[0] = M::knot(cum, bound);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    cum += v[j] * r;
    knot[j + 1] = M::knot(cum, bound);
  }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// knot slopes: exp(softclip(d)) inside, 1 at both ends (transforms.py:482, 486, 490)
template <typename T, int K, class M = MathIEEE<T>, typename Ld> __device__ __forceinline__ void rqs_slopes(Ld ld, T ls, T (&kd)[K + 1]) {
  kd[0] = T(1);
  kd[K] = T(1);
#pragma unroll
  for (int j = 1; j < K; ++j) kd[j] = M::exp(softclip<T, M>(ld(j - 1), ls));
}

// Bin search and corner gather in one sweep: k = #(search knots < v) - 1 (strict compare,
// transforms.py:521-526) and the corners of the bin selected by the SAME compares — the knots are
// increasing, so "knot j < v" is a prefix property and the last true j is the bin.  For v outside
// [first, last] knot (k = -1 or K, or NaN) the reference gathers the wrapped bin k % K
// (transforms.py:502) and then masks the result; any finite bin gives the same masked outputs
// (y = v, ladj = 0 * log(finite) or NaN for non-finite v), so the first / last bin is used here.
template <typename T, int K>
__device__ __forceinline__ int rqs_locate(const T (&ks)[K + 1], const T (&kx)[K + 1], const T (&ky)[K + 1], const T (&kd)[K + 1], T v, bool& inside, T& x0,
                                          T& x1, T& y0, T& y1, T& d0, T& d1) {
  int cnt = (ks[0] < v) ? 1 : 0;
  x0 = kx[0]; x1 = kx[1]; y0 = ky[0]; y1 = ky[1]; d0 = kd[0]; d1 = kd[1];
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const bool c = ks[j] < v;
    cnt += c ? 1 : 0;
    x0 = c ? kx[j] : x0;
    x1 = c ? kx[j + 1] : x1;
    y0 = c ? ky[j] : y0;
    y1 = c ? ky[j + 1] : y1;
    d0 = c ? kd[j] : d0;
    d1 = c ? kd[j + 1] : d1;
  }
  cnt += (ks[K] < v) ? 1 : 0;
  const int k = cnt - 1;
  inside = (k >= 0) && (k < K);
  return k;
}

// forward value + log|dy/dx| (transforms.py:554-567).  Out-of-range / NaN / inf behaviour is
// inherited from the literal `mask * ...` arithmetic of the reference.
template <typename T, int K, class M = MathIEEE<T>>
__device__ __forceinline__ void rqs_fwd(const T (&kx)[K + 1], const T (&ky)[K + 1], const T (&kd)[K + 1], T x, T& y, T& ladj, int& k) {
  bool inside;
  T x0, x1, y0, y1, d0, d1;
  k = rqs_locate<T, K>(kx, kx, ky, kd, x, inside, x0, x1, y0, y1, d0, d1);
  T m = inside ? T(1) : T(0);
  T s = M::div(y1 - y0, x1 - x0);
  T z = M::div(m * (x - x0), x1 - x0);
  T omz = T(1) - z;
  T t = (d0 + d1) - T(2) * s;
  T den = s + (t * z) * omz;
  T num = s * (z * z) + (d0 * z) * omz;
  T yy = y0 + M::div((y1 - y0) * num, den);
  T jac = M::div((s * s) * ((((T(2) * s) * z) * omz + d0 * (omz * omz)) + d1 * (z * z)), den * den);
  y = inside ? yy : x;
  ladj = m * M::log(jac);
}

// inverse value (transforms.py:534-548): bin search on the vertical knots, stable quadratic root
template <typename T, int K, class M = MathIEEE<T>>
__device__ __forceinline__ void rqs_inv(const T (&kx)[K + 1], const T (&ky)[K + 1], const T (&kd)[K + 1], T y, T& x, int& k) {
  bool inside;
  T x0, x1, y0, y1, d0, d1;
  k = rqs_locate<T, K>(ky, kx, ky, kd, y, inside, x0, x1, y0, y1, d0, d1);
  T m = inside ? T(1) : T(0);
  T s = M::div(y1 - y0, x1 - x0);
  T y_ = m * (y - y0);
  T t = (d0 + d1) - T(2) * s;
  T a = (y1 - y0) * (s - d0) + y_ * t;
  T b = (y1 - y0) * d0 - y_ * t;
  T c = (-s) * y_;
  T z = M::div(T(2) * c, (-b) - t_sqrt(b * b - (T(4) * a) * c));
  T xx = x0 + z * (x1 - x0);
  x = inside ? xx : y;
}

// ---------------------------------------------------------------------------------------------
// Lean fp32 evaluation straight from the unconstrained parameters (fast-math policy).  Same map as
// rqs_axis_knots + rqs_slopes + rqs_fwd / rqs_inv, organised around what the VALU has to issue:
//   * softclip and the softmax exponent are one chain  e_j = 2^(u_j * r_j),  r_j = 1 / ((1 + |u_j| c) / log2 e),
//     i.e. fma, v_rcp, (packed) mul, v_exp per value; no max-subtraction — the clipped exponent is
//     confined to (-|ln slope|/2, |ln slope|/2) = (-3.45, 3.45), so nothing can overflow;
//   * both axes ride the two halves of v_pk_{mul,add,fma}_f32;
//   * the bin is found by bisection over the K+1 knots, each level halving the candidate knots of
//     the search axis, of the other axis and of the (zero-padded, still unconstrained) derivative
//     parameters with the same compare: 3 (K/2+1 + K/4+1 + ... + 2) v_cndmask instead of 6 (K-1);
//     for increasing knots the selected bin is exactly #(knots < v) - 1 (transforms.py:521-526);
//   * only the two selected derivative parameters are soft-clipped and exponentiated (the padded
//     zeros give exp(0) = 1 at both ends, transforms.py:486, 490).
// ---------------------------------------------------------------------------------------------
struct RqsLeanConst {
  float bound;
  float c2l;   // (2 / |ln slope|) / log2(e)
  float c1l;   // (1 / |ln slope|) / log2(e)
  float il2e;  // 1 / log2(e) = ln 2
};
static inline RqsLeanConst rqs_lean_const(double bound, double ls) {
  const double ln2 = 0.69314718055994530942, a = ls < 0 ? -ls : ls;
  RqsLeanConst c;
  c.bound = (float)bound; c.c2l = (float)(2.0 / a * ln2); c.c1l = (float)(1.0 / a * ln2); c.il2e = (float)ln2;
  return c;
}

// one bisection level over M+1 candidates (M a power of two >= 2); returns the index of the selected bin
template <int M> struct RqsBisect {
  static __device__ __forceinline__ int run(float v, float (&ks)[M + 1], float (&ko)[M + 1], float (&kr)[M + 1], float& s0, float& s1, float& o0, float& o1,
                                            float& r0, float& r1) {
    const bool c = ks[M / 2] < v;
    float ns[M / 2 + 1], no[M / 2 + 1], nr[M / 2 + 1];
#pragma unroll
    for (int i = 0; i <= M / 2; ++i) {
      ns[i] = c ? ks[M / 2 + i] : ks[i];
      no[i] = c ? ko[M / 2 + i] : ko[i];
      nr[i] = c ? kr[M / 2 + i] : kr[i];
    }
    return (c ? M / 2 : 0) + RqsBisect<M / 2>::run(v, ns, no, nr, s0, s1, o0, o1, r0, r1);
  }
};
template <> struct RqsBisect<1> {
  static __device__ __forceinline__ int run(float, float (&ks)[2], float (&ko)[2], float (&kr)[2], float& s0, float& s1, float& o0, float& o1, float& r0,
                                            float& r1) {
    s0 = ks[0]; s1 = ks[1]; o0 = ko[0]; o1 = ko[1]; r0 = kr[0]; r1 = kr[1];
    return 0;
  }
};

// `ks_out` (optional, K+1 floats): the knots of the SEARCH axis exactly as the bisection compared them
// (the diagnostic entry points zk_rqs_diag / zk_ar_forward_diag expose them, so that the bin index
// can be asserted against the kernel's own knots: k == #(ks < v) - 1, zuko/transforms.py:521-523).
// INV_LADJ (inverse only): also return log|dy/dx| of the FORWARD map at the solution (the incremental inverse kernel
// accumulates it, so that rsample_and_log_prob needs no second pass; zuko/distributions.py:129-138).
template <int K, bool INV, bool INV_LADJ = false, typename LdW, typename LdH, typename LdD>
__device__ __forceinline__ void rqs_lean(LdW ldw, LdH ldh, LdD ldd, const RqsLeanConst& c, float v, float& out, float& ladj, int& k, float* ks_out = nullptr) {
  static_assert((K & (K - 1)) == 0 && K >= 2, "bisection needs a power-of-two bin count");
  float kx[K + 1], ky[K + 1], kr[K + 1];
  f32x2_t acc = {0.f, 0.f};
  f32x2_t cum[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const f32x2_t u = {ldw(j), ldh(j)};
    const f32x2_t r = {__builtin_amdgcn_rcpf(fmaf(fabsf(u.x), c.c2l, c.il2e)), __builtin_amdgcn_rcpf(fmaf(fabsf(u.y), c.c2l, c.il2e))};
    const f32x2_t t = u * r;
    acc += f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    cum[j] = acc;
  }
  const f32x2_t scale = f32x2_t{__builtin_amdgcn_rcpf(acc.x), __builtin_amdgcn_rcpf(acc.y)} * (2.f * c.bound);
  const f32x2_t negB = {-c.bound, -c.bound};
  kx[0] = -c.bound; ky[0] = -c.bound;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const f32x2_t kn = __builtin_elementwise_fma(cum[j], scale, negB);
    kx[j + 1] = kn.x; ky[j + 1] = kn.y;
  }
  kr[0] = 0.f; kr[K] = 0.f;
#pragma unroll
  for (int j = 1; j < K; ++j) kr[j] = ldd(j - 1);

  if (ks_out) {
#pragma unroll
    for (int j = 0; j <= K; ++j) ks_out[j] = INV ? ky[j] : kx[j];
  }
  float x0, x1, y0, y1, r0, r1;
  bool inside, above;
  int bin;
  if (INV) {
    above = ky[K] < v;
    inside = (ky[0] < v) && !above;
    bin = RqsBisect<K>::run(v, ky, kx, kr, y0, y1, x0, x1, r0, r1);
  } else {
    above = kx[K] < v;
    inside = (kx[0] < v) && !above;
    bin = RqsBisect<K>::run(v, kx, ky, kr, x0, x1, y0, y1, r0, r1);
  }
  k = inside ? bin : (above ? K : -1);  // #(knots < v) - 1, NaN -> -1 (dropped by the compiler where unused)
  const f32x2_t rr = {r0, r1};
  const f32x2_t td = rr * f32x2_t{__builtin_amdgcn_rcpf(fmaf(fabsf(r0), c.c1l, c.il2e)), __builtin_amdgcn_rcpf(fmaf(fabsf(r1), c.c1l, c.il2e))};
  const float d0 = __builtin_amdgcn_exp2f(td.x), d1 = __builtin_amdgcn_exp2f(td.y);

  const float m = inside ? 1.f : 0.f;
  const float dx = x1 - x0, dy = y1 - y0;
  const float rdx = __builtin_amdgcn_rcpf(dx);
  float s = dy * rdx;
  if (INV) s = fmaf(fmaf(-dx, s, dy), rdx, s);  // the inverse amplifies errors of s by 1 / slope: one Newton step
  const float t = (d0 + d1) - 2.f * s;
  if (INV) {
    const float y_ = m * (v - y0);
    const float yt = y_ * t;
    const float qa = fmaf(dy, s - d0, yt);
    const float qb = fmaf(dy, d0, -yt);
    const float qc = -s * y_;
    const float disc = fmaf(qb, qb, -4.f * qa * qc);
    const float qd = -qb - __builtin_amdgcn_sqrtf(disc);
    const float rq = __builtin_amdgcn_rcpf(qd);
    float z = (2.f * qc) * rq;
    z = fmaf(fmaf(-qd, z, 2.f * qc), rq, z);
    out = inside ? fmaf(z, dx, x0) : v;
    if constexpr (INV_LADJ) {
      const float omz = 1.f - z;
      const float zz = z * omz;
      const float rden = __builtin_amdgcn_rcpf(fmaf(t, zz, s));
      const float jn = fmaf(d1 * z, z, fmaf(d0 * omz, omz, (2.f * s) * zz));
      const float sr = s * rden;
      ladj = m * (__builtin_amdgcn_logf((sr * sr) * jn) * c.il2e);
    } else {
      ladj = 0.f;
    }
  } else {
    const float z = (m * (v - x0)) * rdx;
    const float omz = 1.f - z;
    const float zz = z * omz;
    const float den = fmaf(t, zz, s);
    const float rden = __builtin_amdgcn_rcpf(den);
    const float num = fmaf(s * z, z, d0 * zz);
    const float yy = fmaf(dy * num, rden, y0);
    const float jn = fmaf(d1 * z, z, fmaf(d0 * omz, omz, (2.f * s) * zz));
    const float sr = s * rden;
    const float jac = (sr * sr) * jn;
    out = inside ? yy : v;
    ladj = m * (__builtin_amdgcn_logf(jac) * c.il2e);
  }
}
template <int K, bool INV, typename LdW, typename LdH, typename LdD>
__device__ __forceinline__ void rqs_lean(LdW ldw, LdH ldh, LdD ldd, const RqsLeanConst& c, float v, float& out, float& ladj) {
  int k;
  rqs_lean<K, INV, false>(ldw, ldh, ldd, c, v, out, ladj, k);
}

// ---------------------------------------------------------------------------------------------
// sum-of-squares polynomial.  Coefficients a[P][L1] are read through `ld(p*L1 + j)`.
// ---------------------------------------------------------------------------------------------
#define ZK_SOS_MAX_NODES 16

template <typename T> struct SosConst {
  T bound;     // 10 (MonotonicTransform default, transforms.py:593)
  T slope;     // additive floor of g (transforms.py:963)
  int P, L1;   // polynomials, degree+1 (= number of quadrature nodes, transforms.py:952)
  T node[ZK_SOS_MAX_NODES];    // Gauss-Legendre nodes on [0,1]  (utils.py:337-339)
  T weight[ZK_SOS_MAX_NODES];  // weights / 2
};

// g(x) = mean_p (1 + sum_j a_pj (x/B)^j)^2 + slope   (transforms.py:958-963)
template <typename T, typename Ld> __device__ __forceinline__ T sos_g(const SosConst<T>& c, Ld ld, T x) {
  T u = x / c.bound;
  T acc = T(0);
  for (int p = 0; p < c.P; ++p) {
    T pw = T(1);  // u**0 == 1 for every u, NaN included (IEEE pow)
    T dot = T(0);
    for (int j = 0; j < c.L1; ++j) {
      dot += ld(p * c.L1 + j) * pw;
      pw *= u;
    }
    T q = T(1) + dot;
    acc += q * q;
  }
  return acc / T(c.P) + c.slope;
}

// f(x) = int_0^x g  by Gauss-Legendre on [0, x]  (transforms.py:911-918, utils.py:349-363)
template <typename T, typename Ld> __device__ __forceinline__ T sos_f(const SosConst<T>& c, Ld ld, T x) {
  T acc = T(0);
  for (int i = 0; i < c.L1; ++i) {
    T w = c.node[i];
    // torch.lerp(0, x, w): a + w*(b-a) below one half, b - (b-a)*(1-w) above
    T pt = (w < T(0.5)) ? (T(0) + w * (x - T(0))) : (x - (x - T(0)) * (T(1) - w));
    acc += c.weight[i] * sos_g<T>(c, ld, pt);
  }
  return (x - T(0)) * acc;
}

// sos_g / sos_f with the polynomial count and the degree as template arguments (same expression trees, fully unrolled): for callers whose
// coefficients live in registers (the fused kernels' epilogue: a run-time loop would index the register array through scratch memory)
template <typename T, int P, int L1, typename Ld> __device__ __forceinline__ T sos_g_static(const SosConst<T>& c, Ld ld, T x) {
  T u = x / c.bound;
  T acc = T(0);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    T pw = T(1);
    T dot = T(0);
#pragma unroll
    for (int j = 0; j < L1; ++j) {
      dot += ld(p * L1 + j) * pw;
      pw *= u;
    }
    T q = T(1) + dot;
    acc += q * q;
  }
  return acc / T(P) + c.slope;
}
template <typename T, int P, int L1, typename Ld> __device__ __forceinline__ T sos_f_static(const SosConst<T>& c, Ld ld, T x) {
  T acc = T(0);
#pragma unroll
  for (int i = 0; i < L1; ++i) {
    T w = c.node[i];
    T pt = (w < T(0.5)) ? (T(0) + w * (x - T(0))) : (x - (x - T(0)) * (T(1) - w));
    acc += c.weight[i] * sos_g_static<T, P, L1>(c, ld, pt);
  }
  return (x - T(0)) * acc;
}

// fixed-count bisection on [-B, B] (utils.py:170-180; n = ceil(log2(2B/eps)), transforms.py:615)
template <typename T, typename Ld> __device__ __forceinline__ T sos_inv(const SosConst<T>& c, Ld ld, T y, int n) {
  T a = -c.bound, b = c.bound;
  for (int it = 0; it < n; ++it) {
    T mid = (a + b) / T(2);
    bool below = sos_f<T>(c, ld, mid) < y;
    a = below ? mid : a;
    b = below ? b : mid;
  }
  return (a + b) / T(2);
}

// ---------------------------------------------------------------------------------------------
// Bernstein polynomial.  NC = number of CONSTRAINED coefficients (order M = NC-1).
// ---------------------------------------------------------------------------------------------
#define ZK_BERN_EPS 1e-6  /* default of zuko/transforms.py:594; the entry points take eps at run time */

template <typename T> __device__ __forceinline__ T softplus(T v) {  // torch softplus, beta=1, threshold=20
  return v > T(20) ? v : t_log1p(t_exp(v));
}

// unbounded: theta = cumsum([t0, sp(t1), sp(t1), sp(t2), ..., sp(t_last), sp(t_last)]) - log(2) * n / 2
// (transforms.py:703-727); n unconstrained -> NC = n + 2
template <typename T, int NC, typename Ld> __device__ __forceinline__ void bern_theta_unbounded(Ld ld, T (&th)[NC]) {
  constexpr int n = NC - 2;
  const T shift = T(0.69314718055994530942 * n / 2.0);
  T cum = ld(0);
  th[0] = cum - shift;
  cum += softplus<T>(ld(1));
  th[1] = cum - shift;
#pragma unroll
  for (int j = 1; j < n; ++j) {
    cum += softplus<T>(ld(j));
    th[j + 1] = cum - shift;
  }
  cum += softplus<T>(ld(n - 1));
  th[n + 1] = cum - shift;
}

// bounded: theta = cumsum([-B, e, e, softmax(t) * (2B - 4e), e, e]),  e = 2B / (n + 4)
// (transforms.py:797-818); n unconstrained -> NC = n + 5
template <typename T, int NC, typename Ld> __device__ __forceinline__ void bern_theta_bounded(Ld ld, T bound, T (&th)[NC]) {
  constexpr int n = NC - 5;
  const T edge = (T(2) * bound) / T(n + 4);
  const T span = T(2) * bound - T(4) * edge;
  T v[n];
  T m;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    v[j] = ld(j);
    m = (j == 0) ? v[0] : (v[j] > m ? v[j] : m);
  }
  T s = T(0);
#pragma unroll
  for (int j = 0; j < n; ++j) {
    v[j] = t_exp(v[j] - m);
    s += v[j];
  }
  T r = T(1) / s;
  T cum = -bound;
  th[0] = cum;
  cum += edge; th[1] = cum;
  cum += edge; th[2] = cum;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    cum += (v[j] * r) * span;
    th[3 + j] = cum;
  }
  cum += edge; th[n + 3] = cum;
  cum += edge; th[n + 4] = cum;
}

// binomial C(n, k) at compile time (exact in double up to n = 56)
constexpr double zk_binom(int n, int k) {
  double c = 1.0;
  for (int i = 1; i <= k; ++i) c = c * (double)(n - k + i) / (double)i;
  return c;
}
template <typename T> __device__ __forceinline__ T zk_ipow(T w, int n) {  // w^n, n a compile-time constant at every call site
  T r = T(1), b = w;
  for (int e = n; e > 0; e >>= 1) {
    if (e & 1) r = r * b;
    b = b * b;
  }
  return r;
}

// de Casteljau: value B(u) = sum_i C(M,i) u^i (1-u)^(M-i) theta_i (closed form of transforms.py:736-740; equality with the Beta-pdf form verified in
// SURVEY 9.1).  Derivative dB/du = M sum_i C(M-1,i) u^i (1-u)^(M-1-i) (theta_{i+1} - theta_i): the DIFFERENCES are all positive (theta is a cumulative sum
// of positive increments), so the sum has no cancellation in any arrangement — it is evaluated by Horner in u / v (u <= 1/2) or v / u (both chains; the
// overflowing one is never selected), 2 M fused multiply-adds.  (Until round 5 the derivative was M (b_1 - b_0) of the value sweep's last two points: two
// numbers of size |theta| that agree in their leading digits wherever the polynomial is flat — the one comparison of the golden sets whose log-derivative sat
// 2.9x further from float64 than the float32 reference's; a second de Casteljau sweep over the differences fixed that at +30 % on the fused BPF layer, this form
// at no cost.)
template <typename T, int NC> __device__ __forceinline__ void bern_eval(const T (&th)[NC], T u, T& val, T& dval) {
  constexpr int M = NC - 1;
  const T v = T(1) - u;
  // the derivative first, straight from theta (no array of differences: with one the fused Bernstein layer spilled 30 registers, and a spilled
  // accumulator of the matrix instruction came back wrong in 1 launch of 8 — tests/test_gpu_flows.py::test_polynomial_flows_run_on_a_fused_split_kernel)
  {
    const T sr = u / v, rr = v / u;
    T hA = (th[M] - th[M - 1]) * T(zk_binom(M - 1, M - 1)), hB = (th[1] - th[0]) * T(zk_binom(M - 1, 0));
#pragma unroll
    for (int i = 1; i < M; ++i) {
      hA = hA * sr + (th[M - i] - th[M - 1 - i]) * T(zk_binom(M - 1, M - 1 - i));
      hB = hB * rr + (th[i + 1] - th[i]) * T(zk_binom(M - 1, i));
    }
    const bool low = u <= T(0.5);
    dval = T(M) * ((low ? hA : hB) * zk_ipow<T>(low ? v : u, M - 1));
  }
  T b[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) b[i] = th[i];
#pragma unroll
  for (int r = 1; r < NC - 1; ++r) {
#pragma unroll
    for (int i = 0; i < NC - r; ++i) b[i] = v * b[i] + u * b[i + 1];
  }
  val = v * b[0] + u * b[1];
}

template <typename T> struct BernTails { T off0, off1, slp0, slp1; };

// offsets/slopes of the linear continuation (transforms.py:685-701 unbounded, :820-831 bounded)
template <typename T, int NC> __device__ __forceinline__ BernTails<T> bern_tails(const T (&th)[NC], bool bounded, T bound, T eps) {
  BernTails<T> t;
  if (bounded) {
    t.off0 = -bound; t.off1 = bound; t.slp0 = T(2) * bound; t.slp1 = T(2) * bound;
  } else {
    bern_eval<T, NC>(th, eps, t.off0, t.slp0);
    bern_eval<T, NC>(th, T(1) - eps, t.off1, t.slp1);
  }
  return t;
}

// y = f(x) and dy/dx (transforms.py:742-760; derivative = what autograd yields at :623-637)
template <typename T, int NC>
__device__ __forceinline__ void bern_fwd(const T (&th)[NC], const BernTails<T>& t, T bound, T x, T& y, T& dydx, T eps) {
  T u = (x + bound) / (T(2) * bound);
  bool lo = u <= eps;
  bool hi = u >= T(1) - eps;
  T safe = (lo || hi) ? T(0.5) : u;
  T val, dval;
  bern_eval<T, NC>(th, safe, val, dval);
  T ylo = t.slp0 * (u - eps) + t.off0;
  T yhi = t.slp1 * ((u - T(1)) + eps) + t.off1;
  y = lo ? ylo : val;
  y = hi ? yhi : y;
  T du = lo ? t.slp0 : dval;
  du = hi ? t.slp1 : du;
  dydx = du / (T(2) * bound);
}

// x = f^{-1}(y): n-step bisection on [-B, B] + closed-form tails (transforms.py:762-777, :609-617)
template <typename T, int NC> __device__ __forceinline__ T bern_inv(const T (&th)[NC], const BernTails<T>& t, T bound, T y, int n, T eps) {
  T a = -bound, b = bound;
  for (int it = 0; it < n; ++it) {
    T mid = (a + b) / T(2);
    T fy, d;
    bern_fwd<T, NC>(th, t, bound, mid, fy, d, eps);
    bool below = fy < y;
    a = below ? mid : a;
    b = below ? b : mid;
  }
  T x = (a + b) / T(2);
  T xlo = (((y - t.off0) / t.slp0 + eps) * T(2)) * bound - bound;
  T xhi = ((((y - t.off1) / t.slp1 - eps) + T(1)) * T(2)) * bound - bound;
  x = (y <= t.off0) ? xlo : x;
  x = (y >= t.off1) ? xhi : x;
  return x;
}

}  // namespace zk
