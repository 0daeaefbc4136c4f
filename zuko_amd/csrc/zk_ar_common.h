// zuko_amd — pieces shared by the fused autoregressive kernels (fused_ar.hip: generic tile-skipping kernel, inverse sweeps;
// fused_ar_static.hip: the static-shape density kernel): launch arguments, activations, univariate epilogues.
#pragma once
#include "zk_univariate.h"

namespace zk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AR_TF 256 /* floats per tile image */
#define AR_WAVES 8
#define AR_T 16   /* activation tiles (256 units) */
#define ARS_ABI 9 /* contract between this library and the generated static-shape kernels (zuko_amd/static_ar.py): bump on any change of ArArgs */

struct ArArgs {
  int64_t N;
  int D, DIN;               // features, conditioner inputs (features + context), DIN % 4 == 0
  const float* x; int64_t ldx;  // [N, DIN] = cat(x, c) zero-padded to a multiple of 4; rows 16-byte aligned
  float* y; int64_t ldy;
  const float* yin; int64_t ldyin;  // INVERSE only: the values to invert (y of the forward map)
  float* ladj; int accumulate;
  const float* stream;
  const float* bias;
  const uint32_t* skip;
  const int32_t* featmap;
  int L, NG, n_chunks, act, bias_floats, dbg;
  const int* sched;  // optional chunk schedule (partial inverse sweeps): stream chunk ids in consumption order
  int n_sched;
  int olim[8];       // per hidden layer: last out-group (of 4 tiles) to evaluate; 3 = all
  int g0, g1;        // last-layer groups [g0, g1) to evaluate
  int xlds;  // x (or y_in) and the result tile are staged in a wave-private LDS region (stride xs words)
  int xs;
  float bound, ls;
  RqsLeanConst lc;   // spline epilogues: constants of rqs_lean
  int64_t n_tiles;
  int32_t* bin_out;  // diagnostic instantiation only: bin index [N, D] and the K+1 search-axis knots [N, D, K+1]
  float* knots_out;
  int l1rev;         // static-shape kernels only: the first layer follows the pattern's ALTERNATIVE input tiles (descending feature order)
  // static-shape kernels, conditioner-only (training) instantiation: the hidden activations [N, width_l] (units in the stream's
  // sorted order) and the packed parameters phi [N, D * total] (module order) are written out; y / ladj are not
  float* act_out[3];
  float* phi_out;
  int64_t ldphi;
  // static-shape dgrad chain (ars_dgrad_kernel): saved activations h_l [N, width_l] whose sign gates the gradient of layer l
  const float* gate[3];
  // fused backward of an autoregressive transform (arxb_kernel): x / featmap as for the forward; the forward's phi and the gradient
  // of phi it writes for the weight gradients [N, D * total] (row stride ldpin); gy [N, D] (row stride ldgy), gl [N]; the input gradient
  // goes to phi_out (row stride ldphi) as in the dgrad chain
  const float* phi_in;
  float* gphi_out;
  int64_t ldpin;
  const float* gy; int64_t ldgy;
  const float* gl;
  // training launches: phi (and its gradient) in the kernels' PACKED order instead of the module's — row n holds, for group g and tile t,
  // 16 floats at (g NT + t) 16: parameter 4 t + r of the features of lane q = 0..3 at 4 q + r (row stride >= NG * NT * 16), which is
  // how a lane holds them in registers: 16-byte accesses, no regrouping (zuko_amd/train.py keeps such a phi to itself)
  int phi_packed;
  // polynomial maps (uni_kind 5, 6; operand-split static-shape kernels only): constants of the SOS quadrature, Bernstein continuation margin
  SosConst<float> sos;
  float eps;
  // two-part (f16) operand-split kernels (fused_ar_half_impl.h): 2^-ew_l of every linear layer, the power of two its weights were stored with
  float wdescale[4];
  // training launches (operand-split static-shape kernels): where to fold the maximum magnitude of every tensor the launch stores for the weight
  // gradients (zk_half.h: 64 slots) — forward: amax[l] for act_out[l]; backward: amax[l] for act_out[l] (gradient of a hidden layer), amax[3] for
  // gphi_out.  null = not wanted.  With them the weight gradients run on two-part f16 operands (csrc/train.hip: wgrad_split_body<true>).
  unsigned* amax[4];
};

__device__ __forceinline__ float act_f32(float v, int act) {
  switch (act) {
    case 1: return v < 0.f ? 0.f : v;
    case 2: return v > 0.f ? v : expm1f(v);
    case 3: return tanhf(v);
    case 4: return v / (1.f + expf(-v));
    case 5: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case 6: return 1.f / (1.f + expf(-v));
    case 7: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}


// ---- univariate epilogues -------------------------------------------------------------------------
struct UniAffine {
  static constexpr int TOTAL = 2, FPL = 2, NT = 1;
  template <bool INV> static __device__ __forceinline__ void poison(float* p, int base, float nan_or_zero) {
    p[base + 0] += nan_or_zero;
    p[base + 1] += nan_or_zero;
  }
  static constexpr int NKNOT = 1;
  template <typename P, typename A> static __device__ __forceinline__ void fwd(const P& p, int base, const A& a, float x, float& y, float& lj, int* k = nullptr, float* ks = nullptr) {
    affine_fwd<float, MathFast>(p(base + 0), p(base + 1), a.ls, x, y, lj);
  }
  template <typename P, typename A> static __device__ __forceinline__ float inv(const P& p, int base, const A& a, float y) {
    return affine_inv<float, MathFast>(p(base + 0), p(base + 1), a.ls, y);
  }
};

// K-bin rational-quadratic spline; CIRC: preceded by the circular shift of NCSF (zuko/flows/spline.py:65-72,
// zuko/transforms.py:344-348: x -> remainder(x, 2B) - B with B = pi passed as `bound`).
template <int K, bool CIRC> struct UniRqs {
  static constexpr int TOTAL = 3 * K - 1, FPL = 1, NT = (TOTAL + 3) / 4;
  static __device__ __forceinline__ float shift(float v, float bound) {
    const float period = 2.f * bound;
    float r = fmodf(v, period);
    r = (r < 0.f) ? r + period : r;  // torch.remainder: result takes the sign of the divisor
    return r - bound;
  }
  // All-NaN parameters (reference: zuko/nn.py:217-218 on a non-finite input) leave knot 0 = -B finite and every other
  // knot NaN: values right of -B land in bin 0 with a NaN corner (y = NaN), values at or left of it, NaN and -inf keep
  // y = v, and log|dy/dx| is NaN everywhere.  NaN widths (heights for the inverse, which searches the other axis) give
  // exactly that: the remaining parameters never reach an output that is not already NaN.
  template <bool INV> static __device__ __forceinline__ void poison(float* p, int base, float nan_or_zero) {
#pragma unroll
    for (int j = 0; j < K; ++j) p[base + (INV ? K : 0) + j] += nan_or_zero;
  }
  static constexpr int NKNOT = K + 1;
  // k / ks (diagnostic instantiation): bin index and search-axis knots of THIS evaluation
  template <typename P, typename A> static __device__ __forceinline__ void fwd(const P& p, int base, const A& a, float x, float& y, float& lj, int* k = nullptr, float* ks = nullptr) {
    int kk;
    rqs_lean<K, false>([&](int j) { return p(base + j); }, [&](int j) { return p(base + K + j); }, [&](int j) { return p(base + 2 * K + j); }, a.lc,
                       CIRC ? shift(x, a.bound) : x, y, lj, kk, ks);
    if (k) *k = kk;
  }
  template <typename P, typename A> static __device__ __forceinline__ float inv(const P& p, int base, const A& a, float y) {
    float x, lj;
    rqs_lean<K, true>([&](int j) { return p(base + j); }, [&](int j) { return p(base + K + j); }, [&](int j) { return p(base + 2 * K + j); }, a.lc, y, x, lj);
    return CIRC ? shift(x, a.bound) : x;
  }
};
// Shifted sum-of-squares polynomial (zuko/transforms.py:905-963 + the learned constant of zuko/flows/polynomial.py:64-70): P polynomials of L1
// coefficients, then the shift.  FORWARD only (the inverse is a bisection: it stays with the layer-wise kernels).
template <int P, int L1> struct UniSos {
  static constexpr int TOTAL = P * L1 + 1, FPL = 1, NT = (TOTAL + 3) / 4, NKNOT = 1;
  template <bool INV> static __device__ __forceinline__ void poison(float* p, int base, float nan_or_zero) {
    p[base] += nan_or_zero;             // (a NaN coefficient makes f and g NaN: y and log|dy/dx| NaN, as the reference's all-NaN parameters do)
    p[base + P * L1] += nan_or_zero;
  }
  template <typename Pa, typename A> static __device__ __forceinline__ void fwd(const Pa& p, int base, const A& a, float x, float& y, float& lj, int* k = nullptr, float* ks = nullptr) {
    auto ld = [&](int j) { return p(base + j); };
    y = sos_f_static<float, P, L1>(a.sos, ld, x) + p(base + P * L1);
    lj = t_log(sos_g_static<float, P, L1>(a.sos, ld, x));
  }
};
// Bounded Bernstein polynomial (zuko/transforms.py:779-831) with NC - 5 unconstrained parameters.  FORWARD only.
template <int NC> struct UniBernBounded {
  static constexpr int TOTAL = NC - 5, FPL = 1, NT = (TOTAL + 3) / 4, NKNOT = 1;
  template <bool INV> static __device__ __forceinline__ void poison(float* p, int base, float nan_or_zero) { p[base] += nan_or_zero; }  // (softmax: one NaN makes every coefficient NaN)
  template <typename Pa, typename A> static __device__ __forceinline__ void fwd(const Pa& p, int base, const A& a, float x, float& y, float& lj, int* k = nullptr, float* ks = nullptr) {
    float th[NC];
    bern_theta_bounded<float, NC>([&](int j) { return p(base + j); }, a.bound, th);
    const BernTails<float> t = bern_tails<float, NC>(th, true, a.bound, a.eps);
    float d;
    bern_fwd<float, NC>(th, t, a.bound, x, y, d, a.eps);
    lj = t_log(d);
  }
};
typedef UniSos<3, 5> UniSos3x5;            // SOSPF defaults (zuko/flows/polynomial.py:51-53)
typedef UniBernBounded<22> UniBern17;      // BPF default degree 16 (zuko/flows/polynomial.py:97)
typedef UniRqs<8, false> UniRqs8;
typedef UniRqs<4, false> UniRqs4;
typedef UniRqs<16, false> UniRqs16;
typedef UniRqs<8, true> UniCircRqs8;

}  // namespace zk
