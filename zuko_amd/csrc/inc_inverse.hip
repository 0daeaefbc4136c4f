// zuko_amd — INCREMENTAL autoregressive inverse: x = f^{-1}(y) of one MaskedAutoregressiveTransform in ONE launch whose
// multiply-add count is ~1.5x that of a density evaluation.
//
// Replaces the loop of AutoregressiveTransform._inverse (zuko/transforms.py:994-1000) — `passes` evaluations of the whole
// conditioner (zuko/nn.py:217-218 x layers) + the inverse univariate map (zuko/transforms.py:534-548 / :443) — for masked
// conditioners whose hidden units fit ALIGNED tiles (zuko_amd/incremental.py): tile j of every hidden layer depends only on
// feature groups <= j, and group j's parameters only on tiles <= j.  Per 16 samples a wavefront then walks the groups once:
//
//   pull   pre-activations of tile j (all layers) and the parameters of group j receive, ONCE, the contributions of the
//          final tiles t < j (v_mfma_f32_16x16x4_f32, weights streamed through the LDS ring in a fixed order);
//   iterate the diagonal weight tiles stay in registers; pass r = 0..3 re-evaluates tile j of every layer and the group's
//          parameters from them, inverts the univariate map for the group's r-th feature (the only one whose inputs just
//          became final) and writes x back to the wave-private LDS tile the first layer reads its B operand from;
//          a fifth pass finalises tile j.
//
// The final hidden activations of all tiles live in registers (3 x 17 tiles x 4 VGPRs): one wavefront per SIMD
// (launch_bounds(256, 1)), four wavefronts = 64 samples per workgroup sharing the weight ring.
// The kernel also accumulates log|dy/dx| of the forward map at the solution, so rsample_and_log_prob
// (zuko/distributions.py:129-138) needs no second pass.
#include "../../include/zuko_amd.h"
#include "zk_univariate.h"
#include <type_traits>
#include <utility>

namespace zk {

typedef float f32x4i __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8i __attribute__((ext_vector_type(8)));

// HALF instantiation (round 6): the PULL phase — every off-diagonal weight tile, multiplied once per sample — runs on v_mfma_f32_16x16x32_f16 with
// the two-part operand split of csrc/fused_ar_half_impl.h (operand = two f16 numbers after an exact power-of-two scaling, three partial products, f32
// accumulation) instead of four v_mfma_f32_16x16x4_f32 per 16 x 16 tile: 48 matrix cycles per 16 x 32 block against 256.  The final activations of
// a layer are held as PAIRS of tiles (2 p, 2 p + 1) — the B operand of a block: 4 values of either tile per lane, as h and l parts (the same 4
// registers per tile as the f32 values) — converted when a tile becomes final with the pair's own scale (max over the sample's 32 values of the pair:
// two shuffles; the even tile is converted alone first, and again with its partner).  A block's three products go to a zeroed temporary that enters
// the pre-activation through ONE fma with 2^-(ew + ea) (ew: the layer's weight scale, host; ea: the pair's).  The first layer (inputs are x itself)
// and the five passes over the diagonal tiles stay on the f32 instruction: their operands change every pass.
struct IncPair { f16x8i h, l; };
__device__ __forceinline__ void inc_pair_convert(const f32x4i& lo, const f32x4i& hi, float amax, float wdesc, IncPair& p, float& dd) {
  amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  int ea = 15 - __builtin_amdgcn_frexp_expf(amax);  // amax 2^ea in [2^14, 2^15); zero / inf / NaN: ea = 15 (zeros stay zeros, non-finite values become NaN)
  ea = ea > 90 ? 90 : (ea < -90 ? -90 : ea);
  const float s = __builtin_amdgcn_ldexpf(1.0f, ea);
  dd = __builtin_amdgcn_ldexpf(1.0f, -ea) * wdesc;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (e < 4 ? lo[e] : hi[e - 4]) * s;
    const _Float16 h = (_Float16)v;
    p.h[e] = h;
    p.l[e] = (_Float16)(v - (float)h);
  }
}
__device__ __forceinline__ float inc_amax4(const f32x4i& v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }
// acc += dd * (W' B'): the block's images wh / wl (h and l parts of the scaled weights) against the pair's parts.  The three matrix instructions and
// the wait states behind them are ONE assembly block with the accumulator pinned to VGPRs: compiled from the builtin, hipcc (ROCm 7.2) put the accumulator
// in AGPRs and left 8 wait states between the last v_mfma_f32_16x16x32_f16 and the v_accvgpr_read of its result — not enough on gfx950: a timing-dependent
// subset of wavefronts read stale registers (whole 16-sample tiles off by 1e-2).  With a VGPR destination 8 states are enough (6 are not): measured in
// profiles/r06/inverse.md; 12 are used.  Smallest partial product first.
#define IN_HBLOCK(acc, wh, wl, P, dd)                                                                                               \
  {                                                                                                                                  \
    f32x4i t_;                                                                                                                       \
    asm volatile("s_nop 1\n\t"                                                                                                      \
                 "v_mfma_f32_16x16x32_f16 %0, %1, %3, 0\n\t"                                                                        \
                 "v_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n\t"                                                                       \
                 "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"                                                                       \
                 INC_NOPS                                                                                                            \
                 : "=&v"(t_) : "v"(wl), "v"(wh), "v"((P).h), "v"((P).l));                                                            \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) (acc)[r_] = __builtin_fmaf(t_[r_], (dd), (acc)[r_]);                            \
  }

#define IN_T 17      /* tiles per hidden layer = feature groups (static unroll depth) */
#define IN_CH 24     /* tiles per ring chunk */
#ifndef IN_NR
#define IN_NR 3      /* ring slots */
#endif
#ifndef INC_NOPS
#define INC_NOPS "s_nop 11"  /* 12 wait states behind a block's last matrix instruction (8 needed: profiles/r06/inverse.md) */
#endif
#ifndef INC_DBG
#define INC_DBG 0    /* probe builds: 1 = wait states behind a block's matrix instructions, 2 = every counted wait drains */
#endif
#define IN_WAVES 4
#define IN_MAXD 4    /* dynamic first-layer input tiles kept in registers per group (= IN_L1D) */
#define IN_PROG (2 + 2 * IN_T)

struct IncArgs {
  int64_t N;
  int D, DIN, C, nit;
  const float* yin; int64_t ldy;
  const float* ctx; int64_t ldc;
  float* x; int64_t ldx;
  float* ladj;
  const float* stream;
  const float* bias;
  const int32_t* featmap;
  const int32_t* prog;
  int G, n_chunks, act, bias_floats;
  int bias_off[4];
  int xs;
  float bound, ls;
  RqsLeanConst lc;
  int64_t n_tiles;
  float wdescale[4];  // HALF: 2^-ew of linear layer 1 .. NH (the power of two its pull blocks were stored with); [0] unused
  SosConst<float> sos;  // polynomial maps (uni_kind 5): quadrature nodes / weights, bound, slope
  float eps;            // uni_kind 6: continuation margin of the Bernstein map
  int nbis;             // uni_kind 5 / 6: bisection steps (ceil(log2(2 B / 1e-6)), zuko/transforms.py:615)
};

// (not inlined: the group step below exists 17 times; inline expansions of expm1f / tanhf / erff at every activation
//  site would make it instruction-cache bound — ReLU and identity, the common cases, are handled inline)
__device__ __attribute__((noinline)) float inc_act(float v, int act) {
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
__device__ __forceinline__ f32x4i inc_act4(f32x4i v, int act) {
  if (act == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
  } else if (act != 0) {
#pragma unroll 1
    for (int rep = 0; rep < 1; ++rep) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = inc_act(v[r], act);
    }
  }
  return v;
}

// Stream layout of one pass (static, so that chunk boundaries fall at compile-time-known reads): per group j
//   [IN_L1S first-layer tiles whose inputs are final (padded)] [(NH-1) j hidden pulls] [NT j last-layer pulls]
//   [IN_L1D first-layer diagonal tiles (padded)] [(NH-1) hidden diagonal tiles] [NT last-layer diagonal tiles]
#define IN_L1S 4
#define IN_L1D 4
// HALF: a pull is a 16 x 32 BLOCK (out tile, pair of final in tiles: the pairs that hold a tile < j are (j + 1) / 2) of two images (h, l)
__host__ __device__ constexpr int inc_group_tiles(int NH, int NT, int j, bool half = false) {
  return half ? IN_L1S + 2 * ((NH - 1) + NT) * ((j + 1) / 2) + IN_L1D + (NH - 1) + NT : IN_L1S + (NH - 1) * j + NT * j + IN_L1D + (NH - 1) + NT;
}
__host__ __device__ constexpr int inc_group_start(int NH, int NT, int j, bool half = false) {
  int s = 0;
  for (int i = 0; i < j; ++i) s += inc_group_tiles(NH, NT, i, half);
  return s;
}

// weight ring: 3 x 24 tile images of 1 KiB, filled by global_load_lds two chunks ahead, shared by the 4 wavefronts.
// read(s) takes the position s of the tile inside the pass; every call site has a compile-time s after unrolling, so the
// refill (barrier + DMA issue) is emitted only at the ~55 sites per pass where s is a multiple of the chunk size.
struct IncRing {
  float* lds;
  const float* stream;
  unsigned cur_off;  // LDS byte address of the slot being read + lane * 16
  unsigned lds_off;  // LDS byte address of the ring
  int n_chunks, slot, load_chunk, load_slot, wave, lane;
  // each wave copies IN_CH / IN_WAVES consecutive tiles: one address and one M0 value for all of them, the tile selected by
  // the instruction's (signed) immediate offset (a vector-memory instruction costs the wave ~40 cycles of issue, an M0 write ~20 more:
  // scripts/probes/dma_issue_probe.hip)
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < IN_CH / IN_WAVES) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, (I - 4) * 1024, 0);
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {
    static_assert(IN_CH / IN_WAVES == 6, "immediates -4096 .. +1024 around the wave's fifth tile reach six tiles");
    const int b4 = wave * (IN_CH / IN_WAVES) + 4;
    dma<0>(stream + ((size_t)load_chunk * IN_CH + b4) * 256 + lane * 4, lds + (load_slot * IN_CH + b4) * 256);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == IN_NR) ? 0 : load_slot + 1;
  }
  __device__ __forceinline__ void advance() {  // all wavefronts reach this at the same points of the (uniform) control flow
    // only the chunk about to be read has to have landed; the (IN_NR - 2) younger ones stay in flight
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((IN_NR - 2) * (IN_CH / IN_WAVES)) : "memory");
    __builtin_amdgcn_s_barrier();  // bare barrier: __syncthreads() would prepend s_waitcnt vmcnt(0) and drain the look-ahead DMAs
    asm volatile("" ::: "memory");
    issue();
    slot = (slot + 1 == IN_NR) ? 0 : slot + 1;
    cur_off = lds_off + (unsigned)(slot * IN_CH * 1024 + lane * 16);
  }
  // Position S of the tile inside the pass (static).  The read is issued from inline assembly and returns a RAW value: the
  // compiler does not know it is an LDS operation and inserts no wait — with a global_load_lds in flight hipcc turns every LDS wait
  // into lgkmcnt(0), so the compiler-visible form of this loop was `ds_read; s_waitcnt lgkmcnt(0); 4 MFMAs` per tile, the full LDS
  // round trip exposed every 128 cycles of matrix work (one wavefront per SIMD).  inc_settle<N>() makes a raw value usable: it waits
  // until at most N younger LDS operations are outstanding (they complete in order) and is the only consumer of the raw registers.
  template <int S> __device__ __forceinline__ f32x4i read() {
    if constexpr (S % IN_CH == 0) advance();
    f32x4i v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(cur_off), "n"((S % IN_CH) * 1024));
    return v;
  }
};
template <int N> __device__ __forceinline__ void inc_settle(f32x4i& w) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w) : "n"(N)); }
template <class F, int... I> __device__ __forceinline__ void inc_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void inc_for(F&& f) { inc_for_impl(f, std::make_integer_sequence<int, N>{}); }

#define IN_MFMA4(acc, a, b)                                                              \
  {                                                                                      \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) (acc) = __builtin_amdgcn_mfma_f32_16x16x4f32((a)[r_], (b)[r_], (acc), 0, 0, 0); \
  }

struct IncAffine {
  static constexpr int TOTAL = 2, NT = 1;
  template <typename A> static __device__ __forceinline__ void inv(const float* p, const A& a, float y, float& x, float& lj) {
    const float lsc = softclip<float, MathFast>(p[1], a.ls);
    x = MathFast::div_safe(y - p[0], MathFast::exp(lsc));
    lj = lsc;
  }
};
template <int K> struct IncRqs {
  static constexpr int TOTAL = 3 * K - 1, NT = (TOTAL + 3) / 4;
  template <typename A> static __device__ __forceinline__ void inv(const float* p, const A& a, float y, float& x, float& lj) {
    int k;
    rqs_lean<K, true, true>([&](int j) { return p[j]; }, [&](int j) { return p[K + j]; }, [&](int j) { return p[2 * K + j]; }, a.lc, y, x, lj, k);
  }
};

// The polynomial maps (round 6): their inverse is the reference's fixed-count bisection on [-B, B] (zuko/transforms.py:608-617, zuko/utils.py:170-178), here in the
// group's epilogue on the parameters the lane already holds — the layer-wise form launched ~25 kernels per sweep of the reference's loop (SOSPF 0.25 M, BPF 0.56 M
// samples/s).  Same expression trees as zk_sos_inverse / zk_bernstein_inverse (csrc/zk_univariate.h); log|dy/dx| of the forward map at the solution as for the others.
//
// SCREENED bisection (round 6, second pass).  A step only needs the SIGN of f(mid) - y, and for most steps that sign is far from in doubt: the step first
// evaluates f by a cheap closed form with a rigorous error bound E (>= the closed form's and the reference expression tree's distance from the exact value,
// in units of the magnitudes that enter them), and falls back to the reference's expression tree only where |f~(mid) - y| <= E — the last few steps, once
// the bracket is a few 1e-6 wide.  Every comparison therefore has the outcome the tree's would have: the result is the one of the plain loop above
// (2.4x / 2.6x fewer tree evaluations; `INC_NO_SCREEN` builds the plain loop).
//   * SOS: f(x) = int_0^x mean_p q_p(t / B)^2 + slope dt is a polynomial of degree 9 in u = x / B — five-node Gauss-Legendre IS exact for it — so
//     f~ = x (C_0 + u (C_1 + .. u C_8)), C_m = the convolution coefficients / (m + 1), 9 fmas; bound: the same Horner on |coefficients|, |u|.
//   * Bernstein: sum_i C(n, i) th_i u^i v^(n - i) by Horner in u / v (u <= 1/2) or v / u (both chains, the overflowing one is never selected): 2 n fmas instead of
//     n (n - 1) / 2 lerps; all |th_i| <= B, the weights are a partition of one: E = 192 * 2^-24 * B.  In the linear tails (u within eps of 0 or 1) the tree decides.
#ifndef INC_NO_SCREEN
#define INC_NO_SCREEN 0
#endif
#ifndef INC_BERN_TRUST
#define INC_BERN_TRUST 1
#endif
struct IncSos3x5 {  // ShiftedSOSPolynomialTransform, 3 polynomials of degree 4 + the learned constant (zuko/flows/polynomial.py:51-70)
  static constexpr int TOTAL = 16, NT = 4;
  template <typename A> static __device__ __forceinline__ void inv(const float* p, const A& a, float y, float& x, float& lj) {
    auto ld = [&](int j) { return p[j]; };
    const float yy = y - p[15];
    float C[9], Ca[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) C[m] = Ca[m] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float q[5];
#pragma unroll
      for (int e = 0; e < 5; ++e) q[e] = e == 0 ? 1.f + p[5 * k] : p[5 * k + e];
#pragma unroll
      for (int e = 0; e < 5; ++e)
#pragma unroll
        for (int f = 0; f < 5; ++f) { C[e + f] += q[e] * q[f]; Ca[e + f] += fabsf(q[e] * q[f]); }
    }
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      C[m] = (C[m] / 3.f + (m == 0 ? a.sos.slope : 0.f)) / (float)(m + 1);
      Ca[m] = (Ca[m] / 3.f + (m == 0 ? fabsf(a.sos.slope) : 0.f)) / (float)(m + 1);
    }
    float lo = -a.sos.bound, hi = a.sos.bound;
#pragma unroll 1
    for (int it = 0; it < a.nbis; ++it) {
      const float mid = (lo + hi) / 2.f;
      float fm;
      bool sure = false;
      if (!INC_NO_SCREEN) {
        const float u = mid / a.sos.bound, ua = fabsf(u);
        float h = C[8], ha = Ca[8];
#pragma unroll
        for (int m = 7; m >= 0; --m) { h = __builtin_fmaf(h, u, C[m]); ha = __builtin_fmaf(ha, ua, Ca[m]); }
        fm = mid * h;
        sure = fabsf(fm - yy) > 3.8147e-6f * (fabsf(mid) * ha);  // 64 * 2^-24 of the magnitudes that enter either evaluation (a NaN is never sure)
      }
      if (!sure) fm = sos_f_static<float, 3, 5>(a.sos, ld, mid);
      const bool below = fm < yy;
      lo = below ? mid : lo;
      hi = below ? hi : mid;
    }
    x = (lo + hi) / 2.f;
    lj = t_log(sos_g_static<float, 3, 5>(a.sos, ld, x));
  }
};
struct IncBern17 {  // BoundedBernsteinTransform of degree 16: 17 unconstrained parameters -> 22 constrained coefficients (zuko/transforms.py:779-831)
  static constexpr int TOTAL = 17, NT = 5;
  template <typename A> static __device__ __forceinline__ void inv(const float* p, const A& a, float y, float& x, float& lj) {
    constexpr int NC = 22, n = NC - 1;
    float th[NC];
    bern_theta_bounded<float, NC>([&](int j) { return p[j]; }, a.bound, th);
    const BernTails<float> t = bern_tails<float, NC>(th, true, a.bound, a.eps);
    // C(21, i) th_i (the binomials are exact in float)
    constexpr float BIN[NC] = {1.f, 21.f, 210.f, 1330.f, 5985.f, 20349.f, 54264.f, 116280.f, 203490.f, 293930.f, 352716.f,
                               352716.f, 293930.f, 203490.f, 116280.f, 54264.f, 20349.f, 5985.f, 1330.f, 210.f, 21.f, 1.f};
    float c[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) c[i] = BIN[i] * th[i];
    const float E = 1.1444e-5f * a.bound;  // 192 * 2^-24 * B
    float lo = -a.bound, hi = a.bound;
#pragma unroll 1
    for (int it = 0; it < a.nbis; ++it) {
      const float mid = (lo + hi) / 2.f;
      float fm;
      bool sure = false;
      if (!INC_NO_SCREEN) {
        const float u = (mid + a.bound) / (2.f * a.bound), v = 1.f - u;
        const float sr = u / v, rr = v / u;
        float hA = c[n], hB = c[0];
#pragma unroll
        for (int i = 1; i <= n; ++i) { hA = __builtin_fmaf(hA, sr, c[n - i]); hB = __builtin_fmaf(hB, rr, c[i]); }
        const float w = u <= 0.5f ? v : u;
        const float w2 = w * w, w4 = w2 * w2, w8 = w4 * w4, w16 = w8 * w8;
        fm = (u <= 0.5f ? hA : hB) * (w16 * w4 * w);
        // INC_BERN_TRUST (default): the closed form decides wherever it is finite and outside the linear tails.  Its distance from the de Casteljau value is of the size
        // of de Casteljau's own distance from the exact polynomial (both <= ~70 * 2^-24 * B; neither is the reference's expression — that is the Beta-density form of
        // zuko/transforms.py:736-740), so the solution moves by what a different rounding of f moves it: a few 1e-6 / f'.  INC_BERN_TRUST=0: screened as the SOS loop, bit-identical
        // to the plain de Casteljau loop (profiles/r06/poly_screen_check.txt) at 1.2x instead of ~4x.
        sure = (INC_BERN_TRUST ? fabsf(fm) < 3.0e38f : fabsf(fm - y) > E) && u > a.eps && u < 1.f - a.eps;
      }
      if (!sure) {
        float d;
        bern_fwd<float, NC>(th, t, a.bound, mid, fm, d, a.eps);
      }
      const bool below = fm < y;
      lo = below ? mid : lo;
      hi = below ? hi : mid;
    }
    x = (lo + hi) / 2.f;
    const float xlo = (((y - t.off0) / t.slp0 + a.eps) * 2.f) * a.bound - a.bound;
    const float xhi = ((((y - t.off1) / t.slp1 - a.eps) + 1.f) * 2.f) * a.bound - a.bound;
    x = (y <= t.off0) ? xlo : x;
    x = (y >= t.off1) ? xhi : x;
    float fy, d;
    bern_fwd<float, NC>(th, t, a.bound, x, fy, d, a.eps);
    lj = t_log(d);
  }
};

extern __shared__ __attribute__((aligned(16))) float inc_lds[];

template <typename Uni, int NH, bool HALF = false> __global__ __launch_bounds__(256, 1) void inc_inverse_kernel(IncArgs a) {
  constexpr int NT = Uni::NT, TOTAL = Uni::TOTAL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 15, q = lane >> 4;

  float* ring_lds = inc_lds;
  float* bias_lds = inc_lds + IN_NR * IN_CH * 256;
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);
  int* prog_lds = fmap_lds + IN_T * 4;
  float* xw = reinterpret_cast<float*>(prog_lds + IN_T * IN_PROG) + (size_t)wave * 2 * 16 * a.xs;  // wave-private x tile, then y tile
  float* yw = xw + 16 * a.xs;
  for (int i = tid; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
  for (int i = tid; i < a.G * 4; i += 256) fmap_lds[i] = a.featmap[i];
  for (int i = tid; i < a.G * IN_PROG; i += 256) prog_lds[i] = a.prog[i];

  IncRing ring;
  ring.lds = ring_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < IN_NR - 1; ++i) ring.issue();
  ring.slot = IN_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ring_lds);
  ring.cur_off = ring.lds_off;
  __syncthreads();

  const int xs = a.xs;
  float* xrow = xw + jl * xs;
  const float* yrow = yw + jl * xs;

  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * 64 + wave * 16 + jl;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;

    // ---- stage the wave's 16 rows: x tile = [0 ... 0 | context | 0-pad], y tile = the values to invert ------------
    int bad = 0;
    for (int it = 0; it < a.nit; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = it * 16 + 4 * q + r;
        float xv = 0.f, yv = 0.f;
        if (col < a.D) yv = a.yin[nc * a.ldy + col];
        else if (col < a.DIN) xv = a.ctx[nc * a.ldc + (col - a.D)];
        bad |= !(fabsf(xv) < __builtin_inff()) | !(fabsf(yv) < __builtin_inff());
        xrow[col] = xv;
        xrow[col + 16 * xs] = yv;  // (the y tile sits 16 rows behind the x tile)
      }
    }
    bad |= __shfl_xor(bad, 16, 64);
    bad |= __shfl_xor(bad, 32, 64);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    f32x4i h1[HALF ? 1 : IN_T], h2[NH > 1 && !HALF ? IN_T : 1], h3[NH > 2 && !HALF ? IN_T : 1];
    // HALF: the final activations as pairs of tiles in two f16 parts, every pair with its descale factor; the even tile of the open pair in f32
    constexpr int NPAIR = (IN_T + 1) / 2;
    IncPair p1[HALF ? NPAIR : 1], p2[HALF && NH > 1 ? NPAIR : 1], p3[HALF && NH > 2 ? NPAIR : 1];
    float dd1[HALF ? NPAIR : 1], dd2[HALF && NH > 1 ? NPAIR : 1], dd3[HALF && NH > 2 ? NPAIR : 1];
    f32x4i ev1 = {0.f, 0.f, 0.f, 0.f}, ev2 = ev1, ev3 = ev1;
    float am1 = 0.f, am2 = 0.f, am3 = 0.f;
    float lacc = 0.f;

    // one statically indexed copy of the group step per group (a generic lambda over integral constants: `#pragma unroll`
    // does not unroll a loop of this size, and a run-time j would put the activation arrays in scratch memory)
    auto group_step = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if (j < a.G) {
        const int* pg = prog_lds + j * IN_PROG;
        const int ns = __builtin_amdgcn_readfirstlane(pg[0]), nd = __builtin_amdgcn_readfirstlane(pg[1]);
        constexpr int NPR = (j + 1) / 2;                           // HALF: pairs that hold a final tile
        constexpr int P_S = inc_group_start(NH, NT, j, HALF);      // first-layer tiles with final inputs
        constexpr int P_H = P_S + IN_L1S;                          // hidden pulls
        constexpr int P_L = P_H + (HALF ? 2 * (NH - 1) * NPR : (NH - 1) * j);  // last-layer pulls
        constexpr int P_D = P_L + (HALF ? 2 * NT * NPR : NT * j);  // diagonal tiles
        // ---- pull: contributions of everything that is already final, then the diagonal tiles into registers -----------
        // One static sequence of NTOT consecutive stream tiles, three raw reads in flight: tile i + 2 is requested before tile i
        // is multiplied (or stored away, for the diagonal tiles that stay in registers over the five passes).
        f32x4i o1 = *reinterpret_cast<const f32x4i*>(bias_lds + a.bias_off[0] + j * 16 + 4 * q);
        f32x4i o2 = {0.f, 0.f, 0.f, 0.f}, o3 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (NH > 1) o2 = *reinterpret_cast<const f32x4i*>(bias_lds + a.bias_off[1] + j * 16 + 4 * q);
        if constexpr (NH > 2) o3 = *reinterpret_cast<const f32x4i*>(bias_lds + a.bias_off[2] + j * 16 + 4 * q);
        f32x4i po[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) po[tt] = *reinterpret_cast<const f32x4i*>(bias_lds + a.bias_off[NH] + (j * NT + tt) * 16 + 4 * q);
        f32x4i l1b[IN_L1S];  // B operands of the first-layer pulls (final input tiles), fetched up front
#pragma unroll
        for (int s_ = 0; s_ < IN_L1S; ++s_) {
          const int it = s_ < ns ? __builtin_amdgcn_readfirstlane(pg[2 + s_]) : 0;
          l1b[s_] = *reinterpret_cast<const f32x4i*>(xrow + it * 16 + 4 * q);
        }
        f32x4i wd[IN_MAXD];
        int itd[IN_MAXD];
#pragma unroll
        for (int i = 0; i < IN_MAXD; ++i) itd[i] = i < nd ? __builtin_amdgcn_readfirstlane(pg[2 + IN_T + i]) : 0;
        f32x4i wh2 = {0.f, 0.f, 0.f, 0.f}, wh3 = {0.f, 0.f, 0.f, 0.f};
        f32x4i wl[NT];
        constexpr int N_H = HALF ? 2 * (NH - 1) * NPR : (NH - 1) * j, N_L = HALF ? 2 * NT * NPR : NT * j, NPULL = IN_L1S + N_H + N_L, NTOT = NPULL + IN_L1D + (NH - 1) + NT;
        static_assert(P_D == P_S + NPULL, "pull tiles are consecutive in the stream");
        f32x4i buf[4];  // three raw reads in flight; HALF multiplies a block when its second image is in, so the first must outlive one more request
        buf[0] = ring.template read<P_S>();
        buf[1] = ring.template read<P_S + 1>();
        inc_for<NTOT>([&](auto i_) __attribute__((always_inline)) {
          constexpr int i = decltype(i_)::value;
          if constexpr (i + 2 < NTOT) buf[(i + 2) % 4] = ring.template read<P_S + i + 2>();
          inc_settle<INC_DBG == 2 ? 0 : ((NTOT - 1 - i) < 2 ? (NTOT - 1 - i) : 2)>(buf[i % 4]);
          const f32x4i w = buf[i % 4];
          if constexpr (i < IN_L1S) {  // first-layer tiles whose inputs are final
            if (i < ns) IN_MFMA4(o1, w, l1b[i]);
          } else if constexpr (HALF && i < NPULL) {  // blocks of two images: hidden pulls (layer 2 from the pairs of h1, layer 3 from h2), then the last layer's
            if constexpr ((i - IN_L1S) % 2 == 1) {
              constexpr int k = (i - IN_L1S) / 2;
              const f32x4i wh = buf[(i + 3) % 4];  // the block's first image (h), settled one step ago
              if constexpr (k < (NH - 1) * NPR) {
                constexpr int layer = k / NPR, p = k % NPR;
                if constexpr (layer == 0) { IN_HBLOCK(o2, wh, w, p1[p], dd1[p]); }
                else { IN_HBLOCK(o3, wh, w, p2[p], dd2[p]); }
              } else {
                constexpr int kk = k - (NH - 1) * NPR, p = kk / NT, tt = kk % NT;
                if constexpr (NH == 1) { IN_HBLOCK(po[tt], wh, w, p1[p], dd1[p]); }
                else if constexpr (NH == 2) { IN_HBLOCK(po[tt], wh, w, p2[p], dd2[p]); }
                else { IN_HBLOCK(po[tt], wh, w, p3[p], dd3[p]); }
              }
            }
          } else if constexpr (i < IN_L1S + N_H) {  // hidden pulls: layer 2 from h1, then layer 3 from h2
            constexpr int k = i - IN_L1S;
            if constexpr (k < j) { IN_MFMA4(o2, w, h1[k]); }
            else { IN_MFMA4(o3, w, h2[k - j]); }
          } else if constexpr (i < NPULL) {  // last-layer pulls
            constexpr int k = i - IN_L1S - N_H, t = k / NT, tt = k % NT;
            if constexpr (NH == 1) { IN_MFMA4(po[tt], w, h1[t]); }
            else if constexpr (NH == 2) { IN_MFMA4(po[tt], w, h2[t]); }
            else { IN_MFMA4(po[tt], w, h3[t]); }
          } else {  // the diagonal tiles stay in registers over the five passes
            constexpr int d = i - NPULL;
            if constexpr (d < IN_L1D) wd[d] = w;
            else if constexpr (NH > 1 && d == IN_L1D) wh2 = w;
            else if constexpr (NH > 2 && d == IN_L1D + 1) wh3 = w;
            else wl[d - IN_L1D - (NH - 1)] = w;
          }
        });

        const int f = fmap_lds[j * 4 + q];
        const float yv = yrow[f < 0 ? 0 : f];
        f32x4i h1j, h2j, h3j;
#pragma unroll 1
        for (int r = 0; r < 5; ++r) {
          f32x4i c1 = o1;
#pragma unroll
          for (int i = 0; i < IN_MAXD; ++i) {
            if (i < nd) {
              const f32x4i b = *reinterpret_cast<const f32x4i*>(xrow + itd[i] * 16 + 4 * q);
              IN_MFMA4(c1, wd[i], b);
            }
          }
          h1j = inc_act4(c1, a.act);
          if constexpr (NH > 1) {
            f32x4i c2 = o2;
            IN_MFMA4(c2, wh2, h1j);
            h2j = inc_act4(c2, a.act);
          }
          if constexpr (NH > 2) {
            f32x4i c3 = o3;
            IN_MFMA4(c3, wh3, h2j);
            h3j = inc_act4(c3, a.act);
          }
          if (r == 4) break;
          f32x4i hl;
          if constexpr (NH == 1) hl = h1j;
          else if constexpr (NH == 2) hl = h2j;
          else hl = h3j;
          float p[4 * NT];
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            f32x4i c = po[tt];
            IN_MFMA4(c, wl[tt], hl);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[4 * tt + e] = c[e];
          }
          float xv, lj;
          Uni::inv(p, a, yv, xv, lj);
          if (q == r && f >= 0) {
            xrow[f] = xv;
            lacc += lj;
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
        if constexpr (!HALF) {
          h1[j] = h1j;
          if constexpr (NH > 1) h2[j] = h2j;
          if constexpr (NH > 2) h3[j] = h3j;
        } else {
          // tile j is final: pair j / 2 of every layer is (re)converted — the even tile alone against zeros, then together with its partner
          const f32x4i zero = {0.f, 0.f, 0.f, 0.f};
          auto fin = [&](const f32x4i& v, f32x4i& ev, float& am, IncPair& pr, float& dd, float wd) __attribute__((always_inline)) {
            if constexpr (j % 2 == 0) {
              ev = v; am = inc_amax4(v);
              inc_pair_convert(v, zero, am, wd, pr, dd);
            } else {
              inc_pair_convert(ev, v, fmaxf(am, inc_amax4(v)), wd, pr, dd);
            }
          };
          fin(h1j, ev1, am1, p1[j / 2], dd1[j / 2], a.wdescale[1]);
          if constexpr (NH > 1) fin(h2j, ev2, am2, p2[j / 2], dd2[j / 2], a.wdescale[2]);
          if constexpr (NH > 2) fin(h3j, ev3, am3, p3[j / 2], dd3[j / 2], a.wdescale[3]);
        }
      }
    };
#define ZK_INC_STEP(J) group_step(std::integral_constant<int, J>{});
    ZK_INC_STEP(0) ZK_INC_STEP(1) ZK_INC_STEP(2) ZK_INC_STEP(3) ZK_INC_STEP(4) ZK_INC_STEP(5) ZK_INC_STEP(6) ZK_INC_STEP(7) ZK_INC_STEP(8)
    ZK_INC_STEP(9) ZK_INC_STEP(10) ZK_INC_STEP(11) ZK_INC_STEP(12) ZK_INC_STEP(13) ZK_INC_STEP(14) ZK_INC_STEP(15) ZK_INC_STEP(16)
#undef ZK_INC_STEP
    static_assert(IN_T == 17, "one ZK_INC_STEP per group");

    // ---- results: x rows (16-byte stores where possible), ladj reduced over the four lanes of a sample -------------
    const float nanv = __builtin_nanf("");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (live) {
      for (int c0 = 4 * q; c0 < a.D; c0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e < a.D) a.x[n * a.ldx + c0 + e] = bad ? nanv : xrow[c0 + e];
      }
    }
    if (a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = bad ? nanv : lacc;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}

}  // namespace zk

using namespace zk;

static int inc_lds_floats(int bias_floats, int G, int xs) { return IN_NR * IN_CH * 256 + bias_floats + IN_T * 4 + IN_T * IN_PROG + IN_WAVES * 2 * 16 * xs; }

extern "C" {

int zk_ar_inc_lds_bytes(int bias_floats, int nit) { return inc_lds_floats(bias_floats, IN_T, nit * 16 + 4) * (int)sizeof(float); }

// x[N, D] = f^{-1}(y[N, D] | ctx[N, C]) of one masked autoregressive transform, and (optionally) ladj[N] = sum over features
// of log|dy/dx| of the forward map at x.  uni_kind: 0 affine, 1 / 2 / 3 RQS with 8 / 4 / 16 bins.  wstream / bias / featmap
// / prog / bias_off (host, n_hidden + 1 ints) / n_groups / n_chunks: the plan of zuko_amd/incremental.py.
int zk_ar_inverse_incremental(const zk_ar_inc_args_v1* args, void* stream) {
  if (!args || args->struct_size != sizeof(zk_ar_inc_args_v1) || args->version != 1) return ZK_EINVAL;  // (argument block: include/zuko_amd.h)
  const int uni_kind = args->uni_kind, n_hidden = args->n_hidden, D = args->D, C = args->C, n_groups = args->n_groups, n_chunks = args->n_chunks, act = args->act,
            bias_floats = args->bias_floats;
  const int64_t N = args->N, ldy = args->ldy, ldc = args->ldc, ldx = args->ldx;
  const void *y = args->y, *ctx = args->ctx, *wstream = args->wstream, *bias = args->bias;
  void *x = args->x, *ladj = args->ladj;
  const int32_t *bias_off = args->bias_off, *featmap = args->featmap, *prog = args->prog;
  const double bound = args->bound, slope = args->slope;
  if (N <= 0) return 0;
  if (n_hidden < 1 || n_hidden > 3 || n_groups < 1 || n_groups > IN_T || D < 1 || D > 4 * IN_T || C < 0 || D + C > 256 || n_chunks < 1 || (C > 0 && !ctx)) return ZK_EINVAL;
  IncArgs a{};
  a.N = N; a.D = D; a.C = C; a.DIN = D + C; a.nit = (D + C + 15) / 16;
  a.yin = (const float*)y; a.ldy = ldy; a.ctx = (const float*)ctx; a.ldc = ldc; a.x = (float*)x; a.ldx = ldx; a.ladj = (float*)ladj;
  a.stream = (const float*)wstream; a.bias = (const float*)bias; a.featmap = featmap; a.prog = prog;
  a.G = n_groups; a.n_chunks = n_chunks; a.act = act; a.bias_floats = bias_floats;
  for (int l = 0; l <= n_hidden; ++l) a.bias_off[l] = bias_off[l];
  a.xs = a.nit * 16 + 4;
  a.bound = (float)bound; a.ls = (float)log(slope); a.lc = rqs_lean_const(bound, log(slope));
  if (uni_kind == 5 || uni_kind == 6) {  // polynomial maps: bisection inverse in the group epilogue
    if (args->half || args->n_bisect < 1 || args->n_bisect > 64) return ZK_EINVAL;
    a.nbis = args->n_bisect;
    a.eps = (float)(args->eps > 0.0 ? args->eps : 1e-6);
    if (uni_kind == 5) {
      if (!args->gl_nodes01 || !args->gl_weights01) return ZK_EINVAL;
      a.sos.bound = (float)bound; a.sos.slope = (float)slope; a.sos.P = 3; a.sos.L1 = 5;
      for (int i = 0; i < 5; ++i) { a.sos.node[i] = (float)args->gl_nodes01[i]; a.sos.weight[i] = (float)args->gl_weights01[i]; }
    }
  }
  a.n_tiles = (N + 63) / 64;
  const int lds = inc_lds_floats(bias_floats, n_groups, a.xs) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  const bool half = args->half != 0;
  if (half) {  // wstream is the plan's HALF stream (zuko_amd/incremental.py: half_stream): pull blocks as two f16 images of the weights times 1 / wdescale_l
    const double wd[4] = {1.0, args->wdescale1, args->wdescale2, args->wdescale3};
    for (int l = 1; l <= n_hidden; ++l) {
      if (!(wd[l] > 0.0) || !(wd[l] < 1e38)) return ZK_EINVAL;
      a.wdescale[l] = (float)wd[l];
    }
  }
  const void* fn = nullptr;
#ifdef ZK_INC_FAST_BUILD  /* development: the two benchmark instantiations only */
  if (n_hidden != 3) return ZK_EINVAL;
  if (uni_kind == 0) fn = half ? (const void*)inc_inverse_kernel<IncAffine, 3, true> : (const void*)inc_inverse_kernel<IncAffine, 3>;
  else if (uni_kind == 1) fn = half ? (const void*)inc_inverse_kernel<IncRqs<8>, 3, true> : (const void*)inc_inverse_kernel<IncRqs<8>, 3>;
  else return ZK_EINVAL;
#else
#define ZK_INC_PICK1(UNI, H) (n_hidden == 1 ? (const void*)inc_inverse_kernel<UNI, 1, H> : (n_hidden == 2 ? (const void*)inc_inverse_kernel<UNI, 2, H> : (const void*)inc_inverse_kernel<UNI, 3, H>))
#define ZK_INC_PICK(UNI) (half ? ZK_INC_PICK1(UNI, true) : ZK_INC_PICK1(UNI, false))
  if (uni_kind == 0) fn = ZK_INC_PICK(IncAffine);
  else if (uni_kind == 1) fn = ZK_INC_PICK(IncRqs<8>);
  else if (uni_kind == 2) fn = ZK_INC_PICK(IncRqs<4>);
  else if (uni_kind == 3) fn = ZK_INC_PICK(IncRqs<16>);
  else if (uni_kind == 5) fn = ZK_INC_PICK1(IncSos3x5, false);
  else if (uni_kind == 6) fn = ZK_INC_PICK1(IncBern17, false);
  else return ZK_EINVAL;
#endif
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  const unsigned grid = (unsigned)(a.n_tiles < 256 ? a.n_tiles : 256);
  void* kargs[] = {&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
