// zuko_amd — standalone (phi-in-HBM) kernels for the univariate monotone transforms + C-ABI.
//
// One launch evaluates forward+ladj (or the inverse) of a transform for x[N, D] whose parameters
// live in HBM, exactly as the conditioner's last layer emits them:  phi[N, D, total]  with element
// (n, d)'s `total` values contiguous (zuko/flows/autoregressive.py:149,212-213; utils.py:616-622).
//
// Data movement (the kernel is HBM-bound: 4 + 4*total + 4 (+4/D) bytes per element):
//   * a wavefront owns <= 64 consecutive elements (whole rows when D <= 64, a 64-wide row segment
//     otherwise), so its slice of phi is ONE contiguous byte range;
//   * that range is copied HBM -> LDS with 16-byte loads (alignment of the range is preserved in
//     the LDS image so that every interior load/store is a full dwordx4), x / y move as 4 B per lane
//     fully coalesced;
//   * each lane then pulls its `total` parameters out of LDS at stride `total` words — 23 and 47
//     are odd, so ds_read_b32 is bank-conflict free — and does all arithmetic in registers;
//   * log|det J| is summed over the feature axis inside the wave (segmented shuffle reduction),
//     matching DependentTransform(., 1) (zuko/transforms.py:210-214), or written un-reduced.
// Parameters that are NOT packed (independent tensors, broadcast strides) take the `strided`
// instantiation which reads them straight from global memory.
#include <stdlib.h>

#include "zk_univariate.h"

namespace zk {

struct Seg {
  const void* p;
  int64_t sN, sD;  // element strides of the [N, D, len] view (0 = broadcast)
};

struct UniArgs {
  int64_t N, D;
  const void* x;
  void* y;
  void* ladj;      // may be null (inverse)
  int reduced;     // 1: ladj[N] summed over D, 0: ladj[N, D]
  int32_t* kout;   // optional bin index output [N, D] (RQS only)
  float* knots_out;  // optional (fp32 lean spline only): the K+1 search-axis knots the bin search compared, [N, D, K+1]
  Seg seg[3];
  int total;       // packed values per element
  int64_t iters;   // wave-tile iterations per wave (uniform across the grid)
  int rows_per_wave;
  int segs;        // 64-wide segments per row (1 when D <= 64)
  double bound, ls;
  int n_bisect;
  int bounded;
  double eps;      // Bernstein: width of the linear continuation's margin in [0, 1] coordinates (zuko/transforms.py:594)
  int K;           // runtime bin count for the generic RQS path
  const void* extra;  // optional additive constant [N, D] strides in seg[2] (shifted SOS)
  RqsLeanConst lc;    // fp32 spline constants (rqs_lean)
};

template <typename T> struct Ld {
  const T* p;
  __device__ __forceinline__ T operator()(int j) const { return p[j]; }
};

// ---- ops ---------------------------------------------------------------------------------------

template <typename T, int K, bool INV> struct RqsOp {
  static constexpr int NSEG = 3;
  static __device__ __forceinline__ int off(int s, int) { return s * K; }  // packed offsets 0, K, 2K
  template <typename A> static __device__ __forceinline__ void run(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) {
    typedef typename MathStd<T>::type M;
    T kx[K + 1], ky[K + 1], kd[K + 1];
    if constexpr (sizeof(T) == 4) {  // fp32: the same arithmetic as the stream kernel and the fused epilogue
      rqs_lean<K, INV>(ld[0], ld[1], ld[2], a.lc, in, out, ladj, k, a.knots_out ? a.knots_out + e * (K + 1) : nullptr);
    } else {
      rqs_axis_knots<T, K, M>(ld[0], T(a.bound), T(a.ls), kx);
      rqs_axis_knots<T, K, M>(ld[1], T(a.bound), T(a.ls), ky);
      rqs_slopes<T, K, M>(ld[2], T(a.ls), kd);
      if (INV) { rqs_inv<T, K, M>(kx, ky, kd, in, out, k); ladj = T(0); }
      else rqs_fwd<T, K, M>(kx, ky, kd, in, out, ladj, k);
    }
  }
};

// test entry: the three "parameter" segments are the already-constrained knots
// (horizontal, vertical, slopes), each of length K+1 -> bit-exact bin index on shared knots.
template <typename T, int K, bool INV> struct RqsKnotsOp {
  static constexpr int NSEG = 3;
  static __device__ __forceinline__ int off(int s, int) { return s * (K + 1); }
  template <typename A> static __device__ __forceinline__ void run(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) {
    T kx[K + 1], ky[K + 1], kd[K + 1];
#pragma unroll
    for (int j = 0; j <= K; ++j) { kx[j] = ld[0](j); ky[j] = ld[1](j); kd[j] = ld[2](j); }
    if (INV) { rqs_inv<T, K>(kx, ky, kd, in, out, k); ladj = T(0); }
    else rqs_fwd<T, K>(kx, ky, kd, in, out, ladj, k);
  }
};

// any bin count up to 64: knots live in local arrays addressed at run time (slow path)
#define ZK_RQS_KMAX 64
template <typename T, bool INV> struct RqsGenericOp {
  static constexpr int NSEG = 3;
  static __device__ __forceinline__ int off(int s, int K) { return s * K; }
  template <typename A> static __device__ void run(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) {
    const int K = a.K;
    const T bound = T(a.bound), ls = T(a.ls);
    T kn[2][ZK_RQS_KMAX + 1], kd[ZK_RQS_KMAX + 1];
    for (int ax = 0; ax < 2; ++ax) {
      T m = softclip2<T>(ld[ax](0), ls);
      for (int j = 1; j < K; ++j) { T v = softclip2<T>(ld[ax](j), ls); m = v > m ? v : m; }
      T s = T(0);
      for (int j = 0; j < K; ++j) { T e = t_exp(softclip2<T>(ld[ax](j), ls) - m); kn[ax][j + 1] = e; s += e; }
      T r = T(1) / s, cum = T(0);
      kn[ax][0] = bound * (T(2) * cum - T(1));
      for (int j = 0; j < K; ++j) { cum += kn[ax][j + 1] * r; kn[ax][j + 1] = bound * (T(2) * cum - T(1)); }
    }
    kd[0] = T(1); kd[K] = T(1);
    for (int j = 1; j < K; ++j) kd[j] = t_exp(softclip<T>(ld[2](j - 1), ls));
    const T* srch = INV ? kn[1] : kn[0];
    int cnt = 0;
    for (int j = 0; j <= K; ++j) cnt += (srch[j] < in) ? 1 : 0;
    k = cnt - 1;
    bool inside = (k >= 0) && (k < K);
    int kw = k < 0 ? k + K : (k >= K ? k - K : k);
    T x0 = kn[0][kw], x1 = kn[0][kw + 1], y0 = kn[1][kw], y1 = kn[1][kw + 1], d0 = kd[kw], d1 = kd[kw + 1];
    T m = inside ? T(1) : T(0);
    T s = (y1 - y0) / (x1 - x0);
    T t = (d0 + d1) - T(2) * s;
    if (INV) {
      T y_ = m * (in - y0);
      T qa = (y1 - y0) * (s - d0) + y_ * t;
      T qb = (y1 - y0) * d0 - y_ * t;
      T qc = (-s) * y_;
      T z = (T(2) * qc) / ((-qb) - t_sqrt(qb * qb - (T(4) * qa) * qc));
      T xx = x0 + z * (x1 - x0);
      out = inside ? xx : in;
      ladj = T(0);
    } else {
      T z = (m * (in - x0)) / (x1 - x0);
      T omz = T(1) - z;
      T den = s + (t * z) * omz;
      T num = s * (z * z) + (d0 * z) * omz;
      T yy = y0 + ((y1 - y0) * num) / den;
      T jac = ((s * s) * ((((T(2) * s) * z) * omz + d0 * (omz * omz)) + d1 * (z * z))) / (den * den);
      out = inside ? yy : in;
      ladj = m * t_log(jac);
    }
  }
};

template <typename T, bool INV> struct AffineOp {
  static constexpr int NSEG = 2;
  static __device__ __forceinline__ int off(int s, int) { return s; }  // packed: [shift, scale]
  template <typename A> static __device__ __forceinline__ void run(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) {
    k = 0;
    typedef typename MathStd<T>::type M;
    if (INV) { out = affine_inv<T, M>(ld[0](0), ld[1](0), T(a.ls), in); ladj = T(0); }
    else affine_fwd<T, M>(ld[0](0), ld[1](0), T(a.ls), in, out, ladj);
  }
};

}  // namespace zk

// SOS constants travel in __constant__-like kernel argument (by value)
namespace zk {

template <typename T, bool INV> struct SosOp {
  static constexpr int NSEG = 2;  // seg0 = a[P*L1], seg1 = additive constant (len 1, optional)
  SosConst<T> c;
  int has_const;
  __device__ __forceinline__ int offr(int s) const { return s * c.P * c.L1; }
  template <typename A> __device__ __forceinline__ void runi(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) const {
    k = 0;
    T cst = has_const ? ld[1](0) : T(0);
    if (c.P == 3 && c.L1 == 5) {
      // SOSPF's layout (3 polynomials of degree 4): the 15 coefficients once into registers, the fully unrolled twins of sos_f / sos_g (same expression trees) —
      // the generic path below re-reads every coefficient at each of the quadrature's nodes (and at each of the 25 bisection steps): 0.36 ms per layer at 2^16 x 64
      T cf[15];
#pragma unroll
      for (int j = 0; j < 15; ++j) cf[j] = ld[0](j);
      auto lr = [&](int j) { return cf[j]; };
      if (INV) {
        const T yy = in - cst;
        T lo = -c.bound, hi = c.bound;
        for (int it = 0; it < a.n_bisect; ++it) {
          const T mid = (lo + hi) / T(2);
          const bool below = sos_f_static<T, 3, 5>(c, lr, mid) < yy;
          lo = below ? mid : lo;
          hi = below ? hi : mid;
        }
        out = (lo + hi) / T(2);
        ladj = T(0);
      } else {
        out = sos_f_static<T, 3, 5>(c, lr, in) + cst;
        ladj = t_log(sos_g_static<T, 3, 5>(c, lr, in));
      }
      return;
    }
    if (INV) {
      out = sos_inv<T>(c, ld[0], in - cst, a.n_bisect);
      ladj = T(0);
    } else {
      out = sos_f<T>(c, ld[0], in) + cst;
      ladj = t_log(sos_g<T>(c, ld[0], in));
    }
  }
};

template <typename T, int NC, bool INV> struct BernOp {
  static constexpr int NSEG = 1;
  static __device__ __forceinline__ int off(int, int) { return 0; }
  template <typename A> static __device__ __forceinline__ void run(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) {
    k = 0;
    T th[NC];
    const T bound = T(a.bound);
    if (a.bounded) bern_theta_bounded<T, NC>(ld[0], bound, th);
    else bern_theta_unbounded<T, NC>(ld[0], th);
    const T eps = T(a.eps);
    BernTails<T> t = bern_tails<T, NC>(th, a.bounded != 0, bound, eps);
    if (INV) { out = bern_inv<T, NC>(th, t, bound, in, a.n_bisect, eps); ladj = T(0); }
    else { T d; bern_fwd<T, NC>(th, t, bound, in, out, d, eps); ladj = t_log(d); }
  }
};

// any number of coefficients up to ZK_BERN_NCMAX: same expression trees as bern_* with run-time loops over local
// arrays (slow path; the register-resident instantiations above cover BPF's default degree and the tested ones)
#define ZK_BERN_NCMAX 72
template <typename T, bool INV> struct BernGenericOp {
  static constexpr int NSEG = 1;
  static __device__ __forceinline__ int off(int, int) { return 0; }
  static __device__ void eval(const T* th, int NC, T u, T& val, T& dval) {
    T b[ZK_BERN_NCMAX];
    for (int i = 0; i < NC; ++i) b[i] = th[i];
    const T v = T(1) - u;
    for (int r = 1; r < NC - 1; ++r)
      for (int i = 0; i < NC - r; ++i) b[i] = v * b[i] + u * b[i + 1];
    dval = T(NC - 1) * (b[1] - b[0]);
    val = v * b[0] + u * b[1];
  }
  static __device__ void fwd(const T* th, int NC, const BernTails<T>& t, T bound, T x, T& y, T& dydx, T eps) {
    T u = (x + bound) / (T(2) * bound);
    bool lo = u <= eps;
    bool hi = u >= T(1) - eps;
    T safe = (lo || hi) ? T(0.5) : u;
    T val, dval;
    eval(th, NC, safe, val, dval);
    T ylo = t.slp0 * (u - eps) + t.off0;
    T yhi = t.slp1 * ((u - T(1)) + eps) + t.off1;
    y = lo ? ylo : val;
    y = hi ? yhi : y;
    T du = lo ? t.slp0 : dval;
    du = hi ? t.slp1 : du;
    dydx = du / (T(2) * bound);
  }
  template <typename A> static __device__ void run(const A& a, const Ld<T> (&ld)[3], T in, T& out, T& ladj, int& k, int64_t e) {
    k = 0;
    const int NC = a.K;  // constrained coefficients
    const T bound = T(a.bound), eps = T(a.eps);
    T th[ZK_BERN_NCMAX];
    if (a.bounded) {  // bern_theta_bounded
      const int n = NC - 5;
      const T edge = (T(2) * bound) / T(n + 4);
      const T span = T(2) * bound - T(4) * edge;
      T m = ld[0](0);
      for (int j = 1; j < n; ++j) { T v = ld[0](j); m = v > m ? v : m; }
      T s = T(0);
      for (int j = 0; j < n; ++j) { T ev = t_exp(ld[0](j) - m); th[3 + j] = ev; s += ev; }
      T r = T(1) / s;
      T cum = -bound;
      th[0] = cum;
      cum += edge; th[1] = cum;
      cum += edge; th[2] = cum;
      for (int j = 0; j < n; ++j) { cum += (th[3 + j] * r) * span; th[3 + j] = cum; }
      cum += edge; th[n + 3] = cum;
      cum += edge; th[n + 4] = cum;
    } else {  // bern_theta_unbounded
      const int n = NC - 2;
      const T shift = T(0.69314718055994530942 * n / 2.0);
      T cum = ld[0](0);
      th[0] = cum - shift;
      cum += softplus<T>(ld[0](1));
      th[1] = cum - shift;
      for (int j = 1; j < n; ++j) { cum += softplus<T>(ld[0](j)); th[j + 1] = cum - shift; }
      cum += softplus<T>(ld[0](n - 1));
      th[n + 1] = cum - shift;
    }
    BernTails<T> t;
    if (a.bounded) { t.off0 = -bound; t.off1 = bound; t.slp0 = T(2) * bound; t.slp1 = T(2) * bound; }
    else { eval(th, NC, eps, t.off0, t.slp0); eval(th, NC, T(1) - eps, t.off1, t.slp1); }
    if (INV) {
      T lo = -bound, hi = bound;
      for (int it = 0; it < a.n_bisect; ++it) {
        T mid = (lo + hi) / T(2);
        T fy, d;
        fwd(th, NC, t, bound, mid, fy, d, eps);
        bool below = fy < in;
        lo = below ? mid : lo;
        hi = below ? hi : mid;
      }
      T x = (lo + hi) / T(2);
      T xlo = (((in - t.off0) / t.slp0 + eps) * T(2)) * bound - bound;
      T xhi = ((((in - t.off1) / t.slp1 - eps) + T(1)) * T(2)) * bound - bound;
      x = (in <= t.off0) ? xlo : x;
      x = (in >= t.off1) ? xhi : x;
      out = x;
      ladj = T(0);
    } else {
      T d;
      fwd(th, NC, t, bound, in, out, d, eps);
      ladj = t_log(d);
    }
  }
};

// ---- the streaming driver ------------------------------------------------------------------------

// copy `count` consecutive T's starting at global `src` into LDS `dst` such that dst keeps src's
// alignment modulo 16 bytes; returns nothing, caller adds `shift` = ((uintptr_t)src / sizeof(T)) % VEC.
template <typename T> __device__ __forceinline__ void stage_contiguous(const T* src, T* dst_aligned, int64_t count, int lane) {
  constexpr int VEC = 16 / sizeof(T);
  const int shift = (int)(((uintptr_t)src / sizeof(T)) % VEC);
  int head = shift ? (VEC - shift) : 0;
  if (head > count) head = (int)count;
  if (lane < head) dst_aligned[shift + lane] = src[lane];
  const int64_t body = (count - head) / VEC;
  const float4* vsrc = reinterpret_cast<const float4*>(src + head);
  float4* vdst = reinterpret_cast<float4*>(dst_aligned + shift + head);
  for (int64_t i = lane; i < body; i += ZK_WAVE) vdst[i] = vsrc[i];
  const int64_t done = head + body * VEC;
  const int tail = (int)(count - done);
  if (lane < tail) dst_aligned[shift + done + lane] = src[done + lane];
}

template <typename T, bool PACKED, typename Op, typename RunFn>
__device__ __forceinline__ void uni_driver(const UniArgs& a, T* sh, RunFn run) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t D = a.D;
  const int RW = a.rows_per_wave;
  const int total = a.total;
  constexpr int VEC = 16 / sizeof(T);
  T* wsh = PACKED ? sh + (size_t)wave * (((size_t)64 * total + 2 * VEC + VEC - 1) / VEC * VEC) : nullptr;
  const T* xg = (const T*)a.x;
  T* yg = (T*)a.y;
  T* lg = (T*)a.ladj;

  // Each wavefront owns a CONTIGUOUS run of `iters` wave tiles: its phi stream is one long sequential
  // read, and (one row per tile, D == 64) the per-row ladj sums are parked in lane (it % 64) and written
  // back 64 rows at a time as one coalesced 256-byte store instead of 64 single-lane stores.
  const int64_t wt0 = ((int64_t)blockIdx.x * 4 + wave) * a.iters;
  const bool park = lg && a.reduced && a.segs == 1 && sizeof(T) == 4 && D == 64;
  float parked = 0.f;
  for (int64_t it = 0; it < a.iters; ++it) {
    const int64_t wt = wt0 + it;  // wave tile
    const int64_t row0 = wt * RW;
    T lacc = T(0);
    for (int sgm = 0; sgm < a.segs; ++sgm) {
      int r, d;
      if (a.segs == 1) { r = lane / (int)D; d = lane - r * (int)D; }
      else { r = 0; d = sgm * 64 + lane; }
      const int64_t row = row0 + r;
      const bool valid = (r < RW) && (row < a.N) && (d < D);
      const int64_t e = row * D + d;
      int shift = 0;
      if (PACKED) {
        // contiguous element range owned by this wave in this step
        const int64_t e0 = row0 * D + (int64_t)sgm * 64;
        int64_t cnt;
        if (a.segs == 1) { int64_t rows = a.N - row0; rows = rows < 0 ? 0 : (rows > RW ? RW : rows); cnt = rows * D; }
        else { cnt = (row0 < a.N) ? (D - (int64_t)sgm * 64) : 0; cnt = cnt > 64 ? 64 : cnt; }
        const T* src = (const T*)a.seg[0].p + e0 * total;
        shift = (int)(((uintptr_t)src / sizeof(T)) % VEC);
        // The LDS image is private to this wavefront and a wave's DS operations execute in order, so
        // no workgroup barrier is needed: fence the compiler and drain the staging stores only.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous step's reads have returned
        __builtin_amdgcn_wave_barrier();
        if (cnt > 0) stage_contiguous<T>(src, wsh, cnt * total, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      T out = T(0), lj = T(0);
      int k = 0;
      if (valid) {
        Ld<T> ld[3];
        if (PACKED) {
          const int le = (a.segs == 1) ? lane : lane;  // element index inside the staged range
          const T* base = wsh + shift + (size_t)le * total;
          ld[0].p = base + Op::off(0, a.K);
          ld[1].p = base + Op::off(1, a.K);
          ld[2].p = base + Op::off(2, a.K);
        } else {
#pragma unroll
          for (int s = 0; s < 3; ++s) ld[s].p = (const T*)a.seg[s].p + row * a.seg[s].sN + d * a.seg[s].sD;
        }
        run(ld, xg[e], out, lj, k, e);
        yg[e] = out;
        if (lg && !a.reduced) lg[e] = lj;
        if (a.kout) a.kout[e] = k;
      }
      if (lg && a.reduced) {
        if (a.segs == 1) {
          T v = valid ? lj : T(0);
          if (park) {  // one row per tile: DPP reduction (result in lane 63), parked in lane it % 64
            const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_dpp_to_lane63((float)v)), 63));
            const int slot = (int)(it & 63);
            if (lane == slot) parked = tot;
            if (slot == 63 || it + 1 == a.iters) {
              const int64_t rbase = row0 - slot;  // rows are consecutive over `it` (RW == 1)
              if (lane <= slot && rbase + lane < a.N) lg[rbase + lane] = (T)parked;
            }
          } else {
            v = segment_sum<T>(v, (int)D, d);
            if (valid && d == 0) lg[row] = v;
          }
        } else {
          lacc += valid ? lj : T(0);
        }
      }
    }
    if (lg && a.reduced && a.segs > 1) {
      T v = wave_sum<T>(lacc);
      if (lane == 0 && row0 < a.N) lg[row0] = v;
    }
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];

template <typename T, bool PACKED, typename Op> __global__ __launch_bounds__(256) void uni_kernel(UniArgs a) {
  T* sh = reinterpret_cast<T*>(zk_dyn_lds);
  uni_driver<T, PACKED, Op>(a, sh, [&](const Ld<T>(&ld)[3], T in, T& out, T& lj, int& k, int64_t e) { Op::run(a, ld, in, out, lj, k, e); });
}

template <typename T, bool PACKED, bool INV> __global__ __launch_bounds__(256) void sos_kernel(UniArgs a, SosOp<T, INV> op) {
  T* sh = reinterpret_cast<T*>(zk_dyn_lds);
  struct OffT {
    static __device__ __forceinline__ int off(int s, int K) { return s * K; }  // K carries P*L1 here
  };
  uni_driver<T, PACKED, OffT>(a, sh, [&](const Ld<T>(&ld)[3], T in, T& out, T& lj, int& k, int64_t e) { op.runi(a, ld, in, out, lj, k, e); });
}

// ---- host side ------------------------------------------------------------------------------------

struct Plan {
  dim3 grid;
  size_t lds;
};

template <typename T> static Plan plan(UniArgs& a, bool packed) {
  const int64_t D = a.D;
  if (D <= 64) { a.rows_per_wave = (int)(64 / D); a.segs = 1; }
  else { a.rows_per_wave = 1; a.segs = (int)((D + 63) / 64); }
  const int64_t wave_tiles = (a.N + a.rows_per_wave - 1) / a.rows_per_wave;
  const int64_t blocks = (wave_tiles + 3) / 4;
  Plan p;
  constexpr int VEC = 16 / sizeof(T);
  const size_t per_wave = (((size_t)64 * a.total + 2 * VEC + VEC - 1) / VEC * VEC);
  p.lds = packed ? 4 * per_wave * sizeof(T) : 0;
  // persistent grid = exactly what is co-resident (256 CUs x blocks/CU by LDS and wave slots), so the
  // grid-stride loop has no partially filled second round
  int64_t per_cu = p.lds ? (int64_t)(160 * 1024) / (int64_t)p.lds : 8;
  per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
  const int64_t cap = 256 * per_cu;
  p.grid = dim3((unsigned)(blocks < cap ? blocks : cap));
  a.iters = (blocks + p.grid.x - 1) / p.grid.x;
  return p;
}

// is the parameter set one packed phi[N, D, total] buffer?
static bool is_packed(const UniArgs& a, const int* lens, int nseg, size_t esz) {
  int64_t total = 0;
  for (int s = 0; s < nseg; ++s) total += lens[s];
  const char* base = (const char*)a.seg[0].p;
  int64_t offs = 0;
  for (int s = 0; s < nseg; ++s) {
    if (a.seg[s].sD != total || a.seg[s].sN != total * a.D) return false;
    if ((const char*)a.seg[s].p != base + offs * (int64_t)esz) return false;
    offs += lens[s];
  }
  return total * 64 * esz * 4 <= 64 * 1024;  // keep the LDS image <= 64 KiB per block
}

template <typename T, typename OpP, typename OpS> static int launch_uni(UniArgs a, const int* lens, int nseg, hipStream_t st) {
  if (a.N <= 0 || a.D <= 0) return 0;
  int total = 0;
  for (int s = 0; s < nseg; ++s) total += lens[s];
  a.total = total;
  const bool packed = is_packed(a, lens, nseg, sizeof(T));
  Plan p = plan<T>(a, packed);
  if (packed) hipLaunchKernelGGL((uni_kernel<T, true, OpP>), p.grid, dim3(256), p.lds, st, a);
  else hipLaunchKernelGGL((uni_kernel<T, false, OpS>), p.grid, dim3(256), 0, st, a);
  return ZK_LAUNCH_CHECK();
}

template <typename T, bool INV> static int launch_rqs(UniArgs a, int K, hipStream_t st) {
  const int lens[3] = {K, K, K - 1};
  a.K = K;
  switch (K) {
    case 4: return launch_uni<T, RqsOp<T, 4, INV>, RqsOp<T, 4, INV>>(a, lens, 3, st);
    case 8: return launch_uni<T, RqsOp<T, 8, INV>, RqsOp<T, 8, INV>>(a, lens, 3, st);
    case 16: return launch_uni<T, RqsOp<T, 16, INV>, RqsOp<T, 16, INV>>(a, lens, 3, st);
    default:
      if (K < 2 || K > ZK_RQS_KMAX) return ZK_EINVAL;
      return launch_uni<T, RqsGenericOp<T, INV>, RqsGenericOp<T, INV>>(a, lens, 3, st);
  }
}

template <typename T, bool INV> static int launch_rqs_knots(UniArgs a, int K, hipStream_t st) {
  const int lens[3] = {K + 1, K + 1, K + 1};
  a.K = K;
  switch (K) {
    case 4: return launch_uni<T, RqsKnotsOp<T, 4, INV>, RqsKnotsOp<T, 4, INV>>(a, lens, 3, st);
    case 8: return launch_uni<T, RqsKnotsOp<T, 8, INV>, RqsKnotsOp<T, 8, INV>>(a, lens, 3, st);
    case 16: return launch_uni<T, RqsKnotsOp<T, 16, INV>, RqsKnotsOp<T, 16, INV>>(a, lens, 3, st);
    default: return ZK_EINVAL;
  }
}

template <typename T, int NC, bool INV> static int launch_bern_nc(UniArgs a, hipStream_t st) {
  const int lens[1] = {a.bounded ? NC - 5 : NC - 2};
  return launch_uni<T, BernOp<T, NC, INV>, BernOp<T, NC, INV>>(a, lens, 1, st);
}

template <typename T, bool INV> static int launch_bern(UniArgs a, int M, hipStream_t st) {
  const int NC = a.bounded ? M + 5 : M + 2;
  switch (NC) {
    case 6: return launch_bern_nc<T, 6, INV>(a, st);
    case 8: return launch_bern_nc<T, 8, INV>(a, st);
    case 10: return launch_bern_nc<T, 10, INV>(a, st);
    case 13: return launch_bern_nc<T, 13, INV>(a, st);
    case 14: return launch_bern_nc<T, 14, INV>(a, st);
    case 18: return launch_bern_nc<T, 18, INV>(a, st);
    case 21: return launch_bern_nc<T, 21, INV>(a, st);
    case 22: return launch_bern_nc<T, 22, INV>(a, st);
    case 34: return launch_bern_nc<T, 34, INV>(a, st);
    case 37: return launch_bern_nc<T, 37, INV>(a, st);
    default: {
      if (NC < 3 || NC > ZK_BERN_NCMAX || (a.bounded ? M < 1 : M < 2)) return ZK_EINVAL;
      const int lens[1] = {M};
      a.K = NC;
      return launch_uni<T, BernGenericOp<T, INV>, BernGenericOp<T, INV>>(a, lens, 1, st);
    }
  }
}

template <typename T, bool INV>
static int launch_sos(UniArgs a, int P, int L1, double slope, const double* nodes01, const double* weights01, int has_const, hipStream_t st) {
  if (a.N <= 0 || a.D <= 0) return 0;
  if (L1 < 1 || L1 > ZK_SOS_MAX_NODES || P < 1) return ZK_EINVAL;
  SosOp<T, INV> op;
  op.c.bound = T(10.0);
  op.c.slope = T(slope);
  op.c.P = P;
  op.c.L1 = L1;
  for (int i = 0; i < L1; ++i) { op.c.node[i] = T(nodes01[i]); op.c.weight[i] = T(weights01[i]); }
  op.has_const = has_const;
  const int lens[2] = {P * L1, 1};
  const int nseg = has_const ? 2 : 1;
  a.total = P * L1 + (has_const ? 1 : 0);
  a.K = P * L1;  // packed offset of the constant
  const bool packed = is_packed(a, lens, nseg, sizeof(T));
  Plan p = plan<T>(a, packed);
  if (packed) hipLaunchKernelGGL((sos_kernel<T, true, INV>), p.grid, dim3(256), p.lds, st, a, op);
  else hipLaunchKernelGGL((sos_kernel<T, false, INV>), p.grid, dim3(256), 0, st, a, op);
  return ZK_LAUNCH_CHECK();
}

// ---- the fp32 spline stream (K1 of SURVEY 2.1 at its benchmark shape) ---------------------------------
// phi[N, D, 3K-1] packed and 16-byte aligned, N*D a multiple of 64 and D | 64 or 64 | D.  A wavefront walks
// a contiguous run of 64-element tiles.  While tile t is evaluated out of the wave's private LDS image,
// tile t+1's 64 x (3K-1) floats are already in flight as dwordx4 loads into registers (and its x values
// with them); they are written to the LDS image once tile t's parameters have been read out.  No
// workgroup barrier anywhere: a wave's LDS operations execute in order.  Arithmetic: rqs_lean.
struct RqsStreamArgs {
  const void* x;     // float, or __bf16 in the BF instantiations (ladj stays float)
  const void* phi;
  void* y;
  float* ladj;       // null: none
  int64_t tiles;     // N * D / 64
  int64_t iters;     // tiles per wave (a multiple of D / 64 when D > 64)
  int cs;            // >= 0: tiles are dealt to the waves in chunks of 2^cs consecutive tiles, round-robin; -1: one contiguous run per wave
  int64_t N;
  int D;
  int reduced;
  RqsLeanConst c;
};

#ifndef ZK_NT
#define ZK_NT NT  /* template parameter: phi is streamed once, non-temporal loads keep it from displacing x / y lines in L2 */
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint32x4_t __attribute__((ext_vector_type(4)));
// LM: 0 no ladj, 1 ladj[N, D], 2 ladj[N] with D == 64 (one row per tile), 3 ladj[N] for any other admissible D
// BF: x, phi and y are bf16 in HBM (cfg5); the LDS image, the arithmetic and ladj stay fp32
// NT: phi through non-temporal loads (default; ZUKO_AMD_K1_NT=0 selects the plain-load instantiation of the fp32 kernels —
// the two are timed side by side by bench.py, the faster one differs between boxes by a few percent)
template <int K, bool INV, int LM, bool BF, bool NT = true> __global__ __launch_bounds__(256) void rqs_stream_kernel(RqsStreamArgs a) {
  constexpr int TOTAL = 3 * K - 1;
  constexpr int TILE_V = (BF ? 8 : 16) * TOTAL;  // 16-byte vectors per tile
  constexpr int NV = (TILE_V + 63) / 64;         // dwordx4 loads per lane per tile
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* img = reinterpret_cast<float*>(zk_dyn_lds) + (size_t)wave * 64 * TOTAL;
  f32x4_t* img4 = reinterpret_cast<f32x4_t*>(img);
  const float* my = img + lane * TOTAL;
  // k-th tile of this wave (increasing in k, so the first tile past the end terminates the wave)
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave, W = (int64_t)gridDim.x * 4;
  const int cs = a.cs;
  const int64_t cmask = cs >= 0 ? ((int64_t)1 << cs) - 1 : 0;
  auto tile_of = [&](int64_t k) -> int64_t { return cs < 0 ? wid * a.iters + k : ((((k >> cs) * W + wid) << cs) | (k & cmask)); };
  int64_t t = tile_of(0);
  if (t >= a.tiles) return;
  const f32x4_t* phi4 = reinterpret_cast<const f32x4_t*>(a.phi);

  f32x4_t nxt[NV];
  float xn;
  // (the last dwordx4 round covers only part of the wave: out-of-range lanes re-read the tile's last vector)
#define ZK_FETCH(t)                                                         \
  {                                                                         \
    const f32x4_t* src = phi4 + (t) * TILE_V;                                \
    _Pragma("unroll") for (int r = 0; r < NV; ++r) {                        \
      const int i = r * 64 + lane;                                          \
      nxt[r] = ZK_NT ? __builtin_nontemporal_load(&src[((r + 1) * 64 <= TILE_V || i < TILE_V) ? i : TILE_V - 1]) : src[((r + 1) * 64 <= TILE_V || i < TILE_V) ? i : TILE_V - 1]; \
    }                                                                       \
    xn = BF ? (float)((const __bf16*)a.x)[(t) * 64 + lane] : ((const float*)a.x)[(t) * 64 + lane]; \
  }
#define ZK_STASH()                                                          \
  {                                                                         \
    _Pragma("unroll") for (int r = 0; r < NV; ++r) {                        \
      const int i = r * 64 + lane;                                          \
      if ((r + 1) * 64 <= TILE_V || i < TILE_V) {                           \
        if (BF) {  /* 8 bf16 -> 8 floats: a bf16 is the high half of its float */ \
          const uint32x4_t u = __builtin_bit_cast(uint32x4_t, nxt[r]);      \
          const uint32x4_t lo = {u.x << 16, u.x & 0xffff0000u, u.y << 16, u.y & 0xffff0000u}; \
          const uint32x4_t hi = {u.z << 16, u.z & 0xffff0000u, u.w << 16, u.w & 0xffff0000u}; \
          img4[2 * i] = __builtin_bit_cast(f32x4_t, lo);                    \
          img4[2 * i + 1] = __builtin_bit_cast(f32x4_t, hi);                \
        } else {                                                            \
          img4[i] = nxt[r];                                                 \
        }                                                                   \
      }                                                                     \
    }                                                                       \
  }
  ZK_FETCH(t);
  ZK_STASH();

  const int D = a.D;
  const int per_row = D >> 6;                       // tiles per row when D >= 64
  float parked = 0.f, lacc = 0.f;
  int row_left = per_row;                           // (a wave's run starts on a row boundary)
  int slot = 0;                                     // LM == 2: parked row sums so far (rows t - slot .. t are consecutive)
  for (int64_t k = 0; k < a.iters; ++k) {
    const float xc = xn;
    const int64_t tn = tile_of(k + 1);
    const bool more = (k + 1 < a.iters) && (tn < a.tiles);   // wave-uniform
    if (more) ZK_FETCH(tn);
    float p[TOTAL];
#pragma unroll
    for (int j = 0; j < TOTAL; ++j) p[j] = my[j];
    float out, lj;
    rqs_lean<K, INV>([&](int j) { return p[j]; }, [&](int j) { return p[K + j]; }, [&](int j) { return p[2 * K + j]; }, a.c, xc, out, lj);
    const int64_t e = t * 64 + lane;
    if (BF) ((__bf16*)a.y)[e] = (__bf16)out;
    else ((float*)a.y)[e] = out;
    if (LM == 1) {
      a.ladj[e] = lj;
    } else if (LM == 2) {  // DPP reduction; 64 row sums are parked in lanes and stored as one 256-byte line
      const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_dpp_to_lane63(lj)), 63));
      if (lane == slot) parked = tot;
      if (slot == 63 || !more || tn != t + 1) {
        if (lane <= slot) a.ladj[t - slot + lane] = parked;
        slot = 0;
      } else {
        ++slot;
      }
    } else if (LM == 3) {
      if (D < 64) {
        const int sh = __builtin_ctz(D);            // D divides 64: a power of two
        const float v = segment_sum<float>(lj, D, lane & (D - 1));
        if ((lane & (D - 1)) == 0) a.ladj[e >> sh] = v;
      } else {
        lacc += lj;
        if (--row_left == 0) {
          const float v = wave_sum<float>(lacc);
          if (lane == 0) a.ladj[(t + 1) / per_row - 1] = v;
          lacc = 0.f;
          row_left = per_row;
        }
      }
    }
    if (!more) break;
    ZK_STASH();
    t = tn;
  }
#undef ZK_FETCH
#undef ZK_STASH
}

// Tiles are dealt to the waves round-robin in chunks of 2^cs consecutive tiles, so that at any moment the
// resident waves sweep ONE compact window of phi (measured at N = 2^20, D = 64, K = 8: 5.6 TB/s against
// 4.95 TB/s for one long run per wave).  Feature-reduced ladj wants 64 consecutive rows per wave (their
// sums leave as one 256-byte store); the other modes are fastest with single tiles.
// ZUKO_AMD_K1_CHUNK overrides cs (-1 = contiguous runs).
static int zk_stream_chunk_shift(int lm) {
  const char* e = getenv("ZUKO_AMD_K1_CHUNK");  // (read per call: the tests switch it)
  return e ? atoi(e) : (lm == 2 ? 6 : 0);
}

static bool zk_stream_nt() {
  const char* e = getenv("ZUKO_AMD_K1_NT");  // (read per call: bench.py times both)
  return !(e && e[0] == '0');
}

template <int K, bool INV, int LM, bool BF = false, bool NT = true> static int launch_rqs_stream_k(RqsStreamArgs a, hipStream_t st) {
  if constexpr (NT && !BF) {
    if (!zk_stream_nt()) return launch_rqs_stream_k<K, INV, LM, BF, false>(a, st);
  }
  constexpr int TOTAL = 3 * K - 1;
  const size_t lds = (size_t)4 * 64 * TOTAL * sizeof(float);
  static int per_cu = 0;  // resident blocks per CU for this instantiation (device query, once)
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rqs_stream_kernel<K, INV, LM, BF, NT>, 256, lds) != hipSuccess || nb < 1) nb = 4;
    per_cu = nb;
  }
  const int64_t per_row = a.D > 64 ? a.D / 64 : 1;
  const int64_t waves_wanted = (a.tiles + per_row - 1) / per_row;     // a wave needs at least one whole row
  int64_t blocks = (waves_wanted + 3) / 4;
  const int64_t cap = (int64_t)256 * per_cu;
  if (blocks > cap) blocks = cap;
  const int64_t W = blocks * 4;
  a.cs = (LM == 3) ? -1 : zk_stream_chunk_shift(LM);
  while (a.cs > 0 && (a.tiles >> a.cs) < W) --a.cs;                   // small inputs: keep every wave busy
  if (a.cs >= 0) {
    const int64_t chunks = (a.tiles + ((int64_t)1 << a.cs) - 1) >> a.cs;
    a.iters = ((chunks + W - 1) / W) << a.cs;
  } else {
    int64_t iters = (a.tiles + W - 1) / W;
    a.iters = (iters + per_row - 1) / per_row * per_row;
  }
  hipLaunchKernelGGL((rqs_stream_kernel<K, INV, LM, BF, NT>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  return ZK_LAUNCH_CHECK();
}
template <int K, bool BF = false> static int launch_rqs_stream_fwd(const RqsStreamArgs& a, hipStream_t st) {
  if (!a.ladj) return launch_rqs_stream_k<K, false, 0, BF>(a, st);
  if (!a.reduced) return launch_rqs_stream_k<K, false, 1, BF>(a, st);
  return a.D == 64 ? launch_rqs_stream_k<K, false, 2, BF>(a, st) : launch_rqs_stream_k<K, false, 3, BF>(a, st);
}

// returns -1 when the call does not have the stream kernel's shape (caller falls back to uni_kernel)
template <bool INV, bool BF = false> static int try_rqs_stream(const UniArgs& u, int K, hipStream_t st) {
  if (K != 4 && K != 8 && K != 16) return -1;
  if (u.kout || u.N <= 0 || u.D <= 0) return -1;
  const int64_t D = u.D, E = u.N * D;
  if (E % 64 != 0 || !((D <= 64 && 64 % D == 0) || D % 64 == 0)) return -1;
  const int lens[3] = {K, K, K - 1};
  if (!is_packed(u, lens, 3, BF ? 2 : sizeof(float))) return -1;
  if (((uintptr_t)u.seg[0].p & 15) != 0) return -1;
  RqsStreamArgs a;
  a.x = u.x; a.phi = u.seg[0].p; a.y = u.y; a.ladj = (float*)u.ladj;
  a.tiles = E / 64; a.iters = 0; a.N = u.N; a.D = (int)D; a.reduced = u.reduced;
  a.c = u.lc;
  if (INV) {
    switch (K) {
      case 4: return launch_rqs_stream_k<4, true, 0, BF>(a, st);
      case 8: return launch_rqs_stream_k<8, true, 0, BF>(a, st);
      default: return launch_rqs_stream_k<16, true, 0, BF>(a, st);
    }
  }
  switch (K) {
    case 4: return launch_rqs_stream_fwd<4, BF>(a, st);
    case 8: return launch_rqs_stream_fwd<8, BF>(a, st);
    default: return launch_rqs_stream_fwd<16, BF>(a, st);
  }
}

// ---- base density: Independent(Normal(loc, scale), 1).log_prob(z) + ladj --------------------------
// (zuko/distributions.py:115-119, 337-363; torch/distributions/normal.py log_prob)
template <typename T> __global__ __launch_bounds__(256) void normal_kernel(int64_t N, int64_t D, const T* z, const T* loc, const T* scale, const T* ladj, T* out, int64_t iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T half_log_2pi = T(0.91893853320467274178);  // log(sqrt(2*pi))
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t row = (it * gridDim.x + blockIdx.x) * 4 + wave;
    if (row >= N) continue;  // wave-uniform
    T acc = T(0);
    for (int64_t d = lane; d < D; d += 64) {
      T s = scale[d];
      T df = z[row * D + d] - loc[d];
      acc += (-(df * df)) / (T(2) * (s * s)) - t_log(s) - half_log_2pi;
    }
    acc = wave_sum<T>(acc);
    if (lane == 0) out[row] = ladj ? acc + ladj[row] : acc;
  }
}

template <typename T> static int launch_normal(int64_t N, int64_t D, const void* z, const void* loc, const void* scale, const void* ladj, void* out, hipStream_t st) {
  if (N <= 0) return 0;
  const int64_t blocks = (N + 3) / 4;
  dim3 grid((unsigned)grid_for(blocks));
  const int64_t iters = (blocks + grid.x - 1) / grid.x;
  hipLaunchKernelGGL((normal_kernel<T>), grid, dim3(256), 0, st, N, D, (const T*)z, (const T*)loc, (const T*)scale, (const T*)ladj, (T*)out, iters);
  return ZK_LAUNCH_CHECK();
}

// ---- sum of a vector into a double accumulator (per-rank partial NLL; K9 / C1 of SURVEY 2.1) ------
template <typename T> __global__ __launch_bounds__(256) void sum_kernel(int64_t N, const T* v, double* partial) {
  __shared__ double ws[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) acc += (double)v[i];
  acc = wave_sum<double>(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(256) void sum_final_kernel(int n, const double* partial, double* out, double scale) {
  __shared__ double ws[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = wave_sum<double>(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((ws[0] + ws[1]) + (ws[2] + ws[3])) * scale;
}

}  // namespace zk

// =====================================================================================================
// C-ABI (declared and documented in include/zuko_amd.h)
// =====================================================================================================
using namespace zk;

static UniArgs base_args(int64_t N, int64_t D, const void* x, void* y, void* ladj, int reduced, int32_t* kout) {
  UniArgs a{};
  a.N = N; a.D = D; a.x = x; a.y = y; a.ladj = ladj; a.reduced = reduced; a.kout = kout;
  return a;
}

#define ZK_DISPATCH(dtype, CALL_F32, CALL_F64) \
  ((dtype) == ZK_DTYPE_F32 ? (CALL_F32) : ((dtype) == ZK_DTYPE_F64 ? (CALL_F64) : ZK_EINVAL))

// ZUKO_AMD_NO_STREAM=1 routes every call through the general uni_kernel (A/B measurements, tests of both paths)
static bool zk_no_stream() {
  const char* e = getenv("ZUKO_AMD_NO_STREAM");  // (read per call: the tests switch it)
  return e && e[0] == '1';
}

extern "C" {

int zk_rqs_forward(int dtype, int64_t N, int64_t D, int K, double bound, double slope, const void* x, const void* widths, int64_t w_sN, int64_t w_sD,
                   const void* heights, int64_t h_sN, int64_t h_sD, const void* derivs, int64_t d_sN, int64_t d_sD, void* y, void* ladj, int ladj_reduced,
                   int32_t* bin_out, void* stream) {
  UniArgs a = base_args(N, D, x, y, ladj, ladj_reduced, bin_out);
  a.seg[0] = {widths, w_sN, w_sD}; a.seg[1] = {heights, h_sN, h_sD}; a.seg[2] = {derivs, d_sN, d_sD};
  a.bound = bound; a.ls = log(slope); a.lc = rqs_lean_const(bound, a.ls);
  if (dtype == ZK_DTYPE_BF16) {  // bf16 x / phi / y, fp32 arithmetic and ladj: stream-kernel shapes only
    const int rc = try_rqs_stream<false, true>(a, K, (hipStream_t)stream);
    return rc >= 0 ? rc : ZK_EINVAL;
  }
  if (dtype == ZK_DTYPE_F32 && !zk_no_stream()) {
    const int rc = try_rqs_stream<false>(a, K, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  return ZK_DISPATCH(dtype, (launch_rqs<float, false>(a, K, (hipStream_t)stream)), (launch_rqs<double, false>(a, K, (hipStream_t)stream)));
}

int zk_rqs_inverse(int dtype, int64_t N, int64_t D, int K, double bound, double slope, const void* y, const void* widths, int64_t w_sN, int64_t w_sD,
                   const void* heights, int64_t h_sN, int64_t h_sD, const void* derivs, int64_t d_sN, int64_t d_sD, void* x, int32_t* bin_out, void* stream) {
  UniArgs a = base_args(N, D, y, x, nullptr, 0, bin_out);
  a.seg[0] = {widths, w_sN, w_sD}; a.seg[1] = {heights, h_sN, h_sD}; a.seg[2] = {derivs, d_sN, d_sD};
  a.bound = bound; a.ls = log(slope); a.lc = rqs_lean_const(bound, a.ls);
  if (dtype == ZK_DTYPE_BF16) {
    const int rc = try_rqs_stream<true, true>(a, K, (hipStream_t)stream);
    return rc >= 0 ? rc : ZK_EINVAL;
  }
  if (dtype == ZK_DTYPE_F32 && !zk_no_stream()) {
    const int rc = try_rqs_stream<true>(a, K, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  return ZK_DISPATCH(dtype, (launch_rqs<float, true>(a, K, (hipStream_t)stream)), (launch_rqs<double, true>(a, K, (hipStream_t)stream)));
}

// Diagnostic twin of zk_rqs_forward / zk_rqs_inverse (fp32 only): the SAME per-element arithmetic the product kernels
// run (rqs_lean) through the general kernel, additionally writing the bin index and the K+1 knots of the search axis.
int zk_rqs_diag(int inverse, int64_t N, int64_t D, int K, double bound, double slope, const void* in, const void* widths, int64_t w_sN, int64_t w_sD,
                const void* heights, int64_t h_sN, int64_t h_sD, const void* derivs, int64_t d_sN, int64_t d_sD, void* out, void* ladj, int32_t* bin_out,
                float* knots_out, void* stream) {
  if (K != 4 && K != 8 && K != 16) return ZK_EINVAL;
  UniArgs a = base_args(N, D, in, out, inverse ? nullptr : ladj, 0, bin_out);
  a.knots_out = knots_out;
  a.seg[0] = {widths, w_sN, w_sD}; a.seg[1] = {heights, h_sN, h_sD}; a.seg[2] = {derivs, d_sN, d_sD};
  a.bound = bound; a.ls = log(slope); a.lc = rqs_lean_const(bound, a.ls);
  return inverse ? launch_rqs<float, true>(a, K, (hipStream_t)stream) : launch_rqs<float, false>(a, K, (hipStream_t)stream);
}

int zk_rqs_from_knots(int dtype, int inverse, int64_t N, int64_t D, int K, const void* in, const void* horizontal, const void* vertical, const void* slopes,
                      int64_t k_sN, int64_t k_sD, void* out, void* ladj, int32_t* bin_out, void* stream) {
  UniArgs a = base_args(N, D, in, out, inverse ? nullptr : ladj, 0, bin_out);
  a.seg[0] = {horizontal, k_sN, k_sD}; a.seg[1] = {vertical, k_sN, k_sD}; a.seg[2] = {slopes, k_sN, k_sD};
  hipStream_t st = (hipStream_t)stream;
  if (inverse) return ZK_DISPATCH(dtype, (launch_rqs_knots<float, true>(a, K, st)), (launch_rqs_knots<double, true>(a, K, st)));
  return ZK_DISPATCH(dtype, (launch_rqs_knots<float, false>(a, K, st)), (launch_rqs_knots<double, false>(a, K, st)));
}

int zk_affine_forward(int dtype, int64_t N, int64_t D, double slope, const void* x, const void* shift, int64_t s_sN, int64_t s_sD, const void* scale,
                      int64_t c_sN, int64_t c_sD, void* y, void* ladj, int ladj_reduced, void* stream) {
  UniArgs a = base_args(N, D, x, y, ladj, ladj_reduced, nullptr);
  a.seg[0] = {shift, s_sN, s_sD}; a.seg[1] = {scale, c_sN, c_sD}; a.seg[2] = {scale, c_sN, c_sD};
  a.ls = log(slope);
  const int lens[2] = {1, 1};
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_uni<float, AffineOp<float, false>, AffineOp<float, false>>(a, lens, 2, st)),
                     (launch_uni<double, AffineOp<double, false>, AffineOp<double, false>>(a, lens, 2, st)));
}

int zk_affine_inverse(int dtype, int64_t N, int64_t D, double slope, const void* y, const void* shift, int64_t s_sN, int64_t s_sD, const void* scale,
                      int64_t c_sN, int64_t c_sD, void* x, void* stream) {
  UniArgs a = base_args(N, D, y, x, nullptr, 0, nullptr);
  a.seg[0] = {shift, s_sN, s_sD}; a.seg[1] = {scale, c_sN, c_sD}; a.seg[2] = {scale, c_sN, c_sD};
  a.ls = log(slope);
  const int lens[2] = {1, 1};
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_uni<float, AffineOp<float, true>, AffineOp<float, true>>(a, lens, 2, st)),
                     (launch_uni<double, AffineOp<double, true>, AffineOp<double, true>>(a, lens, 2, st)));
}

int zk_sos_forward(int dtype, int64_t N, int64_t D, int P, int L1, double slope, const double* gl_nodes01, const double* gl_weights01, const void* x,
                   const void* a_coef, int64_t a_sN, int64_t a_sD, const void* constant, int64_t c_sN, int64_t c_sD, void* y, void* ladj, int ladj_reduced,
                   void* stream) {
  UniArgs a = base_args(N, D, x, y, ladj, ladj_reduced, nullptr);
  a.seg[0] = {a_coef, a_sN, a_sD}; a.seg[1] = {constant ? constant : a_coef, c_sN, c_sD}; a.seg[2] = a.seg[1];
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_sos<float, false>(a, P, L1, slope, gl_nodes01, gl_weights01, constant != nullptr, st)),
                     (launch_sos<double, false>(a, P, L1, slope, gl_nodes01, gl_weights01, constant != nullptr, st)));
}

int zk_sos_inverse(int dtype, int64_t N, int64_t D, int P, int L1, double slope, const double* gl_nodes01, const double* gl_weights01, int n_bisect,
                   const void* y, const void* a_coef, int64_t a_sN, int64_t a_sD, const void* constant, int64_t c_sN, int64_t c_sD, void* x, void* stream) {
  UniArgs a = base_args(N, D, y, x, nullptr, 0, nullptr);
  a.seg[0] = {a_coef, a_sN, a_sD}; a.seg[1] = {constant ? constant : a_coef, c_sN, c_sD}; a.seg[2] = a.seg[1];
  a.n_bisect = n_bisect;
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_sos<float, true>(a, P, L1, slope, gl_nodes01, gl_weights01, constant != nullptr, st)),
                     (launch_sos<double, true>(a, P, L1, slope, gl_nodes01, gl_weights01, constant != nullptr, st)));
}

int zk_bernstein_forward(int dtype, int64_t N, int64_t D, int M, int bounded, double bound, double eps, const void* x, const void* theta, int64_t t_sN, int64_t t_sD,
                         void* y, void* ladj, int ladj_reduced, void* stream) {
  if (!(eps > 0.0 && eps < 0.5)) return ZK_EINVAL;
  UniArgs a = base_args(N, D, x, y, ladj, ladj_reduced, nullptr);
  a.seg[0] = {theta, t_sN, t_sD}; a.seg[1] = a.seg[0]; a.seg[2] = a.seg[0];
  a.bound = bound; a.bounded = bounded; a.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_bern<float, false>(a, M, st)), (launch_bern<double, false>(a, M, st)));
}

int zk_bernstein_inverse(int dtype, int64_t N, int64_t D, int M, int bounded, double bound, double eps, int n_bisect, const void* y, const void* theta, int64_t t_sN,
                         int64_t t_sD, void* x, void* stream) {
  if (!(eps > 0.0 && eps < 0.5)) return ZK_EINVAL;
  UniArgs a = base_args(N, D, y, x, nullptr, 0, nullptr);
  a.seg[0] = {theta, t_sN, t_sD}; a.seg[1] = a.seg[0]; a.seg[2] = a.seg[0];
  a.bound = bound; a.bounded = bounded; a.n_bisect = n_bisect; a.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_bern<float, true>(a, M, st)), (launch_bern<double, true>(a, M, st)));
}

int zk_diag_normal_log_prob(int dtype, int64_t N, int64_t D, const void* z, const void* loc, const void* scale, const void* ladj, void* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  return ZK_DISPATCH(dtype, (launch_normal<float>(N, D, z, loc, scale, ladj, out, st)), (launch_normal<double>(N, D, z, loc, scale, ladj, out, st)));
}

// out[0] = scale * sum_i v[i], accumulated in f64; `workspace` must hold >= 1024 doubles.
int zk_sum_f64(int dtype, int64_t N, const void* v, double scale, double* workspace, double* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int64_t nb = (N + 255) / 256;
  int blocks = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
  if (dtype == ZK_DTYPE_F32) hipLaunchKernelGGL((sum_kernel<float>), dim3(blocks), dim3(256), 0, st, N, (const float*)v, workspace);
  else if (dtype == ZK_DTYPE_F64) hipLaunchKernelGGL((sum_kernel<double>), dim3(blocks), dim3(256), 0, st, N, (const double*)v, workspace);
  else return ZK_EINVAL;
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, blocks, workspace, out, scale);
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
