// zuko_amd — the TWO-SET form of the operand-split fused autoregressive kernel (fused_ar_split_impl.h):
//
//     y, log|dy/dx| = univariate(conditioner(cat(x, c))).call_and_ladj(x)      (zuko/flows/autoregressive.py:207-218)
//
// Same arithmetic, same weight stream and — per output — the same sequence of partial products as arx_kernel (asserted bit-identical in
// tests/test_gpu_flows.py), but a different occupancy model.  arx_kernel runs two wavefronts per SIMD, 16 samples each: the matrix pipe is
// busy 62 % of the time (profiles/r03/split_kernel.md) because the VALU phases of a wavefront — operand conversions, the spline — do not
// overlap its partner's matrix instructions, and every weight image read from LDS feeds only six of them.  Here ONE wavefront per SIMD
// (up to 512 registers) carries TWO 16-sample sets:
//
//   * a weight image is read once and multiplied against both sets (half the LDS reads, ring DMAs and chunk barriers per sample);
//   * a STEP is one in pair against two out tiles (consecutive blocks of the stream), executed as six QUADS: one partial-product term
//     each = four matrix instructions on four different accumulators (2 out tiles x 2 sets), so no instruction waits for its
//     predecessor's result (a dependent v_mfma_f32_16x16x32_bf16 issues after ~38 cycles, an independent one after 16);
//   * the non-matrix work is cut into small units — the ReLU + three-way bf16 split of two activations; a micro-step of the univariate
//     map of the PREVIOUS feature group — and every quad carries a few of them (tables CVQ / SPQ, dealt by zuko_amd/static_ar.py:
//     split2_schedule); inside a quad's scheduling region the compiler is told to alternate them with the matrix instructions
//     (sched_group_barrier), which is where VALU work is free (scripts/probes/coexec_probe.hip: two VALU instructions per matrix
//     instruction of the SAME wavefront cost nothing).
//
// Register plan per set: in[NSLOT] (bf16 h, m, l parts of a layer's input pairs: 12 registers each) + out[2 NOS] (f32 accumulators of
// NOS = 3 out pairs).  A hidden layer walks its out pairs from the last to the first; with the units sorted by dependency count out
// pair P reads in pairs <= P, so when P is complete in pair P is dead, and P's conversion — the next layer's in pair P — is written over
// it (slots dealt by split2_schedule) while out pair P - 1 is being multiplied.  The next layer, and the last layer's feature groups,
// read the pairs in the order they were finished.  Last layer: when a group's accumulators are complete they become the parameter
// registers of its maps, and the next group accumulates while those are evaluated.
#pragma once
#include "fused_ar_split_impl.h"

namespace zk {

// Raw LDS accesses.  While an LDS-DMA (global_load_lds) is in flight the compiler orders EVERY LDS access it knows about behind
// `s_waitcnt vmcnt(0)` (the DMA might write the same bytes) — which drains the weight ring's look-ahead, one to two thousand cycles each
// time.  The ring, the row image and the bias image are disjoint, so the accesses inside the pass are issued from inline assembly: a
// read returns a RAW value, usable only behind a covering `s_waitcnt lgkmcnt` that names it (ArRingS::read has the idiom); a write
// needs nothing — the LDS operations of a wavefront execute in order, so a later read of the same word sees it.
template <int OFF> __device__ __forceinline__ void arx2_raw_read(f32x4& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ float arx2_raw_read1(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF> __device__ __forceinline__ float arx2_raw_read1i(unsigned addr) {  // (one base register + an immediate: nothing per-group for the compiler to hoist and spill)
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void arx2_raw_write1(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void arx2_settle1(float& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ void arx2_tie(int& v) { asm volatile("" : "+v"(v)); }  // orders the uses of a raw value behind the wait in front of this statement

// ---- univariate maps as sequences of micro-steps -------------------------------------------------------------------------------------
// State lives in registers between the quads that carry the steps.  `p(i)` is a reference to parameter i of the feature (an accumulator
// register).  Every step sequence evaluates exactly what Uni::fwd of zk_ar_common.h evaluates, operation by operation.
struct Uni2Io {
  unsigned xr;    // LDS byte address of the row image of this lane's sample: x is read from it, y written to it (raw accesses)
  int f;          // feature id (< 0: padding slot)
  int spare;      // index of a word of the row image no feature owns (the rows are padded to D + 4 words)
  float poison;   // NaN when the sample has a non-finite input, else 0
  int64_t n;      // sample (diagnostic stores)
  bool live;
};

struct Uni2Affine {
  typedef UniAffine Base;
  static constexpr int NSTEP = 2;
  struct State { float x; };
  template <int I, bool DIAG, class P, class A> static __device__ __forceinline__ void step(State& st, const P& p, const A& a, const Uni2Io& io, float& lacc) {
    if constexpr (I == 0) {
      st.x = arx2_raw_read1(io.xr + 4u * (unsigned)(io.f < 0 ? 0 : io.f));
      p(0) += io.poison;
      p(1) += io.poison;
    } else {
      arx2_settle1(st.x);
      float y, lj;  // (no branch: see Uni2Rqs)
      affine_fwd<float, MathFast>(p(0), p(1), a.ls, st.x, y, lj);
      arx2_raw_write1(io.xr + 4u * (unsigned)(io.f >= 0 ? io.f : io.spare), y);
      lacc += io.f >= 0 ? lj : 0.f;
    }
  }
};

template <int K, bool CIRC> struct Uni2Rqs {
  typedef UniRqs<K, CIRC> Base;
  static constexpr int LV = K == 4 ? 2 : (K == 8 ? 3 : 4);
  static constexpr int NKS = (K + 3) / 4;
  // load | K softmax elements | knots, four per step | bisection: level 0 in two halves, then one step per level | slopes (2) | map (2)
  static constexpr int S_SOFT = 1, S_KNOT = 1 + K, S_BIS = S_KNOT + NKS, S_SLOPE = S_BIS + LV + 1, S_EVAL = S_SLOPE + 2, NSTEP = S_EVAL + 2;
  struct State {
    float v;
    f32x2_t acc;
    f32x2_t cum[K];
    f32x2_t scale;
    float kx[K + 1], ky[K + 1], kr[K + 1];
    float ks[K + 1];  // diagnostic instantiation only: the search-axis knots as the bisection compared them
    bool inside, above, c0;
    int bin;
    float d0, d1, m, dx, dy, rdx, s, t;
    float z, omz, zz, rden, out;
  };
  template <int I, bool DIAG, class P, class A> static __device__ __forceinline__ void step(State& st, const P& p, const A& a, const Uni2Io& io, float& lacc) {
    const RqsLeanConst& c = a.lc;
    if constexpr (I == 0) {
      st.v = arx2_raw_read1(io.xr + 4u * (unsigned)(io.f < 0 ? 0 : io.f));  // (raw: settled by the first knot step, long after)
#pragma unroll
      for (int j = 0; j < K; ++j) p(j) += io.poison;  // (UniRqs::poison<false>: the search-axis parameters)
      st.acc = f32x2_t{0.f, 0.f};
    } else if constexpr (I < S_KNOT) {
      constexpr int j = I - S_SOFT;
      const f32x2_t u = {p(j), p(K + j)};
      const f32x2_t r = {__builtin_amdgcn_rcpf(fmaf(fabsf(u.x), c.c2l, c.il2e)), __builtin_amdgcn_rcpf(fmaf(fabsf(u.y), c.c2l, c.il2e))};
      const f32x2_t t = u * r;
      st.acc += f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
      st.cum[j] = st.acc;
    } else if constexpr (I < S_BIS) {
      constexpr int q = I - S_KNOT;
      if constexpr (q == 0) {
        arx2_settle1(st.v);
        if constexpr (CIRC) st.v = Base::shift(st.v, a.bound);
        st.scale = f32x2_t{__builtin_amdgcn_rcpf(st.acc.x), __builtin_amdgcn_rcpf(st.acc.y)} * (2.f * c.bound);
        st.kx[0] = -c.bound; st.ky[0] = -c.bound;
        st.kr[0] = 0.f; st.kr[K] = 0.f;
      }
      const f32x2_t negB = {-c.bound, -c.bound};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * q + jj;  // (a compile-time constant after unrolling: the state must stay in registers)
        if (j < K) {
          const f32x2_t kn = __builtin_elementwise_fma(st.cum[j], st.scale, negB);
          st.kx[j + 1] = kn.x; st.ky[j + 1] = kn.y;
          if (j >= 1) st.kr[j] = p(2 * K + j - 1);
        }
      }
    } else if constexpr (I == S_BIS) {  // bisection, level 0, first half: flags, the compare, the search axis
      if constexpr (DIAG) {
#pragma unroll
        for (int j = 0; j <= K; ++j) st.ks[j] = st.kx[j];
      }
      st.above = st.kx[K] < st.v;
      st.inside = (st.kx[0] < st.v) && !st.above;
      st.c0 = st.kx[K / 2] < st.v;
#pragma unroll
      for (int i = 0; i <= K / 2; ++i) st.kx[i] = st.c0 ? st.kx[K / 2 + i] : st.kx[i];
      st.bin = st.c0 ? K / 2 : 0;
    } else if constexpr (I == S_BIS + 1) {  // second half: the other axis and the derivative parameters
#pragma unroll
      for (int i = 0; i <= K / 2; ++i) {
        st.ky[i] = st.c0 ? st.ky[K / 2 + i] : st.ky[i];
        st.kr[i] = st.c0 ? st.kr[K / 2 + i] : st.kr[i];
      }
    } else if constexpr (I < S_SLOPE) {
      constexpr int lv = I - S_BIS - 1;  // 1 .. LV - 1
      constexpr int M = K >> lv;  // candidates M + 1 -> M / 2 + 1, in place (index i is written from i and M / 2 + i >= i)
      const bool cc = st.kx[M / 2] < st.v;
#pragma unroll
      for (int i = 0; i <= M / 2; ++i) {
        st.kx[i] = cc ? st.kx[M / 2 + i] : st.kx[i];
        st.ky[i] = cc ? st.ky[M / 2 + i] : st.ky[i];
        st.kr[i] = cc ? st.kr[M / 2 + i] : st.kr[i];
      }
      st.bin += cc ? M / 2 : 0;
    } else if constexpr (I == S_SLOPE) {
      const float r0 = st.kr[0], r1 = st.kr[1];
      const f32x2_t rr = {r0, r1};
      const f32x2_t td = rr * f32x2_t{__builtin_amdgcn_rcpf(fmaf(fabsf(r0), c.c1l, c.il2e)), __builtin_amdgcn_rcpf(fmaf(fabsf(r1), c.c1l, c.il2e))};
      st.d0 = __builtin_amdgcn_exp2f(td.x); st.d1 = __builtin_amdgcn_exp2f(td.y);
    } else if constexpr (I == S_SLOPE + 1) {
      st.m = st.inside ? 1.f : 0.f;
      st.dx = st.kx[1] - st.kx[0]; st.dy = st.ky[1] - st.ky[0];
      st.rdx = __builtin_amdgcn_rcpf(st.dx);
      st.s = st.dy * st.rdx;
      st.t = (st.d0 + st.d1) - 2.f * st.s;
    } else if constexpr (I == S_EVAL) {
      const float x0 = st.kx[0], y0 = st.ky[0], v = st.v, s = st.s, d0 = st.d0;
      const float z = (st.m * (v - x0)) * st.rdx;
      const float omz = 1.f - z;
      const float zz = z * omz;
      const float den = fmaf(st.t, zz, s);
      const float rden = __builtin_amdgcn_rcpf(den);
      const float num = fmaf(s * z, z, d0 * zz);
      const float yy = fmaf(st.dy * num, rden, y0);
      st.out = st.inside ? yy : v;
      st.z = z; st.omz = omz; st.zz = zz; st.rden = rden;
    } else {
      const float s = st.s, d0 = st.d0, d1 = st.d1, z = st.z, omz = st.omz;
      const float jn = fmaf(d1 * z, z, fmaf(d0 * omz, omz, (2.f * s) * st.zz));
      const float sr = s * st.rden;
      const float jac = (sr * sr) * jn;
      const float lj = st.m * (__builtin_amdgcn_logf(jac) * c.il2e);
      // NO branch around the result: everything above is used only here, and a conditional block would make the compiler sink the whole
      // map into it (one lump of ~330 VALU instructions behind the last step instead of micro-steps between the matrix instructions).
      // A padding slot (f < 0) writes the row's spare word and adds zero.
      arx2_raw_write1(io.xr + 4u * (unsigned)(io.f >= 0 ? io.f : io.spare), st.out);
      lacc += io.f >= 0 ? lj : 0.f;
      if constexpr (DIAG) {
        if (io.f >= 0) {
          if (io.live) {
            a.bin_out[io.n * a.D + io.f] = st.inside ? st.bin : (st.above ? K : -1);
#pragma unroll
            for (int j = 0; j <= K; ++j) a.knots_out[(io.n * a.D + io.f) * (K + 1) + j] = st.ks[j];
          }
        }
      }
    }
  }
};

template <class U> struct Uni2Of;
template <> struct Uni2Of<UniAffine> { typedef Uni2Affine type; };
template <int K, bool CIRC> struct Uni2Of<UniRqs<K, CIRC>> { typedef Uni2Rqs<K, CIRC> type; };

// ---- conversion units --------------------------------------------------------------------------------------------------------------
// unit u of a layer's input: in pair u / 16, set (u / 8) % 2, value pair e2 = (u / 2) % 4 (values 2 e2, 2 e2 + 1 of the pair's eight per
// lane), half u % 2: activation (ReLU / none), then the bf16 parts h, m, l exactly as arx_split computes them — first half: h and the
// remainder v - h (kept in `rem` for the second half); second half: m, l.
// ReLU on the integer pipe: max(bits, 0) maps every value with the sign bit set to +0 and leaves the others (NaN included) alone — one
// instruction instead of a compare / select pair through VCC.  It differs from `v < 0 ? 0 : v` for -0.0 (+0.0 here) and for NaNs with
// the sign bit set (zero here; the device's own NaNs are positive, and non-finite INPUTS poison the parameters explicitly).
__device__ __forceinline__ float arx2_relu(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

template <int ACT, int E2, int HALF> __device__ __forceinline__ void arx2_convert(const f32x4& lo, const f32x4& hi, ArxB& b, float (&rem)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    constexpr int e0 = 2 * E2;
    const int e = e0 + i;
    if constexpr (HALF == 0) {
      float v = e < 4 ? lo[e] : hi[e - 4];
      if constexpr (ACT == 1) v = arx2_relu(v);
      const __bf16 h = (__bf16)v;
      rem[i] = v - (float)h;
      b.h[e] = h;
    } else {
      const float r1 = rem[i];
      const __bf16 m = (__bf16)r1;
      const float r2 = r1 - (float)m;
      b.m[e] = m;
      b.l[e] = (__bf16)r2;
    }
  }
}


template <class S> struct Arx2Pat {
  static constexpr int n_hidden_steps() { return S::HS_OFF[S::NH]; }
  static constexpr int n_steps() { return S::HS_OFF[S::NH] + S::LS_OFF[S::NG]; }
  static constexpr int layer_of(int gs) {  // hidden layer of global step gs (< n_hidden_steps())
    int l = 0;
    while (gs >= S::HS_OFF[l + 1]) ++l;
    return l;
  }
  static constexpr int step_images(int gs) {  // images of global step gs (hidden steps first, then the last layer's)
    if (gs >= n_steps()) return 0;
    if (gs < n_hidden_steps()) return S::H_OT1[gs] == 255 ? 3 : 6;
    return S::L_T1[gs - n_hidden_steps()] == 255 ? 3 : 6;
  }
  static constexpr int step_pos(int gs) {  // stream position of its first image
    if (gs >= n_steps()) return 0;
    if (gs < n_hidden_steps()) return S::BASE[layer_of(gs)] + 3 * S::H_BLK[gs];
    return S::LAST_BASE + 3 * S::L_BLK[gs - n_hidden_steps()];
  }
  // bias tile a hidden step has to start accumulator `which` of its out pair from, or -1: the pair's first step initialises both
  static constexpr int bias_tile(int gs, int which) {
    if (gs >= n_hidden_steps() || !S::H_INIT[gs]) return -1;
    const int l = layer_of(gs), t = 2 * (S::H_OT0[gs] / 2) + which;
    return t < S::HT[l] ? t : -1;
  }
};

// the scheduling request of one quad: N matrix instructions, each followed by FILL VALU / transcendental instructions (Shape::FILL;
// 0: no request — the compiler's own order)
template <int N, int FILL> __device__ __forceinline__ void arx2_pattern() {
  if constexpr (FILL > 0) {
    ars_for<N>([&](auto) ARS_ALWAYS_INLINE {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, FILL, 0);
    });
  }
}

// one partial-product term of a step: term k of arx_block's sequence (a[0] = h, a[1] = m, a[2] = l images of the weights)
template <int KT> __device__ __forceinline__ void arx2_term(const f32x4& ah, const f32x4& am, const f32x4& al, const ArxB& b, f32x4& c) {
  if (ARX_ABL == 2) {
    asm volatile("" ::"v"(ah), "v"(am), "v"(al));
    return;
  }
  if constexpr (KT == 0) ARX_MFMA(al, b.h, c);
  else if constexpr (KT == 1) ARX_MFMA(ah, b.l, c);
  else if constexpr (KT == 2) ARX_MFMA(am, b.m, c);
  else if constexpr (KT == 3) ARX_MFMA(am, b.h, c);
  else if constexpr (KT == 4) ARX_MFMA(ah, b.m, c);
  else ARX_MFMA(ah, b.h, c);
}

template <class S, typename Uni, bool DIAG> __global__ __launch_bounds__(256, 1) void arx2_kernel(ArArgs a) {
  typedef ArRingS<4, S::CH, S::NR> Ring;
  typedef typename Uni2Of<Uni>::type U2;
  typedef Arx2Pat<S> P2;
  static_assert(S::NH >= 1 && S::CH == 24 && S::NR == 3 && S::NOS == 3 && S::XLDS && (S::ACT == 0 || S::ACT == 1), "two-set operand-split kernel: widths <= 256, LDS-staged rows");
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL;
  constexpr int NG = S::NG, NH = S::NH;
  constexpr int DT = (S::D + 15) / 16;
  constexpr int SPT = 2 * FPL * U2::NSTEP;  // micro-steps of one group's univariate maps: (set, feature slot, step)
  static_assert(SPT == S::SP_TOTAL, "schedule tables were dealt for another step sequence");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;

  Ring ring;
  float* bias_lds = ars_lds + S::NR * S::CH * AR_TF;
  ring.lds = ars_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < S::NR - 1; ++i) ring.issue();
  ring.slot = S::NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ars_lds);
  ring.cur_off = ring.lds_off;

  for (int i = tid; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);  // same LDS layout as the other static-shape kernels
  float* xrow_lds[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) xrow_lds[s] = reinterpret_cast<float*>(fmap_lds + 1024 + 256) + (wave * 32 + s * 16 + j) * a.xs;
  const unsigned bias_off = ring.lds_off + (unsigned)((S::NR * S::CH * AR_TF + 4 * q) * 4);                       // this lane's quad of a bias tile; + tile offsets as immediates
  const unsigned fmap_off = ring.lds_off + (unsigned)((S::NR * S::CH * AR_TF + a.bias_floats + q * FPL) * 4);  // this lane's slots of group 0; + group offsets as immediates
  unsigned xrow_off[2];  // (LDS byte addresses of the same rows, for the raw accesses inside the pass)
#pragma unroll
  for (int s = 0; s < 2; ++s) xrow_off[s] = ring.lds_off + (unsigned)((S::NR * S::CH * AR_TF + a.bias_floats + 1024 + 256 + (wave * 32 + s * 16 + j) * a.xs) * 4);
  for (int i = tid; i < NG * 4 * FPL; i += 256) fmap_lds[i] = a.featmap[i];
  __syncthreads();
  const float* bias_last = bias_lds + NH * S::BIAS_STRIDE;
  int pass_no = 0;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x, ++pass_no) {
    int64_t n[2];
    bool live[2];
    float poison[2] = {0.f, 0.f};
    ArxB in[2][S::NSLOT];
    f32x4 out[2][2 * S::NOS];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      n[s] = tile * 128 + wave * 32 + s * 16 + j;
      live[s] = n[s] < a.N;
      const float* xrow = a.x + (live[s] ? n[s] : a.N - 1) * a.ldx;
      f32x4 xin[S::NIT + 1];
#pragma unroll
      for (int it = 0; it < S::NIT; ++it) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((it + 1) * 16 <= S::DIN || it * 16 + 4 * q < S::DIN) v = *reinterpret_cast<const f32x4*>(xrow + it * 16 + 4 * q);
        xin[it] = v;
      }
      xin[S::NIT] = f32x4{0.f, 0.f, 0.f, 0.f};
      // a NaN / inf input turns ALL parameters of its sample into NaN in the reference (x * 0 = NaN, zuko/nn.py:217-218)
      int bad = 0;
#pragma unroll
      for (int it = 0; it < S::NIT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= !(fabsf(xin[it][r]) < __builtin_inff());
      bad |= __shfl_xor(bad, 16, 64);
      bad |= __shfl_xor(bad, 32, 64);
      if (bad) poison[s] = __builtin_nanf("");
#pragma unroll
      for (int it = 0; it < DT; ++it)
        if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(xrow_lds[s] + it * 16 + 4 * q) = xin[it];
      ars_for<(S::NIT + 1) / 2>([&](auto p_) ARS_ALWAYS_INLINE { arx_split(xin[2 * decltype(p_)::value], xin[2 * decltype(p_)::value + 1], in[s][S::X_SLOT[decltype(p_)::value]]); });
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    f32x4 w[2][6];   // the six images of a step, double-buffered (raw until settled)
    f32x4 bpre[2];   // bias tiles the NEXT step starts accumulators from, requested one step ahead (raw until settled)
    float cvrem[2];  // remainders v - h of the conversion unit whose second half is still to come
    // conversion half-units [LO, HI) of the pass-wide sequence: unit u belongs to the out pair at completion position u / 16 (tables CP_*:
    // its accumulator slot, the in slot it becomes, whether it has a second tile), set (u / 8) % 2, value pair (u / 2) % 4, half u % 2
    auto convert = [&](auto lo_, auto hi_) ARS_ALWAYS_INLINE {
      constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      if (ARX_ABL == 7) return;
      ars_for<HI - LO>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int u = LO + decltype(i_)::value, gp = u / 16, s = (u / 8) % 2, e2 = (u / 2) % 4, half = u % 2;
        constexpr int os = S::CP_OS[gp], is = S::CP_IS[gp];
        if constexpr (S::CP_T1[gp] != 255) arx2_convert<S::ACT, e2, half>(out[s][2 * os], out[s][2 * os + 1], in[s][is], cvrem);
        else arx2_convert<S::ACT, e2, half>(out[s][2 * os], zero, in[s][is], cvrem);
      });
    };
    // COUNT images of global step GS starting with image FIRST
    auto fetch = [&](auto gs_, auto first_, auto count_) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value, FIRST = decltype(first_)::value, COUNT = decltype(count_)::value;
      ars_for<COUNT>([&](auto i_) ARS_ALWAYS_INLINE { w[GS & 1][FIRST + decltype(i_)::value] = ring.template read_acc<P2::step_pos(GS) + FIRST + decltype(i_)::value>(); });
    };
    // bias tiles of global step GS (hidden layers): raw LDS reads
    auto fetch_bias = [&](auto gs_) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value;
      if constexpr (GS < P2::n_hidden_steps()) {
        constexpr int L = P2::layer_of(GS);
        ars_for<2>([&](auto w_) ARS_ALWAYS_INLINE {
          constexpr int T = P2::bias_tile(GS, decltype(w_)::value);
          if constexpr (T >= 0) arx2_raw_read<(L * S::BIAS_STRIDE + T * 16) * 4>(bpre[decltype(w_)::value], bias_off);
        });
      }
    };
    constexpr int NSTEPS_H = P2::n_hidden_steps(), NSTEPS = P2::n_steps();
    if constexpr (NSTEPS > 0) {
      fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, P2::step_images(0)>{});
      fetch_bias(std::integral_constant<int, 0>{});
    }
    // one step: settle its images, six quads; the first three quads also request the images of the next step (two each)
    auto run_step = [&](auto gs_, auto&& quad_extra, f32x4& c00, f32x4& c10, f32x4& c01, f32x4& c11, const ArxB& b0, const ArxB& b1) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value, BUF = GS & 1, NI = P2::step_images(GS);
      constexpr bool TWO = NI == 6;
#ifdef ARX2_TRACE  // probe build (scripts/arx2_trace.py): shader clock at the start of every step of workgroup 0 / wavefront 0, first two passes
      if (a.bin_out && blockIdx.x == 0 && wave == 0 && pass_no < 2) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) a.bin_out[pass_no * (NSTEPS + 1) + GS] = (int)(unsigned)now;
      }
#endif
      if constexpr (TWO) ars_settle_acc<0>(w[BUF][0], w[BUF][1], w[BUF][2], w[BUF][3], w[BUF][4], w[BUF][5]);
      else ars_settle_acc<0>(w[BUF][0], w[BUF][1], w[BUF][2]);
      asm volatile("" : "+v"(bpre[0]), "+v"(bpre[1]));  // (settled by the same wait: LDS operations complete in order)
      __builtin_amdgcn_sched_barrier(0);
      ars_for<6>([&](auto k_) ARS_ALWAYS_INLINE {
        constexpr int KT = decltype(k_)::value;
        if constexpr (GS + 1 < NSTEPS && KT < 3) {
          constexpr int NN = P2::step_images(GS + 1);
          constexpr int FIRST = 2 * KT, COUNT = FIRST >= NN ? 0 : (FIRST + 2 <= NN ? 2 : NN - FIRST);
          if constexpr (COUNT > 0) fetch(std::integral_constant<int, GS + 1>{}, std::integral_constant<int, FIRST>{}, std::integral_constant<int, COUNT>{});
          if constexpr (KT == 2) fetch_bias(std::integral_constant<int, GS + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        quad_extra(k_);
        arx2_term<KT>(w[BUF][0], w[BUF][1], w[BUF][2], b0, c00);
        arx2_term<KT>(w[BUF][0], w[BUF][1], w[BUF][2], b1, c10);
        if constexpr (TWO) {
          arx2_term<KT>(w[BUF][3], w[BUF][4], w[BUF][5], b0, c01);
          arx2_term<KT>(w[BUF][3], w[BUF][4], w[BUF][5], b1, c11);
        }
        arx2_pattern<TWO ? 4 : 2, S::FILL>();
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    // ---- hidden layers: out pairs from the last to the first ----------------------------------------------------------------------------
    ars_for<NSTEPS_H>([&](auto gs_) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value;
      constexpr int OT0 = S::H_OT0[GS], OT1 = S::H_OT1[GS], OS = S::H_OS[GS];
      constexpr bool TWO = OT1 != 255;
      run_step(
          std::integral_constant<int, GS>{},
          [&](auto k_) ARS_ALWAYS_INLINE {
            constexpr int KT = decltype(k_)::value, QI = 6 * GS + KT;
            convert(std::integral_constant<int, S::CVQ[QI]>{}, std::integral_constant<int, S::CVQ[QI + 1]>{});
            if constexpr (KT == 0) {  // a pair's first step: both accumulators start at the bias (after the units that still read the slot's previous tenant)
              if constexpr (P2::bias_tile(GS, 0) >= 0) {
                out[0][2 * OS] = bpre[0];
                out[1][2 * OS] = bpre[0];
              }
              if constexpr (P2::bias_tile(GS, 1) >= 0) {
                out[0][2 * OS + 1] = bpre[1];
                out[1][2 * OS + 1] = bpre[1];
              }
            }
          },
          out[0][2 * OS + (OT0 & 1)], out[1][2 * OS + (OT0 & 1)], out[0][2 * OS + ((TWO ? OT1 : OT0) & 1)], out[1][2 * OS + ((TWO ? OT1 : OT0) & 1)], in[0][S::H_IP[GS]], in[1][S::H_IP[GS]]);
    });

    // ---- last layer + univariate maps: group g + 1 accumulates while the maps of group g are evaluated ------------------------------
    float lacc[2] = {0.f, 0.f};
    f32x4 acc[2][NT];       // [set][tile] of the group that accumulates
    float par[2][4 * NT];   // parameters of the group whose maps are being evaluated (the previous group's accumulators)
    int fnext[FPL], fpar[FPL];  // feature ids of this lane's slots in the accumulating group / in the group whose maps are evaluated
    typename U2::State ust;
    // micro-steps [LO, HI) of group G
    auto spline = [&](auto g_, auto lo_, auto hi_) ARS_ALWAYS_INLINE {
      constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
      if (ARX_ABL == 8) return;
      ars_for<HI - LO>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int u = LO + decltype(i_)::value, s = u / (FPL * U2::NSTEP), fi = (u / U2::NSTEP) % FPL, k = u % U2::NSTEP;
        Uni2Io io;
        io.xr = xrow_off[s];
        io.f = fpar[fi];
        io.poison = poison[s]; io.n = n[s]; io.live = live[s]; io.spare = S::D;
        auto p = [&](int i) ARS_ALWAYS_INLINE -> float& { return par[s][fi * TOTAL + i]; };
        U2::template step<k, DIAG>(ust, p, a, io, lacc[s]);
      });
    };
    auto hand_over = [&]() ARS_ALWAYS_INLINE {  // the finished group's accumulators become the parameters of its maps
#pragma unroll
      for (int s = 0; s < 2; ++s)
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE {
#pragma unroll
          for (int r = 0; r < 4; ++r) par[s][4 * decltype(t)::value + r] = acc[s][t][r];
        });
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) fpar[fi] = fnext[fi];
    };
    ars_for<NG>([&](auto g_) ARS_ALWAYS_INLINE {  // stream position GI holds feature group G (the last group first)
      constexpr int GI = decltype(g_)::value, G = S::G_ORD[GI], GPREV = S::G_ORD[GI > 0 ? GI - 1 : 0], NS = S::LS_OFF[GI + 1] - S::LS_OFF[GI];
      constexpr int SPQ0 = GI + 6 * S::LS_OFF[GI];  // this group's slice of SPQ (one leading entry per group)
      {  // the accumulators start at the bias (raw reads, their own wait: once per group)
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { arx2_raw_read<(NH * S::BIAS_STRIDE + (G * NT + decltype(t)::value) * 16) * 4>(acc[0][t], bias_off); });
        ars_for<FPL>([&](auto fi) ARS_ALWAYS_INLINE { fnext[fi] = __builtin_bit_cast(int, arx2_raw_read1i<(G * 4 * FPL + decltype(fi)::value) * 4>(fmap_off)); });
        if constexpr (NT == 6) ars_settle<0>(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[0][4], acc[0][5]);
        else if constexpr (NT == 3) ars_settle<0>(acc[0][0], acc[0][1], acc[0][2]);
        else ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { ars_settle<0>(acc[0][t]); });
        ars_for<FPL>([&](auto fi) ARS_ALWAYS_INLINE { arx2_tie(fnext[fi]); });  // (covered by the same wait: LDS operations complete in order)
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[1][t] = acc[0][t]; });
      }
      if constexpr (NS == 0 && GI > 0) spline(std::integral_constant<int, GPREV>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, SPT>{});
      ars_for<NS>([&](auto st_) ARS_ALWAYS_INLINE {
        constexpr int ST = decltype(st_)::value, LS = S::LS_OFF[GI] + ST, GS = NSTEPS_H + LS;
        constexpr int T0 = S::L_T0[LS], T1 = S::L_T1[LS], IP = S::L_IP[LS];
        constexpr bool TWO = T1 != 255;
        run_step(
            std::integral_constant<int, GS>{},
            [&](auto k_) ARS_ALWAYS_INLINE {
              constexpr int KT = decltype(k_)::value, QI = 6 * GS + KT, SI = SPQ0 + 6 * ST + KT;
              convert(std::integral_constant<int, S::CVQ[QI]>{}, std::integral_constant<int, S::CVQ[QI + 1]>{});
              if constexpr (GI > 0) spline(std::integral_constant<int, GPREV>{}, std::integral_constant<int, S::SPQ[SI]>{}, std::integral_constant<int, S::SPQ[SI + 1]>{});
            },
            acc[0][T0], acc[1][T0], acc[0][TWO ? T1 : T0], acc[1][TWO ? T1 : T0], in[0][IP], in[1][IP]);
      });
      hand_over();
    });
    if constexpr (NG > 0) spline(std::integral_constant<int, S::G_ORD[NG > 0 ? NG - 1 : 0]>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, SPT>{});

    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (live[s]) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
          if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D)
            *reinterpret_cast<f32x4*>(a.y + n[s] * a.ldy + it * 16 + 4 * q) = *reinterpret_cast<const f32x4*>(xrow_lds[s] + it * 16 + 4 * q);
      }
      if (a.ladj) {
        float l = lacc[s];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (live[s] && q == 0) a.ladj[n[s]] = a.accumulate ? a.ladj[n[s]] + l : l;
      }
    }
#ifdef ARX2_TRACE
    if (a.bin_out && blockIdx.x == 0 && wave == 0 && pass_no < 2) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (lane == 0) a.bin_out[pass_no * (NSTEPS + 1) + NSTEPS] = (int)(unsigned)now;
    }
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // the next pass overwrites the row image
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}

template <class S, typename Uni> static int arx2_launch(const ArArgs* in, int abi, int args_bytes, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;
  ArArgs a = *in;
  if (a.D != S::D || a.DIN != S::DIN || a.L != S::NH + 1 || a.act != S::ACT || a.sched || a.NG != S::NG || a.n_chunks != S::NCHUNK || a.l1rev) return ZK_EINVAL;
  a.n_tiles = (a.N + 127) / 128;
  a.xs = ((S::D + 3) / 4) * 4 + 4;
  if (S::D % 4 || a.ldy % 4 || ((uintptr_t)a.y % 16)) return ZK_EINVAL;
  a.xlds = 1;
  const int lds = (S::NR * S::CH * AR_TF + a.bias_floats + 1024 + 256 + 128 * a.xs) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  if ((a.bin_out != nullptr) != (a.knots_out != nullptr)) return ZK_EINVAL;
#if defined(ARX2_TRACE) || defined(ARX2_ONLY)  // probe builds: the product instantiation only (bin_out, if given, receives the trace)
  const void* fn = (const void*)arx2_kernel<S, Uni, false>;
#else
  const void* fn = a.bin_out ? (const void*)arx2_kernel<S, Uni, true> : (const void*)arx2_kernel<S, Uni, false>;
#endif
  hipError_t e = hipSuccess;
  {
    static std::mutex mu;
    static std::unordered_map<const void*, int> granted;
    std::lock_guard<std::mutex> lock(mu);
    int& g = granted[fn];
    if (g < lds) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      g = lds;
    }
  }
  const unsigned grid = (unsigned)(a.n_tiles < 256 ? a.n_tiles : 256);
  void* kargs[] = {&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

}  // namespace zk
