// zuko_amd — the 32-SAMPLE form of the operand-split fused autoregressive kernel (fused_ar_split_impl.h):
//
//     y, log|dy/dx| = univariate(conditioner(cat(x, c))).call_and_ladj(x)      (zuko/flows/autoregressive.py:207-218)
//
// Same formulation (every f32 operand as three bf16 numbers, six partial products, f32 accumulation), another matrix instruction and
// another occupancy model.  Measured on the MI355X (scripts/probes/mfma32_probe.hip): ONE wavefront issues v_mfma_f32_16x16x32_bf16 at
// 61 % of the bf16 peak at best — where the 8-wavefront kernel (two wavefronts per SIMD, 16 samples each) sits, whatever is done about
// its phases — but v_mfma_f32_32x32x16_bf16 at 85 %, with four to five other instructions behind every matrix instruction for free.
// Here one wavefront per SIMD (up to 512 registers) carries 32 samples, the N dimension of the 32 x 32 form:
//
//   * H^T = W X^T per layer: out TILE = 32 units (16 accumulator registers: lane (sample n = lane % 32, hb = lane / 32) holds units
//     8 g + 4 hb + r), in HALF-TILE = 16 units = eight of those registers in both lane halves — a finished tile is the next layer's
//     B operand of two half-tiles after an in-register conversion, no shuffle, no LDS round trip;
//   * a STEP is one in half-tile against the (up to) two out tiles of an out PAIR: six weight images (read once from LDS, half the
//     reads per sample of the 16-sample kernels), executed as six DUOS — one partial-product term each, two matrix instructions on two
//     accumulators; hidden layers walk their out pairs from the last to the first, and a finished pair is converted over the in
//     half-tiles that have just died (zuko_amd/static_ar.py: split3_tables deals the register slots);
//   * the non-matrix work is cut into small units — ReLU + three-way bf16 split of two activations; a micro-step of a univariate map of
//     the PREVIOUS feature group — and every duo carries a few of them (tables CVQ / SPQ), which the compiler is told to alternate with
//     the duo's matrix instructions (sched_group_barrier);
//   * nothing inside the pass may touch scratch memory or make the compiler wait for vector memory: a spill reload is a vector-memory
//     load, and waiting for it drains the weight ring's look-ahead DMAs (one to two thousand cycles each time).  Hence the raw LDS
//     accesses, the weight images in accumulation registers, and no branch around a map's result (the compiler would sink the whole map
//     into the conditional block).
#pragma once
#include "fused_ar_split_impl.h"

namespace zk {

// Raw LDS accesses.  While an LDS-DMA (global_load_lds) is in flight the compiler orders EVERY LDS access it knows about behind
// `s_waitcnt vmcnt(0)` (the DMA might write the same bytes) — which drains the weight ring's look-ahead, one to two thousand cycles each
// time.  The ring, the row image and the bias image are disjoint, so the accesses inside the pass are issued from inline assembly: a
// read returns a RAW value, usable only behind a covering `s_waitcnt lgkmcnt` that names it (ArRingS::read has the idiom); a write
// needs nothing — the LDS operations of a wavefront execute in order, so a later read of the same word sees it.
template <int OFF> __device__ __forceinline__ void arx3_raw_read(f32x4& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ float arx3_raw_read1(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF> __device__ __forceinline__ float arx3_raw_read1i(unsigned addr) {  // (one base register + an immediate: nothing per-group for the compiler to hoist and spill)
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void arx3_raw_write1(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void arx3_settle1(float& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ void arx3_tie(int& v) { asm volatile("" : "+v"(v)); }  // orders the uses of a raw value behind the wait in front of this statement

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
      st.x = arx3_raw_read1(io.xr + 4u * (unsigned)(io.f < 0 ? 0 : io.f));
      p(0) += io.poison;
      p(1) += io.poison;
    } else {
      arx3_settle1(st.x);
      float y, lj;  // (no branch: see Uni2Rqs)
      affine_fwd<float, MathFast>(p(0), p(1), a.ls, st.x, y, lj);
      arx3_raw_write1(io.xr + 4u * (unsigned)(io.f >= 0 ? io.f : io.spare), y);
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
      st.v = arx3_raw_read1(io.xr + 4u * (unsigned)(io.f < 0 ? 0 : io.f));  // (raw: settled by the first knot step, long after)
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
        arx3_settle1(st.v);
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
      arx3_raw_write1(io.xr + 4u * (unsigned)(io.f >= 0 ? io.f : io.spare), st.out);
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
// half-unit u of the pass-wide sequence: out pair at completion position u / 32, tile (u / 16) % 2 of the pair, value pair vp = (u / 2) % 8
// (accumulator registers 2 vp, 2 vp + 1 of the tile), half u % 2.  Registers 0 .. 7 of a tile are the k-slots of in half-tile 2 T, 8 .. 15
// of half-tile 2 T + 1.  Activation (ReLU / none), then the bf16 parts h, m, l exactly as arx_split computes them — first half: h and
// the remainder v - h (kept in `rem` for the second half); second half: m, l.
// ReLU on the integer pipe: max(bits, 0) maps every value with the sign bit set to +0 and leaves the others (NaN included) alone — one
// instruction instead of a compare / select pair through VCC.  It differs from `v < 0 ? 0 : v` for -0.0 (+0.0 here) and for NaNs with
// the sign bit set (zero here; the device's own NaNs are positive, and non-finite INPUTS poison the parameters explicitly).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float arx3_relu(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

template <int ACT, int VP, int HALF> __device__ __forceinline__ void arx3_convert(const f32x16& acc, ArxB& b, float (&rem)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    constexpr int v0 = 2 * VP;
    const int e = (v0 + i) % 8;
    if constexpr (HALF == 0) {
      float v = acc[v0 + i];
      if constexpr (ACT == 1) v = arx3_relu(v);
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

template <class S> struct Arx3Pat {
  static constexpr int n_hidden_steps() { return S::HS_OFF[S::NH]; }
  static constexpr int n_steps() { return S::HS_OFF[S::NH] + S::LS_OFF[S::NG3]; }
  static constexpr int layer_of(int gs) {  // hidden layer of global step gs (< n_hidden_steps())
    int l = 0;
    while (gs >= S::HS_OFF[l + 1]) ++l;
    return l;
  }
  static constexpr int step_images(int gs) {  // images of global step gs (hidden steps first, then the last layer's)
    if (gs >= n_steps()) return 0;
    if (gs < n_hidden_steps()) return S::H_T1[gs] == 255 ? 3 : 6;
    return S::L_T1[gs - n_hidden_steps()] == 255 ? 3 : 6;
  }
  static constexpr int step_pos(int gs) {  // stream position of its first image
    if (gs >= n_steps()) return 0;
    if (gs < n_hidden_steps()) return S::BASE[layer_of(gs)] + 3 * S::H_BLK[gs];
    return S::LAST_BASE + 3 * S::L_BLK[gs - n_hidden_steps()];
  }
  // Ring refills.  Duo q = 6 gs + k (k < 3) requests images [2 k, 2 k + 2) of step gs + 1; when one of them is the first image of a
  // chunk the ring moves on there (sync), and the six DMAs of the refill are issued one per duo in the six duos that follow.
  static constexpr bool syncs_at(int q) {
    const int gs = q / 6, k = q % 6;
    if (q < 0 || k >= 3 || gs + 1 >= n_steps()) return false;
    const int nn = step_images(gs + 1), first = 2 * k;
    for (int i = first; i < first + 2 && i < nn; ++i)
      if ((step_pos(gs + 1) + i) % S::CH == 0) return true;
    return false;
  }
  static constexpr int dma_piece(int q) {  // DMA of the pending refill to issue in duo q, or -1
    if (q < 6) return q;  // (the pass starts on a chunk boundary: its first step's images are requested, and the ring moved on, before duo 0)
    for (int i = 0; i < 6; ++i)
      if (syncs_at(q - 1 - i)) return i;
    return -1;
  }
  // bias tile accumulator `which` of the out pair of hidden step gs starts from (the pair's FIRST step), or -1
  static constexpr int bias_tile(int gs, int which) {
    if (gs >= n_hidden_steps() || !S::H_INIT[gs]) return -1;
    const int l = layer_of(gs), t = 2 * (S::H_T0[gs] / 2) + which;
    return t < S::HT32[l] ? t : -1;
  }
};

// the scheduling request of one duo: N matrix instructions, each followed by FILL VALU / transcendental instructions (0: no request)
template <int N, int FILL> __device__ __forceinline__ void arx3_pattern() {
  if constexpr (FILL > 0) {
    ars_for<N>([&](auto) ARS_ALWAYS_INLINE {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, FILL, 0);
    });
  }
}

#define ARX3_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), B, C, 0, 0, 0)
// one partial-product term of a step: term k of arx_block's sequence (h, m, l images of the weights against the parts of the activations)
template <int KT> __device__ __forceinline__ void arx3_term(const f32x4& ah, const f32x4& am, const f32x4& al, const ArxB& b, f32x16& c) {
  if (ARX_ABL == 2) {
    asm volatile("" ::"a"(ah), "a"(am), "a"(al));
    return;
  }
  if constexpr (KT == 0) ARX3_MFMA(al, b.h, c);
  else if constexpr (KT == 1) ARX3_MFMA(ah, b.l, c);
  else if constexpr (KT == 2) ARX3_MFMA(am, b.m, c);
  else if constexpr (KT == 3) ARX3_MFMA(am, b.h, c);
  else if constexpr (KT == 4) ARX3_MFMA(ah, b.m, c);
  else ARX3_MFMA(ah, b.h, c);
}

// raw read of 16 consecutive floats (one lane half's rows of a 32-row bias tile are four quads 8 floats apart) into an accumulator tile
template <int OFF> __device__ __forceinline__ void arx3_raw_read_tile(f32x16& dst, unsigned addr) {
  f32x4 q0, q1, q2, q3;
  asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
               : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
               : "v"(addr), "n"(OFF), "n"(OFF + 32), "n"(OFF + 64), "n"(OFF + 96));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
#pragma unroll
  for (int r = 0; r < 4; ++r) { dst[r] = q0[r]; dst[4 + r] = q1[r]; dst[8 + r] = q2[r]; dst[12 + r] = q3[r]; }
}

template <class S, typename Uni, bool DIAG> __global__ __launch_bounds__(256, 1) void arx3_kernel(ArArgs a) {
  typedef ArRingS<4, S::CH, S::NR> Ring;
  typedef typename Uni2Of<Uni>::type U2;
  typedef Arx3Pat<S> P3;
  static_assert(S::NH >= 1 && S::CH == 24 && S::NR == 3 && S::NOS == 3 && (S::ACT == 0 || S::ACT == 1), "32-sample operand-split kernel");
  constexpr int TOTAL = Uni::TOTAL, TP = S::TP, NT3 = S::NT3, FPL3 = S::FPL3;
  constexpr int NG = S::NG3, NH = S::NH;
  constexpr int SPT = FPL3 * U2::NSTEP;  // micro-steps of one group's univariate maps per lane: (feature slot, step)
  static_assert(SPT == S::SP_TOTAL && FPL3 * TP <= 16 * NT3 && TOTAL <= TP, "schedule tables were dealt for another step sequence");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ns = lane & 31, hb = lane >> 5;  // sample of the wave tile, lane half

  Ring ring;
  float* bias_lds = ars_lds + S::NR * S::CH * AR_TF;
  ring.lds = ars_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < S::NR - 1; ++i) ring.issue();
  ring.slot = S::NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ars_lds);
  ring.cur_off = ring.lds_off;

  // LDS: ring | bias image (hidden layers in sorted unit order, BIAS_STRIDE apart; last layer [group][tile][32 rows]) | feature map
  // [group][lane half][slot] | row image [128 samples][xs]
  for (int i = tid; i < S::BIAS_FLOATS; i += 256) bias_lds[i] = a.bias[i];
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + S::BIAS_FLOATS);
  for (int i = tid; i < NG * 2 * FPL3; i += 256) fmap_lds[i] = a.featmap[i];
  // Lane-dependent LDS addresses are RECOMPUTED from the lane id where they are used (two or three instructions) instead of being kept in
  // registers across the pass: the compiler would spill such long-lived values, and reloading a spill waits for vector memory — which
  // drains the ring's look-ahead DMAs.  (An opaque copy of the lane id keeps it from hoisting the sums back out.)
  auto lane_now = [&]() ARS_ALWAYS_INLINE {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  auto bias_off = [&]() ARS_ALWAYS_INLINE { return ring.lds_off + (unsigned)((S::NR * S::CH * AR_TF + 4 * (lane_now() >> 5)) * 4); };                    // this lane half's first quad of a bias tile; + tile / quad offsets as immediates
  auto fmap_off = [&]() ARS_ALWAYS_INLINE { return ring.lds_off + (unsigned)((S::NR * S::CH * AR_TF + S::BIAS_FLOATS + (lane_now() >> 5) * FPL3) * 4); };  // this lane half's slots of group 0; + group offsets as immediates
  auto xrow_off = [&]() ARS_ALWAYS_INLINE { return ring.lds_off + (unsigned)((S::NR * S::CH * AR_TF + S::BIAS_FLOATS + NG * 2 * FPL3 + (wave * 32 + (lane_now() & 31)) * a.xs) * 4); };
  auto xrow_ptr = [&]() ARS_ALWAYS_INLINE { return reinterpret_cast<float*>(fmap_lds + NG * 2 * FPL3) + (wave * 32 + (lane_now() & 31)) * a.xs; };
  __syncthreads();

  int pass_no = 0;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x, ++pass_no) {
    auto row_now = [&]() ARS_ALWAYS_INLINE -> int64_t { return tile * 128 + wave * 32 + (lane_now() & 31); };  // this lane's sample (recomputed: see above)
    const int64_t n = row_now();
    const bool live = n < a.N;
    float poison = 0.f;
    ArxB in[S::NSLOT];
    f32x16 out[2 * S::NOS];
    {
      const float* xrow = a.x + (live ? n : a.N - 1) * a.ldx;
      float* xrow_lds = xrow_ptr();
      int bad = 0;
      // in half-tile H of the input: k-slots (kg = hb, i) <-> columns 16 H + 8 (i / 4) + 4 hb + i % 4: two float4 loads
      ars_for<S::NIH0>([&](auto h_) ARS_ALWAYS_INLINE {
        constexpr int H = decltype(h_)::value;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (16 * H + 4 * hb < S::DIN) lo = *reinterpret_cast<const f32x4*>(xrow + 16 * H + 4 * hb);
        if (16 * H + 8 + 4 * hb < S::DIN) hi = *reinterpret_cast<const f32x4*>(xrow + 16 * H + 8 + 4 * hb);
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= !(fabsf(lo[r]) < __builtin_inff()) | !(fabsf(hi[r]) < __builtin_inff());
        if (16 * H + 4 * hb < S::D) *reinterpret_cast<f32x4*>(xrow_lds + 16 * H + 4 * hb) = lo;
        if (16 * H + 8 + 4 * hb < S::D) *reinterpret_cast<f32x4*>(xrow_lds + 16 * H + 8 + 4 * hb) = hi;
        arx_split(lo, hi, in[S::X_SLOT[H]]);
      });
      // a NaN / inf input turns ALL parameters of its sample into NaN in the reference (x * 0 = NaN, zuko/nn.py:217-218)
      bad |= __shfl_xor(bad, 32, 64);
      if (bad) poison = __builtin_nanf("");
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    f32x4 w[2][6];   // the six images of a step, double-buffered (raw until settled; accumulation registers)
    float cvrem[2];  // remainders v - h of the conversion unit whose second half is still to come
    // conversion half-units [LO, HI) of the pass-wide sequence (tables CP_*: the pair's accumulator slot, its tiles, the in slots it becomes)
    auto convert = [&](auto lo_, auto hi_) ARS_ALWAYS_INLINE {
      constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
      if (ARX_ABL == 7) return;
      ars_for<HI - LO>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int u = LO + decltype(i_)::value, gp = u / 32, tt = (u / 16) % 2, vp = (u / 2) % 8, half = u % 2;
        constexpr int os = S::CP_OS[gp], is = S::CP_IS[4 * gp + 2 * tt + vp / 4];
        if constexpr (is != 255) arx3_convert<S::ACT, vp, half>(out[2 * os + tt], in[is], cvrem);
      });
    };
    auto fetch = [&](auto gs_, auto first_, auto count_) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value, FIRST = decltype(first_)::value, COUNT = decltype(count_)::value;
      ars_for<COUNT>([&](auto i_) ARS_ALWAYS_INLINE { w[GS & 1][FIRST + decltype(i_)::value] = ring.template read_acc_nodma<P3::step_pos(GS) + FIRST + decltype(i_)::value>(); });
    };
    constexpr int NSTEPS_H = P3::n_hidden_steps(), NSTEPS = P3::n_steps();
    if constexpr (NSTEPS > 0) fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, P3::step_images(0)>{});
    // one step: settle its images, six duos; the first three also request the images of the next step (two each)
    auto run_step = [&](auto gs_, auto&& duo_extra, f32x16& c0, f32x16& c1, const ArxB& b) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value, BUF = GS & 1, NI = P3::step_images(GS);
      constexpr bool TWO = NI == 6;
#ifdef ARX3_TRACE  // probe build (scripts/arx2_trace.py): shader clock at the start of every step of workgroup 0 / wavefront 0, first two passes
      if (a.bin_out && blockIdx.x == 0 && wave == 0 && pass_no < 2) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) a.bin_out[pass_no * (NSTEPS + 1) + GS] = (int)(unsigned)now;
      }
#endif
      if constexpr (TWO) ars_settle_acc<0>(w[BUF][0], w[BUF][1], w[BUF][2], w[BUF][3], w[BUF][4], w[BUF][5]);
      else ars_settle_acc<0>(w[BUF][0], w[BUF][1], w[BUF][2]);
      __builtin_amdgcn_sched_barrier(0);
      ars_for<6>([&](auto k_) ARS_ALWAYS_INLINE {
        constexpr int KT = decltype(k_)::value;
        if constexpr (GS + 1 < NSTEPS && KT < 3) {
          constexpr int NN = P3::step_images(GS + 1);
          constexpr int FIRST = 2 * KT, COUNT = FIRST >= NN ? 0 : (FIRST + 2 <= NN ? 2 : NN - FIRST);
          if constexpr (COUNT > 0) fetch(std::integral_constant<int, GS + 1>{}, std::integral_constant<int, FIRST>{}, std::integral_constant<int, COUNT>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        duo_extra(k_);
        if constexpr (P3::dma_piece(6 * GS + KT) >= 0) ring.template piece<(P3::dma_piece(6 * GS + KT) >= 0 ? P3::dma_piece(6 * GS + KT) : 0)>();
        arx3_term<KT>(w[BUF][0], w[BUF][1], w[BUF][2], b, c0);
        if constexpr (TWO) arx3_term<KT>(w[BUF][3], w[BUF][4], w[BUF][5], b, c1);
        arx3_pattern<TWO ? 2 : 1, S::FILL>();
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    // ---- hidden layers: out pairs from the last to the first ----------------------------------------------------------------------------
    ars_for<NSTEPS_H>([&](auto gs_) ARS_ALWAYS_INLINE {
      constexpr int GS = decltype(gs_)::value, L = P3::layer_of(GS);
      constexpr int T0 = S::H_T0[GS], T1 = S::H_T1[GS], OS = S::H_OS[GS];
      constexpr bool TWO = T1 != 255;
      if constexpr (S::H_INIT[GS]) {  // a pair's first step: both accumulators start at the bias (the slot's previous tenant is converted: split3_tables)
        ars_for<2>([&](auto w_) ARS_ALWAYS_INLINE {
          constexpr int T = P3::bias_tile(GS, decltype(w_)::value);
          if constexpr (T >= 0) arx3_raw_read_tile<(L * S::BIAS_STRIDE + T * 32) * 4>(out[2 * OS + decltype(w_)::value], bias_off());
        });
      }
      run_step(
          std::integral_constant<int, GS>{},
          [&](auto k_) ARS_ALWAYS_INLINE {
            constexpr int KT = decltype(k_)::value, QI = 6 * GS + KT;
            convert(std::integral_constant<int, S::CVQ[QI]>{}, std::integral_constant<int, S::CVQ[QI + 1]>{});
          },
          out[2 * OS + (T0 & 1)], out[2 * OS + ((TWO ? T1 : T0) & 1)], in[S::H_IN[GS]]);
    });

    // ---- last layer + univariate maps: group g + 1 accumulates while the maps of group g are evaluated ------------------------------
    float lacc = 0.f;
    f32x16 acc[NT3];        // tiles of the group that accumulates
    float par[16 * NT3];    // parameters of the group whose maps are being evaluated (the previous group's accumulators)
    int fnext[FPL3], fpar[FPL3];  // feature ids of this lane's slots in the accumulating group / in the group whose maps are evaluated
    typename U2::State ust;
    auto spline = [&](auto lo_, auto hi_) ARS_ALWAYS_INLINE {  // micro-steps [LO, HI) of the group whose parameters are in `par`
      constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
      if (ARX_ABL == 8) return;
      ars_for<HI - LO>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int u = LO + decltype(i_)::value, fi = u / U2::NSTEP, k = u % U2::NSTEP;
        Uni2Io io;
        if constexpr (k == 0 || k == U2::NSTEP - 1) io.xr = xrow_off();  // (the steps that touch the row image)
        else io.xr = 0;
        io.f = fpar[fi];
        io.poison = poison; io.spare = S::D;
        if constexpr (DIAG && k == U2::NSTEP - 1) { io.n = row_now(); io.live = io.n < a.N; }
        else { io.n = 0; io.live = false; }
        auto p = [&](int i) ARS_ALWAYS_INLINE -> float& { return par[fi * TP + i]; };
        U2::template step<k, DIAG>(ust, p, a, io, lacc);
      });
    };
    auto hand_over = [&]() ARS_ALWAYS_INLINE {  // the finished group's accumulators become the parameters of its maps
      ars_for<NT3>([&](auto t) ARS_ALWAYS_INLINE {
#pragma unroll
        for (int r = 0; r < 16; ++r) par[16 * decltype(t)::value + r] = acc[t][r];
      });
#pragma unroll
      for (int fi = 0; fi < FPL3; ++fi) fpar[fi] = fnext[fi];
    };
    ars_for<NG>([&](auto g_) ARS_ALWAYS_INLINE {  // stream position GI holds feature group G (the last group first)
      constexpr int GI = decltype(g_)::value, G = S::G_ORD[GI], NS = S::LS_OFF[GI + 1] - S::LS_OFF[GI];
      constexpr int SPQ0 = GI + 6 * S::LS_OFF[GI];  // this group's slice of SPQ (one leading entry per group)
      // the accumulators start at the bias; this lane's feature ids of the group (raw reads, their own wait: once per group)
      ars_for<NT3>([&](auto t) ARS_ALWAYS_INLINE { arx3_raw_read_tile<(S::BIAS_LAST + (G * NT3 + decltype(t)::value) * 32) * 4>(acc[t], bias_off()); });
      ars_for<FPL3>([&](auto fi) ARS_ALWAYS_INLINE { fnext[fi] = __builtin_bit_cast(int, arx3_raw_read1i<(G * 2 * FPL3 + decltype(fi)::value) * 4>(fmap_off())); });
      {
        float t0 = __builtin_bit_cast(float, fnext[0]);
        arx3_settle1(t0);
        fnext[0] = __builtin_bit_cast(int, t0);
      }
      ars_for<FPL3>([&](auto fi) ARS_ALWAYS_INLINE { arx3_tie(fnext[fi]); });  // (covered by the same wait: LDS operations complete in order)
      if constexpr (NS == 0 && GI > 0) spline(std::integral_constant<int, 0>{}, std::integral_constant<int, SPT>{});
      ars_for<NS>([&](auto st_) ARS_ALWAYS_INLINE {
        constexpr int ST = decltype(st_)::value, LS = S::LS_OFF[GI] + ST, GS = NSTEPS_H + LS;
        constexpr int T0 = S::L_T0[LS], T1 = S::L_T1[LS];
        constexpr bool TWO = T1 != 255;
        run_step(
            std::integral_constant<int, GS>{},
            [&](auto k_) ARS_ALWAYS_INLINE {
              constexpr int KT = decltype(k_)::value, QI = 6 * GS + KT, SI = SPQ0 + 6 * ST + KT;
              convert(std::integral_constant<int, S::CVQ[QI]>{}, std::integral_constant<int, S::CVQ[QI + 1]>{});
              if constexpr (GI > 0) spline(std::integral_constant<int, S::SPQ[SI]>{}, std::integral_constant<int, S::SPQ[SI + 1]>{});
            },
            acc[T0], acc[TWO ? T1 : T0], in[S::L_IN[LS]]);
      });
      hand_over();
    });
    if constexpr (NG > 0) spline(std::integral_constant<int, 0>{}, std::integral_constant<int, SPT>{});

    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {  // row n leaves as 16-byte pieces: the two lane halves take alternate pieces
      constexpr int NPIECE = S::D / 4;
      const int64_t n2 = row_now();
      const bool live2 = n2 < a.N;
      const float* xrow_lds = xrow_ptr();
      const int hb2 = lane_now() >> 5;
      if (live2) {
#pragma unroll
        for (int i = 0; i < (NPIECE + 1) / 2; ++i) {
          const int pc = 2 * i + hb2;
          if (pc < NPIECE) *reinterpret_cast<f32x4*>(a.y + n2 * a.ldy + 4 * pc) = *reinterpret_cast<const f32x4*>(xrow_lds + 4 * pc);
        }
      }
      if (a.ladj) {
        float l = lacc;
        l += __shfl_xor(l, 32, 64);
        if (live2 && hb2 == 0) a.ladj[n2] = a.accumulate ? a.ladj[n2] + l : l;
      }
    }
#ifdef ARX3_TRACE
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

template <class S, typename Uni> static int arx3_launch(const ArArgs* in, int abi, int args_bytes, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;
  ArArgs a = *in;
  if (a.D != S::D || a.DIN != S::DIN || a.L != S::NH + 1 || a.act != S::ACT || a.sched || a.NG != S::NG3 || a.n_chunks != S::NCHUNK || a.l1rev || a.bias_floats != S::BIAS_FLOATS) return ZK_EINVAL;
  a.n_tiles = (a.N + 127) / 128;
  a.xs = ((S::D + 3) / 4) * 4 + 4;
  if (S::D % 4 || a.ldy % 4 || ((uintptr_t)a.y % 16)) return ZK_EINVAL;
  a.xlds = 1;
  const int lds = (S::NR * S::CH * AR_TF + S::BIAS_FLOATS + S::NG3 * 2 * S::FPL3 + 128 * a.xs) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  if ((a.bin_out != nullptr) != (a.knots_out != nullptr)) return ZK_EINVAL;
#if defined(ARX3_TRACE) || defined(ARX3_ONLY)  // probe builds: the product instantiation only (bin_out, if given, receives the trace)
  const void* fn = (const void*)arx3_kernel<S, Uni, false>;
#else
  const void* fn = a.bin_out ? (const void*)arx3_kernel<S, Uni, true> : (const void*)arx3_kernel<S, Uni, false>;
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
