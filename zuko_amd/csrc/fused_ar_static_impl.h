// zuko_amd — static-shape twin of the fused autoregressive density kernel (fused_ar.hip), as a TEMPLATE over the block pattern
// of ONE conditioner:
//
//     y, log|dy/dx| = univariate(conditioner(cat(x, c))).call_and_ladj(x)      (zuko/flows/autoregressive.py:207-218)
//
// With the hidden units sorted by dependency count (zuko_amd/fused.py) the masks of MaskedMLP (zuko/nn.py:270-295) are block
// lower triangular at the kernel's 16 x 16 tile granularity, so the weight stream of a given (features, context, hidden widths,
// order, univariate map) has a fixed length and every tile's position in it is a compile-time constant.  The generic kernel finds
// that structure at run time (a wave-uniform bit test and branch per tile block, `s_waitcnt lgkmcnt(0)` at every join, ring
// position in a register); here the pass over a 16-sample wave tile is straight-line code: ring refills only where a position is a
// multiple of the chunk size, the A tiles of step s + 1 requested before the MFMAs of step s, the first tiles of the next feature
// group requested before the epilogue of the current one.  Same arithmetic in the same order per output as the generic kernel
// (asserted bit-identical in tests/test_gpu_flows.py).
//
// The pattern arrives as a `Shape` struct of constexpr tables GENERATED from the plan by zuko_amd/static_ar.py (one small
// translation unit per pattern: built ahead of time for the BASELINE.json configurations, compiled on first use — hipcc, ~10 s —
// for any other conditioner), which includes this header, instantiates zk::ars_kernel<Shape, Uni, ...> and exports one launcher.
//
//   Shape::D, DIN, NIT            features, conditioner inputs (features + context, multiple of 4), input tiles
//   Shape::NH, HT[l]              hidden layers and their 16-unit tiles; TMAX = activation tiles held in registers (multiple of 4)
//   Shape::NS[l], SOFF[l]         steps of hidden layer l: step = (out-group of 4 tiles, one input tile, 4-bit mask of the out tiles
//   Shape::S_OTG/S_IT/S_MASK      that hold non-zero weights), in stream order;  S_ALT: the input tile of the step under the
//                                 alternative first-layer pattern (HAS_ALT: a descending feature order mirrors the input tiles)
//   Shape::BASE[l], LAST_BASE     stream position (in tiles) where each layer starts (layers are padded to whole chunks)
//   Shape::NG, GOFF[g], G_IT      last layer: kept input tiles of every feature group, in stream order
//   Shape::ACT                    activation between the layers (code of zuko_amd/nn.py:_act_code; 1 = ReLU)
//   Shape::NCHUNK, WAVES, XLDS    stream length in chunks; wavefronts per workgroup (8: widths <= 256, two per SIMD; 4: widths <= 512,
//                                 one per SIMD); whether x / y rows are staged through a wave-private LDS image
#pragma once
#include "zk_ar_common.h"
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <utility>

namespace zk {

#define ARS_CH 24
#define ARS_NR 3
#ifndef ARX_ABL
#define ARX_ABL 0  // timing ablations of the operand-split kernel (scripts/split_ablate.py): 1 no DMA, 2 no MFMA, 3 no epilogue, 4 no barrier, 5 no LDS reads, 6 no conversions
#endif
#define ARS_ALWAYS_INLINE __attribute__((always_inline))

template <class F, int... I> __device__ __forceinline__ void ars_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void ars_for(F&& f) {
  if constexpr (N > 0) ars_for_impl(f, std::make_integer_sequence<int, N>{});
}

__host__ __device__ constexpr int ars_popc(unsigned m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u) + ((m >> 3) & 1u)); }

// ---- compile-time views of the generated tables ----------------------------------------------------------------------
template <class S> struct ArsPat {
  static constexpr int otg(int l, int s) { return S::S_OTG[S::SOFF[l] + s]; }
  static constexpr int it(int l, int s) { return S::S_IT[S::SOFF[l] + s]; }
  static constexpr int alt(int l, int s) { return S::S_ALT[S::SOFF[l] + s]; }
  static constexpr unsigned mask(int l, int s) { return S::S_MASK[S::SOFF[l] + s]; }
  static constexpr bool first_of_group(int l, int s) { return s == 0 || otg(l, s - 1) != otg(l, s); }
  static constexpr bool group_has_steps(int l, int g) {
    for (int s = 0; s < S::NS[l]; ++s)
      if (otg(l, s) == g) return true;
    return false;
  }
  static constexpr int pos(int l, int s) {  // tiles of layer l streamed before step s
    int n = 0;
    for (int i = 0; i < s; ++i) n += ars_popc(mask(l, i));
    return n;
  }
  static constexpr int n_last_steps() { return S::GOFF[S::NG]; }
};

template <int WAVES, int CH = ARS_CH, int NR = ARS_NR> struct ArRingS {
  static constexpr int PER = CH / WAVES;  // consecutive tiles a wave copies per chunk: one address, one M0 value, immediate offsets
  static constexpr int PIVOT = PER > 4 ? 4 : 0;  // (signed immediates -4096 .. +1024 around the wave's fifth tile reach six tiles)
  static_assert(PER * WAVES == CH && PER <= 6, "ring geometry");
  float* lds;
  const float* stream;
  unsigned cur_off;  // LDS byte address of the slot being read + lane * 16
  unsigned lds_off;  // LDS byte address of the ring
  int n_chunks, slot, load_chunk, load_slot, wave, lane;
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < PER) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, (I - PIVOT) * AR_TF * 4, 0);
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {
    const int b0 = wave * PER + PIVOT;
    if (ARX_ABL != 1) dma<0>(stream + ((size_t)load_chunk * CH + b0) * AR_TF + lane * 4, lds + (load_slot * CH + b0) * AR_TF);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == NR) ? 0 : load_slot + 1;
  }
  __device__ __forceinline__ void advance() {  // all waves, at the same (static) points of the pass
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NR - 2) * PER) : "memory");
    if (ARX_ABL != 4) __builtin_amdgcn_s_barrier();  // (not __syncthreads(): its fence is s_waitcnt vmcnt(0) and would drain the look-ahead DMAs)
    asm volatile("" ::: "memory");
    issue();
    slot = (slot + 1 == NR) ? 0 : slot + 1;
    cur_off = lds_off + (unsigned)(slot * CH * AR_TF * 4 + lane * 16);
  }
  // Position S inside the pass (static).  The read is issued from inline assembly and returns a RAW value: the compiler does
  // not know it is an LDS operation, so it inserts no wait for it — while a global_load_lds is in flight hipcc turns every
  // LDS wait into lgkmcnt(0), which would make the step wait for the tiles it has just requested for the NEXT step.  The
  // value becomes usable through ars_settle<N>() below, which waits until at most N younger LDS operations are outstanding
  // (LDS operations of a wave complete in order) and is the only consumer of the raw registers.
  template <int S> __device__ __forceinline__ f32x4 read() {
    if constexpr (S % CH == 0) advance();
    f32x4 v;
    if (ARX_ABL == 5) {
      asm volatile("v_mov_b32 %0, %1" : "=v"(v[0]) : "v"(cur_off));
      v[1] = v[2] = v[3] = v[0];
      return v;
    }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(cur_off), "n"((S % CH) * AR_TF * 4));
    return v;
  }
};

template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a0) : "n"(N)); }
template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0, f32x4& a1, f32x4& a2) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a0), "+v"(a1), "+v"(a2) : "n"(N));
}
template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N));
}
template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "n"(N));
}
template <int N, int NT> __device__ __forceinline__ void ars_settle_tiles(f32x4 (&w)[NT]) {
  if constexpr (NT == 1) ars_settle<N>(w[0]);
  else if constexpr (NT == 3) ars_settle<N>(w[0], w[1], w[2]);
  else if constexpr (NT == 6) ars_settle<N>(w[0], w[1], w[2], w[3], w[4], w[5]);
  else static_assert(NT == 1 || NT == 3 || NT == 6, "last-layer tile count");
}

extern __shared__ __attribute__((aligned(16))) float ars_lds[];

// one hidden layer: out = W in + bias over the steps of the generated pattern
template <class S, int L, class Ring> __device__ __forceinline__ void ars_hidden(Ring& ring, const float* bias_q, const f32x4 (&in)[S::TMAX], f32x4 (&out)[S::TMAX], bool rev) {
  typedef ArsPat<S> P;
  constexpr int NS = S::NS[L], BASE = S::BASE[L], NOG = (S::HT[L] + 3) / 4;
  // out-groups without a single weight tile (units that depend on nothing): bias only
  ars_for<NOG>([&](auto g_) ARS_ALWAYS_INLINE {
    constexpr int g = g_;
    if constexpr (!P::group_has_steps(L, g)) {
      ars_for<4>([&](auto t) ARS_ALWAYS_INLINE { out[g * 4 + t] = *reinterpret_cast<const f32x4*>(bias_q + (g * 4 + t) * 16); });
    }
  });
  if constexpr (NS > 0) {
    f32x4 a[2][4];
    {
      constexpr unsigned M0 = P::mask(L, 0);
      ars_for<4>([&](auto t) ARS_ALWAYS_INLINE {
        constexpr int tt = decltype(t)::value;
        if constexpr ((M0 >> tt) & 1u) a[0][tt] = ring.template read<BASE + ars_popc(M0 & ((1u << tt) - 1u))>();
      });
    }
    ars_for<NS>([&](auto s_) ARS_ALWAYS_INLINE {
      constexpr int s = s_, otg = P::otg(L, s), it = P::it(L, s), alt = P::alt(L, s);
      constexpr unsigned M = P::mask(L, s);
      if constexpr (P::first_of_group(L, s)) {
        ars_for<4>([&](auto t) ARS_ALWAYS_INLINE { out[otg * 4 + t] = *reinterpret_cast<const f32x4*>(bias_q + (otg * 4 + t) * 16); });  // accumulators start at the bias
      }
      if constexpr (s + 1 < NS) {
        constexpr unsigned MN = P::mask(L, s + 1);
        constexpr int PN = BASE + P::pos(L, s + 1);
        ars_for<4>([&](auto t) ARS_ALWAYS_INLINE {
          constexpr int tt = decltype(t)::value;
          if constexpr ((MN >> tt) & 1u) a[(s + 1) & 1][tt] = ring.template read<PN + ars_popc(MN & ((1u << tt) - 1u))>();
        });
        ars_settle<ars_popc(MN)>(a[s & 1][0], a[s & 1][1], a[s & 1][2], a[s & 1][3]);  // this step's tiles are in; only the next step's may be outstanding
      } else {
        ars_settle<0>(a[s & 1][0], a[s & 1][1], a[s & 1][2], a[s & 1][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 b;
      if constexpr (S::HAS_ALT && L == 0 && alt != it) {
        // the first layer's columns are in natural feature order: a descending feature order mirrors the input tiles an
        // out-group depends on (same count, same stream positions)
        const f32x4 up = in[it], down = in[alt];
        b = rev ? down : up;
      } else {
        b = in[it];
      }
      ars_for<4>([&](auto r) ARS_ALWAYS_INLINE {
        ars_for<4>([&](auto t) ARS_ALWAYS_INLINE {
          if constexpr ((M >> decltype(t)::value) & 1u) out[otg * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][t][(int)r], b[(int)r], out[otg * 4 + t], 0, 0, 0);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
}

template <class S, int L, class Ring, bool TRAIN> __device__ __forceinline__ void ars_hidden_stack(Ring& ring, const float* bias_lds, int q, f32x4 (&in)[S::TMAX], f32x4 (&out)[S::TMAX], bool rev,
                                                                                                 const ArArgs& a, int64_t n, bool live) {
  if constexpr (L < S::NH) {
    ars_hidden<S, L>(ring, bias_lds + L * S::BIAS_STRIDE + 4 * q, in, out, rev);
    constexpr int HTL = S::HT[L], TO = 4 * ((HTL + 3) / 4);
    if constexpr (S::ACT == 1) {
#pragma unroll
      for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];  // NaN stays NaN, as torch.relu
    } else if constexpr (S::ACT == 0) {
#pragma unroll
      for (int t = 0; t < TO; ++t) in[t] = out[t];
    } else {
      // ELU / tanh / SiLU / GELU / sigmoid / leaky ReLU: the same expressions as the generic kernel (act_f32), inside a loop the compiler
      // must not unroll over the activation's inline expansion (64-128 copies of tanhf / erff made the generic kernel's epilogue
      // instruction-cache bound)
#pragma unroll 1
      for (int rep = 0; rep < 1; ++rep) {
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) in[t][r] = act_f32(out[t][r], S::ACT);
      }
    }
    if constexpr (TRAIN) {
      if (live) {
#pragma unroll
        for (int t = 0; t < HTL; ++t) *reinterpret_cast<f32x4*>(a.act_out[L < 3 ? L : 2] + n * (HTL * 16) + t * 16 + 4 * q) = in[t];
      }
    }
    ars_hidden_stack<S, L + 1, Ring, TRAIN>(ring, bias_lds, q, in, out, rev, a, n, live);
  }
}

// TRAIN: conditioner only — the hidden activations and phi are stored for the backward pass (zuko_amd/train.py), the univariate
// map is not evaluated (the autograd graph applies it to phi itself).
template <class S, typename Uni, bool TRAIN> __global__ __launch_bounds__(64 * S::WAVES, S::WAVES == 8 ? 2 : 1) void ars_kernel(ArArgs a) {
  typedef ArsPat<S> P;
  typedef ArRingS<S::WAVES> Ring;
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL, WAVES = S::WAVES;
  constexpr int NG = S::NG;
  constexpr int NSTEP = P::n_last_steps();
  constexpr bool XLDS = S::XLDS;
  constexpr bool FID_REGS = NG * FPL <= 32;
  constexpr int DT = (S::D + 15) / 16;  // tiles that hold features
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const bool rev = a.l1rev != 0;

  Ring ring;
  float* bias_lds = ars_lds + ARS_NR * ARS_CH * AR_TF;
  ring.lds = ars_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < ARS_NR - 1; ++i) ring.issue();
  ring.slot = ARS_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ars_lds);
  ring.cur_off = ring.lds_off;

  for (int i = tid; i < a.bias_floats; i += 64 * WAVES) bias_lds[i] = a.bias[i];
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);  // same LDS layout as the generic kernel (fused_ar.hip)
  float* xr = reinterpret_cast<float*>(fmap_lds + 1024 + 256) + wave * 16 * a.xs + j * a.xs;
  for (int i = tid; i < NG * 4 * FPL; i += 64 * WAVES) fmap_lds[i] = a.featmap[i];
  __syncthreads();
  const float* bias_last = bias_lds + S::NH * S::BIAS_STRIDE;
  // feature ids of this lane's slots in every group: constant over the launch, kept in registers when they fit (a per-group LDS read
  // puts one exposed LDS round trip in front of the read of x that depends on it)
  int fids[FID_REGS ? NG * FPL : 1];
  if constexpr (FID_REGS) {
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) fids[i * FPL + fi] = fmap_lds[(i * 4 + q) * FPL + fi];
  }

  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * (16 * WAVES) + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    const float* xrow = a.x + nc * a.ldx;

    f32x4 in[S::TMAX], out[S::TMAX];
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((it + 1) * 16 <= S::DIN || it * 16 + 4 * q < S::DIN) v = *reinterpret_cast<const f32x4*>(xrow + it * 16 + 4 * q);
      in[it] = v;
    }
    // a NaN / inf input turns ALL parameters of its sample into NaN in the reference (x * 0 = NaN, zuko/nn.py:217-218)
    float poison = 0.f;
    {
      int bad = 0;
#pragma unroll
      for (int it = 0; it < S::NIT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= !(fabsf(in[it][r]) < __builtin_inff());
      bad |= __shfl_xor(bad, 16, 64);
      bad |= __shfl_xor(bad, 32, 64);
      if (bad) poison = __builtin_nanf("");
    }
    if constexpr (XLDS) {
#pragma unroll
      for (int it = 0; it < DT; ++it)
        if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(xr + it * 16 + 4 * q) = in[it];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }

    // ---- hidden layers ---------------------------------------------------------------------------------------------
    ars_hidden_stack<S, 0, Ring, TRAIN>(ring, bias_lds, q, in, out, rev, a, n, live);

    // ---- last layer + univariate transform, one group of 4 * FPL features at a time --------------------------------
    float lacc = 0.f;
    f32x4 w[2][NT];
    if constexpr (NSTEP > 0) {
      ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { w[0][t] = ring.template read<S::LAST_BASE + decltype(t)::value>(); });
    }
    ars_for<NG>([&](auto g_) ARS_ALWAYS_INLINE {
      constexpr int g = g_, ST0 = S::GOFF[g], GN = S::GOFF[g + 1] - S::GOFF[g];
      // operands of the epilogue are requested before the group's MFMAs: feature ids, x values and the bias from LDS
      int fid[FPL];
      float xin[FPL];
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) {
        if constexpr (FID_REGS) fid[fi] = fids[g * FPL + fi];
        else fid[fi] = fmap_lds[(g * 4 + q) * FPL + fi];
        const int fc = fid[fi] < 0 ? 0 : fid[fi];
        if constexpr (XLDS) xin[fi] = xr[fc];
        else xin[fi] = xrow[fc];
      }
      f32x4 acc[NT];  // the accumulators start at the bias
      {
        const float* bg = bias_last + (g * NT) * 16 + 4 * q;
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[t] = *reinterpret_cast<const f32x4*>(bg + t * 16); });
      }
      ars_for<GN>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int st = ST0 + decltype(i_)::value, it = S::G_IT[st];
        if constexpr (st + 1 < NSTEP) {
          ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { w[(st + 1) & 1][t] = ring.template read<S::LAST_BASE + (st + 1) * NT + decltype(t)::value>(); });
          ars_settle_tiles<NT, NT>(w[st & 1]);
        } else {
          ars_settle_tiles<0, NT>(w[st & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        ars_for<4>([&](auto r) ARS_ALWAYS_INLINE {
          ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[st & 1][t][(int)r], in[it][(int)r], acc[t], 0, 0, 0); });
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      float p[4 * NT];
      ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * t + r] = acc[t][r];
      });
      if constexpr (TRAIN) {
        // the reference multiplies every input by mask * W: a non-finite input makes ALL parameters of its sample NaN
#pragma unroll
        for (int fi = 0; fi < FPL; ++fi) {
          const int f = fid[fi];
          if (f >= 0 && live) {
            float* dst = a.phi_out + n * a.ldphi + f * TOTAL;
#pragma unroll
            for (int i = 0; i < TOTAL; ++i) dst[i] = p[fi * TOTAL + i] + poison;
          }
        }
      } else {
#pragma unroll
        for (int fi = 0; fi < FPL; ++fi) Uni::template poison<false>(p, fi * TOTAL, poison);
        auto ld = [&](int i) { return p[i]; };
#pragma unroll
        for (int fi = 0; fi < FPL; ++fi) {
          const int f = fid[fi];
          if (f >= 0) {
            float yv, lj;
            Uni::fwd(ld, fi * TOTAL, a, xin[fi], yv, lj);
            if constexpr (XLDS) xr[f] = yv;
            else if (live) a.y[n * a.ldy + f] = yv;
            lacc += lj;
          }
        }
      }
    });
    if constexpr (XLDS && !TRAIN) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (live) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
          if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(a.y + n * a.ldy + it * 16 + 4 * q) = *reinterpret_cast<const f32x4*>(xr + it * 16 + 4 * q);
      }
    }
    if (!TRAIN && a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}

// ---- the backward twin: dgrad through the hidden layers in one launch ------------------------------------------------------------
// g_{l-1} = (g_l W_l) * relu'(h_{l-1}) for the hidden layers of the conditioner (what autograd derives from zuko/nn.py:217-218 and the
// ReLUs between them), with the gradients in registers from layer to layer exactly as the activations are in the forward kernel: the
// chain is itself a masked MLP whose layer matrices are the TRANSPOSED sorted weights (block UPPER triangular: the same tile tables,
// generated from the transposed masks by zuko_amd/static_ar.py:chain_tables), whose "activation" multiplies by the sign of the saved
// forward activation, and whose every layer output is stored for the weight gradients.  Input: g of the LAST hidden layer's
// pre-activations [N, width] (the K = features x total product with the last layer's weight stays a stand-alone GEMM); outputs: g of
// every earlier hidden layer [N, width_l] and of the conditioner's input [N, DOUT].
template <class S, int L, class Ring> __device__ __forceinline__ void ars_dgrad_stack(Ring& ring, const float* zero_q, int q, f32x4 (&in)[S::TMAX], f32x4 (&out)[S::TMAX], const ArArgs& a,
                                                                                      int64_t n, int64_t nc, bool live) {
  if constexpr (L < S::NH) {
    ars_hidden<S, L>(ring, zero_q, in, out, false);
    constexpr int HTL = S::HT[L];
    if constexpr (L + 1 < S::NH) {
      const float* grow = a.gate[L] + nc * (HTL * 16) + 4 * q;
#pragma unroll
      for (int t = 0; t < HTL; ++t) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(grow + t * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][r] = out[t][r] * (h[r] > 0.f ? 1.f : 0.f);  // (a product, as autograd's: NaN gradients stay NaN)
      }
      if (live) {
#pragma unroll
        for (int t = 0; t < HTL; ++t) *reinterpret_cast<f32x4*>(a.act_out[L] + n * (HTL * 16) + t * 16 + 4 * q) = in[t];
      }
      ars_dgrad_stack<S, L + 1, Ring>(ring, zero_q, q, in, out, a, n, nc, live);
    } else if (live) {  // gradient w.r.t. the conditioner's input: module column order, DOUT columns
#pragma unroll
      for (int t = 0; t < HTL; ++t)
        if ((t + 1) * 16 <= S::DOUT || t * 16 + 4 * q < S::DOUT) *reinterpret_cast<f32x4*>(a.phi_out + n * a.ldphi + t * 16 + 4 * q) = out[t];
    }
  }
}

template <class S> __global__ __launch_bounds__(512, 2) void ars_dgrad_kernel(ArArgs a) {
  typedef ArRingS<8> Ring;
  static_assert(S::NH >= 1 && S::NH <= 4 && S::TMAX <= 16, "dgrad chain: up to three gated layers + the input layer, widths <= 256");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  Ring ring;
  ring.lds = ars_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < ARS_NR - 1; ++i) ring.issue();
  ring.slot = ARS_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ars_lds);
  ring.cur_off = ring.lds_off;
  float* zero_lds = ars_lds + ARS_NR * ARS_CH * AR_TF;  // "bias image" of a layer without bias
  for (int i = tid; i < S::TMAX * 16 + 16; i += 512) zero_lds[i] = 0.f;
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * 128 + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    const float* grow = a.x + nc * a.ldx;
    f32x4 in[S::TMAX], out[S::TMAX];
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) in[it] = *reinterpret_cast<const f32x4*>(grow + it * 16 + 4 * q);
    ars_dgrad_stack<S, 0, Ring>(ring, zero_lds + 4 * q, q, in, out, a, n, nc, live);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class S> static int ars_dgrad_launch(const ArArgs* in, int abi, int args_bytes, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;
  ArArgs a = *in;
  if (a.DIN != S::DIN || a.D != S::DOUT || a.L != S::NH || a.n_chunks != S::NCHUNK || !a.x || !a.phi_out || a.ldx % 4 || a.ldphi % 4 || ((uintptr_t)a.x % 16) ||
      ((uintptr_t)a.phi_out % 16))
    return ZK_EINVAL;
  for (int l = 0; l + 1 < S::NH; ++l)
    if (!a.gate[l] || !a.act_out[l] || ((uintptr_t)a.gate[l] % 16) || ((uintptr_t)a.act_out[l] % 16)) return ZK_EINVAL;
  a.n_tiles = (a.N + 127) / 128;
  const int lds = (ARS_NR * ARS_CH * AR_TF + S::TMAX * 16 + 16) * (int)sizeof(float);
  const void* fn = (const void*)ars_dgrad_kernel<S>;
  static bool granted = false;
  if (!granted) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    granted = true;
  }
  const unsigned grid = (unsigned)(a.n_tiles < 256 ? a.n_tiles : 256);
  void* kargs[] = {&a};
  hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

// Launch of one instantiation; `a` arrives filled by the main library's argument checks (csrc/fused_ar.hip: zk_ar_forward_static).
template <class S, typename Uni> static int ars_launch(const ArArgs* in, int abi, int args_bytes, int train, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;  // kernel built against another version of the library
  ArArgs a = *in;
  if (a.D != S::D || a.DIN != S::DIN || a.L != S::NH + 1 || a.act != S::ACT || a.sched || a.NG != S::NG || a.n_chunks != S::NCHUNK) return ZK_EINVAL;
  if (a.l1rev && !S::HAS_ALT) return ZK_EINVAL;
  if (a.bin_out || a.knots_out) return ZK_EINVAL;  // (no diagnostic instantiation: these kernels are bit-identical to the generic one, whose twin serves)
  if (train && (!S::TRAIN_OK || !a.phi_out)) return ZK_EINVAL;
  a.n_tiles = (a.N + 16 * S::WAVES - 1) / (16 * S::WAVES);
  a.xs = ((S::D + 3) / 4) * 4 + 4;
  const bool vec_ok = (S::D % 4 == 0) && (train || ((a.ldy % 4 == 0) && ((uintptr_t)a.y % 16 == 0)));
  if (S::XLDS != 0 && !vec_ok) return ZK_EINVAL;
  a.xlds = S::XLDS;
  const int lds = (ARS_NR * ARS_CH * AR_TF + a.bias_floats + 1024 + 256 + (S::XLDS ? S::WAVES * 16 * a.xs : 0)) * (int)sizeof(float);  // ring | bias | feature map | (skip words) | row tiles
  if (lds > 160 * 1024) return ZK_EINVAL;
  const void* fn = nullptr;
  if (train) {
    if constexpr (S::TRAIN_OK) fn = (const void*)ars_kernel<S, Uni, true>;
  } else {
    fn = (const void*)ars_kernel<S, Uni, false>;
  }
  if (!fn) return ZK_EINVAL;
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
  e = hipLaunchKernel(fn, dim3(grid), dim3(64 * S::WAVES), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

}  // namespace zk
