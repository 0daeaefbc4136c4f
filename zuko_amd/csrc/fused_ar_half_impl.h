// zuko_amd — the TWO-PART operand split of the static-shape fused autoregressive kernel (inference forward):
//
//     y, log|dy/dx| = univariate(conditioner(cat(x, c))).call_and_ladj(x)      (zuko/flows/autoregressive.py:207-218, zuko/nn.py:217-218)
//
// fused_ar_split_impl.h writes every f32 operand as three bf16 numbers and needs SIX matrix instructions per 16 x 32 weight block.  Round 6
// measured that on a SIMD whose two wavefronts keep the matrix pipe full, vector instructions add to the matrix time wherever they stand
// (profiles/r06/headline.md): the launch is matrix time + vector issue time + waits, and only FEWER instructions make it shorter.  Here an
// operand is the sum of TWO f16 numbers, h = f16(v), l = f16(v - h) (11 + 11 significant bits: |v - h - l| <= 2^-22 |v| while no part is
// subnormal; the subtraction is exact in f32), and a product a b is the three partial products down to 2^-11 relative size
//
//     a_h b_l + a_l b_h + a_h b_h        (dropped: a_l b_l <= 2^-22 |a b|)
//
// on v_mfma_f32_16x16x32_f16 (f16 x f16 is exact in f32; f32 accumulation): HALF the matrix instructions, two thirds of the weight stream,
// LDS reads and conversions.  f16 has 5 exponent bits, so both operands are brought into its range by POWERS OF TWO (exact):
//   weights      layer l is stored as W_l 2^ew_l with max |W_l| 2^ew_l in [2^14, 2^15) (host: zuko_amd/fused.py, zk_gather_split_f16);
//   activations  every SAMPLE's input vector of a layer is scaled by 2^ea with max_k |a_k| 2^ea in [2^14, 2^15) before it is split (a lane
//                holds values of ONE sample; its four lanes agree on ea through two shuffles);
// and the accumulator returns through ONE fma:  out = fma(acc, 2^-(ew_l + ea), bias)  — products and sums before it carry no rounding but
// the f32 accumulation's.  Values more than 2^18 below their vector's maximum lose relative (not absolute) precision: an element's absolute
// error stays below 2^-40 of the vector's maximum.  Exponent bookkeeping: arh_scale.  Measured against float64 next to the reference's own f32 evaluation (random-init, x30
// "trained", 2^-120 / 2^100 scaled weights: scripts/split_scheme_emulation.py, tests/test_gpu_flows.py): the same error as the f32 path;
// weights whose magnitudes spread over more than the f16 range WITHIN a layer keep the three-part kernel (zuko_amd/fused.py: eligibility).
// A hidden value that overflows f32 becomes NaN for its sample, as in fused_ar_split_impl.h (inf - inf in the low part).
//
// Layout, ring, raw reads and counted waits are those of fused_ar_split_impl.h with TWO 1 KiB images (h, l) per block.
#pragma once
#include "fused_ar_split_impl.h"

#ifndef ARH_LOOK
#define ARH_LOOK 1  // blocks of weight images requested ahead of the matrix instructions that consume them
#endif
#ifndef ARH_RELU_INT
#define ARH_RELU_INT 0  // probe builds: 1 = ReLU as an integer max (one instruction; a NaN with the sign bit set would become 0)
#endif

namespace zk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ArhB {  // B operand of one pair of activation tiles
  f16x8 h, l;
};

// power-of-two scale of a sample: s = 2^ea with amax * s in [2^14, 2^15) for amax in [2^-75, 2^105]; ea stays in [-90, 90] so that the accumulator's
// descale factor 2^-(ew + ea) (ew in [-25, 35]: zuko_amd/fused.py, half_scales) is a normal f32 number.  Beyond: a sample whose largest magnitude
// exceeds 2^105 (4e31) overflows f16 and becomes NaN — as a non-finite value does (inf - inf in the low part) —, smaller ones than 2^-75 lose
// relative precision (absolute error below 2^-100).  amax = 0: zeros stay zeros.
__device__ __forceinline__ void arh_scale(float amax, float& s, float& inv_s) {
  amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  int e = __builtin_amdgcn_frexp_expf(amax);  // amax = f 2^e, f in [0.5, 1); 0 for zero / inf / NaN
  int ea = 15 - e;
  ea = ea > 90 ? 90 : (ea < -90 ? -90 : ea);
  s = __builtin_amdgcn_ldexpf(1.0f, ea);
  inv_s = __builtin_amdgcn_ldexpf(1.0f, -ea);
}

__device__ __forceinline__ void arh_split(const f32x4& lo, const f32x4& hi, float s, ArhB& b) {
  if (ARX_ABL == 6) {
    b.h = __builtin_bit_cast(f16x8, lo); b.l = __builtin_bit_cast(f16x8, hi);
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (e < 4 ? lo[e] : hi[e - 4]) * s;
    const _Float16 h = (_Float16)v;
    const float r = v - (float)h;
    b.h[e] = h;
    b.l[e] = (_Float16)r;
  }
}

#define ARH_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), B, C, 0, 0, 0)
// the three partial products of one block, smallest first (a[0] = h, a[1] = l image of the weights)
__device__ __forceinline__ void arh_block(const f32x4 (&a)[2], const ArhB& b, f32x4& c) {
  if (ARX_ABL == 2) {
    asm volatile("" ::"v"(a[0]), "v"(a[1]));
    return;
  }
  ARH_MFMA(a[1], b.h, c);
  ARH_MFMA(a[0], b.l, c);
  ARH_MFMA(a[0], b.h, c);
}

__device__ __forceinline__ void arh_touch(f32x4& v) { asm volatile("" : "+v"(v)); }  // a raw-read register becomes usable HERE (behind the counted wait that settled it)
template <int N> __device__ __forceinline__ void arh_settle(f32x4& a0, f32x4& a1) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(N)); }
template <int N> __device__ __forceinline__ void arh_settle(f32x4& a0, f32x4& a1, f32x4& a2) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a0), "+v"(a1), "+v"(a2) : "n"(N)); }

// one hidden layer: out = fma(W' in', d, bias) over the blocks of the generated pattern (W', in': the scaled operands; d = 2^-(ew + ea))
template <class S, int L, class Ring> __device__ __forceinline__ void arh_hidden(Ring& ring, const float* bias_q, const ArhB (&in)[S::TMAX / 2], f32x4 (&out)[S::TMAX], float d) {
  typedef ArxPat<S> P;
  constexpr int NB = S::NB[L], BASE = S::BASE[L], HTL = S::HT[L];
  ars_for<HTL>([&](auto t_) ARS_ALWAYS_INLINE {
    constexpr int t = t_;
    if constexpr (!P::tile_has_blocks(L, t)) out[t] = *reinterpret_cast<const f32x4*>(bias_q + t * 16);  // units that depend on nothing: bias only
  });
  if constexpr (NB > 0) {
    constexpr int LOOK = ARH_LOOK < NB ? ARH_LOOK : NB;
    f32x4 a[LOOK + 1][2];
    f32x4 bs;  // the out tile's bias: a raw read in front of the look-ahead request of the tile's LAST block, whose counted wait settles it
    f32x4 acc;
    const unsigned bias_addr = arx_lds_addr(bias_q);
    ars_for<LOOK>([&](auto b_) ARS_ALWAYS_INLINE {
      constexpr int b = b_;
      ars_for<2>([&](auto p) ARS_ALWAYS_INLINE { a[b][p] = ring.template read<BASE + 2 * b + decltype(p)::value>(); });
    });
    ars_for<NB>([&](auto s_) ARS_ALWAYS_INLINE {
      constexpr int s = s_, ot = P::ot(L, s), ip = P::ip(L, s), cur = s % (LOOK + 1);
      constexpr bool first_of_tile = (s == 0 || P::ot(L, s - 1) != ot), last_of_tile = (s + 1 == NB || P::ot(L, s + 1) != ot);
      if constexpr (first_of_tile) acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (last_of_tile) bs = arx_lds_raw<ot * 64>(bias_addr);
      if constexpr (s + LOOK < NB) {
        constexpr int nx = (s + LOOK) % (LOOK + 1);
        ars_for<2>([&](auto p) ARS_ALWAYS_INLINE { a[nx][p] = ring.template read<BASE + 2 * (s + LOOK) + decltype(p)::value>(); });
      }
      constexpr int ahead = (s + LOOK < NB ? LOOK : NB - 1 - s);  // blocks behind this one whose images may still be outstanding
      // (a tile's last block: only THIS step's request is younger than the bias read — the wait that settles it also settles the blocks in between)
      if constexpr (last_of_tile) arh_settle<(s + LOOK < NB ? 2 : 0)>(a[cur][0], a[cur][1], bs);
      else arh_settle<2 * ahead>(a[cur][0], a[cur][1]);
      if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
      arh_block(a[cur], in[ip], acc);
      if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
      if constexpr (last_of_tile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[ot][r] = __builtin_fmaf(acc[r], d, bs[r]);
      }
    });
  }
}

template <class S, int L, class Ring> __device__ __forceinline__ void arh_hidden_stack(Ring& ring, const float* bias_lds, int q, ArhB (&in)[S::TMAX / 2], f32x4 (&out)[S::TMAX], const ArArgs& a,
                                                                                       float& inv_s) {
  if constexpr (L < S::NH) {
    arh_hidden<S, L>(ring, bias_lds + L * S::BIAS_STRIDE + 4 * q, in, out, a.wdescale[L] * inv_s);
    constexpr int HTL = S::HT[L];
    float amax = 0.f;
    if constexpr (S::ACT == 1) {
#pragma unroll
      for (int t = 0; t < HTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (ARH_RELU_INT) out[t][r] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, out[t][r]), 0));
          else out[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];  // NaN stays NaN, as torch.relu
          amax = fmaxf(amax, out[t][r]);                  // (non-negative after the ReLU; a NaN is skipped here and poisons through the split)
        }
    } else {
      if constexpr (S::ACT != 0) {
#pragma unroll 1
        for (int rep = 0; rep < 1; ++rep) {
#pragma unroll
          for (int t = 0; t < HTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[t][r] = act_f32(out[t][r], S::ACT);
        }
      }
#pragma unroll
      for (int t = 0; t < HTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(out[t][r]));
    }
    float s;
    arh_scale(amax, s, inv_s);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < (HTL + 1) / 2; ++p) arh_split(out[2 * p], 2 * p + 1 < HTL ? out[2 * p + 1] : zero, s, in[p]);
    arh_hidden_stack<S, L + 1, Ring>(ring, bias_lds, q, in, out, a, inv_s);
  }
}

// DIAG: the diagnostic twin of the product launch (also writes the bin index the spline USED and the knots it searched), as arx_kernel's.
template <class S, typename Uni, bool DIAG = false> __global__ __launch_bounds__(64 * S::WAVES, S::OCC) void arh_kernel(ArArgs a) {
  typedef ArRingS<S::WAVES, S::CH, S::NR> Ring;
  static_assert(S::WAVES == 8 && S::NR == 3 && S::TMAX <= 16 && S::TMAX % 2 == 0 && S::OCC == 2, "two-part split kernels: widths <= 256, two wavefronts per SIMD");
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL, WAVES = S::WAVES;
  constexpr int NG = S::NG;
  constexpr int NSTEP = S::GOFF[NG];  // (group, in pair) steps of the last layer, NT blocks each
  constexpr bool XLDS = S::XLDS;
  constexpr bool FID_REGS = NG * FPL <= 32;
  constexpr int DT = (S::D + 15) / 16;
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

  for (int i = tid; i < a.bias_floats; i += 64 * WAVES) bias_lds[i] = a.bias[i];
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);  // same LDS layout as the f32 kernels
  float* xr = reinterpret_cast<float*>(fmap_lds + 1024 + 256) + wave * 16 * a.xs + j * a.xs;
  for (int i = tid; i < NG * 4 * FPL; i += 64 * WAVES) fmap_lds[i] = a.featmap[i];
  __syncthreads();
  const float* bias_last = bias_lds + S::NH * S::BIAS_STRIDE;
  const unsigned bias_last_addr = arx_lds_addr(bias_last + 4 * q);
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

    ArhB in[S::TMAX / 2];
    f32x4 out[S::TMAX];
    float poison = 0.f;
    float inv_s;  // 2^-ea of the operands `in` currently holds
    {
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
      float amax = 0.f;
#pragma unroll
      for (int it = 0; it < S::NIT; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bad |= !(fabsf(xin[it][r]) < __builtin_inff());
          amax = fmaxf(amax, fabsf(xin[it][r]));
        }
      bad |= __shfl_xor(bad, 16, 64);
      bad |= __shfl_xor(bad, 32, 64);
      if (bad) poison = __builtin_nanf("");
      if constexpr (XLDS) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
          if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(xr + it * 16 + 4 * q) = xin[it];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      float s;
      arh_scale(amax, s, inv_s);
#pragma unroll
      for (int p = 0; p < (S::NIT + 1) / 2; ++p) arh_split(xin[2 * p], xin[2 * p + 1], s, in[p]);
    }

    // ---- hidden layers ---------------------------------------------------------------------------------------------
    arh_hidden_stack<S, 0, Ring>(ring, bias_lds, q, in, out, a, inv_s);

    // ---- last layer + univariate transform, one group of 4 * FPL features at a time --------------------------------
    const float dl = a.wdescale[S::NH] * inv_s;
    float lacc = 0.f;
    constexpr int NBL = NSTEP * NT;                          // blocks of the last layer
    constexpr int LOOKL = ARH_LOOK < NBL ? ARH_LOOK : NBL;
    f32x4 w[LOOKL + 1][2];
    ars_for<LOOKL>([&](auto b_) ARS_ALWAYS_INLINE {
      constexpr int b = b_;
      ars_for<2>([&](auto p) ARS_ALWAYS_INLINE { w[b][p] = ring.template read<S::LAST_BASE + 2 * b + decltype(p)::value>(); });
    });
    ars_for<NG>([&](auto g_) ARS_ALWAYS_INLINE {
      constexpr int g = g_, ST0 = S::GOFF[g], GN = S::GOFF[g + 1] - S::GOFF[g];
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
      f32x4 acc[NT], bs[NT];  // the group's bias tiles: raw reads in front of the look-ahead request of the group's LAST block, whose counted wait settles them
      if constexpr (GN == 0) {
        const float* bg = bias_last + (g * NT) * 16 + 4 * q;
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { bs[t] = *reinterpret_cast<const f32x4*>(bg + t * 16); });
      }
      ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; });
      ars_for<GN>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int st = ST0 + decltype(i_)::value, ip = S::G_IP[st];
        ars_for<NT>([&](auto t_) ARS_ALWAYS_INLINE {
          constexpr int t = t_, blk = st * NT + t, cur = blk % (LOOKL + 1);
          constexpr bool last_of_group = (decltype(i_)::value == GN - 1 && t == NT - 1);
          if constexpr (last_of_group) {
            ars_for<NT>([&](auto u) ARS_ALWAYS_INLINE { bs[u] = arx_lds_raw<(g * NT + decltype(u)::value) * 64>(bias_last_addr); });
          }
          if constexpr (blk + LOOKL < NBL) {
            constexpr int nx = (blk + LOOKL) % (LOOKL + 1);
            ars_for<2>([&](auto p) ARS_ALWAYS_INLINE { w[nx][p] = ring.template read<S::LAST_BASE + 2 * (blk + LOOKL) + decltype(p)::value>(); });
          }
          constexpr int ahead = (blk + LOOKL < NBL ? LOOKL : NBL - 1 - blk);
          if constexpr (last_of_group) arh_settle<(blk + LOOKL < NBL ? 2 : 0)>(w[cur][0], w[cur][1]);
          else arh_settle<2 * ahead>(w[cur][0], w[cur][1]);
          if constexpr (last_of_group) {  // (the bias tiles are older than this step's request: the same wait has settled them)
#pragma unroll
            for (int u = 0; u < NT; ++u) arh_touch(bs[u]);
          }
          if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
          arh_block(w[cur], in[ip], acc[t]);
          if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
        });
      });
      float p[4 * NT];
      ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * t + r] = __builtin_fmaf(acc[t][r], dl, bs[t][r]);
      });
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) Uni::template poison<false>(p, fi * TOTAL, poison);
      auto ld = [&](int i) { return p[i]; };
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) {
        const int f = fid[fi];
        if (f >= 0) {
          float yv, lj;
          if (ARX_ABL == 3) {
            yv = p[fi * TOTAL] + xin[fi]; lj = p[fi * TOTAL + 1];
#pragma unroll
            for (int i = 2; i < TOTAL; ++i) lj += p[fi * TOTAL + i];
          } else if constexpr (DIAG) {
            int kb = 0;
            float ks[Uni::NKNOT];
            Uni::fwd(ld, fi * TOTAL, a, xin[fi], yv, lj, &kb, ks);
            if (live) {
              a.bin_out[n * S::D + f] = kb;
#pragma unroll
              for (int jj = 0; jj < Uni::NKNOT; ++jj) a.knots_out[(n * S::D + f) * Uni::NKNOT + jj] = ks[jj];
            }
          } else
          Uni::fwd(ld, fi * TOTAL, a, xin[fi], yv, lj);
          if constexpr (XLDS) xr[f] = yv;
          else if (live) a.y[n * a.ldy + f] = yv;
          lacc += lj;
        }
      }
    });
    if constexpr (XLDS) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (live) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
          if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(a.y + n * a.ldy + it * 16 + 4 * q) = *reinterpret_cast<const f32x4*>(xr + it * 16 + 4 * q);
      }
    }
    if (a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}

template <class S, typename Uni> static int arh_launch(const ArArgs* in, int abi, int args_bytes, int train, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;  // kernel built against another version of the library
  ArArgs a = *in;
  if (train || a.D != S::D || a.DIN != S::DIN || a.L != S::NH + 1 || a.act != S::ACT || a.sched || a.NG != S::NG || a.n_chunks != S::NCHUNK || a.l1rev) return ZK_EINVAL;
  for (int l = 0; l <= S::NH; ++l)
    if (!(a.wdescale[l] > 0.f) || !(a.wdescale[l] < __builtin_inff())) return ZK_EINVAL;  // the stream's per-layer scales must come with it
  a.n_tiles = (a.N + 16 * S::WAVES - 1) / (16 * S::WAVES);
  a.xs = ((S::D + 3) / 4) * 4 + 4;
  const bool vec_ok = (S::D % 4 == 0) && (a.ldy % 4 == 0) && ((uintptr_t)a.y % 16 == 0);
  if (S::XLDS != 0 && !vec_ok) return ZK_EINVAL;
  a.xlds = S::XLDS;
  const int lds = (S::NR * S::CH * AR_TF + a.bias_floats + 1024 + 256 + (S::XLDS ? S::WAVES * 16 * a.xs : 0)) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  const void* fn = nullptr;
  if ((a.bin_out != nullptr) != (a.knots_out != nullptr)) return ZK_EINVAL;
  if (a.bin_out) {
    if constexpr (Uni::NKNOT > 1) fn = (const void*)arh_kernel<S, Uni, true>;  // (the diagnostic twin exists for the spline maps only)
  } else {
    fn = (const void*)arh_kernel<S, Uni, false>;
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
