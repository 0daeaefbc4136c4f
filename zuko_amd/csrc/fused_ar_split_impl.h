// zuko_amd — operand-split twin of the static-shape fused autoregressive kernel (fused_ar_static_impl.h):
//
//     y, log|dy/dx| = univariate(conditioner(cat(x, c))).call_and_ladj(x)      (zuko/flows/autoregressive.py:207-218)
//
// gfx950 has no xf32 / tf32 matrix instruction and its f32 one (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate, which is what
// bounded the f32 kernels (matrix pipe 83 % busy, profiles/r03).  Here every f32 operand v is written as the exact sum of three bf16
// numbers h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) (3 x 8 significant bits: |v - h - m - l| <= 2^-25 |v|; both subtractions are
// exact in f32), and a product a b is evaluated as the six partial products down to 2^-18 relative size
//
//     a_l b_h + a_h b_l + a_m b_m + a_m b_h + a_h b_m + a_h b_h        (dropped: a_m b_l + a_l b_m + a_l b_l <= 2^-25 |a b|)
//
// on v_mfma_f32_16x16x32_bf16 (bf16 x bf16 is exact in f32; accumulation in f32, smallest terms first).  The result carries the error
// of an f32 dot product (measured against float64 in tests/ and in bench.py's parity block, next to the reference's own f32 error) at
// 6/16 of the f32 matrix time.  ZUKO_AMD_EXACT_F32=1 selects the f32-instruction kernels instead.
//
// Layout.  A stream BLOCK is one 16 x 32 weight block — out tile ot, the PAIR of in tiles (2 ip, 2 ip + 1) — as three 1 KiB images
// (h, m, l): lane (i = lane % 16, kq = lane / 16) holds 8 bf16 = weights of out unit i against units 4 kq .. 4 kq + 3 of in tile 2 ip
// (elements 0-3) and of in tile 2 ip + 1 (elements 4-7).  That is exactly how a lane of the 16-sample wave tile holds the activations
// (accumulator registers: sample lane % 16, units 4 (lane / 16) + r of every tile), so the B operand of a block is the lane's own
// registers of the two tiles, converted once per layer: no shuffle, no LDS round trip between layers.  The ring, the chunking and
// the raw-read / counted-wait idiom are those of fused_ar_static_impl.h (one block = three consecutive images).
//
//   Shape::NB[l], BOFF[l], B_OT, B_IP   blocks of hidden layer l in stream order (sorted by out tile): out tile, in pair
//   Shape::BASE[l], LAST_BASE           stream position (in images) where each layer starts
//   Shape::GOFF[g], G_IP                last layer: kept in pairs of every feature group (each with Uni::NT blocks)
#pragma once
#include "fused_ar_static_impl.h"
#include "zk_half.h"
#include "zk_univariate_bwd.h"

#ifndef ARX_LOOK
#define ARX_LOOK 1
#endif
#ifndef ARX_FENCE
#define ARX_FENCE 1
#endif

namespace zk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ArxB {  // B operand of one pair of activation tiles
  bf16x8 h, m, l;
};

__device__ __forceinline__ void arx_split(const f32x4& lo, const f32x4& hi, ArxB& b) {
  if (ARX_ABL == 6) {
    b.h = __builtin_bit_cast(bf16x8, lo); b.m = __builtin_bit_cast(bf16x8, hi); b.l = b.h;
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = e < 4 ? lo[e] : hi[e - 4];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    b.h[e] = h;
    b.m[e] = m;
    b.l[e] = (__bf16)r2;
  }
}

#define ARX_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), B, C, 0, 0, 0)
// the six partial products of one block, smallest first (a[0] = h, a[1] = m, a[2] = l images of the weights)
__device__ __forceinline__ void arx_block(const f32x4 (&a)[3], const ArxB& b, f32x4& c) {
  if (ARX_ABL == 2) {
    asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]));
    return;
  }
  ARX_MFMA(a[2], b.h, c);
  ARX_MFMA(a[0], b.l, c);
  ARX_MFMA(a[1], b.m, c);
  ARX_MFMA(a[1], b.h, c);
  ARX_MFMA(a[0], b.m, c);
  ARX_MFMA(a[0], b.h, c);
}

// A bias tile read RAW (as the weight images are, fused_ar_static_impl.h): issued in front of the look-ahead block's images, it is older than them, so the
// counted wait that settles the current block settles it too.  As a compiler-visible load its wait was lgkmcnt(0) — the compiler cannot count the raw reads —
// in front of the tile's first matrix instruction: the look-ahead block drained with it, 64 exposed LDS round trips per pass of the headline conditioner
// (found in the ISA, round 5).
template <int OFF> __device__ __forceinline__ f32x4 arx_lds_raw(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ unsigned arx_lds_addr(const float* p) { return (unsigned)(size_t)((const __attribute__((address_space(3))) float*)p); }

// Wait states between a layer's last matrix instruction and the first vector instruction that reads its accumulator — for the ONE-wavefront-per-SIMD
// instantiations only (Shape::OCC == 1: up to 512 registers, so accumulators may live in AGPRs).  Round 6 found that hipcc (ROCm 7.2) leaves 8 wait states
// between a v_mfma_f32_16x16x32_* whose destination is an AGPR and the v_accvgpr_read of it, and that this is not enough on gfx950 (csrc/inc_inverse.hip's
// two-part pull blocks read stale accumulators in a timing-dependent subset of wavefronts; with the destination in a VGPR the same 8 states suffice:
// profiles/r06/inverse.md).  The two-wavefront kernels hold everything in the 256 architectural VGPRs (agpr_count 0): nothing to guard there.
template <bool ON> __device__ __forceinline__ void arx_mfma_guard(f32x4& acc) {
  if constexpr (ON) asm volatile("s_nop 15" : "+v"(acc));
}
template <class S, class = void> struct ArxOneWave : std::false_type {};  // (the dgrad chain shapes have no OCC: two wavefronts per SIMD)
template <class S> struct ArxOneWave<S, std::void_t<decltype(S::OCC)>> : std::integral_constant<bool, S::OCC == 1> {};

template <class S> struct ArxPat {
  static constexpr int ot(int l, int s) { return S::B_OT[S::BOFF[l] + s]; }
  static constexpr int ip(int l, int s) { return S::B_IP[S::BOFF[l] + s]; }
  static constexpr bool tile_has_blocks(int l, int t) {
    for (int s = 0; s < S::NB[l]; ++s)
      if (ot(l, s) == t) return true;
    return false;
  }
};

// one hidden layer: out = W in + bias over the blocks of the generated pattern
template <class S, int L, class Ring> __device__ __forceinline__ void arx_hidden(Ring& ring, const float* bias_q, const ArxB (&in)[S::TMAX / 2], f32x4 (&out)[S::TMAX]) {
  typedef ArxPat<S> P;
  constexpr int NB = S::NB[L], BASE = S::BASE[L], HTL = S::HT[L];
  ars_for<HTL>([&](auto t_) ARS_ALWAYS_INLINE {
    constexpr int t = t_;
    if constexpr (!P::tile_has_blocks(L, t)) out[t] = *reinterpret_cast<const f32x4*>(bias_q + t * 16);  // units that depend on nothing: bias only
  });
  if constexpr (NB > 0) {
    // weight images are requested ARX_LOOK blocks ahead of the matrix instructions that consume them (probe builds: -DARX_LOOK=1|2|3,
    // -DARX_FENCE=0 drops the scheduling fences around a block's six matrix instructions; profiles/r05/headline.md)
    constexpr int LOOK = ARX_LOOK < NB ? ARX_LOOK : NB;
    f32x4 a[LOOK + 1][3];
    const unsigned bias_addr = arx_lds_addr(bias_q);
    ars_for<LOOK>([&](auto b_) ARS_ALWAYS_INLINE {
      constexpr int b = b_;
      ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { a[b][p] = ring.template read<BASE + 3 * b + decltype(p)::value>(); });
    });
    ars_for<NB>([&](auto s_) ARS_ALWAYS_INLINE {
      constexpr int s = s_, ot = P::ot(L, s), ip = P::ip(L, s), cur = s % (LOOK + 1);
      constexpr bool first_of_tile = (s == 0 || P::ot(L, s - 1) != ot);
      if constexpr (first_of_tile) out[ot] = arx_lds_raw<ot * 64>(bias_addr);  // accumulators start at the bias (raw read: see arx_lds_raw)
      if constexpr (s + LOOK < NB) {
        constexpr int nx = (s + LOOK) % (LOOK + 1);
        ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { a[nx][p] = ring.template read<BASE + 3 * (s + LOOK) + decltype(p)::value>(); });
      }
      constexpr int ahead = (s + LOOK < NB ? LOOK : NB - 1 - s);  // blocks behind this one whose images may still be outstanding
      // (with the tile's bias: only the images requested in THIS step are younger than it — equal to 3 * ahead at the default look-ahead of one block)
      if constexpr (first_of_tile) ars_settle<(s + LOOK < NB ? 3 : 0)>(a[cur][0], a[cur][1], a[cur][2], out[ot]);
      else ars_settle<3 * ahead>(a[cur][0], a[cur][1], a[cur][2]);
      if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
      arx_block(a[cur], in[ip], out[ot]);
      if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
    });
    arx_mfma_guard<ArxOneWave<S>::value>(out[P::ot(L, NB - 1)]);  // (the layer's last accumulator is read by the activation next)
  }
}

__device__ __forceinline__ void arx_amax_flush(const ArArgs& a, const float (&mx)[4], unsigned which, int lane) {
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (a.amax[l]) {  // (wave-uniform)
      const float m = gh_wave_max(mx[l]);
      if (lane == 0) gh_amax_put(a.amax[l], which, m);
    }
}
// mx[l]: running maximum magnitude of what this lane stored to act_out[l] (training launches with ArArgs::amax; folded into the device maxima at the kernel's end)
template <class S, int L, class Ring, bool TRAIN> __device__ __forceinline__ void arx_hidden_stack(Ring& ring, const float* bias_lds, int q, ArxB (&in)[S::TMAX / 2], f32x4 (&out)[S::TMAX],
                                                                                                 const ArArgs& a, int64_t n, bool live, float (&mx)[4]) {
  if constexpr (L < S::NH) {
    arx_hidden<S, L>(ring, bias_lds + L * S::BIAS_STRIDE + 4 * q, in, out);
    constexpr int HTL = S::HT[L];
    if constexpr (S::ACT == 1) {
#pragma unroll
      for (int t = 0; t < HTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];  // NaN stays NaN, as torch.relu
    } else if constexpr (S::ACT != 0) {
      // (a loop the compiler must not unroll over the activation's inline expansion, as in fused_ar_static_impl.h)
#pragma unroll 1
      for (int rep = 0; rep < 1; ++rep) {
#pragma unroll
        for (int t = 0; t < HTL; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) out[t][r] = act_f32(out[t][r], S::ACT);
      }
    }
    if constexpr (TRAIN) {
      if (live) {
#pragma unroll
        for (int t = 0; t < HTL; ++t) *reinterpret_cast<f32x4*>(a.act_out[L < 3 ? L : 2] + n * (HTL * 16) + t * 16 + 4 * q) = out[t];
        if (a.amax[L < 3 ? L : 2]) {
#pragma unroll
          for (int t = 0; t < HTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[L < 3 ? L : 2] = fmaxf(mx[L < 3 ? L : 2], fabsf(out[t][r]));
        }
      }
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < (HTL + 1) / 2; ++p) arx_split(out[2 * p], 2 * p + 1 < HTL ? out[2 * p + 1] : zero, in[p]);
    arx_hidden_stack<S, L + 1, Ring, TRAIN>(ring, bias_lds, q, in, out, a, n, live, mx);
  }
}

// TRAIN: the hidden activations and phi are stored for the backward pass (zuko_amd/train.py); y / ladj only when a.y is given.
// DIAG: the diagnostic twin of the product launch (same arithmetic, same instruction order up to two extra stores per feature): also
// writes the bin index the spline USED and the search-axis knots it searched (bin_out [N, D], knots_out [N, D, NKNOT]), as
// zk_ar_forward_diag does for the generic kernel — the parity bar on the bin index is asserted on the product path (tests/test_gpu_bins.py).
// Shape::OCC = wavefronts per SIMD the registers are budgeted for: 2 (widths <= 256: 16 activation tiles + 8 operand pairs per wavefront), 1 for
// conditioners 257 - 512 wide (32 tiles + 16 pairs = 320 registers; four wavefronts per workgroup, one workgroup per CU).
template <class S, typename Uni, bool TRAIN, bool DIAG = false> __global__ __launch_bounds__(64 * S::WAVES, S::OCC) void arx_kernel(ArArgs a) {
  typedef ArRingS<S::WAVES, S::CH, S::NR> Ring;
  static_assert((S::WAVES == 8 || S::WAVES == 4) && (S::NR == 2 || S::NR == 3) && (S::TMAX <= 16 || (S::TMAX <= 32 && S::OCC == 1 && S::WAVES == 4)) && S::TMAX % 2 == 0,
                "operand-split kernels: widths <= 256 with two wavefronts per SIMD, <= 512 with one");
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL, WAVES = S::WAVES;
  constexpr int NG = S::NG;
  constexpr int NSTEP = S::GOFF[NG];  // (group, in pair) steps of the last layer, NT blocks each
  constexpr bool XLDS = S::XLDS;
  constexpr bool FID_REGS = NG * FPL <= 32 && !TRAIN;  // (the training instantiation holds phi for its stores as well: the feature ids stay in LDS there)
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

  const bool uni_on = !TRAIN || a.y != nullptr;  // (training launch: y, ladj beside phi and the activations when the caller passes y)
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * (16 * WAVES) + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    const float* xrow = a.x + nc * a.ldx;

    ArxB in[S::TMAX / 2];
    f32x4 out[S::TMAX];
    float poison = 0.f;
    {
      f32x4 xin[S::NIT + 1];
#pragma unroll
      for (int it = 0; it < S::NIT; ++it) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((it + 1) * 16 <= S::DIN || it * 16 + 4 * q < S::DIN) v = *reinterpret_cast<const f32x4*>(xrow + it * 16 + 4 * q);
        xin[it] = v;
        if constexpr (TRAIN) {
          if (live && a.amax[3]) {  // (training: the maximum of the conditioner's input, the first layer's operand of the weight gradients)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[3] = fmaxf(mx[3], fabsf(v[r]));
          }
        }
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
      if (bad) poison = __builtin_nanf("");
      if constexpr (XLDS) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
          if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(xr + it * 16 + 4 * q) = xin[it];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int p = 0; p < (S::NIT + 1) / 2; ++p) arx_split(xin[2 * p], xin[2 * p + 1], in[p]);
    }

    // ---- hidden layers ---------------------------------------------------------------------------------------------
    arx_hidden_stack<S, 0, Ring, TRAIN>(ring, bias_lds, q, in, out, a, n, live, mx);

    // ---- last layer + univariate transform, one group of 4 * FPL features at a time --------------------------------
    float lacc = 0.f;
    constexpr int NBL = NSTEP * NT;                          // blocks of the last layer
    constexpr int LOOKL = ARX_LOOK < NBL ? ARX_LOOK : NBL;   // (read ahead as in arx_hidden; the groups' epilogues sit between the blocks, so the
    f32x4 w[LOOKL + 1][3];                                   //  images of the next group's first blocks travel while the spline is evaluated)
    ars_for<LOOKL>([&](auto b_) ARS_ALWAYS_INLINE {
      constexpr int b = b_;
      ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { w[b][p] = ring.template read<S::LAST_BASE + 3 * b + decltype(p)::value>(); });
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
      f32x4 acc[NT];  // the accumulators start at the bias (raw reads, older than every image requested below: the first block's wait settles them all)
      if constexpr (GN > 0) {
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[t] = arx_lds_raw<(g * NT + decltype(t)::value) * 64>(bias_last_addr); });
      } else {
        const float* bg = bias_last + (g * NT) * 16 + 4 * q;
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[t] = *reinterpret_cast<const f32x4*>(bg + t * 16); });
      }
      ars_for<GN>([&](auto i_) ARS_ALWAYS_INLINE {
        constexpr int st = ST0 + decltype(i_)::value, ip = S::G_IP[st];
        ars_for<NT>([&](auto t_) ARS_ALWAYS_INLINE {
          constexpr int t = t_, blk = st * NT + t, cur = blk % (LOOKL + 1);
          if constexpr (blk + LOOKL < NBL) {
            constexpr int nx = (blk + LOOKL) % (LOOKL + 1);
            ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { w[nx][p] = ring.template read<S::LAST_BASE + 3 * (blk + LOOKL) + decltype(p)::value>(); });
          }
          constexpr int ahead = (blk + LOOKL < NBL ? LOOKL : NBL - 1 - blk);
          // (the group's first blocks: their accumulator's bias with them; at the very first only this step's request is younger than the bias reads)
          if constexpr (decltype(i_)::value == 0 && t == 0) ars_settle<(blk + LOOKL < NBL ? 3 : 0)>(w[cur][0], w[cur][1], w[cur][2], acc[t]);
          else if constexpr (decltype(i_)::value == 0) ars_settle<3 * ahead>(w[cur][0], w[cur][1], w[cur][2], acc[t]);
          else ars_settle<3 * ahead>(w[cur][0], w[cur][1], w[cur][2]);
          if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
          arx_block(w[cur], in[ip], acc[t]);
          if (ARX_FENCE) __builtin_amdgcn_sched_barrier(0);
        });
      });
      float p[4 * NT];
      if constexpr (GN > 0) arx_mfma_guard<ArxOneWave<S>::value>(acc[NT - 1]);
      ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * t + r] = acc[t][r];
      });
      if constexpr (TRAIN) {
        if (a.phi_packed) {  // the accumulators as they are: 16 bytes per lane and tile (zk_ar_common.h: ArArgs::phi_packed)
          if (live) {
            float* dst = a.phi_out + n * a.ldphi + (g * NT) * 16 + 4 * q;
            ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE {
              *reinterpret_cast<f32x4*>(dst + decltype(t)::value * 16) = f32x4{p[4 * t] + poison, p[4 * t + 1] + poison, p[4 * t + 2] + poison, p[4 * t + 3] + poison};
            });
          }
        } else {
#pragma unroll
          for (int fi = 0; fi < FPL; ++fi) {
            const int f = fid[fi];
            if (f >= 0 && live) {
              float* dst = a.phi_out + n * a.ldphi + f * TOTAL;
#pragma unroll
              for (int i = 0; i < TOTAL; ++i) dst[i] = p[fi * TOTAL + i] + poison;
            }
          }
        }
      }
      if (uni_on) {
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
      }
    });
    if constexpr (XLDS) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (live && uni_on) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
          if ((it + 1) * 16 <= S::D || it * 16 + 4 * q < S::D) *reinterpret_cast<f32x4*>(a.y + n * a.ldy + it * 16 + 4 * q) = *reinterpret_cast<const f32x4*>(xr + it * 16 + 4 * q);
      }
    }
    if (uni_on && a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
  }
  if constexpr (TRAIN) arx_amax_flush(a, mx, blockIdx.x * WAVES + wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}

// ---- the backward twin: dgrad through EVERY linear layer of the conditioner in one launch -------------------------------------------
// g_{l-1} = (g_l (W_l * mask_l)) * relu'(h_{l-1}) from the gradient of the packed parameters phi down to the conditioner's input (what
// autograd derives from zuko/nn.py:217-218 and the ReLUs between the layers), with the operand split of this file.  The chain is itself
// a masked MLP whose layer matrices are the TRANSPOSED sorted weights (zuko_amd/static_ar.py: chain_split_tables).  Its first layer has
// features x total inputs (1472 for the 8-bin spline head of 64 features): far more than a wavefront can hold, so that layer walks
// its blocks IN-PAIR MAJOR and reads the two tiles of g_phi it needs per pair straight from global memory, DEPTH pairs ahead
// (each element of g_phi is read exactly once per launch); the other layers keep the gradient tiles in registers like the forward.
//   Shape::DIN0, NP0, P0[]        width of g_phi, in pairs of layer 0 that have blocks, and which pairs these are (in stream order)
//   Shape::NB[l], BOFF, B_OT, B_IP, BASE   blocks per chain layer (layer 0: in-pair major; others: out-tile major)
//   Shape::HT[l]                  out tiles of chain layer l (= tiles of the hidden layer whose gradient it yields); DOUT: conditioner inputs
#ifndef ARXB_AHEAD
#define ARXB_AHEAD 1  // steps of look-ahead of the first layer's inputs (phi, x, gy) in registers
#endif
#ifndef ARXB_ABL
#define ARXB_ABL 0  // timing ablations of the backward kernels (wrong results; scripts/build_chain_variant.py): 1 no g_phi stores, 2 no phi loads, 3 neither, 4 no adjoint arithmetic, 5 no g_h stores, 6 no gate loads, 7 = 5 + 6, 8 = 3 + 7
#endif
template <class S> struct ArxdPat {
  static constexpr int pair_index(int s) {  // position in P0 of the pair block s of layer 0 belongs to
    int n = 0;
    for (int i = 1; i <= s; ++i) n += S::B_IP[i] != S::B_IP[i - 1];
    return n;
  }
};

template <class S, class Ring> __device__ __forceinline__ void arxd_first(Ring& ring, const float* grow_q, int q, f32x4 (&out)[S::TMAX]) {
  typedef ArxdPat<S> P;
  constexpr int NB = S::NB[0], BASE = S::BASE[0], HTL = S::HT[0], NP = S::NP0, DEPTH = NP < 6 ? NP : 6;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < HTL; ++t) out[t] = zero;
  f32x4 pf[DEPTH][2];
  auto fetch = [&](auto j_) ARS_ALWAYS_INLINE {
    constexpr int j = decltype(j_)::value, ip = S::P0[j];
    ars_for<2>([&](auto h_) ARS_ALWAYS_INLINE {
      constexpr int it = 2 * ip + decltype(h_)::value;
      f32x4 v = zero;
      if constexpr ((it + 1) * 16 <= S::DIN0) v = *reinterpret_cast<const f32x4*>(grow_q + it * 16);
      else if constexpr (it * 16 < S::DIN0) {
        if (it * 16 + 4 * q < S::DIN0) v = *reinterpret_cast<const f32x4*>(grow_q + it * 16);
      }
      pf[j % DEPTH][h_] = v;
    });
  };
  ars_for<DEPTH>([&](auto j_) ARS_ALWAYS_INLINE { fetch(j_); });
  ArxB b;
  f32x4 a[2][3];
  ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { a[0][p] = ring.template read<BASE + decltype(p)::value>(); });
  ars_for<NB>([&](auto s_) ARS_ALWAYS_INLINE {
    constexpr int s = s_, ot = S::B_OT[s], j = P::pair_index(s);
    if constexpr (s == 0 || S::B_IP[s - 1] != S::B_IP[s]) {  // first block of a pair: its two tiles become the B operand, their slots are refilled
      arx_split(pf[j % DEPTH][0], pf[j % DEPTH][1], b);
      if constexpr (j + DEPTH < NP) fetch(std::integral_constant<int, j + DEPTH>{});
    }
    if constexpr (s + 1 < NB) {
      ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { a[(s + 1) & 1][p] = ring.template read<BASE + 3 * (s + 1) + decltype(p)::value>(); });
      ars_settle<3>(a[s & 1][0], a[s & 1][1], a[s & 1][2]);
    } else {
      ars_settle<0>(a[s & 1][0], a[s & 1][1], a[s & 1][2]);
    }
    __builtin_amdgcn_sched_barrier(0);
    arx_block(a[s & 1], b, out[ot]);
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <class S, int L, class Ring> __device__ __forceinline__ void arxd_stack(Ring& ring, const float* zero_q, int q, ArxB (&in)[S::TMAX / 2], f32x4 (&out)[S::TMAX], const ArArgs& a,
                                                                                 int64_t n, int64_t nc, bool live, float (&mx)[4], const float* xadd_q = nullptr) {
  if constexpr (L < S::NH) {
    if constexpr (L > 0) arx_hidden<S, L>(ring, zero_q, in, out);
    constexpr int HTL = S::HT[L];
    if constexpr (L + 1 < S::NH) {
      const float* grow = a.gate[L] + nc * (HTL * 16) + 4 * q;
#pragma unroll
      for (int t = 0; t < HTL; ++t) {
        const f32x4 h = (ARXB_ABL == 6 || ARXB_ABL == 7 || ARXB_ABL == 8) ? f32x4{1.f, -1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(grow + t * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = out[t][r] * (h[r] > 0.f ? 1.f : 0.f);  // (a product, as autograd's: NaN gradients stay NaN)
      }
      if (live && ARXB_ABL != 5 && ARXB_ABL != 7 && ARXB_ABL != 8) {
#pragma unroll
        for (int t = 0; t < HTL; ++t) *reinterpret_cast<f32x4*>(a.act_out[L] + n * (HTL * 16) + t * 16 + 4 * q) = out[t];
        if (a.amax[L < 3 ? L : 2]) {
#pragma unroll
          for (int t = 0; t < HTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[L < 3 ? L : 2] = fmaxf(mx[L < 3 ? L : 2], fabsf(out[t][r]));
        }
      }
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < (HTL + 1) / 2; ++p) arx_split(out[2 * p], 2 * p + 1 < HTL ? out[2 * p + 1] : zero, in[p]);
      arxd_stack<S, L + 1, Ring>(ring, zero_q, q, in, out, a, n, nc, live, mx, xadd_q);
    } else if (live) {  // gradient w.r.t. the conditioner's input: module column order, DOUT columns (accumulate: added to what the buffer holds)
#pragma unroll
      for (int t = 0; t < HTL; ++t)
        if ((t + 1) * 16 <= S::DOUT || t * 16 + 4 * q < S::DOUT) {
          f32x4* dst = reinterpret_cast<f32x4*>(a.phi_out + n * a.ldphi + t * 16 + 4 * q);
          if (xadd_q) out[t] += *reinterpret_cast<const f32x4*>(xadd_q + t * 16);  // (wave-private LDS row: the univariate map's own d/dx term)
          *dst = a.accumulate ? *dst + out[t] : out[t];
        }
    }
  }
}

template <class S> __global__ __launch_bounds__(512, 2) void arxd_kernel(ArArgs a) {
  typedef ArRingS<8, S::CH> Ring;
  static_assert(S::NH >= 2 && S::NH <= 4 && S::TMAX <= 16 && S::TMAX % 2 == 0 && S::WAVES == 8, "dgrad chain: the last layer + up to three gated layers, widths <= 256");
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
  float* zero_lds = ars_lds + ARS_NR * S::CH * AR_TF;  // "bias image" of a layer without bias
  for (int i = tid; i < S::TMAX * 16 + 16; i += 512) zero_lds[i] = 0.f;
  __syncthreads();
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * 128 + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    ArxB in[S::TMAX / 2];
    f32x4 out[S::TMAX];
    arxd_first<S>(ring, a.x + nc * a.ldx + 4 * q, q, out);
    arxd_stack<S, 0, Ring>(ring, zero_lds + 4 * q, q, in, out, a, n, nc, live, mx);
  }
  arx_amax_flush(a, mx, blockIdx.x * 8 + wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the whole backward of one autoregressive transform up to the weight gradients -------------------------------------------------
// (gy, gl) -> g_phi (univariate adjoint, zk_univariate_bwd.h) -> dgrad chain -> gx, in ONE launch: the chain's first layer is walked in
// the FORWARD kernel's packed order of phi — K tile (g, t) = parameters 4 t .. 4 t + 3 of the 4 * FPL features of group g, so that lane
// (sample j, q) holds exactly its own features' parameters, as the forward's accumulators did — and every lane computes the adjoint of
// its own features from phi, x, gy, gl: the results ARE its B operand of the group's in pairs (no shuffle, no LDS).  phi arrives and
// g_phi leaves in that same packed order (ArArgs::phi_packed: the training forward stores its accumulators as they are, the weight
// gradients read the packed gradient through a row table), 16 bytes per lane and tile — lane-wise dword accesses to module-order rows
// (23 per feature, each instruction scattering 64 dwords over 16 rows) cost this kernel a third of its time (profiles/r04).  The map's
// direct d/dx term waits in a wave-private LDS row and is added to the chain's input gradient at the end (the conditioner's input may carry
// context columns behind the D features: x is cat(x, c), the input gradient covers both, the direct term only the features).
//   Shape::NG, PB[]     feature groups; first block (of layer 0, in-pair major) of every packed pair, PB[NG * NT / 2] = NB[0]
template <typename Uni, typename A> __device__ __forceinline__ void arxb_adjoint(const float* p, const A& a, float x, float gy, float gl, float& gx, float* g) {
  if constexpr (Uni::TOTAL == 2) affine_backward_element(p, x, gy, gl, a.ls, gx, g);
  else rqs_backward_element<(Uni::TOTAL + 1) / 3>(p, x, gy, gl, a.bound, a.ls, gx, g);
}

template <class S, typename Uni, class Ring>
__device__ __forceinline__ void arxb_first(Ring& ring, const ArArgs& a, const int* fmap_lds, float* xr, int q, int64_t n, int64_t nc, bool live, f32x4 (&out)[S::TMAX], float (&mx)[4]) {
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL;
  constexpr int SG = (NT % 2) ? 2 : 1;  // groups per step: a step ends on a pair boundary
  constexpr int NSTEP = S::NG / SG, PPS = SG * NT / 2, NF = SG * FPL;
  constexpr int NB = S::NB[0], BASE = S::BASE[0], HTL = S::HT[0];
  static_assert(S::NG % SG == 0 && FPL * TOTAL <= 4 * NT, "packed order: whole pairs per step");
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < HTL; ++t) out[t] = zero;
  const float glv = a.gl[nc];
  const float* phirow = a.phi_in + nc * a.ldpin + 4 * q;  // packed rows: tile (g, t) of this lane = 4 floats at (g NT + t) 16 + 4 q
  float* gphirow = a.gphi_out + n * a.ldpin + 4 * q;
  const float* xrow = a.x + nc * a.ldx;
  const float* gyrow = a.gy + nc * a.ldgy;
  // ARXB_AHEAD steps of look-ahead in registers (buffer = step % ARXB_AHEAD): the loads of step s + ARXB_AHEAD are issued when step s's values are consumed
  f32x4 ph4[ARXB_AHEAD][SG * NT];
  float xv[ARXB_AHEAD][NF], gyv[ARXB_AHEAD][NF];
  int fid[ARXB_AHEAD][NF];
  auto fetch = [&](auto s_) ARS_ALWAYS_INLINE {
    constexpr int s = decltype(s_)::value, bf = s % ARXB_AHEAD;
#pragma unroll
    for (int t = 0; t < SG * NT; ++t)
      ph4[bf][t] = (ARXB_ABL == 2 || ARXB_ABL == 3 || ARXB_ABL == 8) ? f32x4{0.01f, 0.02f, 0.03f, 0.04f} : *reinterpret_cast<const f32x4*>(phirow + (s * SG * NT + t) * 16);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = fmap_lds[((s * SG + i / FPL) * 4 + q) * FPL + i % FPL];
      const int fc = f < 0 ? 0 : f;
      fid[bf][i] = f;
      xv[bf][i] = xrow[fc];
      gyv[bf][i] = gyrow[fc];
    }
  };
  ars_for<(ARXB_AHEAD < NSTEP ? ARXB_AHEAD : NSTEP)>([&](auto s_) ARS_ALWAYS_INLINE { fetch(s_); });
  f32x4 w[2][3];
  if constexpr (NB > 0) {
    ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { w[0][p] = ring.template read<BASE + decltype(p)::value>(); });
  }
  ars_for<NSTEP>([&](auto s_) ARS_ALWAYS_INLINE {
    constexpr int s = s_, bf = s % ARXB_AHEAD;
    float gq[SG * NT * 4];
#pragma unroll
    for (int i = 0; i < SG * NT * 4; ++i) gq[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = fid[bf][i];
      float ph[TOTAL], g[TOTAL], gxv;
#pragma unroll
      for (int k = 0; k < TOTAL; ++k) ph[k] = ph4[bf][(i / FPL) * NT + ((i % FPL) * TOTAL + k) / 4][((i % FPL) * TOTAL + k) % 4];
      if (ARXB_ABL == 4) {
        gxv = xv[bf][i] + gyv[bf][i];
#pragma unroll
        for (int k = 0; k < TOTAL; ++k) g[k] = ph[k] + glv;
      } else
      arxb_adjoint<Uni>(ph, a, xv[bf][i], gyv[bf][i], glv, gxv, g);
      if (f >= 0) xr[f] = gxv;
#pragma unroll
      for (int k = 0; k < TOTAL; ++k) gq[(i / FPL) * NT * 4 + (i % FPL) * TOTAL + k] = f >= 0 ? g[k] : 0.f;
    }
    // this step's inputs are consumed: their registers take the loads of step s + ARXB_AHEAD — issued BEFORE this step's stores, so that waiting
    // for them later does not wait for the stores (one in-order counter for both)
    if constexpr (s + ARXB_AHEAD < NSTEP) fetch(std::integral_constant<int, s + ARXB_AHEAD>{});
    if (live && ARXB_ABL != 1 && ARXB_ABL != 3 && ARXB_ABL != 8) {  // the gradient in the same packed order (padding slots zero), for the weight gradients
#pragma unroll
      for (int t = 0; t < SG * NT; ++t) *reinterpret_cast<f32x4*>(gphirow + (s * SG * NT + t) * 16) = f32x4{gq[4 * t], gq[4 * t + 1], gq[4 * t + 2], gq[4 * t + 3]};
      if (a.amax[3]) {
#pragma unroll
        for (int t = 0; t < 4 * SG * NT; ++t) mx[3] = fmaxf(mx[3], fabsf(gq[t]));
      }
    }
    ars_for<PPS>([&](auto pl_) ARS_ALWAYS_INLINE {
      constexpr int pl = pl_, pp = s * PPS + pl, B0 = S::PB[pp], B1 = S::PB[pp + 1];
      if constexpr (B1 > B0) {
        ArxB b;
        arx_split(f32x4{gq[8 * pl], gq[8 * pl + 1], gq[8 * pl + 2], gq[8 * pl + 3]}, f32x4{gq[8 * pl + 4], gq[8 * pl + 5], gq[8 * pl + 6], gq[8 * pl + 7]}, b);
        ars_for<B1 - B0>([&](auto i_) ARS_ALWAYS_INLINE {
          constexpr int blk = B0 + decltype(i_)::value, ot = S::B_OT[blk];
          if constexpr (blk + 1 < NB) {
            ars_for<3>([&](auto p) ARS_ALWAYS_INLINE { w[(blk + 1) & 1][p] = ring.template read<BASE + 3 * (blk + 1) + decltype(p)::value>(); });
            ars_settle<3>(w[blk & 1][0], w[blk & 1][1], w[blk & 1][2]);
          } else {
            ars_settle<0>(w[blk & 1][0], w[blk & 1][1], w[blk & 1][2]);
          }
          __builtin_amdgcn_sched_barrier(0);
          arx_block(w[blk & 1], b, out[ot]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
  });
}

template <class S, typename Uni> __global__ __launch_bounds__(512, 2) void arxb_kernel(ArArgs a) {
  typedef ArRingS<8, S::CH> Ring;
  static_assert(S::NH >= 2 && S::NH <= 4 && S::TMAX <= 16 && S::TMAX % 2 == 0 && S::WAVES == 8, "dgrad chain: the last layer + up to three gated layers, widths <= 256");
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
  constexpr int NFMAP = ((S::NG * 4 * Uni::FPL + 3) / 4) * 4;
  float* zero_lds = ars_lds + ARS_NR * S::CH * AR_TF;
  int* fmap_lds = reinterpret_cast<int*>(zero_lds + S::TMAX * 16 + 16);
  float* xr = reinterpret_cast<float*>(fmap_lds + NFMAP) + (wave * 16 + j) * a.xs;
  for (int i = tid; i < S::TMAX * 16 + 16; i += 512) zero_lds[i] = 0.f;
  for (int i = tid; i < S::NG * 4 * Uni::FPL; i += 512) fmap_lds[i] = a.featmap[i];
  __syncthreads();
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * 128 + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    ArxB in[S::TMAX / 2];
    f32x4 out[S::TMAX];
    if (a.D < S::DOUT) {  // context columns (and padding) of the conditioner's input have no direct term
#pragma unroll 1
      for (int i = a.D + q; i < a.xs - 4; i += 4) xr[i] = 0.f;
    }
    arxb_first<S, Uni>(ring, a, fmap_lds, xr, q, n, nc, live, out, mx);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    arxd_stack<S, 0, Ring>(ring, zero_lds + 4 * q, q, in, out, a, n, nc, live, mx, xr + 4 * q);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  arx_amax_flush(a, mx, blockIdx.x * 8 + wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class S, typename Uni> static int arxb_launch(const ArArgs* in, int abi, int args_bytes, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;
  ArArgs a = *in;
  if (a.DIN != S::DOUT || a.D < 1 || a.D > S::DOUT || a.D > S::NG * 4 * Uni::FPL || a.L != S::NH || a.NG != S::NG || a.n_chunks != S::NCHUNK || !a.x || !a.phi_out || !a.phi_in || !a.gphi_out || !a.gy || !a.gl || !a.featmap ||
      a.ldx % 4 || a.ldphi % 4 || a.ldpin < (int64_t)S::NG * Uni::NT * 16 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.phi_out % 16))
    return ZK_EINVAL;
  for (int l = 0; l + 1 < S::NH; ++l)
    if (!a.gate[l] || !a.act_out[l] || ((uintptr_t)a.gate[l] % 16) || ((uintptr_t)a.act_out[l] % 16)) return ZK_EINVAL;
  a.n_tiles = (a.N + 127) / 128;
  a.xs = ((S::DOUT + 3) / 4) * 4 + 4;
  constexpr int NFMAP = ((S::NG * 4 * Uni::FPL + 3) / 4) * 4;
  const int lds = (ARS_NR * S::CH * AR_TF + S::TMAX * 16 + 16 + NFMAP + 8 * 16 * a.xs) * (int)sizeof(float);
  if (lds > 160 * 1024 || a.ldpin % 4 || ((uintptr_t)a.phi_in % 16) || ((uintptr_t)a.gphi_out % 16)) return ZK_EINVAL;
  const void* fn = (const void*)arxb_kernel<S, Uni>;
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

template <class S> static int arxd_launch(const ArArgs* in, int abi, int args_bytes, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;
  ArArgs a = *in;
  if (a.DIN != S::DIN0 || a.D != S::DOUT || a.L != S::NH || a.n_chunks != S::NCHUNK || !a.x || !a.phi_out || a.ldx % 4 || a.ldphi % 4 || ((uintptr_t)a.x % 16) ||
      ((uintptr_t)a.phi_out % 16))
    return ZK_EINVAL;
  for (int l = 0; l + 1 < S::NH; ++l)
    if (!a.gate[l] || !a.act_out[l] || ((uintptr_t)a.gate[l] % 16) || ((uintptr_t)a.act_out[l] % 16)) return ZK_EINVAL;
  a.n_tiles = (a.N + 127) / 128;
  const int lds = (ARS_NR * S::CH * AR_TF + S::TMAX * 16 + 16) * (int)sizeof(float);
  const void* fn = (const void*)arxd_kernel<S>;
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

template <class S, typename Uni> static int arx_launch(const ArArgs* in, int abi, int args_bytes, int train, void* stream) {
  if (abi != ARS_ABI || args_bytes != (int)sizeof(ArArgs)) return ZK_EINVAL;  // kernel built against another version of the library
  ArArgs a = *in;
  if (a.D != S::D || a.DIN != S::DIN || a.L != S::NH + 1 || a.act != S::ACT || a.sched || a.NG != S::NG || a.n_chunks != S::NCHUNK || a.l1rev) return ZK_EINVAL;
  if (train && (!S::TRAIN_OK || !a.phi_out || (a.phi_packed && (a.ldphi % 4 || a.ldphi < (int64_t)S::NG * Uni::NT * 16 || ((uintptr_t)a.phi_out % 16))))) return ZK_EINVAL;
  a.n_tiles = (a.N + 16 * S::WAVES - 1) / (16 * S::WAVES);
  a.xs = ((S::D + 3) / 4) * 4 + 4;
  const bool vec_ok = (S::D % 4 == 0) && ((train && !a.y) || ((a.ldy % 4 == 0) && ((uintptr_t)a.y % 16 == 0)));
  if (S::XLDS != 0 && !vec_ok) return ZK_EINVAL;
  a.xlds = S::XLDS;
  const int lds = (S::NR * S::CH * AR_TF + a.bias_floats + 1024 + 256 + (S::XLDS ? S::WAVES * 16 * a.xs : 0)) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  const void* fn = nullptr;
  if ((a.bin_out != nullptr) != (a.knots_out != nullptr) || (a.bin_out && train)) return ZK_EINVAL;
  if (train) {
    if constexpr (S::TRAIN_OK) fn = (const void*)arx_kernel<S, Uni, true>;
  } else if (a.bin_out) {
    if constexpr (Uni::NKNOT > 1) fn = (const void*)arx_kernel<S, Uni, false, true>;  // (the diagnostic twin exists for the spline maps only)
  } else {
    fn = (const void*)arx_kernel<S, Uni, false>;
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
  constexpr int MAXG = (S::WAVES == 4 && S::OCC == 2) ? 512 : 256;  // 4 wavefronts at two per SIMD: two independent workgroups per CU (one wavefront per SIMD each)
  const unsigned grid = (unsigned)(a.n_tiles < MAXG ? a.n_tiles : MAXG);
  void* kargs[] = {&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(64 * S::WAVES), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

}  // namespace zk
