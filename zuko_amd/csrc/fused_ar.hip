// zuko_amd — fused masked-autoregressive transform: conditioner + univariate transform + ladj.
//
// Replaces, for one autoregressive layer of MAF / NSF (zuko/flows/autoregressive.py:207-218),
//     phi = MaskedMLP(cat(x, c));  y, ladj = univariate(*unpack(phi)).call_and_ladj(x);  ladj.sum(-1)
// i.e. zuko/nn.py:217-218 (x4), ReLU (x3), zuko/transforms.py:469-490 + 554-567 (RQS) or
// :436-446 (affine) and the feature reduction of :210-214 — ONE launch, phi never touches HBM.
//
// Execution model (gfx950):
//   * a workgroup = 8 wavefronts = 128 samples; a wavefront owns 16 samples for the whole network.
//   * every layer is computed TRANSPOSED, H^T = W . X^T, on v_mfma_f32_16x16x4_f32 (exact fp32):
//     A = 16x4 slice of W (rows = out units), B = 4x16 slice of the activations (cols = samples).
//     The D fragment (lane (j, q) holds out units 4q..4q+3 of sample j) has exactly the layout the
//     NEXT layer needs as its B operand (lane (j, q) supplies k = q of a 4-deep step) once the k
//     axis is enumerated as k-step r <-> units {r, 4+r, 8+r, 12+r}; the weight stream is laid out
//     accordingly on the host, so activations stay in VGPRs from the input load to the spline.
//   * weights (1.2 MB per transform after tile skipping, L2-resident) are streamed through a
//     3 x 24 KiB LDS ring with global_load_lds_dwordx4 (one 1 KiB tile image per wave-instruction),
//     shared by the 8 waves and refilled two chunks ahead; A fragments are conflict-free
//     ds_read_b128 (lane-linear image).
//   * hidden units are sorted by MADE degree on the host => masks are block lower-triangular and
//     all-zero 16x16 tiles are skipped via per-group bitmasks (44% of the tiles at cfg2).
//   * last layer: rows are regrouped so lane (j, q) accumulates all `total` parameters of one
//     (RQS) or two (affine) features of sample j; the univariate transform runs on those
//     registers; y is stored, ladj reduced over q with two shuffles and over groups in a register.
//
// Host-side planning (permutations, stream order, skip masks): zuko_amd/fused.py.
#include "../../include/zuko_amd.h"
#include "zk_ar_common.h"
#include <cstring>
#include <mutex>
#include <type_traits>
#include <unordered_map>

// -DZK_AR_TIMING=1 compiles in the s_memtime/printf phase probes (dbg bits 3 and 4); off in the product build
#ifndef ZK_AR_TIMING
#define ZK_AR_TIMING 0
#endif

namespace zk {


template <int CH, int NR> struct RingT {
  static constexpr int kChunk = CH, kSlots = NR;
  float* lds;
  const float* stream;
  int n_chunks, pos, slot, load_chunk, load_slot, wave, lane, dbg;
  const int* sched;  // when set, load_chunk indexes this list (length n_chunks) instead of the stream

  // Each wave copies CH / AR_WAVES CONSECUTIVE tiles: one LDS base (M0) and one address per chunk, the tiles selected by
  // the instruction's immediate offset (it moves the global and the LDS address alike).  A vector-memory instruction costs
  // the issuing wave ~40 cycles of its instruction stream and every M0 write ~20 more (scripts/probes/dma_issue_probe.hip).
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < CH / AR_WAVES) {
#ifndef ZK_AR_NO_RING_DMA  // probe builds only (wrong results): what the ring DMAs cost the instruction stream
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, I * AR_TF * 4, 0);
#endif
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {
    static_assert((CH / AR_WAVES - 1) * AR_TF * 4 < 4096, "immediate offset range");
    const int b0 = wave * (CH / AR_WAVES);
    const int chunk_id = sched ? sched[load_chunk] : load_chunk;
    const float* g = stream + ((size_t)chunk_id * CH + b0) * AR_TF + lane * 4;
    float* l = lds + (load_slot * CH + b0) * AR_TF;
    dma<0>(g, l);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == NR) ? 0 : load_slot + 1;
  }
  // all 8 waves call this at the same point of the (uniform) control flow
  __device__ __forceinline__ void advance() {
    // my DMAs of the chunk about to be read have landed (younger chunks stay in flight); my reads of the chunk being released returned
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NR - 2) * (CH / AR_WAVES)) : "memory");
    __builtin_amdgcn_s_barrier();  // bare barrier: __syncthreads() would prepend s_waitcnt vmcnt(0) and drain the look-ahead DMAs
    asm volatile("" ::: "memory");
    issue();                                           // refill the slot that was just released
    slot = (slot + 1 == NR) ? 0 : slot + 1;
    pos = 0;
  }
  __device__ __forceinline__ f32x4 tile(int t) const {
    return *reinterpret_cast<const f32x4*>(lds + (slot * CH + pos + t) * AR_TF + lane * 4);
  }
  template <int G> __device__ __forceinline__ void begin() {
    if (pos == CH) advance();
  }
  template <int G> __device__ __forceinline__ void commit() { pos += G; }
  __device__ __forceinline__ void end_layer() {
    if (pos != 0) pos = CH;
  }
  __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

extern __shared__ __attribute__((aligned(16))) float ar_lds[];

// launcher exported by a generated static-shape kernel (csrc/fused_ar_static_impl.h: ars_launch; zuko_amd/static_ar.py builds them)
typedef int (*ars_launch_fn)(const ArArgs* a, int abi, int args_bytes, int train, void* stream);

// one masked layer with <= 256 inputs / outputs: out = W in + bias, tiles skipped per (group of 4 out tiles, in tile)
template <class Src>
__device__ __forceinline__ void hidden_layer(Src& ring, const uint32_t* __restrict__ skip4, int olim, const float* bias_q, const f32x4 (&in)[AR_T], f32x4 (&out)[AR_T]) {
#pragma unroll
  for (int otg = 0; otg < 4; ++otg) {
    const uint32_t bits = otg <= olim ? skip4[otg] : 0u;  // partial sweeps evaluate a prefix of the out-groups
#pragma unroll
    for (int t = 0; t < 4; ++t) out[otg * 4 + t] = *reinterpret_cast<const f32x4*>(bias_q + (otg * 4 + t) * 16);  // accumulators start at the bias
#pragma unroll
    for (int it = 0; it < AR_T; ++it) {
      if (bits & (1u << it)) {
        ring.template begin<4>();
        f32x4 a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = ring.tile(t);
        ring.template commit<4>();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) out[otg * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][r], in[it][r], out[otg * 4 + t], 0, 0, 0);
      }
    }
  }
  ring.end_layer();
}

template <typename Uni, bool INVERSE, class Src, bool XLDS, bool DIAG = false> __global__ __launch_bounds__(512, 2) void ar_kernel(ArArgs a) {
  constexpr bool DIRECT = false;
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  float* ring_lds = ar_lds;
  float* bias_lds = ar_lds;  // (DIRECT: the bias image is the only LDS user)

  Src ring;
  if constexpr (DIRECT) {
    ring.init(a.stream, lane, a.n_chunks);  // n_chunks == number of tile images (chunk size 1)
  } else {
    bias_lds = ar_lds + Src::kSlots * Src::kChunk * AR_TF;
    ring.dbg = a.dbg; ring.lds = ring_lds; ring.stream = a.stream; ring.sched = a.sched; ring.n_chunks = a.sched ? a.n_sched : a.n_chunks; ring.wave = wave; ring.lane = lane;
    ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
    for (int i = 0; i < Src::kSlots - 1; ++i) ring.issue();
    ring.slot = Src::kSlots - 1;
    ring.pos = Src::kChunk;
  }

  for (int i = tid; i < a.bias_floats; i += 512) bias_lds[i] = a.bias[i];
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);  // feature map of the last layer
  // wave-private [16 samples x D] tile: the epilogue's input values on the way in, the results on the
  // way out.  Keeps the per-group operand fetch on the LDS (lgkmcnt) queue — a global load there would
  // have to be waited for with vmcnt(0), i.e. behind the ring DMAs in flight (measured: 5 k cycles per
  // group) — and turns 4-byte scattered result stores into coalesced 16-byte row stores.
  // skip words of the last layer's groups: read per group, so they must not come through a vector-memory
  // load (its s_waitcnt vmcnt(0) would also drain the ring DMAs in flight) — LDS copy, lgkmcnt queue
  int* skip_lds = fmap_lds + 1024;
  float* xr = reinterpret_cast<float*>(fmap_lds + 1024 + 256) + wave * 16 * a.xs + j * a.xs;
  for (int i = tid; i < a.NG * 4 * FPL; i += 512) fmap_lds[i] = a.featmap[i];
  for (int i = tid; i < a.NG; i += 512) skip_lds[i] = (int)a.skip[(a.L - 1) * 4 + i];
  __syncthreads();

  const float* bias_last = bias_lds + (a.L - 1) * 256;

  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * 128 + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    const float* xrow = a.x + nc * a.ldx;

    // ---- input tile -> B-operand registers: in[it][r] = input[16 it + 4 q + r] --------------------
    f32x4 in[AR_T], out[AR_T];
#pragma unroll
    for (int it = 0; it < AR_T; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (it * 16 < a.DIN) {
        const int i0 = it * 16 + 4 * q;
        if (i0 < a.DIN) v = *reinterpret_cast<const f32x4*>(xrow + i0);
      }
      in[it] = v;
    }
    // The reference multiplies every input by (mask * W): a NaN or +-inf input therefore turns ALL
    // parameters of its sample into NaN (x * 0 = NaN), including those whose mask excludes that input
    // (zuko/nn.py:217-218).  Skipped tiles would not reproduce that, so the sample is flagged instead.
    float poison = 0.f;
    {
      int bad = 0;
#pragma unroll
      for (int it = 0; it < AR_T; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= !(fabsf(in[it][r]) < __builtin_inff());
      bad |= __shfl_xor(bad, 16, 64);
      bad |= __shfl_xor(bad, 32, 64);
      if (bad) poison = __builtin_nanf("");
    }
    if (XLDS) {
#pragma unroll
      for (int it = 0; it < AR_T; ++it) {
        if (it * 16 < a.D) {
          const int i0 = it * 16 + 4 * q;
          if (i0 < a.D) {
            f32x4 v = in[it];
            if (INVERSE) v = *reinterpret_cast<const f32x4*>(a.yin + nc * a.ldyin + i0);
            *reinterpret_cast<f32x4*>(xr + i0) = v;
          }
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }

    // ---- hidden layers ---------------------------------------------------------------------------
    unsigned long long tstamp[6];  // dbg bit3: phase timestamps (s_memtime) printed by two waves of block 0
    const bool tprobe = ZK_AR_TIMING && (a.dbg & 8) && blockIdx.x == 0 && tile == (int64_t)gridDim.x && lane == 0 && (wave == 0 || wave == 4);
    if (ZK_AR_TIMING && (a.dbg & 8)) tstamp[0] = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < a.L - 1; ++l) {
      hidden_layer(ring, a.skip + l * 4, a.olim[l < 8 ? l : 7], bias_lds + l * 256 + 4 * q, in, out);
      if (ZK_AR_TIMING && (a.dbg & 8)) tstamp[1 + (l < 3 ? l : 3)] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int t = 0; t < AR_T; ++t) in[t] = out[t];
      // the activation id is wave-uniform: ONE switch around 64-element loops (not 64 switches)
      switch (a.act) {
        case 1:
#pragma unroll
          for (int t = 0; t < AR_T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) in[t][r] = in[t][r] < 0.f ? 0.f : in[t][r];  // NaN stays NaN, as torch.relu
          break;
        case 0: break;
        default:
#pragma unroll 1
          for (int rep = 0; rep < 1; ++rep) {
#pragma unroll
            for (int t = 0; t < AR_T; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) in[t][r] = act_f32(in[t][r], a.act);
          }
          break;
      }
    }

    // ---- last layer + univariate transform, one group of 4*FPL features at a time ----------------
    float lacc = 0.f;
    for (int g = a.g0; g < a.g1; ++g) {
      const uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane(skip_lds[g]);
      if (a.sched) ring.end_layer();  // aligned plan: every group starts on a chunk boundary
      unsigned long long tg0 = 0, tg1 = 0;
      if (ZK_AR_TIMING && (a.dbg & 16)) tg0 = __builtin_amdgcn_s_memtime();
      // operands of the epilogue are requested BEFORE the group's MFMAs so their latency is hidden:
      // feature ids + bias from LDS, x[n, f] from global/L2 (one dependent load)
      int fid[FPL];
      float xin[FPL];
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) {
        fid[fi] = fmap_lds[(g * 4 + q) * FPL + fi];
        const int fc = fid[fi] < 0 ? 0 : fid[fi];
        if (XLDS) xin[fi] = xr[fc];
        else xin[fi] = INVERSE ? a.yin[nc * a.ldyin + fc] : xrow[fc];
      }
      f32x4 acc[NT];  // the accumulators start at the bias
      {
        const float* bg = bias_last + (g * NT) * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(bg + t * 16);
      }
#pragma unroll
      for (int it = 0; it < AR_T; ++it) {
        if (bits & (1u << it)) {
          ring.template begin<NT>();
          f32x4 w[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) w[t] = ring.tile(t);
          ring.template commit<NT>();
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][r], in[it][r], acc[t], 0, 0, 0);
        }
      }
      if (ZK_AR_TIMING && (a.dbg & 16)) tg1 = __builtin_amdgcn_s_memtime();
      // Pin the accumulators: without this use the register allocator is free to give the two sides of every skip
      // branch above different VGPR tuples and reconcile them with ~28 v_mov per executed block (+6 % kernel time,
      // measured: 5.80 vs 5.45 ms).  tests/test_codegen.py guards the instruction mix of this loop.
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
      float p[4 * NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * t + r] = acc[t][r];
      // a sample with a non-finite input has NaN parameters throughout in the reference; making the parameters of
      // the SEARCH axis NaN reproduces every output of that case (see Uni::poison) at a third of the additions
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) Uni::template poison<INVERSE>(p, fi * TOTAL, poison);
      auto ld = [&](int i) { return p[i]; };
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) {
        const int f = fid[fi];
        if (f >= 0) {
          const float xv = xin[fi];
          float yv, lj;
          if (ZK_AR_TIMING && (a.dbg & 1)) { yv = xv + p[fi * TOTAL]; lj = p[fi * TOTAL + 1]; }  // ablation: no univariate math
          else if (INVERSE) { yv = Uni::inv(ld, fi * TOTAL, a, xv); lj = 0.f; }
          else if (DIAG) {
            int kb = 0;
            float ks[Uni::NKNOT];
            Uni::fwd(ld, fi * TOTAL, a, xv, yv, lj, &kb, ks);
            if (live) {
              a.bin_out[n * a.D + f] = kb;
#pragma unroll
              for (int jj = 0; jj < Uni::NKNOT; ++jj) a.knots_out[(n * a.D + f) * Uni::NKNOT + jj] = ks[jj];
            }
          } else Uni::fwd(ld, fi * TOTAL, a, xv, yv, lj);
          if (XLDS && !a.sched) xr[f] = yv;
          else if (live) a.y[n * a.ldy + f] = yv;  // (partial sweeps touch a few features only: direct stores)
          lacc += lj;
        }
      }
      if ((ZK_AR_TIMING && (a.dbg & 16)) && blockIdx.x == 0 && tile == (int64_t)gridDim.x && lane == 0 && wave == 4) {
        const unsigned long long tg2 = __builtin_amdgcn_s_memtime();
        printf("grp %2d bits %04x: mfma %llu  epilogue %llu\n", g, bits, tg1 - tg0, tg2 - tg1);
      }
    }
    ring.end_layer();
    if (XLDS && !a.sched) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < AR_T; ++it) {
        if (it * 16 < a.D) {
          const int i0 = it * 16 + 4 * q;
          if (i0 < a.D && live) *reinterpret_cast<f32x4*>(a.y + n * a.ldy + i0) = *reinterpret_cast<const f32x4*>(xr + i0);
        }
      }
    }
    if (ZK_AR_TIMING && (a.dbg & 8)) {
      tstamp[5] = __builtin_amdgcn_s_memtime();
      if (tprobe)
        printf("wave %d: L1 %llu  L2 %llu  L3 %llu  (+bias/act each)  last layer + 16 epilogues %llu   total %llu cycles\n", wave, tstamp[1] - tstamp[0],
               tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[5] - tstamp[3], tstamp[5] - tstamp[0]);
    }
    if (!INVERSE && a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
  }
  ring.drain();  // look-ahead DMAs must land before the LDS is released
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ src, const uint8_t* __restrict__ mask, const int32_t* __restrict__ idx, int64_t n, float* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t k = idx[i];
    float v = 0.f;
    if (k >= 0 && (!mask || mask[k])) v = src[k];
    dst[i] = v;
  }
}

// Weight stream of the operand-split kernels (fused_ar_split_impl.h): per block of 64 lanes x 8 weights, three 1 KiB bf16 images
// h = bf16(w), m = bf16(w - h), l = bf16(w - h - m) (round to nearest even; both differences are exact in f32).
__global__ __launch_bounds__(256) void gather_split_kernel(const float* __restrict__ src, const uint8_t* __restrict__ mask, const int32_t* __restrict__ idx, int64_t n_lanes,
                                                           uint4* __restrict__ dst) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_lanes; i += (int64_t)gridDim.x * 256) {
    const int4 k0 = *reinterpret_cast<const int4*>(idx + i * 8), k1 = *reinterpret_cast<const int4*>(idx + i * 8 + 4);
    const int k[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    bf16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = 0.f;
      if (k[e] >= 0 && (!mask || mask[k[e]])) v = src[k[e]];
      const __bf16 hh = (__bf16)v;
      const float r1 = v - (float)hh;
      const __bf16 mm = (__bf16)r1;
      const float r2 = r1 - (float)mm;
      h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
    }
    const int64_t b = i >> 6;
    const int lane = (int)(i & 63);
    dst[(b * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, h);
    dst[(b * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, m);
    dst[(b * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, l);
  }
}

// Weight stream of the two-part split kernels (fused_ar_half_impl.h): per block two 1 KiB f16 images h = f16(w s), l = f16(w s - h), s a power of two.
__global__ __launch_bounds__(256) void gather_split_f16_kernel(const float* __restrict__ src, const uint8_t* __restrict__ mask, const int32_t* __restrict__ idx, int64_t n_lanes,
                                                               uint4* __restrict__ dst, float scale) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_lanes; i += (int64_t)gridDim.x * 256) {
    const int4 k0 = *reinterpret_cast<const int4*>(idx + i * 8), k1 = *reinterpret_cast<const int4*>(idx + i * 8 + 4);
    const int k[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = 0.f;
      if (k[e] >= 0 && (!mask || mask[k[e]])) v = src[k[e]] * scale;
      const _Float16 hh = (_Float16)v;
      h[e] = hh; l[e] = (_Float16)(v - (float)hh);
    }
    const int64_t b = i >> 6;
    const int lane = (int)(i & 63);
    dst[(b * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, h);
    dst[(b * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, l);
  }
}

// Up to eight gathers (f32 elements or operand-split blocks) in one launch: the streams and bias images of a conditioner are a dozen tiny
// gathers that the GPU finishes faster than the host queues them (training re-gathers every step).
struct GatherMulti {
  int n;
  int start[9];  // first block of every gather, start[n] = total
  struct One { const float* src; const uint8_t* mask; const int32_t* idx; int64_t count; void* dst; int split; } g[8];
};
__global__ __launch_bounds__(256) void gather_multi_kernel(GatherMulti m) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const int b = (int)blockIdx.x;
  GatherMulti::One g = m.g[0];
  int base = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k)
    if (k < m.n && b >= m.start[k]) { g = m.g[k]; base = m.start[k]; }
  const int64_t i = (int64_t)(b - base) * 256 + threadIdx.x;
  if (!g.split) {
    if (i >= g.count) return;
    const int32_t k = g.idx[i];
    float v = 0.f;
    if (k >= 0 && (!g.mask || g.mask[k])) v = g.src[k];
    reinterpret_cast<float*>(g.dst)[i] = v;
    return;
  }
  if (i >= g.count * 64) return;
  const int4 k0 = *reinterpret_cast<const int4*>(g.idx + i * 8), k1 = *reinterpret_cast<const int4*>(g.idx + i * 8 + 4);
  const int k[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
  bf16x8 h, mm_, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (k[e] >= 0 && (!g.mask || g.mask[k[e]])) v = g.src[k[e]];
    const __bf16 hh = (__bf16)v;
    const float r1 = v - (float)hh;
    const __bf16 mm = (__bf16)r1;
    h[e] = hh; mm_[e] = mm; l[e] = (__bf16)(r1 - (float)mm);
  }
  const int64_t blk = i >> 6;
  const int lane = (int)(i & 63);
  uint4* dst = reinterpret_cast<uint4*>(g.dst);
  dst[(blk * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, h);
  dst[(blk * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, mm_);
  dst[(blk * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, l);
}

}  // namespace zk

using namespace zk;

extern "C" {

// dst[i] = idx[i] < 0 ? 0 : (mask && !mask[idx[i]] ? 0 : src[idx[i]])   (weight-stream / bias-image prep)
int zk_gather_f32(const void* src, const uint8_t* mask, const int32_t* idx, int64_t n, void* dst, void* stream) {
  if (n <= 0) return 0;
  int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, (const float*)src, mask, idx, n, (float*)dst);
  return ZK_LAUNCH_CHECK();
}

// Weight stream of an operand-split static-shape kernel: idx [n_blocks * 512] (lane-major, 8 per lane; -1 = zero) into src, mask as
// zk_gather_f32; dst receives 3 KiB per block (bf16 images h, m, l of 64 lanes x 8 values).
int zk_gather_split_bf16(const void* src, const uint8_t* mask, const int32_t* idx, int64_t n_blocks, void* dst, void* stream) {
  if (n_blocks <= 0) return 0;
  if (!src || !idx || !dst || ((uintptr_t)idx % 16) || ((uintptr_t)dst % 16)) return ZK_EINVAL;
  const int64_t n_lanes = n_blocks * 64, nb = (n_lanes + 255) / 256;
  hipLaunchKernelGGL(gather_split_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, (const float*)src, mask, idx, n_lanes, (uint4*)dst);
  return ZK_LAUNCH_CHECK();
}

// Weight stream of a two-part operand-split kernel: as zk_gather_split_bf16 with the weights multiplied by `scale` (a power of two) and two f16 images
// per block (2 KiB).
int zk_gather_split_f16(const void* src, const uint8_t* mask, const int32_t* idx, int64_t n_blocks, void* dst, double scale, void* stream) {
  if (n_blocks <= 0) return 0;
  if (!src || !idx || !dst || ((uintptr_t)idx % 16) || ((uintptr_t)dst % 16) || !(scale > 0.0) || !(scale < 1e38)) return ZK_EINVAL;
  const int64_t n_lanes = n_blocks * 64, nb = (n_lanes + 255) / 256;
  hipLaunchKernelGGL(gather_split_f16_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, (const float*)src, mask, idx, n_lanes, (uint4*)dst, (float)scale);
  return ZK_LAUNCH_CHECK();
}

// Up to eight zk_gather_f32 / zk_gather_split_bf16 in one launch; `descs`: HOST array (include/zuko_amd.h: zk_gather_desc_v1).
int zk_gather_multi(int n, const zk_gather_desc_v1* descs, void* stream) {
  if (n < 1 || n > 8 || !descs) return ZK_EINVAL;
  GatherMulti m{};
  m.n = n;
  int blocks = 0;
  for (int k = 0; k < n; ++k) {
    const zk_gather_desc_v1& d = descs[k];
    if (d.struct_size != sizeof(zk_gather_desc_v1) || d.count < 0 || (d.count > 0 && (!d.src || !d.idx || !d.dst))) return ZK_EINVAL;
    if (d.split && (((uintptr_t)d.idx % 16) || ((uintptr_t)d.dst % 16))) return ZK_EINVAL;
    m.g[k] = {(const float*)d.src, d.mask, d.idx, d.count, d.dst, d.split};
    m.start[k] = blocks;
    const int64_t threads = d.split ? d.count * 64 : d.count;
    if (threads > (int64_t)0x7fffffff) return ZK_EINVAL;
    blocks += (int)((threads + 255) / 256);
  }
  m.start[n] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(gather_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m);
  return ZK_LAUNCH_CHECK();
}

#ifndef AR_CH
#define AR_CH 24
#define AR_NR 3
#endif
typedef RingT<AR_CH, AR_NR> Ring24x3;  // 3 x 24 KiB; 2 x 48 and 3 x 48 tiles measured within +-1 %
static int ar_base_lds_floats(int bias_floats) { return AR_CH * AR_NR * AR_TF + bias_floats + 1024 + 256; }  // ring + bias + feature map + skip words
int zk_ar_lds_bytes(int variant, int bias_floats) { return (ar_base_lds_floats(bias_floats) + 8 * 16 * 260) * (int)sizeof(float); }  // upper bound incl. x/y tiles

// uni_kind: 0 = affine (total 2), 1 = RQS with 8 bins (total 23); contract in include/zuko_amd.h.
struct ArPartial {  // optional: evaluate only last-layer groups [g0, g1) and the prefix of the network they depend on
  const void* static_fn = nullptr;  // launcher of a generated static-shape kernel: the launch goes there after the argument checks
  int rev = 0;                      // ... with the alternative first-layer pattern (descending feature order)
  float* act_out[3] = {nullptr, nullptr, nullptr};  // ... conditioner-only (training) instantiation: hidden activations and phi
  float* phi_out = nullptr;
  int64_t ldphi = 0;
  int phi_packed = 0;
  const double* gl_nodes01 = nullptr;  // uni_kind 5 (static-shape kernels): quadrature of the SOS map
  const double* gl_weights01 = nullptr;
  double eps = 0.0;                    // uni_kind 6: Bernstein continuation margin
  int32_t* bin_out = nullptr;  // diagnostic launch (forward, spline maps): bin index + search knots
  float* knots_out = nullptr;
  double wdescale[4] = {0.0, 0.0, 0.0, 0.0};  // two-part split kernels: 1 / (the power of two every layer's weights were stored with)
  unsigned* amax[4] = {nullptr, nullptr, nullptr, nullptr};  // training launches: maxima of the stored tensors (ArArgs::amax)
  const int* sched = nullptr;
  int n_sched = 0;
  const int* olim = nullptr;  // host array, one entry per hidden layer
  int g0 = 0, g1 = -1;
};

static int ar_launch(const ArPartial& part, bool inverse, int uni_kind, int64_t N, int D, int DIN, const void* x, int64_t ldx, const void* yin, int64_t ldyin, void* y, int64_t ldy,
                     void* ladj, int accumulate, const void* wstream, const void* bias, int bias_floats, const uint32_t* skip, const int32_t* featmap,
                     int n_layers, int n_groups, int n_chunks, int act, double bound, double slope, int variant, void* stream) {
  if (N <= 0) return 0;
  if (n_groups * 8 > 1024 || n_groups > 256) return ZK_EINVAL;
  if (n_layers < 2 || DIN > (part.static_fn ? 512 : 256) || DIN < D || DIN % 4 || ldx % 4 || ((uintptr_t)x % 16) || n_chunks < 1) return ZK_EINVAL;
  ArArgs a{};
  a.N = N; a.D = D; a.DIN = DIN;
  a.x = (const float*)x; a.ldx = ldx;
  a.yin = (const float*)yin; a.ldyin = ldyin;
  a.y = (float*)y; a.ldy = ldy; a.ladj = (float*)ladj; a.accumulate = accumulate;
  a.stream = (const float*)wstream; a.bias = (const float*)bias; a.skip = skip; a.featmap = featmap;
  a.L = n_layers; a.NG = n_groups; a.n_chunks = n_chunks; a.act = act; a.bias_floats = bias_floats;
  a.bound = (float)bound; a.ls = (float)log(slope);
  a.lc = rqs_lean_const(bound, log(slope));
  a.n_tiles = (N + 127) / 128;
  a.g0 = 0; a.g1 = n_groups;
  for (int l = 0; l < 8; ++l) a.olim[l] = 3;
  if (part.sched) {
    if (part.n_sched < 1 || part.g0 < 0 || part.g1 > n_groups || part.g0 >= part.g1 || n_layers - 1 > 8) return ZK_EINVAL;
    a.sched = part.sched; a.n_sched = part.n_sched; a.g0 = part.g0; a.g1 = part.g1;
    for (int l = 0; l < n_layers - 1; ++l) a.olim[l] = part.olim[l];
  }
#if ZK_AR_TIMING
  a.dbg = (variant >> 8) & 0xff;  // probe build only (-DZK_AR_TIMING=1): bit0 skip univariate math, bit3 / bit4 phase timestamps
  if ((variant & 0xff) != 0) return ZK_EINVAL;
#else
  if (variant != 0) return ZK_EINVAL;  // reserved
#endif
  if (part.static_fn) {  // a generated static-shape kernel: it derives its own tiling / LDS size from its Shape and re-checks the dimensions
    if (inverse || part.sched || ((part.bin_out != nullptr) != (part.knots_out != nullptr))) return ZK_EINVAL;
    a.bin_out = part.bin_out; a.knots_out = part.knots_out;  // (diagnostic twin of an operand-split kernel; the f32 static kernels decline)
    a.l1rev = part.rev;
    for (int l = 0; l < 3; ++l) a.act_out[l] = part.act_out[l];
    a.phi_out = part.phi_out; a.ldphi = part.ldphi; a.phi_packed = part.phi_packed;
    if (uni_kind == 5) {  // shifted SOS polynomial, 3 x 5 coefficients (zuko/transforms.py:905-963): MonotonicTransform's bound 10 unless given
      const bool conditioner_only = part.phi_out != nullptr && y == nullptr;  // (training launch without the map: no quadrature needed)
      if ((!part.gl_nodes01 || !part.gl_weights01) && !conditioner_only) return ZK_EINVAL;
      a.sos.bound = (float)bound; a.sos.slope = (float)slope; a.sos.P = 3; a.sos.L1 = 5;
      if (part.gl_nodes01 && part.gl_weights01)
        for (int i = 0; i < 5; ++i) { a.sos.node[i] = (float)part.gl_nodes01[i]; a.sos.weight[i] = (float)part.gl_weights01[i]; }
    }
    a.eps = (float)(part.eps > 0.0 ? part.eps : 1e-6);
    for (int l = 0; l < 4; ++l) a.wdescale[l] = (float)part.wdescale[l];
    for (int l = 0; l < 4; ++l) a.amax[l] = part.amax[l];
    return ((ars_launch_fn)part.static_fn)(&a, ARS_ABI, (int)sizeof(ArArgs), part.phi_out != nullptr, stream);
  }
  // stage x / results through LDS when rows are float4-addressable and the tiles fit beside the ring
  a.xs = ((D + 3) / 4) * 4 + 4;  // +4 words: 16-byte aligned rows whose stride is not a multiple of 32 banks
  const bool vec_ok = (D % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)y % 16 == 0) && (!inverse || ((ldyin % 4 == 0) && ((uintptr_t)yin % 16 == 0)));
  a.xlds = vec_ok && (ar_base_lds_floats(bias_floats) + 8 * 16 * a.xs) * 4 <= 160 * 1024;
  const int lds = (ar_base_lds_floats(bias_floats) + (a.xlds ? 8 * 16 * a.xs : 0)) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  const unsigned grid = (unsigned)(a.n_tiles < 256 ? a.n_tiles : 256);
  const void* fn = nullptr;
#define ZK_AR_PICK(UNI)                                                                                                                     \
  (inverse ? (a.xlds ? (const void*)ar_kernel<UNI, true, Ring24x3, true> : (const void*)ar_kernel<UNI, true, Ring24x3, false>)            \
           : (a.xlds ? (const void*)ar_kernel<UNI, false, Ring24x3, true> : (const void*)ar_kernel<UNI, false, Ring24x3, false>))
  // kinds 2-4 (4 / 16 bins, circular 8 bins) are built for the LDS-staged variant only
#define ZK_AR_PICK_X(UNI) (inverse ? (const void*)ar_kernel<UNI, true, Ring24x3, true> : (const void*)ar_kernel<UNI, false, Ring24x3, true>)
  if (uni_kind == 0) fn = ZK_AR_PICK(UniAffine);
  else if (uni_kind == 1) fn = ZK_AR_PICK(UniRqs8);
  else if (uni_kind >= 2 && uni_kind <= 4 && !a.xlds) return ZK_EINVAL;
  else if (uni_kind == 2) fn = ZK_AR_PICK_X(UniRqs4);
  else if (uni_kind == 3) fn = ZK_AR_PICK_X(UniRqs16);
  else if (uni_kind == 4) fn = ZK_AR_PICK_X(UniCircRqs8);
  else return ZK_EINVAL;
  if (part.bin_out || part.knots_out) {  // diagnostic twin: same template, same arithmetic, extra stores
    if (inverse || part.sched || !a.xlds || !part.bin_out || !part.knots_out) return ZK_EINVAL;
    a.bin_out = part.bin_out; a.knots_out = part.knots_out;
    if (uni_kind == 1) fn = (const void*)ar_kernel<UniRqs8, false, Ring24x3, true, true>;
    else if (uni_kind == 2) fn = (const void*)ar_kernel<UniRqs4, false, Ring24x3, true, true>;
    else if (uni_kind == 3) fn = (const void*)ar_kernel<UniRqs16, false, Ring24x3, true, true>;
    else return ZK_EINVAL;
  }
  hipError_t e = hipSuccess;
  {  // the opt-in to > 64 KiB of dynamic LDS is per function: set it once (and again only if a larger size is asked for)
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
  void* kargs[] = {&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

// (argument block: include/zuko_amd.h — every entry point checks struct_size / version before it reads a field)
// A caller compiled against the first layout of version 1 (which ended with gh3) passes a SHORTER block: it is accepted and the fields
// it does not have (phi_packed, gl_nodes01, gl_weights01, eps) read as zero — the point of carrying struct_size.  A block longer than this
// library knows, another version, or one cut inside the original fields is ZK_EINVAL.
static bool ar_args_norm(const zk_ar_args_v1* p, zk_ar_args_v1* out) {
  if (!p || p->version != 1 || p->struct_size < offsetof(zk_ar_args_v1, phi_packed) || p->struct_size > sizeof(zk_ar_args_v1)) return false;
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out, p, p->struct_size);
  out->struct_size = sizeof(zk_ar_args_v1);
  return true;
}
#define AR_ARGS_OK(args) (ar_args_norm(args, &args##_norm_) ? ((args) = &args##_norm_, true) : false)

static int ar_launch_v1(const ArPartial& part, bool inverse, const zk_ar_args_v1& p, void* stream) {
  return ar_launch(part, inverse, p.uni_kind, p.N, p.D, p.DIN, p.x, p.ldx, inverse ? p.y_in : nullptr, inverse ? p.ldy : 0, inverse ? p.x_out : p.y, inverse ? p.ldo : p.ldy,
                   inverse ? nullptr : p.ladj, p.accumulate, p.wstream, p.bias, p.bias_floats, p.skip, p.featmap, p.n_layers, p.n_groups, p.n_chunks, p.act, p.bound, p.slope, 0,
                   stream);
}

int zk_ar_forward(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args)) return ZK_EINVAL;
  return ar_launch_v1(ArPartial{}, false, *args, stream);
}

// Diagnostic twin of zk_ar_forward for the spline maps (uni_kind 1-3): identical kernel template and arithmetic, plus
// bin_out[N, D] (k = #(knots < x) - 1, zuko/transforms.py:521-523) and knots_out[N, D, K+1] (the horizontal knots the
// search compared).  Lets the tests assert the bin index of the FUSED path on its own knots.
int zk_ar_forward_diag(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args)) return ZK_EINVAL;
  ArPartial part;
  part.bin_out = args->bin_out; part.knots_out = args->knots_out;
  zk_ar_args_v1 p = *args;
  p.accumulate = 0;
  return ar_launch_v1(part, false, p, stream);
}

// zk_ar_forward through a generated static-shape kernel (zuko_amd/static_ar.py): `launcher` is the address of the `zk_ars_launch`
// symbol of the kernel's shared object, `rev` selects the alternative first-layer pattern the kernel was generated with (descending
// feature order).  wstream is the PER-TILE stream of the plan (ArPlan.fine_gather), n_chunks its length.  The kernel re-checks
// D / DIN / n_layers / n_groups / n_chunks against the shape it was generated for and returns hipErrorInvalidValue on a mismatch.
int zk_ar_forward_static(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args) || !args->launcher) return ZK_EINVAL;
  ArPartial part;
  part.static_fn = args->launcher; part.rev = args->rev;
  part.gl_nodes01 = args->gl_nodes01; part.gl_weights01 = args->gl_weights01; part.eps = args->eps;
  part.bin_out = args->bin_out; part.knots_out = args->knots_out;  // both set: the kernel's diagnostic twin (operand-split kernels only)
  part.wdescale[0] = args->wdescale0; part.wdescale[1] = args->wdescale1; part.wdescale[2] = args->wdescale2; part.wdescale[3] = args->wdescale3;
  zk_ar_args_v1 p = *args;
  p.skip = nullptr;  // (act: checked by the kernel against the activation it was generated for)
  return ar_launch_v1(part, false, p, stream);
}

// Training launch of a generated static-shape kernel: phi [N, D * total] = net(x) in module order plus the hidden activations
// h_l [N, width_l] (up to three; units in the stream's sorted order), for the backward pass of zuko_amd/train.py.  With y != NULL (operand-split
// kernels only) the same launch also yields y [N, D] and ladj [N] as zk_ar_forward_static does (bound, slope, accumulate are read then).
int zk_ar_forward_train(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args) || !args->launcher || !args->phi || !args->h1) return ZK_EINVAL;
  const int n_layers = args->n_layers;
  if (n_layers < 2 || n_layers > 4 || (n_layers > 2 && !args->h2) || (n_layers > 3 && !args->h3)) return ZK_EINVAL;
  if (((uintptr_t)args->h1 % 16) || ((uintptr_t)args->h2 % 16) || ((uintptr_t)args->h3 % 16)) return ZK_EINVAL;
  ArPartial part;
  part.static_fn = args->launcher; part.rev = args->rev;
  part.act_out[0] = (float*)args->h1; part.act_out[1] = (float*)args->h2; part.act_out[2] = (float*)args->h3; part.phi_out = (float*)args->phi; part.ldphi = args->ldphi;
  part.phi_packed = args->phi_packed;
  part.amax[0] = args->amax0; part.amax[1] = args->amax1; part.amax[2] = args->amax2; part.amax[3] = args->amax3;
  zk_ar_args_v1 p = *args;
  p.act = 1; p.skip = nullptr;
  if (!p.y) {  // conditioner only
    p.ldy = 0; p.ladj = nullptr; p.accumulate = 0; p.bound = 1.0; p.slope = 1e-3;
  }
  return ar_launch_v1(part, false, p, stream);
}

// The backward twin of zk_ar_forward_train for a ReLU conditioner: dgrad through every linear layer but the last in ONE launch of a
// generated kernel (zuko_amd/static_ar.py: chain_kernel; `launcher` = its zk_ars_dgrad_launch).  x = g of the LAST hidden layer's
// pre-activations [N, DIN = its width]; h1.. = the forward's hidden activations (gates), gh1.. = where the gradient of every earlier
// hidden layer's pre-activations goes ([N, width_l], sorted unit order), y [N, D] (row stride ldy) = the gradient w.r.t. the
// conditioner's input.  wstream = the kernel's weight stream (transposed masked weights in its tile order), n_chunks its length.
typedef int (*ars_dgrad_fn)(const ArArgs* a, int abi, int args_bytes, void* stream);
int zk_ar_dgrad_chain(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args) || !args->launcher || !args->x || !args->y || !args->wstream) return ZK_EINVAL;
  const int n = args->n_layers;
  if (n < 2 || n > 4 || args->N < 0 || args->N > 0x7fffffff) return ZK_EINVAL;
  if (args->N == 0) return 0;
  const void* hs[3] = {args->h1, args->h2, args->h3};
  void* gs[3] = {args->gh1, args->gh2, args->gh3};
  ArArgs a{};
  a.x = (const float*)args->x; a.ldx = args->ldx; a.N = args->N; a.D = args->D; a.DIN = args->DIN; a.L = n - 1; a.n_chunks = args->n_chunks;
  a.stream = (const float*)args->wstream;
  a.phi_out = (float*)args->y; a.ldphi = args->ldy;
  for (int c = 0; c + 2 < n; ++c) {  // chain layer c gates with (and yields the gradient of) hidden layer n - 2 - c (1-based)
    a.gate[c] = (const float*)hs[n - 3 - c];
    a.act_out[c] = (float*)gs[n - 3 - c];
  }
  return ((ars_dgrad_fn)args->launcher)(&a, ARS_ABI, (int)sizeof(ArArgs), stream);
}

// zk_ar_dgrad_chain extended to the LAST linear layer: x = gradient of the packed parameters phi [N, DIN = features * total] (row stride
// ldx), h1 .. h_{n-1} the forward's hidden activations (gates), gh1 .. gh_{n-1} receive the gradients of all hidden layers'
// pre-activations, y the gradient w.r.t. the conditioner's input; `launcher` = zk_ars_dgrad_launch of an operand-split chain kernel
// (zuko_amd/static_ar.py: chain_split_tables), whose first layer streams x from global memory.
int zk_ar_dgrad_full(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args) || !args->launcher || !args->x || !args->y || !args->wstream) return ZK_EINVAL;
  const int n = args->n_layers;
  if (n < 2 || n > 4 || args->N < 0 || args->N > 0x7fffffff) return ZK_EINVAL;
  if (args->N == 0) return 0;
  const void* hs[3] = {args->h1, args->h2, args->h3};
  void* gs[3] = {args->gh1, args->gh2, args->gh3};
  ArArgs a{};
  a.x = (const float*)args->x; a.ldx = args->ldx; a.N = args->N; a.D = args->D; a.DIN = args->DIN; a.L = n; a.n_chunks = args->n_chunks;
  a.stream = (const float*)args->wstream;
  a.phi_out = (float*)args->y; a.ldphi = args->ldy; a.accumulate = args->accumulate;
  for (int c = 0; c + 1 < n; ++c) {  // chain layer c gates with (and yields the gradient of) hidden layer n - 1 - c (1-based)
    a.gate[c] = (const float*)hs[n - 2 - c];
    a.act_out[c] = (float*)gs[n - 2 - c];
  }
  return ((ars_dgrad_fn)args->launcher)(&a, ARS_ABI, (int)sizeof(ArArgs), stream);
}

// The whole backward of one autoregressive transform (x = cat(features, context) as for the forward) up to the weight gradients, in one launch of a generated kernel
// (zuko_amd/static_ar.py: chain_split_tables(packed=...), csrc/fused_ar_split_impl.h: arxb_kernel): from (gy, gl) = d loss / d (y, ladj)
// the univariate adjoint gives d loss / d phi — written to x_out for the weight gradients — and the map's own d/dx term; the dgrad chain
// runs from there through every linear layer (gh1 .. = gradients of the hidden pre-activations); y = d loss / dx = chain + direct term.
int zk_ar_backward_full(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args) || !args->launcher || !args->x || !args->y || !args->wstream || !args->phi || !args->x_out || !args->y_in || !args->ladj || !args->featmap) return ZK_EINVAL;
  const int n = args->n_layers;
  if (n < 2 || n > 4 || args->N < 0 || args->N > 0x7fffffff || args->uni_kind < 0 || args->uni_kind > 1 || args->DIN < args->D || args->D < 1) return ZK_EINVAL;
  if (args->N == 0) return 0;
  const void* hs[3] = {args->h1, args->h2, args->h3};
  void* gs[3] = {args->gh1, args->gh2, args->gh3};
  ArArgs a{};
  a.x = (const float*)args->x; a.ldx = args->ldx; a.N = args->N; a.D = args->D; a.DIN = args->DIN; a.L = n; a.NG = args->n_groups; a.n_chunks = args->n_chunks;
  a.stream = (const float*)args->wstream; a.featmap = args->featmap;
  a.phi_out = (float*)args->y; a.ldphi = args->ldy; a.accumulate = args->accumulate;
  a.gy = (const float*)args->y_in; a.ldgy = args->ldo; a.gl = (const float*)args->ladj;
  a.phi_in = (const float*)args->phi; a.gphi_out = (float*)args->x_out; a.ldpin = args->ldphi; a.phi_packed = 1;
  a.bound = (float)args->bound; a.ls = (float)log(args->slope);
  for (int c = 0; c + 1 < n; ++c) {
    a.gate[c] = (const float*)hs[n - 2 - c];
    a.act_out[c] = (float*)gs[n - 2 - c];
  }
  a.amax[0] = args->amax0; a.amax[1] = args->amax1; a.amax[2] = args->amax2; a.amax[3] = args->amax3;
  return ((ars_dgrad_fn)args->launcher)(&a, ARS_ABI, (int)sizeof(ArArgs), stream);
}

// One sweep of the autoregressive inverse (zuko/transforms.py:997-998): x_out = univariate(conditioner(x_cond)).inv(y).
// x_out may alias x_cond (a wave reads its rows of x_cond completely before it writes them).
int zk_ar_inverse_sweep(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args)) return ZK_EINVAL;
  return ar_launch_v1(ArPartial{}, true, *args, stream);
}

// Partial inverse sweep: as zk_ar_inverse_sweep, but only the features of last-layer groups [g0, g1)
// are updated, and only the prefix of the conditioner they depend on is evaluated: `sched` (device,
// n_sched chunk ids of a plan built with group-aligned chunks) lists the weight-stream chunks in
// consumption order, `olim` (host, one int per hidden layer) the last out-group of 4 tiles to compute.
// After sweep p of the reference loop only the features of order <= p are final and only they matter
// to later sweeps, so running, for s = 0..passes-1, the partial sweep of the groups holding order s
// yields the same x as `passes` full sweeps at a fraction of the work (hard part 4).
int zk_ar_inverse_partial(const zk_ar_args_v1* args, void* stream) {
  zk_ar_args_v1 args_norm_;
  if (!AR_ARGS_OK(args) || !args->sched || !args->olim) return ZK_EINVAL;
  ArPartial part;
  part.sched = args->sched; part.n_sched = args->n_sched; part.olim = args->olim; part.g0 = args->g0; part.g1 = args->g1;
  return ar_launch_v1(part, true, *args, stream);
}

}  // extern "C"
