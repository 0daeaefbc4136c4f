// zuko_amd — fused COUPLING transform: dense conditioner MLP + affine map + log|det J| in one launch.
//
// Replaces, for one GeneralCouplingTransform of NICE / RealNVP (zuko/flows/coupling.py:128-136 `meta`,
// zuko/transforms.py:1040-1048 / :1068-1073 `CouplingTransform`):
//     x_a, x_b = split(x);  phi = MLP(cat(x_a, c))  (zuko/nn.py:13-15 per layer + activations);
//     y_b, ladj = MonotonicAffineTransform(*unpack(phi)).call_and_ladj(x_b)  (zuko/transforms.py:436-446);  y = merge(x_a, y_b)
// Layer by layer this moves every hidden activation through HBM ([N, 512] fp32 = 1 GiB per layer at cfg4's 2^19 rows per
// GPU); here a wavefront carries 16 samples through all layers with the activations in registers, exactly as the
// autoregressive kernel (fused_ar.hip) does, but for hidden widths up to 512:
//   * every layer is computed transposed, H^T = W X^T, on v_mfma_f32_16x16x4_f32 (exact fp32); the D fragment of a layer
//     is the B operand of the next one, so nothing is shuffled or staged between layers;
//   * 32 + 32 activation tiles = 256 VGPRs/AGPRs: one wavefront per SIMD (launch_bounds(256, 1)), four wavefronts = 64
//     samples per workgroup share the weight stream (2.9 MB per transform at cfg4, L2-resident) through a 3 x 24 KiB LDS
//     ring filled by global_load_lds two chunks ahead;
//   * the split / merge of CouplingTransform is index arithmetic on a wave-private LDS image of the 16 input rows: the
//     conditioner's B operands are gathered from it (idx_a), the affine map overwrites the moved columns (idx_b) in
//     place, and the image leaves as whole rows.
// Host-side planning (stream order, bias image, index maps): zuko_amd/coupling_plan.py.
#include "../../include/zuko_amd.h"
#include "zk_univariate.h"

#include <type_traits>
#include <utility>

namespace zk {

typedef float f32x4c __attribute__((ext_vector_type(4)));

#define CP_T 32      /* activation tiles (hidden width <= 512) */
#define CP_IT 16     /* input tiles (conditioner inputs <= 256) */
#define CP_CH 24
#define CP_NR 3
#define CP_WAVES 4
#define CP_MAXL 8

struct CpArgs {
  int64_t N;
  int D, C;                     // features of x, context width
  const float* x; int64_t ldx;
  const float* ctx; int64_t ldc;
  float* y; int64_t ldy;
  float* ladj; int accumulate;
  const float* stream;
  const float* bias;
  const int32_t* amap;          // [nit * 16]: column of x (>= 0), -(2 + c) for context column c, -1 for padding
  const int32_t* fmap;          // [NG * 8]: column of x the slot's feature lives in, -1 = padding
  int L;                        // linear layers (>= 2)
  int nit;                      // input tiles of the first layer
  int wt[CP_MAXL];              // output tiles of every hidden layer
  int width[CP_MAXL];           // hidden widths (units beyond are padding)
  int NG;                       // groups of 8 moved features
  int n_chunks, act, bias_floats;
  int bias_off[CP_MAXL];
  int xs;                       // LDS row stride (floats) of the wave-private row image
  int vec4;                     // rows of y can be written 16 bytes per lane
  int vec4_in;                  // rows of x / context can be fetched 16 bytes per lane
  int inverse;                  // x holds y: the moved half is mapped back (x_b = (y_b - shift) exp(-scale)); ladj stays that of the forward map
  float ls;
  int64_t n_tiles;
  float wdescale[CP_MAXL];      // two-part (f16) kernel: 2^-ew of every linear layer, the power of two its weights were stored with
};

__device__ __attribute__((noinline)) float cp_act_slow(float v, int act) {
  switch (act) {
    case 2: return v > 0.f ? v : expm1f(v);
    case 3: return tanhf(v);
    case 4: return v / (1.f + expf(-v));
    case 5: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case 6: return 1.f / (1.f + expf(-v));
    case 7: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}

struct CpRing {
  float* lds;
  const float* stream;
  int n_chunks, pos, slot, load_chunk, load_slot, wave, lane;
  // each wave copies CP_CH / CP_WAVES consecutive tiles: one address and one M0 value per four of them, the tile selected by
  // the instruction's immediate offset (scripts/probes/dma_issue_probe.hip: ~40 cycles of issue per vector-memory
  // instruction, ~20 more per M0 write)
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < CP_CH / CP_WAVES) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, (I - 4) * 1024, 0);
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {
    static_assert(CP_CH / CP_WAVES == 6, "immediates -4096 .. +1024 around the wave's fifth tile reach six tiles");
    const int b4 = wave * (CP_CH / CP_WAVES) + 4;
    dma<0>(stream + ((size_t)load_chunk * CP_CH + b4) * 256 + lane * 4, lds + (load_slot * CP_CH + b4) * 256);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == CP_NR) ? 0 : load_slot + 1;
  }
  __device__ __forceinline__ void advance() {  // all wavefronts reach this at the same points of the (uniform) control flow
    // the oldest chunk in flight is the one about to be read: the younger (CP_NR - 2) chunks' DMAs may stay outstanding
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((CP_NR - 2) * (CP_CH / CP_WAVES)) : "memory");
    __builtin_amdgcn_s_barrier();  // bare barrier: __syncthreads() would prepend s_waitcnt vmcnt(0) and drain the look-ahead DMAs
    asm volatile("" ::: "memory");
    issue();
    slot = (slot + 1 == CP_NR) ? 0 : slot + 1;
    pos = 0;
  }
  template <int G> __device__ __forceinline__ void begin() {
    if (pos == CP_CH) advance();
  }
  __device__ __forceinline__ f32x4c tile(int t) const { return *reinterpret_cast<const f32x4c*>(lds + (slot * CP_CH + pos + t) * 256 + lane * 4); }
  template <int G> __device__ __forceinline__ void commit() { pos += G; }
  __device__ __forceinline__ void end_layer() {
    if (pos != 0) pos = CP_CH;
  }
};

// the affine map of one moved feature (zuko/transforms.py:412-446), forward or inverse; l = log|dy/dx| of the FORWARD map either way
__device__ __forceinline__ void cp_affine(const CpArgs& a, float shift, float scale, float v, float& out, float& l) {
  const float lsc = softclip<float, MathFast>(scale, a.ls);
  out = a.inverse ? MathFast::div_safe(v - shift, MathFast::exp(lsc)) : v * MathFast::exp(lsc) + shift;
  l = lsc;
}

// The wave's 16 rows [x | context] -> its LDS image, by LDS-DMA (4 bytes per lane, consecutive lanes = consecutive columns of
// one row): all row pieces are in flight together and are waited for once.  (A load-to-register / ds_write loop pays one
// memory round trip per piece, each queued behind the ring DMAs in flight: measured ~20 % of the pass at D = 256.)
__device__ __forceinline__ void cp_stage_rows(const CpArgs& a, float* xw, int xs, int64_t n0, int lane) {
  const int DC = a.D + a.C;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous pass's reads of the image have returned
  if (a.vec4_in) {  // 16 bytes per lane: a row of 256 columns is one DMA
#pragma unroll 1
  for (int r = 0; r < 16; ++r) {
      const int64_t nr = (n0 + r < a.N) ? n0 + r : a.N - 1;
      const float* xg = a.x + nr * a.ldx;
      const float* cg = a.C ? a.ctx + nr * a.ldc - a.D : xg;
#pragma unroll 1
      for (int c0 = 0; c0 < DC; c0 += 256) {
        const int c = c0 + lane * 4;
        if (c < DC) {
          const float* g = (c < a.D ? xg : cg) + c;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(xw + r * xs + c0), 16, 0, 0);
        }
      }
    }
  } else
#pragma unroll 1
  for (int r = 0; r < 16; ++r) {
    const int64_t nr = (n0 + r < a.N) ? n0 + r : a.N - 1;
    const float* xg = a.x + nr * a.ldx;
    const float* cg = a.C ? a.ctx + nr * a.ldc - a.D : xg;
#pragma unroll 1
    for (int c0 = 0; c0 < DC; c0 += 64) {
      const int c = c0 + lane;
      if (c < DC) {
        const float* g = (c < a.D ? xg : cg) + c;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(xw + r * xs + c0), 4, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Results: whole rows out of the image (16 bytes per lane when the row layout allows it).
__device__ __forceinline__ void cp_store_rows(const CpArgs& a, const float* xw, int xs, int64_t n0, int lane) {
  if (a.vec4) {
#pragma unroll 1
  for (int r = 0; r < 16; ++r) {
      if (n0 + r < a.N)
#pragma unroll 1
        for (int c = lane * 4; c < a.D; c += 256) *reinterpret_cast<f32x4c*>(a.y + (n0 + r) * a.ldy + c) = *reinterpret_cast<const f32x4c*>(xw + r * xs + c);
    }
  } else {
#pragma unroll 1
  for (int r = 0; r < 16; ++r) {
      if (n0 + r < a.N)
#pragma unroll 1
        for (int c = lane; c < a.D; c += 64) a.y[(n0 + r) * a.ldy + c] = xw[r * xs + c];
    }
  }
}

extern __shared__ __attribute__((aligned(16))) float cp_lds[];

// 32 activation tiles as TWO arrays of 16: a single 512-byte array is not promoted to registers by the compiler (it stays in
// scratch memory — measured 4x slower), two 256-byte ones are.  `t` is a compile-time constant at every use after unrolling.
struct CpAct {
  f32x4c (&lo)[16];
  f32x4c (&hi)[16];
  __device__ __forceinline__ f32x4c& operator[](int t) const { return t < 16 ? lo[t & 15] : hi[t & 15]; }
};

// one dense layer: out[ot] = bias + sum_it W[ot, it] in[it]; NIN = static bound of the input tiles
template <int NIN> __device__ __forceinline__ void cp_layer(CpRing& ring, int n_in, int n_out, const float* bias_q, const CpAct& in, const CpAct& out) {
#pragma unroll
  for (int otg = 0; otg < CP_T / 4; ++otg) {
    if (otg * 4 < n_out) {
#pragma unroll
      for (int t = 0; t < 4; ++t) out[otg * 4 + t] = *reinterpret_cast<const f32x4c*>(bias_q + (otg * 4 + t) * 16);
#pragma unroll
      for (int it = 0; it < NIN; ++it) {
        if (it < n_in) {
          ring.template begin<4>();
          f32x4c a[4];
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
  }
  ring.end_layer();
}

__device__ __forceinline__ void cp_activate(const CpAct& h, int n_out, int width, int act, int q) {
#pragma unroll
  for (int t = 0; t < CP_T; ++t) {
    if (t < n_out) {
      if (act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = h[t][r] < 0.f ? 0.f : h[t][r];  // NaN stays NaN, as torch.relu
      } else if (act != 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = cp_act_slow(h[t][r], act);
      }
      // units beyond the layer's width are padding: keep them at exactly 0 (0 * non-finite would poison the next layer)
      if ((t + 1) * 16 > width) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * 16 + 4 * q + r >= width) h[t][r] = 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(256, 1) void coupling_kernel(CpArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 15, q = lane >> 4;

  float* bias_lds = cp_lds + CP_NR * CP_CH * 256;
  int* amap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);
  int* fmap_lds = amap_lds + CP_IT * 16;
  float* xw = reinterpret_cast<float*>(fmap_lds + a.NG * 8 + 3 * CP_MAXL) + (size_t)wave * 16 * a.xs;
  // per-layer scalars are read through LDS: indexing the by-value kernel argument arrays with a run-time layer index
  // would make the compiler spill the whole argument block to scratch memory
  int* lay_lds = fmap_lds + a.NG * 8;  // [3 * CP_MAXL]: tiles, widths, bias offsets
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < CP_MAXL; ++l) { lay_lds[l] = a.wt[l]; lay_lds[CP_MAXL + l] = a.width[l]; lay_lds[2 * CP_MAXL + l] = a.bias_off[l]; }
  }
  for (int i = tid; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
  for (int i = tid; i < a.nit * 16; i += 256) amap_lds[i] = a.amap[i];
  for (int i = tid; i < a.NG * 8; i += 256) fmap_lds[i] = a.fmap[i];

  CpRing ring;
  ring.lds = cp_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < CP_NR - 1; ++i) ring.issue();
  ring.slot = CP_NR - 1;
  ring.pos = CP_CH;
  __syncthreads();

  const int xs = a.xs;
  float* xrow = xw + jl * xs;
  const int DC = a.D + a.C;

  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * 64 + wave * 16;
    const int64_t n = n0 + jl;
    const bool live = n < a.N;

    // ---- the wave's 16 rows [x | context] -> LDS image (coalesced: a row is read by consecutive lanes) ------------
    cp_stage_rows(a, xw, xs, n0, lane);

    // ---- first layer: B operands gathered from the image through idx_a ---------------------------------------------
    f32x4c out_lo[16], out_hi[16], in_lo[16], in_hi[16];
    const CpAct out{out_lo, out_hi}, in{in_lo, in_hi};
#pragma unroll
    for (int it = 0; it < CP_T; ++it) in[it] = f32x4c{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < CP_IT; ++it) {
      f32x4c v = {0.f, 0.f, 0.f, 0.f};
      if (it < a.nit) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int src = amap_lds[it * 16 + 4 * q + r];
          v[r] = src >= 0 ? xrow[src] : (src <= -2 ? xrow[a.D + (-2 - src)] : 0.f);
        }
      }
      in[it] = v;
    }
    int n_prev = __builtin_amdgcn_readfirstlane(lay_lds[0]);
    cp_layer<CP_IT>(ring, a.nit, n_prev, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL]) + 4 * q, in, out);
    cp_activate(out, n_prev, __builtin_amdgcn_readfirstlane(lay_lds[CP_MAXL]), a.act, q);
    // ---- further hidden layers --------------------------------------------------------------------------------------
    for (int l = 1; l < a.L - 1; ++l) {
#pragma unroll
      for (int t = 0; t < CP_T; ++t) in[t] = out[t];
      const int n_out = __builtin_amdgcn_readfirstlane(lay_lds[l]);
      cp_layer<CP_T>(ring, n_prev, n_out, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + l]) + 4 * q, in, out);
      cp_activate(out, n_out, __builtin_amdgcn_readfirstlane(lay_lds[CP_MAXL + l]), a.act, q);
      n_prev = n_out;
    }
    // ---- last layer + affine map, one group of 8 moved features at a time (lane (j, q): slots 2 q, 2 q + 1) -------
    const int n_last_in = n_prev;
    const float* bias_last = bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + a.L - 1]) + 4 * q;
    float lacc = 0.f;
    for (int g = 0; g < a.NG; ++g) {
      const int f0 = fmap_lds[g * 8 + 2 * q], f1 = fmap_lds[g * 8 + 2 * q + 1];
      const float x0 = xrow[f0 < 0 ? 0 : f0], x1 = xrow[f1 < 0 ? 0 : f1];
      f32x4c acc0 = *reinterpret_cast<const f32x4c*>(bias_last + g * 16), acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: a dependent one would wait 40 cycles per MFMA
#pragma unroll
      for (int it = 0; it < CP_T; it += 2) {
        if (it < n_last_in) {
          ring.template begin<1>();
          const f32x4c w0 = ring.tile(0);
          ring.template commit<1>();
#pragma unroll
          for (int r = 0; r < 4; ++r) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], out[it][r], acc0, 0, 0, 0);
        }
        if (it + 1 < n_last_in) {
          ring.template begin<1>();
          const f32x4c w1 = ring.tile(0);
          ring.template commit<1>();
#pragma unroll
          for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], out[it + 1][r], acc1, 0, 0, 0);
        }
      }
      const f32x4c p = acc0 + acc1;  // (shift, scale) of slot 2 q, then of slot 2 q + 1
      float y0, y1, l0, l1;
      cp_affine(a, p[0], p[1], x0, y0, l0);
      cp_affine(a, p[2], p[3], x1, y1, l1);
      if (f0 >= 0) { xrow[f0] = y0; lacc += l0; }
      if (f1 >= 0) { xrow[f1] = y1; lacc += l1; }
    }
    ring.end_layer();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- results: whole rows out of the image -------------------------------------------------------------------------
    cp_store_rows(a, xw, xs, n0, lane);
    if (a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}


// ---- static-shape fast path ----------------------------------------------------------------------------------------------
// Same data flow, with the layer shapes as template parameters (NIT input tiles, every hidden layer HT tiles wide, NG
// groups): no per-tile guards, and the position of every weight tile inside the pass is a compile-time constant, so the ring
// refill (barrier + DMA issue) is emitted only where that position is a multiple of the chunk size and the code between
// two refills is one basic block the scheduler can software-pipeline (ds_read of the next tiles above the current MFMAs).
#ifndef ZK_CP_ABLATE
#define ZK_CP_ABLATE 0  // probe builds only (wrong results): 2 = no chunk barriers / refills, 4 = no row staging / stores, 8 = no s_barrier, 32 = no ring DMAs
#endif
#ifndef ZK_CP_TIMING
#define ZK_CP_TIMING 0  // -DZK_CP_TIMING=1: s_memtime phase probes printed by wave 0 of block 0 (probe build only)
#endif
#define CP_ALWAYS_INLINE __attribute__((always_inline))
template <class F, int... I> __device__ __forceinline__ void cp_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void cp_for(F&& f) { cp_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct CpRingS {
  unsigned long long t_wait = 0, t_bar = 0, t_iss = 0;
  float* lds;
  const float* stream;
  unsigned cur_off;  // LDS byte address of the slot being read + lane * 16
  unsigned lds_off;  // LDS byte address of the ring
  int n_chunks, slot, load_chunk, load_slot, wave, lane;
  static constexpr int kPerWave = CP_CH / CP_WAVES;
  static_assert(kPerWave == 6, "immediates -4096 .. +1024 around the wave's fifth tile reach six tiles");
  // Each wave copies kPerWave consecutive tiles of a chunk, six DMAs back to back on ONE address / M0 value (that of its fifth
  // tile) with the instruction's signed immediate offset: ~40 cycles of issue for the first, ~15 for each further one
  // (scripts/probes/dma_issue_probe.hip).  Spreading them over the chunk (one per step) was measured: no better.
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < kPerWave) {
      if (!(ZK_CP_ABLATE & 32)) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, (I - 4) * 1024, 0);
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {
    const int b4 = wave * kPerWave + 4;
    dma<0>(stream + ((size_t)load_chunk * CP_CH + b4) * 256 + lane * 4, lds + (load_slot * CP_CH + b4) * 256);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == CP_NR) ? 0 : load_slot + 1;
  }
  __device__ __forceinline__ void advance() {
    unsigned long long p0 = 0, p1 = 0, p2 = 0;
    if (ZK_CP_TIMING) p0 = __builtin_amdgcn_s_memtime();
    // my DMAs of the chunk about to be read have landed (the CP_NR - 2 younger chunks stay in flight: a DMA takes ~5 k cycles,
    // one chunk is consumed in ~3 k); my (prefetching) reads of the slot to be refilled have returned
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((CP_NR - 2) * (CP_CH / CP_WAVES)) : "memory");
    if (ZK_CP_TIMING) p1 = __builtin_amdgcn_s_memtime();
    if (!(ZK_CP_ABLATE & 8)) __builtin_amdgcn_s_barrier();  // bare barrier: __syncthreads() would prepend s_waitcnt vmcnt(0) and drain the look-ahead DMAs
    asm volatile("" ::: "memory");
    if (ZK_CP_TIMING) {
      p2 = __builtin_amdgcn_s_memtime();
      t_wait += p1 - p0; t_bar += p2 - p1;
    }
    issue();  // the slot just released
    slot = (slot + 1 == CP_NR) ? 0 : slot + 1;
    cur_off = lds_off + (unsigned)(slot * CP_CH * 1024 + lane * 16);
  }
  // Position S inside the layer (layers start on chunk boundaries; static).  The read is issued from inline assembly and returns
  // a RAW value: the compiler does not know it is an LDS operation and inserts no wait for it — with a global_load_lds in flight
  // hipcc turns every LDS wait into lgkmcnt(0), which would make a step wait for the tiles it has just requested for the NEXT
  // step.  cp_settle<N>() makes the value usable: it waits until at most N younger LDS operations are outstanding (LDS
  // operations of a wave complete in order) and is the only consumer of the raw registers (tests/test_codegen.py checks the ISA).
  template <int S> __device__ __forceinline__ f32x4c read() {
    if constexpr (S % CP_CH == 0) {
      if (!(ZK_CP_ABLATE & 2)) advance();
    }
    f32x4c v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(cur_off), "n"((S % CP_CH) * 1024));
    return v;
  }
};
template <int N> __device__ __forceinline__ void cp_settle(f32x4c& a0, f32x4c& a1, f32x4c& a2, f32x4c& a3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N));
}

template <int NIN, int HT> __device__ __forceinline__ void cp_layer_static(CpRingS& ring, const float* bias_q, const CpAct& in, const CpAct& out) {
  // software pipeline: the four A tiles of step s + 1 are requested (into the other register set) before the 16 MFMAs of
  // step s are issued, so the LDS round trip hides behind 512 cycles of matrix work (one wavefront per SIMD: there is no
  // partner wave to hide it)
  constexpr int STEPS = (HT / 4) * NIN;
  f32x4c a[2][4];
  cp_for<4>([&](auto t) CP_ALWAYS_INLINE { a[0][t] = ring.template read<decltype(t)::value>(); });
  cp_for<STEPS>([&](auto st_) CP_ALWAYS_INLINE {
    constexpr int st = st_, otg = st / NIN, it = st % NIN;
    if constexpr (it == 0) {
      cp_for<4>([&](auto t) CP_ALWAYS_INLINE { out[otg * 4 + t] = *reinterpret_cast<const f32x4c*>(bias_q + (otg * 4 + t) * 16); });
    }
    if constexpr (st + 1 < STEPS) {
      cp_for<4>([&](auto t) CP_ALWAYS_INLINE { a[(st + 1) & 1][t] = ring.template read<(st + 1) * 4 + decltype(t)::value>(); });
      cp_settle<4>(a[st & 1][0], a[st & 1][1], a[st & 1][2], a[st & 1][3]);
    } else {
      cp_settle<0>(a[st & 1][0], a[st & 1][1], a[st & 1][2], a[st & 1][3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    cp_for<4>([&](auto r) CP_ALWAYS_INLINE {
      cp_for<4>([&](auto t) CP_ALWAYS_INLINE { out[otg * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st & 1][t][(int)r], in[it][(int)r], out[otg * 4 + t], 0, 0, 0); });
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int NIT, int HT> __global__ __launch_bounds__(256, 1) void coupling_kernel_static(CpArgs a) {
  static_assert(HT % 4 == 0 && HT <= CP_T && NIT <= CP_IT, "shape");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 15, q = lane >> 4;
  float* bias_lds = cp_lds + CP_NR * CP_CH * 256;
  int* amap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);
  int* fmap_lds = amap_lds + CP_IT * 16;
  int* lay_lds = fmap_lds + a.NG * 8;
  float* xw = reinterpret_cast<float*>(lay_lds + 3 * CP_MAXL) + (size_t)wave * 16 * a.xs;
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < CP_MAXL; ++l) lay_lds[2 * CP_MAXL + l] = a.bias_off[l];
  }
  for (int i = tid; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
  for (int i = tid; i < NIT * 16; i += 256) amap_lds[i] = a.amap[i];
  for (int i = tid; i < a.NG * 8; i += 256) fmap_lds[i] = a.fmap[i];
  CpRingS ring;
  ring.lds = cp_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < CP_NR - 1; ++i) ring.issue();
  ring.slot = CP_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)cp_lds);
  ring.cur_off = ring.lds_off;
  __syncthreads();

  const int xs = a.xs;
  float* xrow = xw + jl * xs;
  const int DC = a.D + a.C;
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, tacc[5] = {0, 0, 0, 0, 0};
  int n_pass = 0;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * 64 + wave * 16;
    const int64_t n = n0 + jl;
    const bool live = n < a.N;
    if (ZK_CP_TIMING) ts[0] = __builtin_amdgcn_s_memtime();
    if (!(ZK_CP_ABLATE & 4)) cp_stage_rows(a, xw, xs, n0, lane);
    if (ZK_CP_TIMING) ts[1] = __builtin_amdgcn_s_memtime();

    f32x4c out_lo[16], out_hi[16], in_lo[16], in_hi[16];
    const CpAct out{out_lo, out_hi}, in{in_lo, in_hi};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      f32x4c v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int src = amap_lds[it * 16 + 4 * q + r];
        v[r] = src >= 0 ? xrow[src] : (src <= -2 ? xrow[a.D + (-2 - src)] : 0.f);
      }
      in[it] = v;
    }
    cp_layer_static<NIT, HT>(ring, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL]) + 4 * q, in, out);
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];
    if (ZK_CP_TIMING) ts[2] = __builtin_amdgcn_s_memtime();
    for (int l = 1; l < a.L - 1; ++l) {
#pragma unroll
      for (int t = 0; t < HT; ++t) in[t] = out[t];
      cp_layer_static<HT, HT>(ring, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + l]) + 4 * q, in, out);
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];
    }
    if (ZK_CP_TIMING) ts[3] = __builtin_amdgcn_s_memtime();
    const float* bias_last = bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + a.L - 1]) + 4 * q;
    float lacc = 0.f;
    // groups are walked in pairs so that the tile positions are static: 2 * HT tiles per pair (HT = 32: 64 tiles, not a
    // multiple of the chunk — the position is carried in a run-time base that only changes by multiples of the chunk)
    static_assert((3 * HT) % CP_CH == 0, "three groups of the last layer must fill whole chunks");
    for (int g3 = 0; g3 < a.NG; g3 += 3) {
      cp_for<3>([&](auto gg_) CP_ALWAYS_INLINE {
        constexpr int gg = gg_;
        const int g = g3 + gg;
        if (g < a.NG) {
          const int f0 = fmap_lds[g * 8 + 2 * q], f1 = fmap_lds[g * 8 + 2 * q + 1];
          const float x0 = xrow[f0 < 0 ? 0 : f0], x1 = xrow[f1 < 0 ? 0 : f1];
          // four accumulators (one per tile of the step): a dependent MFMA every fourth issue, as in the hidden layers
          f32x4c acc[4] = {*reinterpret_cast<const f32x4c*>(bias_last + g * 16), {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          f32x4c w[2][4];  // four tiles per step, the next step requested before this step's MFMAs
          cp_for<4>([&](auto t) CP_ALWAYS_INLINE { w[0][t] = ring.template read<gg * HT + decltype(t)::value>(); });
          cp_for<HT / 4>([&](auto k_) CP_ALWAYS_INLINE {
            constexpr int k = k_, it = 4 * k;
            if constexpr (it + 4 < HT) {
              cp_for<4>([&](auto t) CP_ALWAYS_INLINE { w[(k + 1) & 1][t] = ring.template read<gg * HT + it + 4 + decltype(t)::value>(); });
              cp_settle<4>(w[k & 1][0], w[k & 1][1], w[k & 1][2], w[k & 1][3]);
            } else {
              cp_settle<0>(w[k & 1][0], w[k & 1][1], w[k & 1][2], w[k & 1][3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            cp_for<4>([&](auto r) CP_ALWAYS_INLINE {
              cp_for<4>([&](auto t) CP_ALWAYS_INLINE { acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k & 1][t][(int)r], out[it + t][(int)r], acc[t], 0, 0, 0); });
            });
            __builtin_amdgcn_sched_barrier(0);
          });
          const f32x4c p = (acc[0] + acc[1]) + (acc[2] + acc[3]);
          float y0, y1, l0, l1;
          cp_affine(a, p[0], p[1], x0, y0, l0);
          cp_affine(a, p[2], p[3], x1, y1, l1);
          if (f0 >= 0) { xrow[f0] = y0; lacc += l0; }
          if (f1 >= 0) { xrow[f1] = y1; lacc += l1; }
        }
      });
    }
    if (ZK_CP_TIMING) ts[4] = __builtin_amdgcn_s_memtime();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (!(ZK_CP_ABLATE & 4)) cp_store_rows(a, xw, xs, n0, lane);
    if (a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (ZK_CP_TIMING) {
      ts[5] = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < 5; ++i) tacc[i] += ts[i + 1] - ts[i];
      ++n_pass;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if ZK_CP_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("cp timing (memtime ticks, %d passes): stage %llu layer0 %llu hidden %llu last %llu store %llu | advance: wait %llu barrier %llu issue %llu\n", n_pass,
           tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], ring.t_wait, ring.t_bar, ring.t_iss);
#endif
}

// ---- operand-split twin of the static-shape path ----------------------------------------------------------------------------------
// The f32 matrix instruction bounds the kernels above (0.81 of its peak at cfg4, profiles/r03) and runs at 1/16 of the bf16 rate;
// gfx950 has no tf32.  As in csrc/fused_ar_split_impl.h every f32 operand is written as h + m + l in bf16 (exact to 2^-25) and a
// product is the six partial products down to 2^-18 on v_mfma_f32_16x16x32_bf16 with f32 accumulation: the error of an f32 dot
// product at 6/16 of the matrix time.  A stream BLOCK is the 16 x 32 weight block (out tile, PAIR of in tiles) as three 1 KiB bf16
// images; a lane's 8 values are [4 units of in tile 2 ip | the same 4 units of in tile 2 ip + 1], which is how the activations sit in
// the accumulator registers, so a layer's D fragments become the next layer's B operands by an in-register conversion.  One
// wavefront per SIMD: a step is 4 out tiles x 1 in pair (12 images, 24 matrix instructions issued term by term, so the same
// accumulator is touched every fourth instruction and never waits for its predecessor).
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
struct CpBv {  // B operands (h, m, l parts) of 16 activation pairs, as arrays small enough to be promoted to registers
  cbf16x8 (&hlo)[8]; cbf16x8 (&hhi)[8]; cbf16x8 (&mlo)[8]; cbf16x8 (&mhi)[8]; cbf16x8 (&llo)[8]; cbf16x8 (&lhi)[8];
  __device__ __forceinline__ cbf16x8& h(int p) const { return p < 8 ? hlo[p & 7] : hhi[p & 7]; }
  __device__ __forceinline__ cbf16x8& m(int p) const { return p < 8 ? mlo[p & 7] : mhi[p & 7]; }
  __device__ __forceinline__ cbf16x8& l(int p) const { return p < 8 ? llo[p & 7] : lhi[p & 7]; }
};
__device__ __forceinline__ void cp_split(const f32x4c& lo, const f32x4c& hi, cbf16x8& h, cbf16x8& m, cbf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = e < 4 ? lo[e] : hi[e - 4];
    const __bf16 hh = (__bf16)v;
    const float r1 = v - (float)hh;
    const __bf16 mm = (__bf16)r1;
    h[e] = hh; m[e] = mm; l[e] = (__bf16)(r1 - (float)mm);
  }
}
template <int N> __device__ __forceinline__ void cp_settle12(f32x4c (&a)[12]) {
  asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]) : "n"(N));
}
template <int N> __device__ __forceinline__ void cp_settle6(f32x4c (&a)[6]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "n"(N));
}
// the step's images together with the four bias tiles of an out-group's first step (read raw in front of the look-ahead images: see cp_layer_split)
template <int N> __device__ __forceinline__ void cp_settle12o(f32x4c (&a)[12], f32x4c& o0, f32x4c& o1, f32x4c& o2, f32x4c& o3) {
  asm volatile("s_waitcnt lgkmcnt(%16)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]),
               "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3) : "n"(N));
}
template <int OFF> __device__ __forceinline__ f32x4c cp_lds_raw(unsigned addr) {
  f32x4c v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
#define CP_XMFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8, A), B, C, 0, 0, 0)

// one dense layer: out[ot] = bias + sum_ip W[ot, ip] in[ip]; NP in pairs, HT out tiles
template <int NP, int HT> __device__ __forceinline__ void cp_layer_split(CpRingS& ring, const float* bias_q, const CpBv& in, const CpAct& out) {
  constexpr int STEPS = (HT / 4) * NP;
  f32x4c a[2][12];  // images (t, part) of the step: a[.][3 t + part], part 0 = h, 1 = m, 2 = l
  cp_for<12>([&](auto i) CP_ALWAYS_INLINE { a[0][i] = ring.template read<decltype(i)::value>(); });
  // The out-group's accumulators start at the bias, read RAW in front of the next step's images: older than those twelve, the counted wait of the
  // step settles them too.  (As compiler-visible loads their wait was lgkmcnt(0) — the compiler cannot count the raw reads — in front of the group's
  // first matrix instruction: the twelve look-ahead images drained with it, once per out-group, with no second wavefront on the SIMD to hide it.)
  const unsigned bias_addr = (unsigned)(size_t)((const __attribute__((address_space(3))) float*)bias_q);
  cp_for<STEPS>([&](auto st_) CP_ALWAYS_INLINE {
    constexpr int st = st_, otg = st / NP, ip = st % NP;
    if constexpr (ip == 0) {
      cp_for<4>([&](auto t) CP_ALWAYS_INLINE { out[otg * 4 + t] = cp_lds_raw<(otg * 4 + decltype(t)::value) * 64>(bias_addr); });
    }
    if constexpr (st + 1 < STEPS) {
      cp_for<12>([&](auto i) CP_ALWAYS_INLINE { a[(st + 1) & 1][i] = ring.template read<(st + 1) * 12 + decltype(i)::value>(); });
      if constexpr (ip == 0) cp_settle12o<12>(a[st & 1], out[otg * 4 + 0], out[otg * 4 + 1], out[otg * 4 + 2], out[otg * 4 + 3]);
      else cp_settle12<12>(a[st & 1]);
    } else {
      if constexpr (ip == 0) cp_settle12o<0>(a[st & 1], out[otg * 4 + 0], out[otg * 4 + 1], out[otg * 4 + 2], out[otg * 4 + 3]);
      else cp_settle12<0>(a[st & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const cbf16x8 bh = in.h(ip), bm = in.m(ip), bl = in.l(ip);
    // six partial products, smallest first, each over the four out tiles of the step
    cp_for<4>([&](auto t) CP_ALWAYS_INLINE { CP_XMFMA(a[st & 1][3 * t + 2], bh, out[otg * 4 + t]); });
    cp_for<4>([&](auto t) CP_ALWAYS_INLINE { CP_XMFMA(a[st & 1][3 * t + 0], bl, out[otg * 4 + t]); });
    cp_for<4>([&](auto t) CP_ALWAYS_INLINE { CP_XMFMA(a[st & 1][3 * t + 1], bm, out[otg * 4 + t]); });
    cp_for<4>([&](auto t) CP_ALWAYS_INLINE { CP_XMFMA(a[st & 1][3 * t + 1], bh, out[otg * 4 + t]); });
    cp_for<4>([&](auto t) CP_ALWAYS_INLINE { CP_XMFMA(a[st & 1][3 * t + 0], bm, out[otg * 4 + t]); });
    cp_for<4>([&](auto t) CP_ALWAYS_INLINE { CP_XMFMA(a[st & 1][3 * t + 0], bh, out[otg * 4 + t]); });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ReLU + conversion of the HT out tiles into the next layer's B operands
template <int HT> __device__ __forceinline__ void cp_convert(const CpAct& out, const CpBv& in) {
  cp_for<HT / 2>([&](auto p_) CP_ALWAYS_INLINE {
    constexpr int p = p_;
    f32x4c lo = out[2 * p], hi = out[2 * p + 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lo[r] = lo[r] < 0.f ? 0.f : lo[r];  // NaN stays NaN, as torch.relu
      hi[r] = hi[r] < 0.f ? 0.f : hi[r];
    }
    cp_split(lo, hi, in.h(p), in.m(p), in.l(p));
  });
}

template <int NIT, int HT> __global__ __launch_bounds__(256, 1) void coupling_kernel_split(CpArgs a) {
  static_assert(HT % 4 == 0 && HT <= CP_T && NIT <= CP_IT && NIT % 2 == 0 && (12 * (NIT / 2) * (HT / 4)) % CP_CH == 0 && (12 * (HT / 2) * (HT / 4)) % CP_CH == 0 && (3 * (HT / 2)) % CP_CH == 0,
                "every layer and every group of the last layer fills whole chunks");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 15, q = lane >> 4;
  float* bias_lds = cp_lds + CP_NR * CP_CH * 256;
  int* amap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);
  int* fmap_lds = amap_lds + CP_IT * 16;
  int* lay_lds = fmap_lds + a.NG * 8;
  float* xw = reinterpret_cast<float*>(lay_lds + 3 * CP_MAXL) + (size_t)wave * 16 * a.xs;
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < CP_MAXL; ++l) lay_lds[2 * CP_MAXL + l] = a.bias_off[l];
  }
  for (int i = tid; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
  for (int i = tid; i < NIT * 16; i += 256) amap_lds[i] = a.amap[i];
  for (int i = tid; i < a.NG * 8; i += 256) fmap_lds[i] = a.fmap[i];
  CpRingS ring;
  ring.lds = cp_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < CP_NR - 1; ++i) ring.issue();
  ring.slot = CP_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)cp_lds);
  ring.cur_off = ring.lds_off;
  __syncthreads();

  const int xs = a.xs;
  float* xrow = xw + jl * xs;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * 64 + wave * 16;
    const int64_t n = n0 + jl;
    const bool live = n < a.N;
    cp_stage_rows(a, xw, xs, n0, lane);

    f32x4c out_lo[16], out_hi[16];
    cbf16x8 bhl[8], bhh[8], bml[8], bmh[8], bll[8], blh[8];
    const CpAct out{out_lo, out_hi};
    const CpBv in{bhl, bhh, bml, bmh, bll, blh};
    // first layer: B operands gathered from the row image through idx_a, converted pair by pair
    cp_for<NIT / 2>([&](auto p_) CP_ALWAYS_INLINE {
      constexpr int p = p_;
      f32x4c v[2];
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int src = amap_lds[(2 * p + hlf) * 16 + 4 * q + r];
          v[hlf][r] = src >= 0 ? xrow[src] : (src <= -2 ? xrow[a.D + (-2 - src)] : 0.f);
        }
      cp_split(v[0], v[1], in.h(p), in.m(p), in.l(p));
    });
    cp_layer_split<NIT / 2, HT>(ring, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL]) + 4 * q, in, out);
    cp_convert<HT>(out, in);
    for (int l = 1; l < a.L - 1; ++l) {
      cp_layer_split<HT / 2, HT>(ring, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + l]) + 4 * q, in, out);
      cp_convert<HT>(out, in);
    }
    // last layer + affine map: one group of 8 moved features = one out tile = HT / 2 blocks, two blocks per step on six accumulators
    const float* bias_last = bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + a.L - 1]) + 4 * q;
    float lacc = 0.f;
    for (int g = 0; g < a.NG; ++g) {
      const int f0 = fmap_lds[g * 8 + 2 * q], f1 = fmap_lds[g * 8 + 2 * q + 1];
      const float x0 = xrow[f0 < 0 ? 0 : f0], x1 = xrow[f1 < 0 ? 0 : f1];
      const f32x4c zero = {0.f, 0.f, 0.f, 0.f};
      f32x4c cH[2] = {*reinterpret_cast<const f32x4c*>(bias_last + g * 16), zero}, cM[2] = {zero, zero}, cS[2] = {zero, zero};
      f32x4c w[2][6];  // images of the step's two blocks: w[.][3 b + part]
      constexpr int KS = HT / 4;  // steps per group
      cp_for<6>([&](auto i) CP_ALWAYS_INLINE { w[0][i] = ring.template read<decltype(i)::value>(); });
      cp_for<KS>([&](auto k_) CP_ALWAYS_INLINE {
        constexpr int k = k_;
        if constexpr (k + 1 < KS) {
          cp_for<6>([&](auto i) CP_ALWAYS_INLINE { w[(k + 1) & 1][i] = ring.template read<(k + 1) * 6 + decltype(i)::value>(); });
          cp_settle6<6>(w[k & 1]);
        } else {
          cp_settle6<0>(w[k & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        cp_for<2>([&](auto b) CP_ALWAYS_INLINE { CP_XMFMA(w[k & 1][3 * b + 2], in.h(2 * k + b), cS[b]); });
        cp_for<2>([&](auto b) CP_ALWAYS_INLINE { CP_XMFMA(w[k & 1][3 * b + 1], in.h(2 * k + b), cM[b]); });
        cp_for<2>([&](auto b) CP_ALWAYS_INLINE { CP_XMFMA(w[k & 1][3 * b + 0], in.h(2 * k + b), cH[b]); });
        cp_for<2>([&](auto b) CP_ALWAYS_INLINE { CP_XMFMA(w[k & 1][3 * b + 0], in.l(2 * k + b), cS[b]); });
        cp_for<2>([&](auto b) CP_ALWAYS_INLINE { CP_XMFMA(w[k & 1][3 * b + 0], in.m(2 * k + b), cM[b]); });
        cp_for<2>([&](auto b) CP_ALWAYS_INLINE { CP_XMFMA(w[k & 1][3 * b + 1], in.m(2 * k + b), cS[b]); });
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 15" : "+v"(cS[1]));  // wait states behind the group's last matrix instruction: one wavefront per SIMD, accumulators may sit in AGPRs (csrc/fused_ar_split_impl.h: arx_mfma_guard)
      const f32x4c p = ((cS[0] + cS[1]) + (cM[0] + cM[1])) + (cH[0] + cH[1]);  // (shift, scale) of slot 2 q, then of slot 2 q + 1
      float y0, y1, l0, l1;
      cp_affine(a, p[0], p[1], x0, y0, l0);
      cp_affine(a, p[2], p[3], x1, y1, l1);
      if (f0 >= 0) { xrow[f0] = y0; lacc += l0; }
      if (f1 >= 0) { xrow[f1] = y1; lacc += l1; }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    cp_store_rows(a, xw, xs, n0, lane);
    if (a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- TWO-PART operand split (round 6; csrc/fused_ar_half_impl.h has the arithmetic's full description) ------------------------------
// Every f32 operand as two f16 numbers after an exact power-of-two scaling (weights: per layer, host; activations: per SAMPLE and layer, from the
// largest magnitude of the sample's input vector — a lane holds one sample, two shuffles), three partial products lh + hl + hh on
// v_mfma_f32_16x16x32_f16, the accumulator back through one fma with 2^-(ew + ea) and the bias: half the matrix instructions of
// coupling_kernel_split, two 1 KiB images per 16 x 32 block (16-image chunks: every layer and every group of the last layer fills whole chunks).
// One wavefront per SIMD, more than 256 registers: the accumulators of a step are pinned to VGPRs and the step's matrix instructions are ONE
// assembly block (hipcc under-counts the wait states behind a v_mfma_f32_16x16x32_f16 whose destination is an AGPR: profiles/r06/inverse.md),
// with 12 wait states in front of the vector instructions that read them.
typedef _Float16 cf16x8 __attribute__((ext_vector_type(8)));
#define CPH_CH 16
struct CpRingH {
  float* lds;
  const float* stream;
  unsigned cur_off, lds_off;
  int n_chunks, slot, load_chunk, load_slot, wave, lane;
  static constexpr int kPerWave = CPH_CH / CP_WAVES;
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < kPerWave) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, (I - 2) * 1024, 0);
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {
    const int b2 = wave * kPerWave + 2;
    dma<0>(stream + ((size_t)load_chunk * CPH_CH + b2) * 256 + lane * 4, lds + (load_slot * CPH_CH + b2) * 256);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == CP_NR) ? 0 : load_slot + 1;
  }
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((CP_NR - 2) * (CPH_CH / CP_WAVES)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    slot = (slot + 1 == CP_NR) ? 0 : slot + 1;
    cur_off = lds_off + (unsigned)(slot * CPH_CH * 1024 + lane * 16);
  }
  template <int S> __device__ __forceinline__ f32x4c read() {
    if constexpr (S % CPH_CH == 0) advance();
    f32x4c v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(cur_off), "n"((S % CPH_CH) * 1024));
    return v;
  }
};
struct CpBh {  // B operands (h, l parts) of 16 activation pairs
  cf16x8 (&hlo)[8]; cf16x8 (&hhi)[8]; cf16x8 (&llo)[8]; cf16x8 (&lhi)[8];
  __device__ __forceinline__ cf16x8& h(int p) const { return p < 8 ? hlo[p & 7] : hhi[p & 7]; }
  __device__ __forceinline__ cf16x8& l(int p) const { return p < 8 ? llo[p & 7] : lhi[p & 7]; }
};
__device__ __forceinline__ void cph_split(const f32x4c& lo, const f32x4c& hi, float s, cf16x8& h, cf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (e < 4 ? lo[e] : hi[e - 4]) * s;
    const _Float16 hh = (_Float16)v;
    h[e] = hh; l[e] = (_Float16)(v - (float)hh);
  }
}
// s = 2^ea with amax 2^ea in [2^14, 2^15) (|ea| <= 90; zero / non-finite amax: ea = 15), inv_s = 2^-ea   (arh_scale of fused_ar_half_impl.h)
__device__ __forceinline__ void cph_scale(float amax, float& s, float& inv_s) {
  amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  int ea = 15 - __builtin_amdgcn_frexp_expf(amax);
  ea = ea > 90 ? 90 : (ea < -90 ? -90 : ea);
  s = __builtin_amdgcn_ldexpf(1.0f, ea);
  inv_s = __builtin_amdgcn_ldexpf(1.0f, -ea);
}
template <int N> __device__ __forceinline__ void cph_settle8(f32x4c (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "n"(N));
}
template <int N> __device__ __forceinline__ void cph_settle8o(f32x4c (&a)[8], f32x4c& o0, f32x4c& o1, f32x4c& o2, f32x4c& o3) {
  asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3) : "n"(N));
}
template <int N> __device__ __forceinline__ void cph_settle4(f32x4c (&a)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N)); }
// the twelve matrix instructions of a step (4 out tiles x 3 partial products, term by term: an accumulator is touched every fourth instruction),
// images a[2 t] = h, a[2 t + 1] = l of out tile t; FIRST: the accumulators start at zero; LAST: wait states for the vector instructions that follow
template <bool FIRST, bool LAST> __device__ __forceinline__ void cph_step(f32x4c (&acc)[4], f32x4c (&a)[8], const cf16x8& bh, const cf16x8& bl) {
  if constexpr (FIRST) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %5, %12, 0\n\tv_mfma_f32_16x16x32_f16 %1, %7, %12, 0\n\tv_mfma_f32_16x16x32_f16 %2, %9, %12, 0\n\tv_mfma_f32_16x16x32_f16 %3, %11, %12, 0\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %4, %13, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %13, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %13, %2\n\tv_mfma_f32_16x16x32_f16 %3, %10, %13, %3\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %12, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, %10, %12, %3"
                 : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(bh), "v"(bl));
  } else {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %5, %12, %0\n\tv_mfma_f32_16x16x32_f16 %1, %7, %12, %1\n\tv_mfma_f32_16x16x32_f16 %2, %9, %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, %11, %12, %3\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %4, %13, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %13, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %13, %2\n\tv_mfma_f32_16x16x32_f16 %3, %10, %13, %3\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %12, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, %10, %12, %3"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(bh), "v"(bl));
  }
  if constexpr (LAST) asm volatile("s_nop 11" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}

// one dense layer: out[ot] = fma(W' in', d, bias) over (4 out tiles, 1 in pair) steps; NP in pairs, HT out tiles
template <int NP, int HT> __device__ __forceinline__ void cph_layer(CpRingH& ring, const float* bias_q, const CpBh& in, const CpAct& out, float d) {
  constexpr int STEPS = (HT / 4) * NP;
  f32x4c a[2][8];
  f32x4c acc[4], bs[4];
  cp_for<8>([&](auto i) CP_ALWAYS_INLINE { a[0][i] = ring.template read<decltype(i)::value>(); });
  const unsigned bias_addr = (unsigned)(size_t)((const __attribute__((address_space(3))) float*)bias_q);
  cp_for<STEPS>([&](auto st_) CP_ALWAYS_INLINE {
    constexpr int st = st_, otg = st / NP, ip = st % NP;
    constexpr bool first = ip == 0, last = ip == NP - 1;
    if constexpr (last) {  // the out-group's bias tiles: raw reads in front of the look-ahead request of the group's LAST step, whose counted wait settles them
      cp_for<4>([&](auto t) CP_ALWAYS_INLINE { bs[t] = cp_lds_raw<(otg * 4 + decltype(t)::value) * 64>(bias_addr); });
    }
    if constexpr (st + 1 < STEPS) {
      cp_for<8>([&](auto i) CP_ALWAYS_INLINE { a[(st + 1) & 1][i] = ring.template read<(st + 1) * 8 + decltype(i)::value>(); });
      if constexpr (last) cph_settle8o<8>(a[st & 1], bs[0], bs[1], bs[2], bs[3]);
      else cph_settle8<8>(a[st & 1]);
    } else {
      if constexpr (last) cph_settle8o<0>(a[st & 1], bs[0], bs[1], bs[2], bs[3]);
      else cph_settle8<0>(a[st & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    cph_step<first, last>(acc, a[st & 1], in.h(ip), in.l(ip));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (last) {
      cp_for<4>([&](auto t) CP_ALWAYS_INLINE {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[otg * 4 + t][r] = __builtin_fmaf(acc[t][r], d, bs[t][r]);
      });
    }
  });
}

// ReLU + per-sample scale + conversion of the HT out tiles into the next layer's B operands; returns 2^-ea
template <int HT> __device__ __forceinline__ float cph_convert(const CpAct& out, const CpBh& in) {
  float amax = 0.f;
  cp_for<HT>([&](auto t_) CP_ALWAYS_INLINE {
    constexpr int t = t_;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      out[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];  // NaN stays NaN, as torch.relu
      amax = fmaxf(amax, out[t][r]);
    }
  });
  float s, inv_s;
  cph_scale(amax, s, inv_s);
  cp_for<HT / 2>([&](auto p_) CP_ALWAYS_INLINE {
    constexpr int p = p_;
    cph_split(out[2 * p], out[2 * p + 1], s, in.h(p), in.l(p));
  });
  return inv_s;
}

template <int NIT, int HT> __global__ __launch_bounds__(256, 1) void coupling_kernel_half(CpArgs a) {
  static_assert(HT % 4 == 0 && HT <= CP_T && NIT <= CP_IT && NIT % 2 == 0 && (8 * (NIT / 2) * (HT / 4)) % CPH_CH == 0 && (8 * (HT / 2) * (HT / 4)) % CPH_CH == 0 && (2 * (HT / 2)) % CPH_CH == 0,
                "every layer and every group of the last layer fills whole chunks");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 15, q = lane >> 4;
  float* bias_lds = cp_lds + CP_NR * CPH_CH * 256;
  int* amap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);
  int* fmap_lds = amap_lds + CP_IT * 16;
  int* lay_lds = fmap_lds + a.NG * 8;
  float* xw = reinterpret_cast<float*>(lay_lds + 3 * CP_MAXL) + (size_t)wave * 16 * a.xs;
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < CP_MAXL; ++l) lay_lds[2 * CP_MAXL + l] = a.bias_off[l];
  }
  for (int i = tid; i < a.bias_floats; i += 256) bias_lds[i] = a.bias[i];
  for (int i = tid; i < NIT * 16; i += 256) amap_lds[i] = a.amap[i];
  for (int i = tid; i < a.NG * 8; i += 256) fmap_lds[i] = a.fmap[i];
  CpRingH ring;
  ring.lds = cp_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < CP_NR - 1; ++i) ring.issue();
  ring.slot = CP_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)cp_lds);
  ring.cur_off = ring.lds_off;
  __syncthreads();

  const int xs = a.xs;
  float* xrow = xw + jl * xs;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * 64 + wave * 16;
    const int64_t n = n0 + jl;
    const bool live = n < a.N;
    cp_stage_rows(a, xw, xs, n0, lane);

    f32x4c out_lo[16], out_hi[16];
    cf16x8 bhl[8], bhh[8], bll[8], blh[8];
    const CpAct out{out_lo, out_hi};
    const CpBh in{bhl, bhh, bll, blh};
    float inv_s;
    {  // first layer: B operands gathered from the row image through idx_a, scaled by the sample's own power of two, converted pair by pair
      f32x4c v[NIT];
      float amax = 0.f;
      cp_for<NIT>([&](auto t_) CP_ALWAYS_INLINE {
        constexpr int t = t_;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int src = amap_lds[t * 16 + 4 * q + r];
          v[t][r] = src >= 0 ? xrow[src] : (src <= -2 ? xrow[a.D + (-2 - src)] : 0.f);
          amax = fmaxf(amax, fabsf(v[t][r]));
        }
      });
      float s;
      cph_scale(amax, s, inv_s);
      cp_for<NIT / 2>([&](auto p_) CP_ALWAYS_INLINE { constexpr int p = p_; cph_split(v[2 * p], v[2 * p + 1], s, in.h(p), in.l(p)); });
    }
    cph_layer<NIT / 2, HT>(ring, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL]) + 4 * q, in, out, a.wdescale[0] * inv_s);
    inv_s = cph_convert<HT>(out, in);
    for (int l = 1; l < a.L - 1; ++l) {
      cph_layer<HT / 2, HT>(ring, bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + l]) + 4 * q, in, out, a.wdescale[l] * inv_s);
      inv_s = cph_convert<HT>(out, in);
    }
    // last layer + affine map: one group of 8 moved features = one out tile = HT / 2 blocks, two blocks per step on two accumulators
    const float* bias_last = bias_lds + __builtin_amdgcn_readfirstlane(lay_lds[2 * CP_MAXL + a.L - 1]) + 4 * q;
    const float dl = a.wdescale[a.L - 1] * inv_s;
    float lacc = 0.f;
    for (int g = 0; g < a.NG; ++g) {
      const int f0 = fmap_lds[g * 8 + 2 * q], f1 = fmap_lds[g * 8 + 2 * q + 1];
      const float x0 = xrow[f0 < 0 ? 0 : f0], x1 = xrow[f1 < 0 ? 0 : f1];
      const f32x4c bg = *reinterpret_cast<const f32x4c*>(bias_last + g * 16);
      f32x4c c0, c1;
      f32x4c w[2][4];  // images of the step's two blocks: w[.][2 b + part]
      constexpr int KS = HT / 4;  // steps per group
      cp_for<4>([&](auto i) CP_ALWAYS_INLINE { w[0][i] = ring.template read<decltype(i)::value>(); });
      cp_for<KS>([&](auto k_) CP_ALWAYS_INLINE {
        constexpr int k = k_;
        if constexpr (k + 1 < KS) {
          cp_for<4>([&](auto i) CP_ALWAYS_INLINE { w[(k + 1) & 1][i] = ring.template read<(k + 1) * 4 + decltype(i)::value>(); });
          cph_settle4<4>(w[k & 1]);
        } else {
          cph_settle4<0>(w[k & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k == 0) {
          asm volatile("s_nop 1\n\t"
                       "v_mfma_f32_16x16x32_f16 %0, %3, %6, 0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %8, 0\n\t"
                       "v_mfma_f32_16x16x32_f16 %0, %2, %7, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %9, %1\n\t"
                       "v_mfma_f32_16x16x32_f16 %0, %2, %6, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %8, %1"
                       : "=&v"(c0), "=&v"(c1)
                       : "v"(w[0][0]), "v"(w[0][1]), "v"(w[0][2]), "v"(w[0][3]), "v"(in.h(0)), "v"(in.l(0)), "v"(in.h(1)), "v"(in.l(1)));
        } else {
          asm volatile("s_nop 1\n\t"
                       "v_mfma_f32_16x16x32_f16 %0, %3, %6, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
                       "v_mfma_f32_16x16x32_f16 %0, %2, %7, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %9, %1\n\t"
                       "v_mfma_f32_16x16x32_f16 %0, %2, %6, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %8, %1"
                       : "+v"(c0), "+v"(c1)
                       : "v"(w[k & 1][0]), "v"(w[k & 1][1]), "v"(w[k & 1][2]), "v"(w[k & 1][3]), "v"(in.h(2 * k)), "v"(in.l(2 * k)), "v"(in.h(2 * k + 1)), "v"(in.l(2 * k + 1)));
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 11" : "+v"(c0), "+v"(c1));
      f32x4c p;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = __builtin_fmaf(c0[r] + c1[r], dl, bg[r]);  // (shift, scale) of slot 2 q, then of slot 2 q + 1
      float y0, y1, l0, l1;
      cp_affine(a, p[0], p[1], x0, y0, l0);
      cp_affine(a, p[2], p[3], x1, y1, l1);
      if (f0 >= 0) { xrow[f0] = y0; lacc += l0; }
      if (f1 >= 0) { xrow[f1] = y1; lacc += l1; }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    cp_store_rows(a, xw, xs, n0, lane);
    if (a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace zk

using namespace zk;

extern "C" {

// y[N, D] = merge(x_a, affine(x_b | MLP(cat(x_a, ctx)))), ladj[N] (+)= sum log|dy_b/dx_b| for one coupling transform.
// n_layers linear layers; tiles / widths: HOST arrays with the output tiles and widths of the n_layers - 1 hidden layers;
// bias_off: HOST array of n_layers offsets into the bias image; amap [nit * 16], fmap [n_groups * 8]: DEVICE index maps;
// wstream / bias: the plan of zuko_amd/coupling_plan.py.  Limits: D + C <= 1024 columns in the row image with
// 4 * 16 * (D + C + 4) * 4 bytes of LDS beside the 72 KiB ring, conditioner inputs <= 256, hidden widths <= 512.
static int cp_launch(int inverse, int64_t N, int D, int C, const void* x, int64_t ldx, const void* ctx, int64_t ldc, void* y, int64_t ldy, void* ladj, int accumulate,
                        const void* wstream, const void* bias, int bias_floats, const int32_t* bias_off, const int32_t* amap, int nit, const int32_t* fmap,
                        int n_groups, int n_layers, const int32_t* tiles, const int32_t* widths, int n_chunks, int act, double slope, int static_ok, void* stream,
                        const double* wdescale = nullptr) {
  if (N <= 0) return 0;
  if (n_layers < 2 || n_layers > CP_MAXL || nit < 1 || nit > CP_IT || n_groups < 1 || n_chunks < 1 || D < 2 || C < 0 || (C > 0 && !ctx)) return ZK_EINVAL;
  CpArgs a{};
  a.inverse = inverse;
  a.N = N; a.D = D; a.C = C; a.x = (const float*)x; a.ldx = ldx; a.ctx = (const float*)ctx; a.ldc = ldc; a.y = (float*)y; a.ldy = ldy;
  a.ladj = (float*)ladj; a.accumulate = accumulate; a.stream = (const float*)wstream; a.bias = (const float*)bias; a.amap = amap; a.fmap = fmap;
  a.L = n_layers; a.nit = nit; a.NG = n_groups; a.n_chunks = n_chunks; a.act = act; a.bias_floats = bias_floats;
  for (int l = 0; l < n_layers - 1; ++l) {
    if (tiles[l] < 1 || tiles[l] > CP_T) return ZK_EINVAL;
    a.wt[l] = tiles[l]; a.width[l] = widths[l];
  }
  for (int l = 0; l < n_layers; ++l) a.bias_off[l] = bias_off[l];
  a.xs = ((D + C + 3) / 4) * 4 + 4;
  a.vec4 = (D % 4 == 0) && (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  a.vec4_in = (D % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
              (C == 0 || ((C % 4 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(ctx) & 15) == 0)));
  a.ls = (float)log(slope);
  a.n_tiles = (N + 63) / 64;
  const int lds = (CP_NR * CP_CH * 256 + bias_floats + CP_IT * 16 + n_groups * 8 + 3 * CP_MAXL + CP_WAVES * 16 * a.xs) * (int)sizeof(float);
  if (lds > 160 * 1024) return ZK_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)coupling_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const unsigned grid = (unsigned)(a.n_tiles < 256 ? a.n_tiles : 256);
  // static-shape instantiation: ReLU, every hidden layer 512 wide (32 tiles), 128 conditioner inputs (8 tiles), groups in
  // whole triples padded by the plan (cfg4 of BASELINE.json: RealNVP(256, hidden [512] * 3)); `static_ok` is the plan's word
  // that its stream has the matching layout (every layer and every triple of groups starts on a chunk boundary)
  bool same = act == 1 && nit == 8;
  for (int l = 0; l < n_layers - 1; ++l) same = same && tiles[l] == 32 && widths[l] == 512;
  if (static_ok == 3) {  // the plan's TWO-PART stream (coupling_plan.py: half stream): f16 images, 16-image chunks, per-layer descale factors
    if (!same || n_layers > 4 || !wdescale) return ZK_EINVAL;
    for (int l = 0; l < n_layers; ++l) {
      if (!(wdescale[l] > 0.0) || !(wdescale[l] < 1e38)) return ZK_EINVAL;
      a.wdescale[l] = (float)wdescale[l];
    }
    static bool attr4 = false;
    if (!attr4) {
      hipError_t e = hipFuncSetAttribute((const void*)coupling_kernel_half<8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr4 = true;
    }
    hipLaunchKernelGGL((coupling_kernel_half<8, 32>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    return ZK_LAUNCH_CHECK();
  }
  if (same && static_ok == 2) {  // the plan's stream is the operand-split one (coupling_plan.py: split_gather): bf16 images, (4 out tiles, in pair) steps
    static bool attr3 = false;
    if (!attr3) {
      hipError_t e = hipFuncSetAttribute((const void*)coupling_kernel_split<8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr3 = true;
    }
    hipLaunchKernelGGL((coupling_kernel_split<8, 32>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    return ZK_LAUNCH_CHECK();
  }
  if (static_ok == 2) return ZK_EINVAL;  // a split stream cannot be read by the f32 kernels
  if (same && static_ok) {
    static bool attr2 = false;
    if (!attr2) {
      hipError_t e = hipFuncSetAttribute((const void*)coupling_kernel_static<8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr2 = true;
    }
    hipLaunchKernelGGL((coupling_kernel_static<8, 32>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    return ZK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(coupling_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  return ZK_LAUNCH_CHECK();
}


static int cp_launch_v1(int inverse, const zk_coupling_args_v1* p, void* stream) {
  if (!p || p->struct_size != sizeof(zk_coupling_args_v1) || p->version != 1) return ZK_EINVAL;
  return cp_launch(inverse, p->N, p->D, p->C, p->in, p->ldx, p->ctx, p->ldc, p->out, p->ldy, p->ladj, p->accumulate, p->wstream, p->bias, p->bias_floats, p->bias_off, p->amap,
                   p->nit, p->fmap, p->n_groups, p->n_layers, p->tiles, p->widths, p->n_chunks, p->act, p->slope, p->static_ok, stream, &p->wdescale0);
}

int zk_coupling_forward(const zk_coupling_args_v1* args, void* stream) { return cp_launch_v1(0, args, stream); }

// CouplingTransform._inverse (zuko/transforms.py:1050-1056): same launch with the affine map of the moved half inverted; ladj = the
// FORWARD map's log-determinant at the solution.
int zk_coupling_inverse(const zk_coupling_args_v1* args, void* stream) { return cp_launch_v1(1, args, stream); }

}  // extern "C"
