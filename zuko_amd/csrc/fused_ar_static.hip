// zuko_amd — static-shape twin of the fused autoregressive density kernel (fused_ar.hip) for the block pattern of
// MaskedAutoregressiveTransform(features = 64, hidden_features = [256] * 3) without context, i.e. the conditioner of
// BASELINE.json's cfg2 (NSF, 8-bin spline) and cfg3 (MAF, affine map):
//
//     y, log|dy/dx| = univariate(conditioner(x)).call_and_ladj(x)      (zuko/flows/autoregressive.py:207-218)
//
// With the hidden units sorted by dependency count (zuko_amd/fused.py) the masks of that conditioner
// (zuko/nn.py:270-295) are block lower triangular at the kernel's tile granularity, for EVERY feature order:
//
//     layer 1 (64 -> 256):   out-group o (4 tiles of 16 units) multiplies o + 1 of the 4 input tiles
//     layers 2, 3 (256^2):   out-group o multiplies input tiles 0 .. 4 (o + 1) - 1
//     last layer:            feature group g (4 lanes x FPL features) multiplies hidden tiles 0 .. FPL (g + 1) - 1
//
// so the (per-tile) weight stream of fused.py has a fixed length (48 chunks of 24 tiles for the spline, 17 for the affine map) and
// every tile's position in it is a compile-time constant.  The generic kernel finds that structure at run time (a
// wave-uniform bit test and branch per tile block, `s_waitcnt lgkmcnt(0)` at every join, ring position in a register);
// here the pass over a 16-sample wave tile is straight-line code: ring refills only where a position is a multiple of the
// chunk size, the A tiles of step s + 1 requested before the MFMAs of step s, the first tiles of the next feature group
// requested before the epilogue of the current one.  Same arithmetic in the same order per output as the generic kernel
// (asserted bit-identical in tests/test_gpu_flows.py).  The host (zuko_amd/fused.py) selects this kernel only after
// comparing the plan's skip words with zk_ar_static_skip(); anything else takes the generic kernel.
#include "zk_ar_common.h"
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <utility>

namespace zk {

#define ARS_CH 24
#define ARS_NR 3
#define ARS_ALWAYS_INLINE __attribute__((always_inline))

template <class F, int... I> __device__ __forceinline__ void ars_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void ars_for(F&& f) { ars_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- the block pattern -------------------------------------------------------------------------------------------
// Hidden layers: step s of a layer = (out-group otg of 4 tiles, j-th input tile of that group); layer 1 multiplies whole
// 4-tile blocks, layers 2 and 3 only the 16x16 tiles that hold non-zero weights: inside the diagonal 64x64 block of out-group o
// its k-th input tile (it = 4 o + k) reaches out tiles t >= k (the units are sorted by dependency count), except that zuko
// gives the first degrees five units each, which makes tile (0, 1) non-empty as well (zuko_amd/fused.py: ArPlan.fine_tilemask
// is compared with zk_ar_static_tiles() before this kernel is selected).
__host__ __device__ constexpr int ars_hid_steps(int l) { return l == 0 ? 10 : 40; }
__host__ __device__ constexpr int ars_hid_first(int l, int otg) { return l == 0 ? otg * (otg + 1) / 2 : 2 * otg * (otg + 1); }
__host__ __device__ constexpr int ars_hid_otg(int l, int s) { return s < ars_hid_first(l, 1) ? 0 : s < ars_hid_first(l, 2) ? 1 : s < ars_hid_first(l, 3) ? 2 : 3; }
__host__ __device__ constexpr unsigned ars_tmask(int l, int otg, int j) {  // bit t: out tile 4 otg + t is multiplied in step (otg, j)
  if (l == 0) return 0xFu;
  const int k = j - 4 * otg;
  return k < 0 ? 0xFu : (otg == 0 && k == 1) ? 0xFu : (0xFu << k) & 0xFu;
}
__host__ __device__ constexpr int ars_popc(unsigned m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u) + ((m >> 3) & 1u)); }
__host__ __device__ constexpr unsigned ars_step_mask(int l, int s) { return ars_tmask(l, ars_hid_otg(l, s), s - ars_hid_first(l, ars_hid_otg(l, s))); }
__host__ __device__ constexpr int ars_hid_pos(int l, int s) {  // tiles of layer l streamed before step s
  int n = 0;
  for (int i = 0; i < s; ++i) n += ars_popc(ars_step_mask(l, i));
  return n;
}
__host__ __device__ constexpr int ars_hid_base(int l) { return l == 0 ? 0 : l == 1 ? 48 : 192; }  // 40 -> 48 and 137 -> 144 tiles: layers are padded to whole chunks
__host__ __device__ constexpr int ars_last_base() { return 336; }
__host__ __device__ constexpr int ars_last_first(int fpl, int g) { return fpl * g * (g + 1) / 2; }  // steps before group g
static_assert(ars_hid_pos(0, 10) == 40 && ars_hid_pos(1, 40) == 137 && ars_hid_pos(2, 40) == 137, "tiles per hidden layer");

struct ArRingS {
  float* lds;
  const float* stream;
  unsigned cur_off;  // LDS byte address of the slot being read + lane * 16
  unsigned lds_off;  // LDS byte address of the ring
  int n_chunks, slot, load_chunk, load_slot, wave, lane;
  template <int I> __device__ __forceinline__ void dma(const float* g, float* l) {
    if constexpr (I < ARS_CH / AR_WAVES) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, I * AR_TF * 4, 0);
      dma<I + 1>(g, l);
    }
  }
  __device__ __forceinline__ void issue() {  // three consecutive tiles per wave: one address, one M0 value, immediate offsets
    const int b0 = wave * (ARS_CH / AR_WAVES);
    dma<0>(stream + ((size_t)load_chunk * ARS_CH + b0) * AR_TF + lane * 4, lds + (load_slot * ARS_CH + b0) * AR_TF);
    load_chunk = (load_chunk + 1 == n_chunks) ? 0 : load_chunk + 1;
    load_slot = (load_slot + 1 == ARS_NR) ? 0 : load_slot + 1;
  }
  __device__ __forceinline__ void advance() {  // all 8 waves, at the same (static) points of the pass
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((ARS_NR - 2) * (ARS_CH / AR_WAVES)) : "memory");
    __builtin_amdgcn_s_barrier();  // (not __syncthreads(): its fence is s_waitcnt vmcnt(0) and would drain the look-ahead DMAs)
    asm volatile("" ::: "memory");
    issue();
    slot = (slot + 1 == ARS_NR) ? 0 : slot + 1;
    cur_off = lds_off + (unsigned)(slot * ARS_CH * AR_TF * 4 + lane * 16);
  }
  // Position S inside the pass (static).  The read is issued from inline assembly and returns a RAW value: the compiler does
  // not know it is an LDS operation, so it inserts no wait for it — while a global_load_lds is in flight hipcc turns every
  // LDS wait into lgkmcnt(0), which would make the step wait for the tiles it has just requested for the NEXT step.  The
  // value becomes usable through ars_settle<N>() below, which waits until at most N younger LDS operations are outstanding
  // (LDS operations of a wave complete in order) and is the only consumer of the raw registers.
  template <int S> __device__ __forceinline__ f32x4 read() {
    if constexpr (S % ARS_CH == 0) advance();
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(cur_off), "n"((S % ARS_CH) * AR_TF * 4));
    return v;
  }
};

template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N));
}
template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a0) : "n"(N)); }
template <int N> __device__ __forceinline__ void ars_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "n"(N));
}
template <int N, int NT> __device__ __forceinline__ void ars_settle_tiles(f32x4 (&w)[NT]) {
  if constexpr (NT == 1) ars_settle<N>(w[0]);
  else if constexpr (NT == 6) ars_settle<N>(w[0], w[1], w[2], w[3], w[4], w[5]);
  else static_assert(NT == 1 || NT == 6, "last-layer tile count");
}

extern __shared__ __attribute__((aligned(16))) float ars_lds[];

template <int L> __device__ __forceinline__ void ars_hidden(ArRingS& ring, const float* bias_q, const f32x4 (&in)[AR_T], f32x4 (&out)[AR_T], bool rev) {
  constexpr int NS = ars_hid_steps(L), BASE = ars_hid_base(L);
  f32x4 a[2][4];
  ars_for<4>([&](auto t) ARS_ALWAYS_INLINE { a[0][t] = ring.template read<BASE + decltype(t)::value>(); });  // (step 0 multiplies all four tiles)
  static_assert(ars_step_mask(L, 0) == 0xFu, "first step");
  ars_for<NS>([&](auto s_) ARS_ALWAYS_INLINE {
    constexpr int s = s_, otg = ars_hid_otg(L, s), j = s - ars_hid_first(L, otg);
    constexpr unsigned M = ars_step_mask(L, s);
    if constexpr (j == 0) {
      ars_for<4>([&](auto t) ARS_ALWAYS_INLINE { out[otg * 4 + t] = *reinterpret_cast<const f32x4*>(bias_q + (otg * 4 + t) * 16); });  // accumulators start at the bias
    }
    if constexpr (s + 1 < NS) {
      constexpr unsigned MN = ars_step_mask(L, s + 1);
      constexpr int PN = BASE + ars_hid_pos(L, s + 1);
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
    if constexpr (L == 0) {
      // the first layer's columns are in natural feature order: an ascending feature order makes out-group o depend on input
      // tiles 0 .. o, a descending one on tiles 3 - o .. 3 (same count, same stream positions)
      const f32x4 up = in[j], down = in[3 - otg + j];
      b = rev ? down : up;
    } else {
      b = in[j];
    }
    ars_for<4>([&](auto r) ARS_ALWAYS_INLINE {
      ars_for<4>([&](auto t) ARS_ALWAYS_INLINE {
        if constexpr ((M >> decltype(t)::value) & 1u) out[otg * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][t][(int)r], b[(int)r], out[otg * 4 + t], 0, 0, 0);
      });
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// TRAIN: conditioner only — the three hidden activations and phi are stored for the backward pass (zuko_amd/train.py), the
// univariate map is not evaluated (the autograd graph applies it to phi itself).
template <typename Uni, bool TRAIN = false> __global__ __launch_bounds__(512, 2) void ar_static_kernel(ArArgs a) {
  constexpr int NT = Uni::NT, FPL = Uni::FPL, TOTAL = Uni::TOTAL;
  constexpr int NG = 64 / (4 * FPL);                         // feature groups
  constexpr int NSTEP = ars_last_first(FPL, NG);             // steps of the last layer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const bool rev = a.l1rev != 0;

  ArRingS ring;
  float* bias_lds = ars_lds + ARS_NR * ARS_CH * AR_TF;
  ring.lds = ars_lds; ring.stream = a.stream; ring.n_chunks = a.n_chunks; ring.wave = wave; ring.lane = lane;
  ring.load_chunk = 0; ring.load_slot = 0;
#pragma unroll
  for (int i = 0; i < ARS_NR - 1; ++i) ring.issue();
  ring.slot = ARS_NR - 1;
  ring.lds_off = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ars_lds);
  ring.cur_off = ring.lds_off;

  for (int i = tid; i < a.bias_floats; i += 512) bias_lds[i] = a.bias[i];
  int* fmap_lds = reinterpret_cast<int*>(bias_lds + a.bias_floats);  // same LDS layout as the generic kernel (fused_ar.hip)
  float* xr = reinterpret_cast<float*>(fmap_lds + 1024 + 256) + wave * 16 * a.xs + j * a.xs;
  for (int i = tid; i < NG * 4 * FPL; i += 512) fmap_lds[i] = a.featmap[i];
  __syncthreads();
  const float* bias_last = bias_lds + 3 * 256;
  // feature ids of this lane's slots in every group: constant over the launch, kept in registers (a per-group LDS read would put
  // one exposed LDS round trip in front of the read of x that depends on it)
  int fids[NG * FPL];
#pragma unroll
  for (int i = 0; i < NG; ++i)
#pragma unroll
    for (int fi = 0; fi < FPL; ++fi) fids[i * FPL + fi] = fmap_lds[(i * 4 + q) * FPL + fi];

  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n = tile * 128 + wave * 16 + j;
    const bool live = n < a.N;
    const int64_t nc = live ? n : a.N - 1;
    const float* xrow = a.x + nc * a.ldx;

    f32x4 in[AR_T], out[AR_T];
#pragma unroll
    for (int it = 0; it < 4; ++it) in[it] = *reinterpret_cast<const f32x4*>(xrow + it * 16 + 4 * q);
    // a NaN / inf input turns ALL parameters of its sample into NaN in the reference (x * 0 = NaN, zuko/nn.py:217-218)
    float poison = 0.f;
    {
      int bad = 0;
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= !(fabsf(in[it][r]) < __builtin_inff());
      bad |= __shfl_xor(bad, 16, 64);
      bad |= __shfl_xor(bad, 32, 64);
      if (bad) poison = __builtin_nanf("");
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(xr + it * 16 + 4 * q) = in[it];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- hidden layers ---------------------------------------------------------------------------------------------
    ars_hidden<0>(ring, bias_lds + 0 * 256 + 4 * q, in, out, rev);
#pragma unroll
    for (int t = 0; t < AR_T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) in[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];  // NaN stays NaN, as torch.relu
    if constexpr (TRAIN) {
      if (live) {
#pragma unroll
        for (int t = 0; t < AR_T; ++t) *reinterpret_cast<f32x4*>(a.act_out[0] + n * 256 + t * 16 + 4 * q) = in[t];
      }
    }
    ars_hidden<1>(ring, bias_lds + 1 * 256 + 4 * q, in, out, rev);
#pragma unroll
    for (int t = 0; t < AR_T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) in[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];
    if constexpr (TRAIN) {
      if (live) {
#pragma unroll
        for (int t = 0; t < AR_T; ++t) *reinterpret_cast<f32x4*>(a.act_out[1] + n * 256 + t * 16 + 4 * q) = in[t];
      }
    }
    ars_hidden<2>(ring, bias_lds + 2 * 256 + 4 * q, in, out, rev);
#pragma unroll
    for (int t = 0; t < AR_T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) in[t][r] = out[t][r] < 0.f ? 0.f : out[t][r];
    if constexpr (TRAIN) {
      if (live) {
#pragma unroll
        for (int t = 0; t < AR_T; ++t) *reinterpret_cast<f32x4*>(a.act_out[2] + n * 256 + t * 16 + 4 * q) = in[t];
      }
    }

    // ---- last layer + univariate transform, one group of 4 * FPL features at a time --------------------------------
    float lacc = 0.f;
    f32x4 w[2][NT];
    ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { w[0][t] = ring.template read<ars_last_base() + decltype(t)::value>(); });
    ars_for<NG>([&](auto g_) ARS_ALWAYS_INLINE {
      constexpr int g = g_, ST0 = ars_last_first(FPL, g);
      // operands of the epilogue are requested before the group's MFMAs: feature ids, x values and the bias from LDS
      int fid[FPL];
      float xin[FPL];
#pragma unroll
      for (int fi = 0; fi < FPL; ++fi) {
        fid[fi] = fids[g * FPL + fi];
        xin[fi] = xr[fid[fi] < 0 ? 0 : fid[fi]];
      }
      f32x4 acc[NT];  // the accumulators start at the bias
      {
        const float* bg = bias_last + (g * NT) * 16 + 4 * q;
        ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { acc[t] = *reinterpret_cast<const f32x4*>(bg + t * 16); });
      }
      ars_for<FPL*(g + 1)>([&](auto it_) ARS_ALWAYS_INLINE {
        constexpr int it = it_, st = ST0 + it;
        if constexpr (st + 1 < NSTEP) {
          ars_for<NT>([&](auto t) ARS_ALWAYS_INLINE { w[(st + 1) & 1][t] = ring.template read<ars_last_base() + (st + 1) * NT + decltype(t)::value>(); });
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
            xr[f] = yv;
            lacc += lj;
          }
        }
      }
    });
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (!TRAIN && live) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.y + n * a.ldy + it * 16 + 4 * q) = *reinterpret_cast<const f32x4*>(xr + it * 16 + 4 * q);
    }
    if (!TRAIN && a.ladj) {
      lacc += __shfl_xor(lacc, 16, 64);
      lacc += __shfl_xor(lacc, 32, 64);
      if (live && q == 0) a.ladj[n] = a.accumulate ? a.ladj[n] + lacc : lacc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must land before the LDS is released
}

// Launch of the static kernel (called by ar_launch in fused_ar.hip after the generic argument checks).  variant: 1 = first-layer
// pattern of an ascending feature order, 2 = of a descending one.
int ar_static_launch(ArArgs& a, int uni_kind, int variant, int lds, unsigned grid, hipStream_t stream) {
  if (a.D != 64 || a.DIN != 64 || a.L != 4 || a.act != 1 || !a.xlds || a.sched || (variant != 1 && variant != 2)) return ZK_EINVAL;
  const bool train = a.phi_out != nullptr;
  const void* fn = nullptr;
  if (uni_kind == 1 && a.NG == 16 && a.n_chunks == 48) fn = train ? (const void*)ar_static_kernel<UniRqs8, true> : (const void*)ar_static_kernel<UniRqs8>;
  else if (uni_kind == 0 && a.NG == 8 && a.n_chunks == 17) fn = train ? (const void*)ar_static_kernel<UniAffine, true> : (const void*)ar_static_kernel<UniAffine>;
  else return ZK_EINVAL;
  a.l1rev = variant == 2;
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
  void* kargs[] = {&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, lds, stream);
  if (e != hipSuccess) return (int)e;
  return ZK_LAUNCH_CHECK();
}

}  // namespace zk

extern "C" {

// The skip words (zuko_amd/fused.py: ArPlan.skip, 4 per hidden layer + one per feature group) a plan must have for
// zk_ar_forward(..., variant = 1 | 2, ...) to be valid.  Returns the number of words written, or 0 if (uni_kind, variant)
// has no static kernel.  out must hold 28 words.
int zk_ar_static_skip(int uni_kind, int variant, uint32_t* out) {
  if ((uni_kind != 0 && uni_kind != 1) || (variant != 1 && variant != 2)) return 0;
  const int fpl = uni_kind == 0 ? 2 : 1, ng = 64 / (4 * fpl);
  int n = 0;
  for (int o = 0; o < 4; ++o) out[n++] = variant == 1 ? (1u << (o + 1)) - 1u : (0xfu << (3 - o)) & 0xfu;
  for (int l = 1; l < 3; ++l)
    for (int o = 0; o < 4; ++o) out[n++] = (1u << (4 * (o + 1))) - 1u;
  for (int g = 0; g < ng; ++g) out[n++] = (1u << (fpl * (g + 1))) - 1u;
  return n;
}

// Conditioner-only (training) launch of the static kernel: phi [N, D * total] = net(x) in module order plus the three hidden
// activations h1, h2, h3 [N, 256] in the stream's sorted unit order, for the backward pass of zuko_amd/train.py.  Same stream,
// bias image and feature map as zk_ar_forward(variant = 1 | 2); x rows 16-byte addressable, ldx % 4 == 0.
int zk_ar_forward_train(int uni_kind, int64_t N, const void* x, int64_t ldx, void* h1, void* h2, void* h3, void* phi, int64_t ldphi, const void* wstream,
                        const void* bias, int bias_floats, const int32_t* featmap, int n_chunks, int variant, void* stream) {
  if (N <= 0) return 0;
  if ((uni_kind != 0 && uni_kind != 1) || !h1 || !h2 || !h3 || !phi || ldx % 4 || ((uintptr_t)x % 16) || ((uintptr_t)h1 % 16) || ((uintptr_t)h2 % 16) || ((uintptr_t)h3 % 16)) return ZK_EINVAL;
  zk::ArArgs a{};
  a.N = N; a.D = 64; a.DIN = 64; a.x = (const float*)x; a.ldx = ldx;
  a.stream = (const float*)wstream; a.bias = (const float*)bias; a.featmap = featmap;
  a.L = 4; a.NG = uni_kind == 1 ? 16 : 8; a.n_chunks = n_chunks; a.act = 1; a.bias_floats = bias_floats;
  a.n_tiles = (N + 127) / 128;
  a.xs = 68; a.xlds = 1;
  a.act_out[0] = (float*)h1; a.act_out[1] = (float*)h2; a.act_out[2] = (float*)h3; a.phi_out = (float*)phi; a.ldphi = ldphi;
  const int lds = (ARS_NR * ARS_CH * AR_TF + bias_floats + 1024 + 256 + 8 * 16 * a.xs) * (int)sizeof(float);  // ring | bias | feature map | (skip words) | row tiles
  if (lds > 160 * 1024) return ZK_EINVAL;
  const unsigned grid = (unsigned)(a.n_tiles < 256 ? a.n_tiles : 256);
  return zk::ar_static_launch(a, uni_kind, variant, lds, grid, (hipStream_t)stream);
}

// The per-tile pattern of the stream the static kernel consumes (ArPlan.fine_tilemask of zuko_amd/fused.py): out[(l * 4 + otg) * 16 + it]
// = 4-bit mask of the out tiles of group otg that input tile it of hidden layer l is multiplied into (0: block skipped).
// Returns 192 (3 layers x 4 out-groups x 16 input tiles), or 0 if (uni_kind, variant) has no static kernel.
int zk_ar_static_tiles(int uni_kind, int variant, uint8_t* out) {
  if ((uni_kind != 0 && uni_kind != 1) || (variant != 1 && variant != 2)) return 0;
  for (int i = 0; i < 192; ++i) out[i] = 0;
  for (int o = 0; o < 4; ++o)
    for (int j = 0; j <= o; ++j) out[(0 * 4 + o) * 16 + (variant == 1 ? j : 3 - o + j)] = 0xF;
  for (int l = 1; l < 3; ++l)
    for (int o = 0; o < 4; ++o)
      for (int it = 0; it < 4 * (o + 1); ++it) out[(l * 4 + o) * 16 + it] = (uint8_t)zk::ars_tmask(l, o, it);
  return 192;
}

}  // extern "C"
