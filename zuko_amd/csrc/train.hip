// zuko_amd — conditioner GEMMs of the TRAINING path: forward, dgrad and wgrad of the masked layers
// F.linear(x, mask * W, b) (zuko/nn.py:217-218) in fp32 on v_mfma_f32_32x32x2_f32, mask-aware by TILE SKIPPING.
//
// The reference differentiates `x @ (mask * W).T + b` with autograd: two dense GEMMs per layer that multiply the
// masked-out zeros (dgrad) or compute-and-discard them (wgrad).  Here the host (zuko_amd/train.py) works in a
// reparametrised network whose hidden units are sorted by MADE degree — rows of W_l / b_l and columns of W_{l+1}
// permuted together, an exact reparametrisation — where every mask is block lower-triangular.  Then
//   * forward  Y = act(X Ws^T + bs)            skips the (128-row out block, 32-wide k tile) pairs the mask zeroes;
//   * dgrad    Gin = (Gout Ws) . act'(H)        is the same kernel on Ws^T with ITS skip map and the activation
//                                               derivative applied in the epilogue (no separate pass over [N, width]);
//   * wgrad    dWs = Gout^T H                   is a split-K GEMM over sample slices that only visits the (128 x 128)
//                                               output blocks the mask keeps; partials are reduced in a fixed order
//                                               (deterministic) with the mask applied, bias gradients by column sums.
// Operand layout notes are next to each kernel.  fp32 MFMA is bitwise an fmaf chain: results differ from any other fp32
// GEMM by summation order only.
#include "zk_common.h"
#include "zk_half.h"
#include <stdlib.h>

#include "../../include/zuko_amd.h"

namespace zk {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t4 __attribute__((ext_vector_type(4)));

// derivative of the activation written in terms of its OUTPUT v (relu, elu, tanh, sigmoid, leaky relu; 0 = identity)
__device__ __forceinline__ float act_grad_out(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? 1.f : 0.f;
    case 2: return v > 0.f ? 1.f : v + 1.f;
    case 3: return 1.f - v * v;
    case 6: return v * (1.f - v);
    case 7: return v > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}
__device__ __forceinline__ float act_fwd(float v, int act) {
  switch (act) {
    case 1: return v < 0.f ? 0.f : v;
    case 2: return v > 0.f ? v : expm1f(v);
    case 3: return tanhf(v);
    case 6: return 1.f / (1.f + expf(-v));
    case 7: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}

struct GemmArgs {
  int64_t N;
  int IN, OUT;
  const float* x; int64_t ldx;   // [N, IN]
  const float* w;                // [OUT, IN] row-major, ALREADY masked (zeros where the mask is false)
  const uint64_t* kskip;         // [ceil(OUT / 128)]: bit kt set = k tile kt (32 inputs) of that out block has non-zero weights; null = all (IN <= 2048 with a map)
  const float* bias;             // [OUT] or null
  int act;                       // activation applied to x w^T + b (forward)
  const float* gate; int64_t ldg;  // optional [N, OUT]: the result is multiplied by act'(gate) (dgrad: gate = the saved activation output)
  int gate_act;
  float* y; int64_t ldy;
  int nbx, nby;
};

#define TBM 128
#define TBN 128
#define TBK 32
#define TPAD 4

// 128 x 128 x 32 LDS tile, 4 waves (2 x 2) x (2 x 2) v_mfma_f32_32x32x2_f32, transposed product (A operand = weight
// rows) so a lane owns four consecutive outputs of one sample; register-staged prefetch of the next live k tile.
template <bool VEC> __global__ __launch_bounds__(256) void gemm_f32_skip(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float As[TBM][TBK + TPAD];
  __shared__ __attribute__((aligned(16))) float Bs[TBN][TBK + TPAD];
  // blocks that share a row panel get consecutive logical ids, one contiguous range per XCD (block b runs on XCD b % 8)
  int bx, by;
  {
    const int nwg = a.nbx * a.nby, orig = blockIdx.x;
    const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    bx = logical / a.nby;
    by = logical % a.nby;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t row0 = (int64_t)bx * TBM;
  const int col0 = by * TBN;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (a.IN + TBK - 1) / TBK;
  const uint64_t lv = a.kskip ? a.kskip[by] : ~0ull;
  const uint64_t live = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(lv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lv);

  float4 ga[4], gb[4];
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // 32 rows x 8 float4 per pass
  constexpr bool vec = VEC;  // IN % 4 == 0, ldx % 4 == 0, x / w 16-byte aligned (checked by the launcher)
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 32 * p;
      const int64_t gr = row0 + r;
      const int k = k0 + lc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < a.N && k < a.IN) {
        const float* src = a.x + gr * a.ldx + k;
        if (vec) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; if (k + 1 < a.IN) v.y = src[1]; if (k + 2 < a.IN) v.z = src[2]; if (k + 3 < a.IN) v.w = src[3]; }
      }
      ga[p] = v;
      const int gc = col0 + r;
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gc < a.OUT && k < a.IN) {
        const float* src = a.w + (int64_t)gc * a.IN + k;
        if (vec) u = *reinterpret_cast<const float4*>(src);
        else { u.x = src[0]; if (k + 1 < a.IN) u.y = src[1]; if (k + 2 < a.IN) u.z = src[2]; if (k + 3 < a.IN) u.w = src[3]; }
      }
      gb[p] = u;
    }
  };

  // next live k tile >= k (the map covers the first 64 tiles; beyond that — only without a map — every tile is live)
  auto next_live = [&](int k) -> int {
    if (k >= nk) return -1;
    if (k >= 64) return k;
    const uint64_t m = live >> k;
    if (m) { const int t = k + (int)__builtin_ctzll(m); return t < nk ? t : -1; }
    return nk > 64 ? 64 : -1;
  };
  int kt = next_live(0);
  if (kt >= 0) gload(kt * TBK);
  while (kt >= 0) {
    const int ktn = next_live(kt + 1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<float4*>(&As[lr + 32 * p][lc]) = ga[p];
      *reinterpret_cast<float4*>(&Bs[lr + 32 * p][lc]) = gb[p];
    }
    __syncthreads();
    if (ktn >= 0) gload(ktn * TBK);
    const int fi = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < TBK / 8; ++kk) {
      f32x4_t4 af[2], bf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const f32x4_t4*>(&As[wr * 64 + m * 32 + fi][kk * 8 + 4 * fh]);
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const f32x4_t4*>(&Bs[wc * 64 + n * 32 + fi][kk * 8 + 4 * fh]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[n][r], af[m][r], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
    kt = ktn;
  }

  // epilogue: acc[m][n][r] = Y[sample m*32 + (lane&31)][output n*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
  const bool vec_y = (a.ldy % 4 == 0) && ((((uintptr_t)a.y) & 15) == 0);
  const bool vec_g = a.gate && (a.ldg % 4 == 0) && ((((uintptr_t)a.gate) & 15) == 0);
  const bool relu = a.act == 1, relu_gate = a.gate_act == 1;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = col0 + wc * 64 + n * 32 + 8 * q + 4 * (lane >> 5);
      float bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t] = (a.bias && col + t < a.OUT) ? a.bias[col + t] : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int64_t row = row0 + wr * 64 + m * 32 + (lane & 31);
        if (row >= a.N) continue;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v[t] = acc[m][n][4 * q + t] + bv[t];
          if (relu) v[t] = v[t] < 0.f ? 0.f : v[t];  // NaN stays NaN, as torch.relu
        }
        if (a.gate) {
          float gv[4] = {0.f, 0.f, 0.f, 0.f};
          const float* gp = a.gate + row * a.ldg + col;
          if (vec_g && col + 4 <= a.OUT) { const float4 t4 = *reinterpret_cast<const float4*>(gp); gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w; }
          else {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (col + t < a.OUT) gv[t] = gp[t];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] *= relu_gate ? (gv[t] > 0.f ? 1.f : 0.f) : act_grad_out(gv[t], a.gate_act);
        }
        float* dst = a.y + row * a.ldy + col;
        if (vec_y && col + 4 <= a.OUT) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
          for (int t = 0; t < 4; ++t) if (col + t < a.OUT) dst[t] = v[t];
        }
      }
    }
  }
  // (other activations: a ROLLED loop over the values this thread just stored — their inline expansions, unrolled 64
  //  times, make the epilogue instruction-cache bound; forward only, a gate is never combined with them)
  if (a.act > 1) {
#pragma unroll 1
    for (int e = 0; e < 64; ++e) {
      const int n = e >> 5, m = (e >> 4) & 1, r = e & 15;
      const int col = col0 + wc * 64 + n * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
      const int64_t row = row0 + wr * 64 + m * 32 + (lane & 31);
      if (col < a.OUT && row < a.N) {
        float* p = a.y + row * a.ldy + col;
        *p = act_fwd(*p, a.act);
      }
    }
  }
}

// ---- wgrad: dW[OUT, IN] = G^T H over the samples, split-K ------------------------------------------------------------
// Block (pair p, slice s) accumulates the 128 x 128 output block pairs[p] = (ob, ib) over rows [s S, (s+1) S) of
// G[N, OUT] and H[N, IN].  v_mfma_f32_32x32x2_f32 contracts two SAMPLES per instruction: its A operand wants
// A[i = lane % 32][k = lane / 32] = G[n0 + k][o0 + i] and B[k][j = lane % 32] = H[n0 + k][i0 + j], i.e. 32 consecutive
// floats of one row per half-wave — both operands load straight from global memory in MFMA layout, coalesced, with
// no LDS staging; the four waves of a block (2 x 2, 64 x 64 each) share the rows through L1.
struct WgradArgs {
  int64_t N;
  int OUT, IN;
  const float* g; int64_t ldg;
  const float* h; int64_t ldh;
  const int32_t* pairs;   // [npairs][2] = (out block, in block)
  int npairs;
  int64_t S;              // rows per slice (even)
  int nslices;
  float* partial;         // [nslices][npairs][128 x 128]
  // optional bias gradient (column sums of g) from the same pass over g: blocks with cs_flag[p] != 0 (one per out block) also
  // add up the 128 columns of their G tiles; cs_partial [nslices][cs_ld]
  const uint8_t* cs_flag;
  float* cs_partial;
  int cs_ld;
  // optional (operand-split kernel): the maxima of |g| and |h| on the device (zk_half.h) — both given = TWO-part f16 operands with per-tensor
  // power-of-two scales (three partial products) instead of three-part bf16 (six)
  const unsigned* g_amax;
  const unsigned* h_amax;
};

__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradArgs a) {
  // k tile = 32 samples: G[32 x 128] and H[32 x 128] are staged through LDS in their natural [sample][unit] layout
  // (register-staged float4 loads, next tile in flight during the MFMAs); the MFMA operands are then plain
  // conflict-free ds_read_b32: A[i][k] = Gs[k][o], B[k][j] = Hs[k][c] with 32 consecutive units per half-wave.
  __shared__ __attribute__((aligned(16))) float Gs[32][128 + 4];
  __shared__ __attribute__((aligned(16))) float Hs[32][128 + 4];
  // XCD-aware walk: workgroups go round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  XCD x takes a CONTIGUOUS range of the
  // (slice, pair) work list, so the blocks that share a G tile (the in blocks of one out block) and an H tile (all out blocks of a slice) run
  // on the same L2 at about the same time: G[N, OUT] and H[N, IN] then come from HBM about once instead of once per sharing block.
  const int total = a.nslices * a.npairs, per_xcd = (total + 7) / 8;
  const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (work >= total || (int)(blockIdx.x >> 3) >= per_xcd) return;
  const int p = work % a.npairs, s = work / a.npairs;
  const int ob = a.pairs[2 * p], ib = a.pairs[2 * p + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lk = lane >> 5;
  const int64_t n_begin = (int64_t)s * a.S;
  int64_t n_end = n_begin + a.S;
  n_end = n_end < a.N ? n_end : a.N;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // loader: thread -> (row = tid / 32 + 8 pass, 4 consecutive units at (tid % 32) * 4)
  const int lrow = tid >> 5, lcol = (tid & 31) * 4;
  const int go = ob * 128 + lcol, hc = ib * 128 + lcol;
  const bool vg = (a.ldg % 4 == 0) && ((((uintptr_t)a.g) & 15) == 0) && (go + 4 <= a.OUT);
  const bool vh = (a.ldh % 4 == 0) && ((((uintptr_t)a.h) & 15) == 0) && (hc + 4 <= a.IN);
  float4 rg[4], rh[4];
  // bias gradient: in the designated blocks thread t < 128 adds up column t of every G tile out of LDS (one register; a float4 share per
  // loader thread would take the kernel past 128 VGPRs, i.e. from four to three blocks per CU)
  const bool do_cs = a.cs_flag && a.cs_flag[p] && tid < 128;
  float csum = 0.f;
  auto gload = [&](int64_t n0) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int64_t n = n0 + lrow + 8 * ps;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), u = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < n_end) {
        const float* sg = a.g + n * a.ldg + go;
        if (vg) v = *reinterpret_cast<const float4*>(sg);
        else { if (go < a.OUT) v.x = sg[0]; if (go + 1 < a.OUT) v.y = sg[1]; if (go + 2 < a.OUT) v.z = sg[2]; if (go + 3 < a.OUT) v.w = sg[3]; }
        const float* sh = a.h + n * a.ldh + hc;
        if (vh) u = *reinterpret_cast<const float4*>(sh);
        else { if (hc < a.IN) u.x = sh[0]; if (hc + 1 < a.IN) u.y = sh[1]; if (hc + 2 < a.IN) u.z = sh[2]; if (hc + 3 < a.IN) u.w = sh[3]; }
      }
      rg[ps] = v; rh[ps] = u;
    }
  };
  gload(n_begin);
  for (int64_t n0 = n_begin; n0 < n_end; n0 += 32) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      *reinterpret_cast<float4*>(&Gs[lrow + 8 * ps][lcol]) = rg[ps];
      *reinterpret_cast<float4*>(&Hs[lrow + 8 * ps][lcol]) = rh[ps];
    }
    __syncthreads();
    if (n0 + 32 < n_end) gload(n0 + 32);
    if (do_cs) {
#pragma unroll 8
      for (int k = 0; k < 32; ++k) csum += Gs[k][tid];
    }
#pragma unroll 4
    for (int kp = 0; kp < 16; ++kp) {
      float av[2], bv[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) av[m] = Gs[2 * kp + lk][wr * 64 + m * 32 + li];
#pragma unroll
      for (int n = 0; n < 2; ++n) bv[n] = Hs[2 * kp + lk][wc * 64 + n * 32 + li];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_cs) a.cs_partial[(size_t)s * a.cs_ld + ob * 128 + tid] = csum;
  // acc[m][n][r] = D[i = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)][j = lane & 31] of the (m, n) 32 x 32 sub-block
  float* dst = a.partial + ((size_t)s * a.npairs + p) * (128 * 128);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = wr * 64 + m * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        const int j = wc * 64 + n * 32 + (lane & 31);
        dst[i * 128 + j] = acc[m][n][r];
      }
}

// The same block on the bf16 matrix instruction with three-way split operands (csrc/fused_ar_split_impl.h: every f32 value as h + m + l
// in bf16, six partial products, f32 accumulation — the f32 instruction above runs at 1/16 of the bf16 rate).  The contraction runs over
// SAMPLES, which are the strided direction of G[N, OUT] and H[N, IN]: each loader thread therefore gathers 8 consecutive samples of ONE
// unit with 8 dword loads (consecutive threads = consecutive units: every load instruction is coalesced), splits them in registers and
// writes the three 16-byte bf16 vectors straight into the operand images of v_mfma_f32_16x16x32_bf16 (lane = 16 * sample octet + unit,
// 8 samples per lane) — no transposition in LDS, operands are plain ds_read_b128.  k tile = 32 samples; a wavefront owns 64 x 64 of the
// block (4 x 4 tiles) and issues the six terms tile after tile, so no accumulator is touched twice in a row.
#ifndef ZK_WG_ABL
#define ZK_WG_ABL 0  // probe builds (wrong results): 1 no conversion, 2 no global loads, 3 no MFMA, 4 no LDS operand reads
#endif
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wsplit8(const float (&v)[8], wbf16x8& h, wbf16x8& m, wbf16x8& l) {
  if (ZK_WG_ABL == 1) {
    h = __builtin_bit_cast(wbf16x8, f32x4_t4{v[0], v[1], v[2], v[3]}); m = __builtin_bit_cast(wbf16x8, f32x4_t4{v[4], v[5], v[6], v[7]}); l = h;
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)v[e];
    const float r1 = v[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    h[e] = hh; m[e] = mm; l[e] = (__bf16)(r1 - (float)mm);
  }
}

__device__ __forceinline__ void wsplit8_half(const float (&v)[8], float sc, gh16x8& h, gh16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e] * sc;
    const _Float16 hh = (_Float16)x;
    h[e] = hh; l[e] = (_Float16)(x - (float)hh);
  }
}

// HALF: two-part f16 operands (a.g_amax / a.h_amax given; a rows-contraction has no per-row scale to factor out: one power of two per tensor, an
// element 2^-18 below its tensor's maximum still carries 22 bits, and the sum over rows is dominated by the large ones), images [unit block][h, l][lane]
template <bool HALF> __device__ __forceinline__ void wgrad_split_body(const WgradArgs& a, int work, uint4* Gs, uint4* Hs) {
  constexpr int NPART = HALF ? 2 : 3;
  int eg = 0, eh = 0;
  if constexpr (HALF) { eg = gh_exp(a.g_amax); eh = gh_exp(a.h_amax); }
  const float sg = __builtin_amdgcn_ldexpf(1.0f, eg), sh = __builtin_amdgcn_ldexpf(1.0f, eh);
  const int p = work % a.npairs, s = work / a.npairs;
  const int ob = a.pairs[2 * p], ib = a.pairs[2 * p + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t n_begin = (int64_t)s * a.S;
  int64_t n_end = n_begin + a.S;
  n_end = n_end < a.N ? n_end : a.N;

  f32x4_t4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t4{0.f, 0.f, 0.f, 0.f};

  // loader: thread -> unit u of the 128, sample octets oc0 and oc0 + 2 of the 4 in a k tile
  const int u = tid & 127, oc0 = tid >> 7;
  const int go = ob * 128 + u, hc = ib * 128 + u;
  const bool g_ok = go < a.OUT, h_ok = hc < a.IN;
  float rg[2][8], rh[2][8];
  auto gload_to = [&](float (&rg_)[2][8], float (&rh_)[2][8], int64_t n0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t n = n0 + (oc0 + 2 * t) * 8 + e;
        const bool in = n < n_end;
        rg_[t][e] = (ZK_WG_ABL != 2 && in && g_ok) ? a.g[n * a.ldg + go] : (ZK_WG_ABL == 2 ? (float)n : 0.f);
        rh_[t][e] = (ZK_WG_ABL != 2 && in && h_ok) ? a.h[n * a.ldh + hc] : (ZK_WG_ABL == 2 ? (float)e : 0.f);
      }
  };
  const bool do_cs = a.cs_flag && a.cs_flag[p];
  float csum = 0.f;
  const int img = (u >> 4) * NPART * 64 + (u & 15);  // + part * 64 + octet * 16
  // registers of one k tile -> operand images in LDS
  auto stage = [&](float (&rg_)[2][8], float (&rh_)[2][8]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int at = img + (oc0 + 2 * t) * 16;
      if constexpr (HALF) {
        gh16x8 hh, ll;
        wsplit8_half(rg_[t], sg, hh, ll);
        Gs[at] = __builtin_bit_cast(uint4, hh); Gs[at + 64] = __builtin_bit_cast(uint4, ll);
        wsplit8_half(rh_[t], sh, hh, ll);
        Hs[at] = __builtin_bit_cast(uint4, hh); Hs[at + 64] = __builtin_bit_cast(uint4, ll);
      } else {
        wbf16x8 hh, mm, ll;
        wsplit8(rg_[t], hh, mm, ll);
        Gs[at] = __builtin_bit_cast(uint4, hh); Gs[at + 64] = __builtin_bit_cast(uint4, mm); Gs[at + 128] = __builtin_bit_cast(uint4, ll);
        wsplit8(rh_[t], hh, mm, ll);
        Hs[at] = __builtin_bit_cast(uint4, hh); Hs[at + 64] = __builtin_bit_cast(uint4, mm); Hs[at + 128] = __builtin_bit_cast(uint4, ll);
      }
      if (do_cs) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum += rg_[t][e];
      }
    }
  };
  auto mma = [&]() {
    if constexpr (HALF) {
      gh16x8 A[4][2], B[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          A[i][part] = __builtin_bit_cast(gh16x8, Gs[((wr * 4 + i) * 2 + part) * 64 + lane]);
          B[i][part] = __builtin_bit_cast(gh16x8, Hs[((wc * 4 + i) * 2 + part) * 64 + lane]);
        }
      // three partial products, smallest first: (l, h) (h, l) (h, h)
#define ZK_HTERM(PA, PB)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i][PA], B[j][PB], acc[i][j], 0, 0, 0);
      ZK_HTERM(1, 0) ZK_HTERM(0, 1) ZK_HTERM(0, 0)
#undef ZK_HTERM
    } else {
      wbf16x8 A[4][3], B[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          A[i][part] = __builtin_bit_cast(wbf16x8, Gs[ZK_WG_ABL == 4 ? lane : ((wr * 4 + i) * 3 + part) * 64 + lane]);
          B[i][part] = __builtin_bit_cast(wbf16x8, Hs[ZK_WG_ABL == 4 ? lane : ((wc * 4 + i) * 3 + part) * 64 + lane]);
        }
      // six partial products, smallest first: (l, h) (h, l) (m, m) (m, h) (h, m) (h, h)
#define ZK_WTERM(PA, PB)                                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
      if (ZK_WG_ABL == 3) { asm volatile("" ::"v"(A[i][PA]), "v"(B[j][PB])); } else                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][PA], B[j][PB], acc[i][j], 0, 0, 0);
      ZK_WTERM(2, 0) ZK_WTERM(0, 2) ZK_WTERM(1, 1) ZK_WTERM(1, 0) ZK_WTERM(0, 1) ZK_WTERM(0, 0)
#undef ZK_WTERM
    }
  };
  // (a second k tile of look-ahead in registers — tried for the two-part operands, whose fragments take 64 registers instead of 96 — still spills: 151 VGPRs)
  gload_to(rg, rh, n_begin);
  for (int64_t n0 = n_begin; n0 < n_end; n0 += 32) {
    stage(rg, rh);
    __syncthreads();
    if (n0 + 32 < n_end) gload_to(rg, rh, n0 + 32);
    mma();
    __syncthreads();
  }
  if (a.cs_flag) {  // (uniform per block: cs_flag[p])
    if (do_cs) {
      float* cs = reinterpret_cast<float*>(Gs);
      cs[tid] = csum;
      __syncthreads();
      if (tid < 128) a.cs_partial[(size_t)s * a.cs_ld + ob * 128 + tid] = cs[tid] + cs[tid + 128];
    }
  }
  // acc[i][j][r] = D[row 4 (lane >> 4) + r][col lane & 15] of tile (i, j)
  const float dg = __builtin_amdgcn_ldexpf(1.0f, -eg), dh = __builtin_amdgcn_ldexpf(1.0f, -eh);
  float* dst = a.partial + ((size_t)s * a.npairs + p) * (128 * 128);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(wr * 64 + i * 16 + 4 * (lane >> 4) + r) * 128 + wc * 64 + j * 16 + (lane & 15)] = HALF ? acc[i][j][r] * dg * dh : acc[i][j][r];
}

// XCD-aware walk: workgroups go round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  XCD x takes a CONTIGUOUS range of the
// (slice, pair) work list, so the blocks that share a G tile (the in blocks of one out block) and an H tile (all out blocks of a slice) run
// on the same L2 at about the same time.
__device__ __forceinline__ int wgrad_work(int total) {
  const int per_xcd = (total + 7) / 8;
  const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  return work < total ? work : -1;
}

__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 Gs[8 * 3 * 64];  // [unit block of 16][part h, m, l][lane]: 24 KiB
  __shared__ __attribute__((aligned(16))) uint4 Hs[8 * 3 * 64];
  const int work = wgrad_work(a.nslices * a.npairs);
  if (work < 0) return;
  wgrad_split_body<false>(a, work, Gs, Hs);
}

// The weight gradients of ALL layers of a conditioner in one launch (up to four: the layers' work lists follow each other; a layer with two
// or three live blocks is otherwise a launch of its own that the GPU finishes faster than the host can queue the next one).
struct WgradMulti {
  int n;
  int start[5];  // first work item of every layer, start[n] = total
  WgradArgs a[4];
};
template <bool HALF> __global__ __launch_bounds__(256, 2) void wgrad_split_multi_kernel(WgradMulti m) {
  __shared__ __attribute__((aligned(16))) uint4 Gs[8 * 3 * 64];
  __shared__ __attribute__((aligned(16))) uint4 Hs[8 * 3 * 64];
  const int work = wgrad_work(m.start[m.n]);
  if (work < 0) return;
  // (a chain of selects on the by-value argument block: indexing it with a run-time layer would spill the whole block to scratch)
  WgradArgs a = m.a[0];
  int base = 0;
  if (m.n > 1 && work >= m.start[1]) { a = m.a[1]; base = m.start[1]; }
  if (m.n > 2 && work >= m.start[2]) { a = m.a[2]; base = m.start[2]; }
  if (m.n > 3 && work >= m.start[3]) { a = m.a[3]; base = m.start[3]; }
  wgrad_split_body<HALF>(a, work - base, Gs, Hs);
}

// dW[o, i] (+)= mask[o, i] * sum_s partial[s][p][...] in slice order (deterministic); 64 blocks of 256 elements per pair
// rows / cols (optional): the gradient is computed on a row / column PERMUTED weight (zuko_amd/train.py: units sorted by dependency
// count) but written where the module keeps it: element (o, c) goes to dw[rows[o], cols[c]] — no scatter pass afterwards
struct WredArgs {
  int OUT, IN, npairs, nslices, accumulate;
  const int32_t* pairs;
  const float* partial;
  const uint8_t* mask;
  float* dw;
  const int32_t *rows, *cols;
  // bias gradient (optional): column sums of g from the same pass
  const float* cs_partial;
  float* db;
  int cs_ld;
};
__device__ __forceinline__ void wgrad_reduce_body(int OUT, int IN, const int32_t* pairs, int npairs, int nslices, const float* partial, const uint8_t* mask, float* dw, int accumulate,
                                                  const int32_t* rows, const int32_t* cols, int block) {
  const int p = block >> 6;
  const int e = ((block & 63) << 8) + threadIdx.x;
  const int ob = pairs[2 * p], ib = pairs[2 * p + 1];
  const int i = e >> 7, j = e & 127;
  const int o = ob * 128 + i, c = ib * 128 + j;
  if (o >= OUT || c >= IN) return;
  const float* src = partial + (size_t)p * (128 * 128) + e;
  const size_t stride = (size_t)npairs * (128 * 128);
  float sum = 0.f;
  int s = 0;
  for (; s + 8 <= nslices; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; s < nslices; ++s) sum += src[(size_t)s * stride];
  const size_t idx = (size_t)o * IN + c;
  if (mask && !mask[idx]) sum = 0.f;
  if (rows && rows[o] < 0) return;  // (a padding slot of a packed gradient: no row of the weight behind it)
  const size_t dst = (size_t)(rows ? rows[o] : o) * IN + (cols ? cols[c] : c);
  dw[dst] = accumulate ? dw[dst] + sum : sum;
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int OUT, int IN, const int32_t* pairs, int npairs, int nslices, const float* partial,
                                                           const uint8_t* mask, float* dw, int accumulate, const int32_t* rows, const int32_t* cols) {
  wgrad_reduce_body(OUT, IN, pairs, npairs, nslices, partial, mask, dw, accumulate, rows, cols, (int)blockIdx.x);
}
struct WredMulti {
  int n;
  int start[5];     // first reduce block (64 per live pair) of every layer
  int cs_start[5];  // first column-sum block (256 columns each) of every layer, after all reduce blocks
  WredArgs r[4];
};
// dW of every layer (+ the bias gradients) from the partial sums of wgrad_split_multi_kernel: one launch
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WredMulti m) {
  const int b = (int)blockIdx.x;
  if (b < m.start[m.n]) {
    WredArgs r = m.r[0];
    int base = 0;
    if (m.n > 1 && b >= m.start[1]) { r = m.r[1]; base = m.start[1]; }
    if (m.n > 2 && b >= m.start[2]) { r = m.r[2]; base = m.start[2]; }
    if (m.n > 3 && b >= m.start[3]) { r = m.r[3]; base = m.start[3]; }
    wgrad_reduce_body(r.OUT, r.IN, r.pairs, r.npairs, r.nslices, r.partial, r.mask, r.dw, r.accumulate, r.rows, r.cols, b - base);
    return;
  }
  const int cb = b - m.start[m.n];
  WredArgs r = m.r[0];
  int base = 0;
  if (m.n > 1 && cb >= m.cs_start[1]) { r = m.r[1]; base = m.cs_start[1]; }
  if (m.n > 2 && cb >= m.cs_start[2]) { r = m.r[2]; base = m.cs_start[2]; }
  if (m.n > 3 && cb >= m.cs_start[3]) { r = m.r[3]; base = m.cs_start[3]; }
  if (!r.db) return;
  const int c = (cb - base) * 256 + threadIdx.x;
  if (c >= r.OUT) return;
  float sum = 0.f;
  int sl = 0;
  for (; sl + 8 <= r.nslices; sl += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r.cs_partial[(size_t)(sl + u) * r.cs_ld + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; sl < r.nslices; ++sl) sum += r.cs_partial[(size_t)sl * r.cs_ld + c];
  if (r.rows && r.rows[c] < 0) return;
  r.db[r.rows ? r.rows[c] : c] = sum;
}

// column sums: out[c] = sum_n x[n, c] (bias gradients), two passes with a fixed reduction order
__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t N, int C, const float* x, int64_t ld, int64_t S, float* partial) {
  // block = 64 columns x one slice of S rows; thread = (column quad q = tid % 16, row phase = tid / 16): float4 loads,
  // 8 rows in flight per thread
  const int s = blockIdx.y;
  const int q = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + q * 4;
  __shared__ float sh[16][64];
  const int64_t n0 = (int64_t)s * S;
  int64_t n1 = n0 + S;
  n1 = n1 < N ? n1 : N;
  const bool vec = (ld % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && (c0 + 4 <= C);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    int64_t n = n0 + ph;
    for (; n + 16 * 7 < n1; n += 16 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* src = x + (n + 16 * u) * ld + c0;
        if (vec) v[u] = *reinterpret_cast<const float4*>(src);
        else v[u] = make_float4(src[0], c0 + 1 < C ? src[1] : 0.f, c0 + 2 < C ? src[2] : 0.f, c0 + 3 < C ? src[3] : 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w; }
    }
    for (; n < n1; n += 16) {
      const float* src = x + n * ld + c0;
      acc[0] += src[0];
      if (c0 + 1 < C) acc[1] += src[1];
      if (c0 + 2 < C) acc[2] += src[2];
      if (c0 + 3 < C) acc[3] += src[3];
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) sh[ph][q * 4 + t] = acc[t];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += sh[r][threadIdx.x];
    if (c < C) partial[(size_t)s * C + c] = sum;
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(int C, int nslices, const float* partial, float* out, int accumulate, int ld = 0, const int32_t* rows = nullptr) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const size_t stride = ld ? (size_t)ld : (size_t)C;
  // eight loads in flight (a single dependent chain over ~100 slices took 29 us per call: latency, not bandwidth); the order of the
  // additions is fixed, so the result is still deterministic
  float sum = 0.f;
  int s = 0;
  for (; s + 8 <= nslices; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(s + u) * stride + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; s < nslices; ++s) sum += partial[(size_t)s * stride + c];
  const int d = rows ? rows[c] : c;  // (sorted-domain column -> where the module keeps it)
  if (d < 0) return;                 // padding slot of a packed row table (include/zuko_amd.h): no destination
  out[d] = accumulate ? out[d] + sum : sum;
}

}  // namespace zk

using namespace zk;

extern "C" {

// y = act(x w^T + bias) [ * act'_{gate_act}(gate) ],  w [OUT, IN] pre-masked, kskip: one 64-bit word per 128-row out block (or NULL)
int zk_gemm_f32_skip(int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* w, const uint64_t* kskip, const void* bias, int act,
                     const void* gate, int64_t ldg, int gate_act, void* y, int64_t ldy, void* stream) {
  if (N <= 0 || out_features <= 0) return 0;
  if (in_features <= 0 || (kskip && in_features > 2048) || !(act == 0 || act == 1 || act == 2 || act == 3 || act == 6 || act == 7)) return ZK_EINVAL;
  if (gate && !(gate_act == 0 || gate_act == 1 || gate_act == 2 || gate_act == 3 || gate_act == 6 || gate_act == 7)) return ZK_EINVAL;
  GemmArgs a{};
  a.N = N; a.IN = in_features; a.OUT = out_features; a.x = (const float*)x; a.ldx = ldx; a.w = (const float*)w; a.kskip = kskip; a.bias = (const float*)bias;
  a.act = act; a.gate = (const float*)gate; a.ldg = ldg; a.gate_act = gate_act; a.y = (float*)y; a.ldy = ldy;
  a.nbx = (int)((N + TBM - 1) / TBM);
  a.nby = (out_features + TBN - 1) / TBN;
  const bool vec = (in_features % 4 == 0) && (ldx % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)w)) % 16 == 0);
  if (vec) hipLaunchKernelGGL(gemm_f32_skip<true>, dim3((unsigned)(a.nbx * a.nby)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gemm_f32_skip<false>, dim3((unsigned)(a.nbx * a.nby)), dim3(256), 0, (hipStream_t)stream, a);
  return ZK_LAUNCH_CHECK();
}

// rows per slice / number of slices zk_wgrad_f32 will use for N rows and npairs live blocks (the caller sizes `partial`
// as nslices * npairs * 128 * 128 floats)
int zk_wgrad_slices(int64_t N, int npairs) {
  if (N <= 0 || npairs <= 0) return 0;
  // at most BPC (default 4: the kernel's LDS allows four blocks per CU; measured 0.72 -> 0.52 ms for 1472 x 256 at N = 2^16) blocks per CU, rounded DOWN: with a few blocks more than a whole number per CU (e.g. 520 on 256 CUs) the CUs that
  // receive an extra block finish 1.5x later and the launch waits for them
  static const int bpc = [] { const char* e = getenv("ZUKO_AMD_WGRAD_BPC"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 4; }();
  int64_t want = (bpc * 256) / npairs;
  want = want < 1 ? 1 : want;
  int64_t S = (N + want - 1) / want;
  S = S < 512 ? 512 : S;
  S = (S + 15) / 16 * 16;
  return (int)((N + S - 1) / S);
}

// dw[OUT, IN] (+)= mask .* (g^T h): g [N, OUT], h [N, IN]; pairs = the (128 x 128) blocks of dw to compute (device, int32 [npairs][2]);
// the blocks not listed are left untouched (the caller zero-fills dw once).  Deterministic.
static int wgrad_launch(int64_t N, int out_features, int in_features, const void* g, int64_t ldg, const void* h, int64_t ldh, const int32_t* pairs, int npairs,
                        float* partial, const uint8_t* mask, void* dw, int accumulate, const uint8_t* cs_flag, float* cs_partial, void* db, const int32_t* rows, const int32_t* cols,
                        void* stream) {
  if (N <= 0 || npairs <= 0) return 0;
  WgradArgs a{};
  a.cs_flag = cs_flag; a.cs_partial = cs_partial; a.cs_ld = (out_features + 127) / 128 * 128;
  a.N = N; a.OUT = out_features; a.IN = in_features; a.g = (const float*)g; a.ldg = ldg; a.h = (const float*)h; a.ldh = ldh; a.pairs = pairs; a.npairs = npairs;
  a.nslices = zk_wgrad_slices(N, npairs);
  a.S = ((N + a.nslices - 1) / a.nslices + 15) / 16 * 16;
  a.nslices = (int)((N + a.S - 1) / a.S);
  a.partial = partial;
  hipStream_t st = (hipStream_t)stream;
  // (operand-split kernel unless ZUKO_AMD_EXACT_F32=1 asks for the f32 matrix instruction; k tiles of 32 samples need S % 32 == 0)
  static const bool exact = [] { const char* e = getenv("ZUKO_AMD_EXACT_F32"); return e && e[0] == '1'; }();
  if (exact) {
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3((unsigned)(8 * ((a.nslices * npairs + 7) / 8))), dim3(256), 0, st, a);
  } else {
    a.S = (a.S + 31) / 32 * 32;
    a.nslices = (int)((N + a.S - 1) / a.S);
    hipLaunchKernelGGL(wgrad_split_kernel, dim3((unsigned)(8 * ((a.nslices * npairs + 7) / 8))), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)npairs * 64), dim3(256), 0, st, out_features, in_features, pairs, npairs, a.nslices, (const float*)partial, mask,
                     (float*)dw, accumulate, rows, cols);
  if (cs_flag) hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((out_features + 255) / 256)), dim3(256), 0, st, out_features, a.nslices, (const float*)cs_partial, (float*)db, 0, a.cs_ld, rows);
  return ZK_LAUNCH_CHECK();
}

int zk_wgrad_f32(int64_t N, int out_features, int in_features, const void* g, int64_t ldg, const void* h, int64_t ldh, const int32_t* pairs, int npairs,
                 float* partial, const uint8_t* mask, void* dw, int accumulate, const int32_t* rows, const int32_t* cols, void* stream) {
  return wgrad_launch(N, out_features, in_features, g, ldg, h, ldh, pairs, npairs, partial, mask, dw, accumulate, nullptr, nullptr, nullptr, rows, cols, stream);
}

// zk_wgrad_f32 plus the bias gradient db[rows[o]] = sum_n g[n, o] (db[o] without rows) from the same pass over g (it replaces a separate zk_colsum_f32 over g):
// cs_flag [npairs] (device, uint8) marks ONE pair per out block (every out block must have one); cs_partial: workspace of
// zk_wgrad_slices(N, npairs) * ceil(OUT / 128) * 128 floats.  Deterministic.
int zk_wgrad_bias_f32(int64_t N, int out_features, int in_features, const void* g, int64_t ldg, const void* h, int64_t ldh, const int32_t* pairs, int npairs,
                      float* partial, const uint8_t* mask, void* dw, int accumulate, const uint8_t* cs_flag, float* cs_partial, void* db, const int32_t* rows,
                      const int32_t* cols, void* stream) {
  if (!cs_flag || !cs_partial || !db) return ZK_EINVAL;
  return wgrad_launch(N, out_features, in_features, g, ldg, h, ldh, pairs, npairs, partial, mask, dw, accumulate, cs_flag, cs_partial, db, rows, cols, stream);
}

// zk_wgrad_bias_f32 for up to four layers in TWO launches (operand-split kernel; the reductions of all layers in one kernel).  `layers`:
// HOST array of n descriptors (include/zuko_amd.h: zk_wgrad_layer_v1).  Every layer needs cs_flag / cs_partial / db (bias gradient) or none
// of the three; dw / db are written, not accumulated.
int zk_wgrad_multi(int n, const zk_wgrad_layer_v1* layers, int64_t N, void* stream) {
  if (n < 1 || n > 4 || !layers) return ZK_EINVAL;
  if (N <= 0) return 0;
  WgradMulti m{};
  WredMulti r{};
  m.n = r.n = n;
  int work = 0, red = 0, cs = 0;
  bool half = true;  // every layer comes with the maxima of its operands: two-part f16 products
  // ONE slice length for all layers, so that every block of the launch runs equally long and the blocks fill a whole number of rounds:
  // about zk_wgrad_slices' blocks-per-CU target over the live blocks of ALL layers (never more slices than a layer's own launch would use:
  // the callers size `partial` for that)
  int64_t pairs_total = 0;
  for (int l = 0; l < n; ++l) pairs_total += layers[l].npairs > 0 ? layers[l].npairs : 0;
  const int common = pairs_total > 0 ? zk_wgrad_slices(N, (int)pairs_total) : 1;
  for (int l = 0; l < n; ++l) {
    const zk_wgrad_layer_v1& d = layers[l];
    if (d.struct_size != sizeof(zk_wgrad_layer_v1) || d.npairs <= 0 || !d.g || !d.h || !d.pairs || !d.partial || !d.dw) return ZK_EINVAL;
    if ((d.db != nullptr) != (d.cs_flag != nullptr) || (d.db != nullptr) != (d.cs_partial != nullptr)) return ZK_EINVAL;
    WgradArgs& a = m.a[l];
    a.N = N; a.OUT = d.out_features; a.IN = d.in_features; a.g = (const float*)d.g; a.ldg = d.ldg; a.h = (const float*)d.h; a.ldh = d.ldh;
    a.pairs = d.pairs; a.npairs = d.npairs; a.partial = (float*)d.partial;
    a.cs_flag = d.cs_flag; a.cs_partial = (float*)d.cs_partial; a.cs_ld = (d.out_features + 127) / 128 * 128;
    a.g_amax = d.g_amax; a.h_amax = d.h_amax;
    half = half && d.g_amax && d.h_amax;
    a.nslices = zk_wgrad_slices(N, d.npairs);
    a.nslices = a.nslices < common ? a.nslices : common;
    a.S = (((N + a.nslices - 1) / a.nslices) + 31) / 32 * 32;
    a.nslices = (int)((N + a.S - 1) / a.S);
    m.start[l] = work; work += a.nslices * a.npairs;
    WredArgs& q = r.r[l];
    q.OUT = a.OUT; q.IN = a.IN; q.npairs = a.npairs; q.nslices = a.nslices; q.accumulate = 0; q.pairs = a.pairs; q.partial = a.partial; q.mask = d.mask; q.dw = (float*)d.dw;
    q.rows = d.rows; q.cols = d.cols; q.cs_partial = a.cs_partial; q.db = (float*)d.db; q.cs_ld = a.cs_ld;
    r.start[l] = red; red += a.npairs * 64;
    r.cs_start[l] = cs; cs += (a.OUT + 255) / 256;
  }
  m.start[n] = work; r.start[n] = red; r.cs_start[n] = cs;
  hipStream_t st = (hipStream_t)stream;
  if (half) hipLaunchKernelGGL(wgrad_split_multi_kernel<true>, dim3((unsigned)(8 * ((work + 7) / 8))), dim3(256), 0, st, m);
  else hipLaunchKernelGGL(wgrad_split_multi_kernel<false>, dim3((unsigned)(8 * ((work + 7) / 8))), dim3(256), 0, st, m);
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)(red + cs)), dim3(256), 0, st, r);
  return ZK_LAUNCH_CHECK();
}

// out[c] (+)= sum_n x[n, c];  workspace: >= zk_colsum_slices(N) * C floats
int zk_colsum_slices(int64_t N) {
  int64_t s = (N + 511) / 512;  // (2048 rows per slice left a 256-column sum with 128 blocks on 256 CUs)
  return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}
int zk_colsum_f32(int64_t N, int C, const void* x, int64_t ld, float* workspace, void* out, int accumulate, void* stream) {
  if (C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int ns = N > 0 ? zk_colsum_slices(N) : 1;
  const int64_t S = N > 0 ? (N + ns - 1) / ns : 1;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)ns), dim3(256), 0, st, N, C, (const float*)x, ld, S, workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, C, ns, (const float*)workspace, (float*)out, accumulate);
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
