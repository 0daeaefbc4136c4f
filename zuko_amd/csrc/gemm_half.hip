// zuko_amd — DENSE conditioner GEMMs of the training path on the f16 matrix instruction with TWO-PART f32 operands.
//
// What it replaces: `F.linear(x, W, b)` + activation of a coupling conditioner (zuko/nn.py:15 `MLP`, called from
// zuko/flows/coupling.py:128-136) and the two GEMMs autograd derives from it.  Until round 5 these ran on v_mfma_f32_32x32x2_f32
// (csrc/train.hip: gemm_f32_skip, 1/16 of the f16 rate): 8.1 of the 17.2 ms of a RealNVP cfg4 training step at 2^14 rows.
//
// Arithmetic (the scheme of csrc/fused_ar_half_impl.h, with PER-TENSOR instead of per-sample scales — a GEMM tile does not see a whole row):
//   every f32 operand x is multiplied by a power of two 2^e (exact) chosen so that the tensor's largest magnitude lands in [2^14, 2^15),
//   and written as h = f16(x 2^e), l = f16(x 2^e - h): 22 significand bits for every element within 2^-18 of the tensor's maximum, an
//   absolute error of 2^-40 of that maximum below.  A product is three v_mfma_f32_16x16x32_f16 (l h, h l, h h: smallest first, f32
//   accumulation), de-scaled by one multiplication per output.  The maxima are DEVICE scalars (bit patterns of non-negative floats under
//   atomicMax): a GEMM reads the maxima of its operands and — optionally — leaves the maximum of its own output for its consumer, so a
//   chain of layers needs no host synchronisation.
//
// Layout: workgroup tile (64 WM) x (64 WN) out of WM x WN wavefronts of 64 rows x 64 units (4 x 4 MFMA tiles) each — (2, 4), (2, 2) or (1, 2), the widest that
// still gives every CU a workgroup (a 256-wide tile converts every activation once per 256 outputs) — K in steps of 32.
//   * the weights come PRE-SPLIT (zk_wsplit_f16, once per optimiser step) as the 1 KiB lane images the matrix instruction reads
//     ([128-unit tile][k step][16-unit tile][h | l][lane][8 halves]) and are moved global -> LDS by `global_load_lds_dwordx4`, no VGPR round trip;
//   * the f32 activation tile travels global -> LDS by the same DMA as it is (three slots, requested three steps ahead), is read back with ordinary LDS
//     loads, scaled + split in registers and stored as the same kind of image (raw ds_write_b64) while the step before it is multiplied;
//     image slot of (row j, k-quarter kq) = 4 j + (kq ^ ((-(j >> 2)) & 3)): the 16 lanes of a ds_write_b64 group cover 128 contiguous bytes
//     and the 16 lanes of every ds_read_b128 group 16 distinct slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS table; counters: 0 conflicts);
//   * one barrier per k step.  Why everything in flight is a DMA, what the counters and removal probes say: DESIGN.md 3.7, profiles/r06/.
//   Lane (j = lane & 15, q = lane >> 4) of an accumulator tile owns out units 4 q .. 4 q + 3 of row j: 16-byte stores.
#include "zk_common.h"
#include "zk_half.h"
#include <stdlib.h>

#include "../../include/zuko_amd.h"

namespace zk {

typedef float ghf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gh_split8(const float (&v)[8], float s, gh16x8& h, gh16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e] * s;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    l[e] = (_Float16)(x - (float)hh);
  }
}
// ---- maxima of |x| ---------------------------------------------------------------------------------------------------------------
struct AmItem { const float* src; int64_t rows; int cols; int64_t ld; unsigned* out; };
struct AmMulti { AmItem it[8]; };
__global__ __launch_bounds__(256) void amax_kernel(AmMulti m) {
  const AmItem& it = m.it[blockIdx.y];
  const int64_t total = it.rows * it.cols;
  float mx = 0.f;
  if (it.ld == it.cols && it.cols % 4 == 0 && (((uintptr_t)it.src) & 15) == 0) {
    const float4* p = reinterpret_cast<const float4*>(it.src);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total / 4; i += (int64_t)gridDim.x * 256) {
      const float4 v = p[i];
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(it.src[(i / it.cols) * it.ld + i % it.cols]));
  }
  // (fmaxf drops NaNs: a NaN operand then shows up in the product itself, as in any GEMM)
  mx = gh_wave_max(mx);
  if ((threadIdx.x & 63) == 0) gh_amax_put(it.out, blockIdx.x * 4 + (threadIdx.x >> 6), mx);
}

// ---- weight images -----------------------------------------------------------------------------------------------------------------
struct WsItem { const float* src; const uint8_t* mask; const unsigned* amax; uint4* dst; int U, K; int64_t su, sk; int64_t slots; int nks; };
struct WsMulti { WsItem it[8]; };
__global__ __launch_bounds__(256) void wsplit_kernel(WsMulti m) {
  const WsItem& it = m.it[blockIdx.y];
  const float s = __builtin_amdgcn_ldexpf(1.0f, gh_exp(it.amax));
  for (int64_t sl = (int64_t)blockIdx.x * 256 + threadIdx.x; sl < it.slots; sl += (int64_t)gridDim.x * 256) {
    const int lane = (int)(sl & 63), ut = (int)((sl >> 6) & 7);
    const int64_t rest = sl >> 9;  // n tile * nks + k step
    const int ks = (int)(rest % it.nks), nt = (int)(rest / it.nks);
    const int u = nt * 128 + ut * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t at = (int64_t)u * it.su + (int64_t)(k0 + e) * it.sk;
      v[e] = (u < it.U && k0 + e < it.K && (!it.mask || it.mask[at])) ? it.src[at] : 0.f;
    }
    gh16x8 h, l;
    gh_split8(v, s, h, l);
    uint4* d = it.dst + (rest * 16 + ut * 2) * 64 + lane;
    d[0] = __builtin_bit_cast(uint4, h);
    d[64] = __builtin_bit_cast(uint4, l);
  }
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------------------------
struct GhArgs {
  int64_t M; int K, N;
  const float* a; int64_t lda;   // [M, K]
  const unsigned* a_amax;        // >= max |a| (bit pattern)
  const uint4* w;                // images of zk_wsplit_f16 for (N units, K)
  const unsigned* w_amax;        // the maximum the images were scaled with
  const float* bias;             // [N] or null
  int act;                       // NONE, RELU, ELU, TANH, SIGMOID, LEAKY
  const float* gate; int64_t ldg; int gate_act;  // optional [M, N]: result *= act'(gate), the derivative written in the activation's OUTPUT (ReLU: gate > 0)
  float* c; int64_t ldc;
  unsigned* c_amax;              // or null: atomicMax of |c|
  int nbm, nbn, nks;
};

#ifndef GH_ABL
#define GH_ABL 0  // probe builds only (wrong results): 1 no conversion arithmetic, 2 no matrix instructions, 3 no raw-tile DMA, 4 no weight DMA, 5 no stores, 6 no fragment reads
#endif
#define GH_IMG 64  /* uint4 per image */
#define GH_WR 2    /* LDS slots of weight images: step s + 1 is requested while step s is multiplied (the weights sit in L2) */
#define GH_AR 3    /* LDS slots of raw f32 activation tiles: step s + 3 is requested while step s is multiplied (they come from HBM) */

extern __shared__ __attribute__((aligned(16))) uint4 gh_lds[];

typedef _Float16 gh16x4 __attribute__((ext_vector_type(4)));
typedef unsigned ghu2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gh_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void gh_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WM x WN wavefronts of 64 rows x 64 out units each: workgroup tile (64 WM) x (64 WN), k step 32.  Everything that is in flight across a k step is
// an LDS-DMA (hipcc cannot keep an ordinary load in flight across the loop without copying its destination registers, and drains every DMA at
// the first use of one): the f32 activation tile travels global -> LDS as it is, is read back (ordinary LDS loads), scaled, split and stored as
// lane images while the step before it is multiplied.
// OTHER: an activation beyond NONE / RELU is in play (a separate instantiation: their inline expansions, unrolled over the 64 outputs of a lane, do not belong in
// the instruction stream of the ReLU kernel)
template <int WM, int WN, bool OTHER> __global__ __launch_bounds__(64 * WM * WN, 2) void gemm_half_kernel(GhArgs a) {
  constexpr int NT = 64 * WM * WN;     // threads
  constexpr int NP = 8 / WN;           // 16-byte pieces of the activation tile per thread and step
  constexpr int NW = 8 / WM;           // weight images per wavefront and step
  constexpr int AIMG = 8 * WM;         // activation images per step (4 WM row tiles x {h, l})
  constexpr int WIMG = 8 * WN;
  constexpr int RAW = 8 * WM * GH_IMG; // uint4 of a raw tile (64 WM rows x 128 bytes)
  // LDS (uint4): [GH_WR x W images | GH_AR x raw tiles | 2 x A images]
  constexpr int OFF_RAW = GH_WR * WIMG * GH_IMG, OFF_IMG = OFF_RAW + GH_AR * RAW;
  // logical block id: blocks that share a row panel are consecutive and stay on one XCD (block b runs on XCD b % 8)
  int bm, bn;
  {
    const int nwg = a.nbm * a.nbn, orig = blockIdx.x;
    const int xcd = orig % 8, qq = nwg / 8, rr = nwg % 8;
    const int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + orig / 8;
    bm = logical / a.nbn;
    bn = logical % a.nbn;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int j = lane & 15, q = lane >> 4;
  const int64_t m0 = (int64_t)bm * (64 * WM);
  const int n0 = bn * (64 * WN);
  const int ea = gh_exp(a.a_amax), ew = gh_exp(a.w_amax);
  const float sa = __builtin_amdgcn_ldexpf(1.0f, ea);

  // raw tile: row-major [64 WM rows][8 pieces of 16 bytes]; thread t moves and later converts pieces t, t + NT, ..: row (t >> 3) + p NT / 8, piece t & 7
  // (8 consecutive lanes = one full 128-byte line).  A row past M reads row 0 (its outputs are never stored), a piece past K reads piece 0 and converts to 0.
  const int sr = tid >> 3, sc = tid & 7;
  const float* a_src[NP];
  int a_wr[NP];  // byte offset of the 8-byte half-slot this thread writes inside an image pair set
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int r = sr + p * (NT / 8);
    const int rj = r & 15, kq = sc >> 1;
    const int slot = 4 * rj + (kq ^ ((-(rj >> 2)) & 3));
    a_wr[p] = (((r >> 4) * 2) * GH_IMG + slot) * 16 + (sc & 1) * 8;
    const int64_t row = m0 + r;
    a_src[p] = a.a + (row < a.M ? row : 0) * a.lda;
  }
  const int a_rd = 4 * j + (q ^ ((-(j >> 2)) & 3));
  auto adma = [&](int ks, int slot) {
    const int k = ks * 32 + 4 * sc;
    const int kk = k < a.K ? k : 0;
    uint4* l = gh_lds + OFF_RAW + slot * RAW + wave * GH_IMG;
#pragma unroll
    for (int p = 0; p < NP; ++p) if (GH_ABL != 3) gh_dma16(a_src[p] + kk, l + p * (NT / 64) * GH_IMG);
  };
  // weights: wavefront w moves images NW w .. NW w + NW - 1 of the step
  //          (the images are laid out per 128-unit tile: [tile][k step][16 images]; a tile past the last one is clamped — its outputs are never stored)
  const int nt_last = (a.N + 127) / 128 - 1;
  const int nt_w = bn * (WN / 2) + (wave * NW) / 16;
  const uint4* wsrc = a.w + ((int64_t)(nt_w < nt_last ? nt_w : nt_last) * a.nks * 16 + (wave * NW) % 16) * GH_IMG + lane;
  auto wdma = [&](int ks, int slot) {
    const uint4* g = wsrc + (int64_t)ks * 16 * GH_IMG;
    uint4* l = gh_lds + slot * WIMG * GH_IMG + wave * NW * GH_IMG;
#pragma unroll
    for (int i = 0; i < NW; ++i) if (GH_ABL != 4) gh_dma16(g + i * GH_IMG, l + i * GH_IMG);
  };
  // (raw ds_write: behind an ordinary LDS store hipcc drains every LDS-DMA in flight — vmcnt(0) — as a possible alias)
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) uint4*)gh_lds);
  ghf4 raw[NP];
  auto rawload = [&](int slot) {
    const uint4* l = gh_lds + OFF_RAW + slot * RAW + tid;
#pragma unroll
    for (int p = 0; p < NP; ++p) raw[p] = __builtin_bit_cast(ghf4, l[p * NT]);
  };
  auto convert = [&](int ks, int islot) {
    const bool kin = ks * 32 + 4 * sc < a.K;
    const unsigned base = lds0 + (unsigned)(OFF_IMG + islot * AIMG * GH_IMG) * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      gh16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (GH_ABL == 1) { h[e] = __builtin_bit_cast(_Float16, (unsigned short)__builtin_bit_cast(unsigned, raw[p][e])); l[e] = h[e]; continue; }
        const float x = kin ? raw[p][e] * sa : 0.f;
        const _Float16 hh = (_Float16)x;
        h[e] = hh;
        l[e] = (_Float16)(x - (float)hh);
      }
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:1024" ::"v"(base + a_wr[p]), "v"(__builtin_bit_cast(ghu2, h)), "v"(__builtin_bit_cast(ghu2, l)) : "memory");
    }
  };
  auto sync = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  ghf4 acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[u][s] = ghf4{0.f, 0.f, 0.f, 0.f};

  const int nks = a.nks;
  // prologue: weights of step 0, raw tiles of steps 0..2 requested; step 0 converted
  wdma(0, 0);
  adma(0, 0);
  if (nks > 1) adma(1, 1);
  if (nks > 2) adma(2, 2);
  if (nks > 2) gh_wait_vm<2 * NP>(); else if (nks > 1) gh_wait_vm<NP>(); else gh_wait_vm<0>();
  sync();
  rawload(0);
  convert(0, 0);
  if (nks > 2) gh_wait_vm<NP>(); else gh_wait_vm<0>();
  sync();

  int ws = 0, as = 0;  // ring slots of step ks: weights (mod GH_WR), raw tile (mod GH_AR); image slot = ks & 1
  for (int ks = 0; ks < nks; ++ks) {
    const int ws1 = ws ^ 1, as1 = as + 1 == GH_AR ? 0 : as + 1;
    const bool more = ks + 1 < nks, req = ks + 3 < nks;
    if (more) wdma(ks + 1, ws1);
    if (req) adma(ks + 3, as);  // (the slot of step ks: read back at step ks - 1)
    if (more) rawload(as1);
    const uint4* const wcur = gh_lds + ws * WIMG * GH_IMG;
    const uint4* const acur = gh_lds + OFF_IMG + (ks & 1) * AIMG * GH_IMG;
    gh16x8 wh[4], wl[4], ah[4], al[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wh[u] = __builtin_bit_cast(gh16x8, wcur[GH_ABL == 6 ? lane : ((wn * 4 + u) * 2) * GH_IMG + lane]);
      wl[u] = __builtin_bit_cast(gh16x8, wcur[GH_ABL == 6 ? lane : ((wn * 4 + u) * 2 + 1) * GH_IMG + lane]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ah[s] = __builtin_bit_cast(gh16x8, acur[GH_ABL == 6 ? lane : ((wm * 4 + s) * 2) * GH_IMG + a_rd]);
      al[s] = __builtin_bit_cast(gh16x8, acur[GH_ABL == 6 ? lane : ((wm * 4 + s) * 2 + 1) * GH_IMG + a_rd]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) if (GH_ABL == 2) { asm volatile("" ::"v"(wl[u]), "v"(ah[s])); } else acc[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[u], ah[s], acc[u][s], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) if (GH_ABL == 2) { asm volatile("" ::"v"(wh[u]), "v"(al[s])); } else acc[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u], al[s], acc[u][s], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) if (GH_ABL == 2) { asm volatile("" ::"v"(wh[u]), "v"(ah[s])); } else acc[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u], ah[s], acc[u][s], 0, 0, 0);
    if (more) convert(ks + 1, (ks + 1) & 1);
    // the weights of step ks + 1 and the raw tile of step ks + 2 have landed once only this step's raw request is outstanding
    if (req) gh_wait_vm<NP>(); else gh_wait_vm<0>();
    sync();
    ws = ws1;
    as = as1;
  }

  // epilogue: acc[u][s][r] = C[m0 + wm 64 + s 16 + j][n0 + wn 64 + u 16 + 4 q + r] 2^(ea + ew)
  const float d0 = __builtin_amdgcn_ldexpf(1.0f, -ea), d1 = __builtin_amdgcn_ldexpf(1.0f, -ew);
  const bool vec_c = (a.ldc % 4 == 0) && ((((uintptr_t)a.c) & 15) == 0);
  const bool vec_g = a.gate && (a.ldg % 4 == 0) && ((((uintptr_t)a.gate) & 15) == 0);
  float mx = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int col = n0 + wn * 64 + u * 16 + 4 * q;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (a.bias && col + r < a.N) ? a.bias[col + r] : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t row = m0 + wm * 64 + s * 16 + j;
      if (row >= a.M || col >= a.N || (GH_ABL == 5 && acc[u][s][0] != 123.f)) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[u][s][r] * d0 * d1 + bv[r];
        if (a.act == 1) v[r] = v[r] < 0.f ? 0.f : v[r];  // NaN stays NaN, as torch.relu
        else if (OTHER && a.act > 1) v[r] = gh_act_fwd(v[r], a.act);
      }
      if (a.gate) {
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
        const float* gp = a.gate + row * a.ldg + col;
        if (vec_g && col + 4 <= a.N) { const float4 t4 = *reinterpret_cast<const float4*>(gp); gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w; }
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (col + r < a.N) gv[r] = gp[r];
        }
        if (a.gate_act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= gv[r] > 0.f ? 1.f : 0.f;
        } else if (OTHER && a.gate_act > 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= gh_act_grad_out(gv[r], a.gate_act);
        }
      }
      float* dst = a.c + row * a.ldc + col;
      if (vec_c && col + 4 <= a.N) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(v[r]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (col + r < a.N) { dst[r] = v[r]; mx = fmaxf(mx, fabsf(v[r])); }
      }
    }
  }
  if (a.c_amax) {
    mx = gh_wave_max(mx);
    if (lane == 0) gh_amax_put(a.c_amax, blockIdx.x * (WM * WN) + wave, mx);
  }
}

template <int WM, int WN, bool OTHER> static int gh_launch_(GhArgs& g, hipStream_t st) {
  constexpr int LDS = (GH_WR * 8 * WN + (GH_AR + 2) * 8 * WM) * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_half_kernel<WM, WN, OTHER>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  g.nbm = (int)((g.M + 64 * WM - 1) / (64 * WM));
  g.nbn = (g.N + 64 * WN - 1) / (64 * WN);
  const int64_t blocks = (int64_t)g.nbm * g.nbn;
  if (blocks > 0x7fffffff) return ZK_EINVAL;
  gemm_half_kernel<WM, WN, OTHER><<<dim3((unsigned)blocks), 64 * WM * WN, LDS, st>>>(g);
  return ZK_LAUNCH_CHECK();
}
template <int WM, int WN> static int gh_launch(GhArgs& g, hipStream_t st) {
  return (g.act > 1 || (g.gate && g.gate_act > 1)) ? gh_launch_<WM, WN, true>(g, st) : gh_launch_<WM, WN, false>(g, st);
}

// ---- the glue of a coupling transform around its conditioner (zuko/transforms.py:1037-1073: split, merge) -----------------------------------
// half[d] >= 0: feature d is moved (slot half[d] of the b half); half[d] < 0: kept (slot -1 - half[d] of the a half)
struct SplitArgs { int64_t N; int D, C, na, nb; const float* x; int64_t ldx; const float* ctx; int64_t ldc; const int* idx_a; const int* idx_b; float* inp; float* xb; unsigned* amax; };
__global__ __launch_bounds__(256) void coupling_split_kernel(SplitArgs a) {
  // a block walks rows, its threads the columns of [inp | xb] (no 64-bit division per element)
  const int wi = a.na + a.C, w = wi + a.nb;
  float mx = 0.f;
  for (int64_t n = blockIdx.x; n < a.N; n += gridDim.x) {
    const float* xr = a.x + n * a.ldx;
    for (int col = threadIdx.x; col < w; col += 256) {
      if (col < a.na) { const float v = xr[a.idx_a[col]]; a.inp[n * wi + col] = v; mx = fmaxf(mx, fabsf(v)); }
      else if (col < wi) { const float v = a.ctx[n * a.ldc + (col - a.na)]; a.inp[n * wi + col] = v; mx = fmaxf(mx, fabsf(v)); }
      else a.xb[n * a.nb + (col - wi)] = xr[a.idx_b[col - wi]];
    }
  }
  mx = gh_wave_max(mx);
  if ((threadIdx.x & 63) == 0) gh_amax_put(a.amax, blockIdx.x * 4 + (threadIdx.x >> 6), mx);
}
// out[n, d] = half[d] >= 0 ? b[n, half[d]] : base[n, d] (+ add[n, -1 - half[d]] when add != null); base == null reads as zero
struct MergeArgs { int64_t N; int D; const float* base; int64_t ldbase; const float* b; int nb; const float* add; int64_t ldadd; const int* half; float* out; };
__global__ __launch_bounds__(256) void coupling_merge_kernel(MergeArgs a) {
  for (int d = threadIdx.x; d < a.D; d += 256) {
    const int h = a.half[d];
    for (int64_t n = blockIdx.x; n < a.N; n += gridDim.x) {
      float v;
      if (h >= 0) v = a.b[n * a.nb + h];
      else {
        v = a.base ? a.base[n * a.ldbase + d] : 0.f;
        if (a.add) v += a.add[n * a.ldadd + (-1 - h)];
      }
      a.out[n * a.D + d] = v;
    }
  }
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_amax_f32(int n, const zk_amax_desc_v1* descs, void* stream) {
  if (n < 0 || (n > 0 && !descs)) return ZK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += 8) {
    AmMulti m{};
    const int cnt = n - i0 < 8 ? n - i0 : 8;
    int64_t most = 0;
    for (int i = 0; i < cnt; ++i) {
      const zk_amax_desc_v1& d = descs[i0 + i];
      if (d.struct_size != sizeof(zk_amax_desc_v1) || d.rows < 0 || d.cols < 0 || !d.out || ((d.rows && d.cols) && !d.src)) return ZK_EINVAL;
      m.it[i] = AmItem{(const float*)d.src, d.rows, d.cols, d.ld, (unsigned*)d.out};
      const int64_t tot = d.rows * d.cols;
      most = tot > most ? tot : most;
    }
    if (most == 0) continue;
    const int64_t blocks = (most / 4 + 255) / 256;
    const int gx = (int)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
    amax_kernel<<<dim3(gx, cnt), 256, 0, st>>>(m);
    const int e = ZK_LAUNCH_CHECK();
    if (e) return e;
  }
  return 0;
}

int zk_wsplit_f16(int n, const zk_wsplit_desc_v1* descs, void* stream) {
  if (n < 0 || (n > 0 && !descs)) return ZK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += 8) {
    WsMulti m{};
    const int cnt = n - i0 < 8 ? n - i0 : 8;
    int64_t most = 0;
    for (int i = 0; i < cnt; ++i) {
      const zk_wsplit_desc_v1& d = descs[i0 + i];
      if (d.struct_size != sizeof(zk_wsplit_desc_v1) || d.units <= 0 || d.k <= 0 || !d.src || !d.amax || !d.dst) return ZK_EINVAL;
      const int nks = (d.k + 31) / 32, nt = (d.units + 127) / 128;
      const int64_t slots = (int64_t)nt * nks * 8 * 64;
      m.it[i] = WsItem{(const float*)d.src, d.mask, (const unsigned*)d.amax, (uint4*)d.dst, d.units, d.k, d.unit_stride, d.k_stride, slots, nks};
      most = slots > most ? slots : most;
    }
    const int64_t blocks = (most + 255) / 256;
    const int gx = (int)(blocks > 2048 ? 2048 : blocks);
    wsplit_kernel<<<dim3(gx, cnt), 256, 0, st>>>(m);
    const int e = ZK_LAUNCH_CHECK();
    if (e) return e;
  }
  return 0;
}

int zk_gemm_f16x2(int64_t M, int K, int N, const void* a, int64_t lda, const uint32_t* a_amax, const void* w_images, const uint32_t* w_amax, const void* bias, int act,
                  const void* gate, int64_t ldg, int gate_act, void* c, int64_t ldc, uint32_t* c_amax, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || !a_amax || !w_images || !w_amax || !c || (M > 0 && !a)) return ZK_EINVAL;
  auto known = [](int c) { return c == 0 || c == 1 || c == 2 || c == 3 || c == 6 || c == 7; };  // NONE, RELU, ELU, TANH, SIGMOID, LEAKY (derivative from the output)
  if (!known(act) || (gate && !known(gate_act))) return ZK_EINVAL;
  if (lda < K || ldc < N || (gate && ldg < N)) return ZK_EINVAL;
  if (K % 4 != 0 || lda % 4 != 0 || (((uintptr_t)a) & 15) != 0) return ZK_EINVAL;  // 16-byte pieces of a row (zk_gemm_f32_skip has no such limit)
  if (M == 0) return 0;
  GhArgs g{};
  g.M = M; g.K = K; g.N = N;
  g.a = (const float*)a; g.lda = lda; g.a_amax = a_amax;
  g.w = (const uint4*)w_images; g.w_amax = w_amax;
  g.bias = (const float*)bias; g.act = act;
  g.gate = (const float*)gate; g.ldg = ldg; g.gate_act = gate_act;
  g.c = (float*)c; g.ldc = ldc; g.c_amax = c_amax;
  g.nks = (K + 31) / 32;
  // the widest tile that still gives (nearly) every CU a workgroup: a 256-wide tile converts every activation once per 256 outputs
  const int64_t rows128 = (M + 127) / 128;
  static const int force = getenv("ZUKO_AMD_GEMM_TILE") ? atoi(getenv("ZUKO_AMD_GEMM_TILE")) : 0;  // 24 / 22 / 12: probe builds
  const int pick = force ? force : (N > 128 && rows128 * ((N + 255) / 256) >= 192 ? 24 : (rows128 * ((N + 127) / 128) >= 192 ? 22 : 12));
  if (pick == 24) return gh_launch<2, 4>(g, (hipStream_t)stream);
  if (pick == 22) return gh_launch<2, 2>(g, (hipStream_t)stream);
  return gh_launch<1, 2>(g, (hipStream_t)stream);
}

int zk_coupling_split(int64_t N, int D, int C, const void* x, int64_t ldx, const void* ctx, int64_t ldc, const int32_t* idx_a, int na, const int32_t* idx_b, int nb, void* inp,
                      void* xb, uint32_t* inp_amax, void* stream) {
  if (N < 0 || D <= 0 || C < 0 || na <= 0 || nb <= 0 || na + nb != D || !idx_a || !idx_b || !inp || !xb || !inp_amax || (N > 0 && !x) || (C > 0 && !ctx)) return ZK_EINVAL;
  if (N == 0) return 0;
  SplitArgs a{N, D, C, na, nb, (const float*)x, ldx, (const float*)ctx, ldc, idx_a, idx_b, (float*)inp, (float*)xb, inp_amax};
  coupling_split_kernel<<<dim3((unsigned)grid_for(N)), 256, 0, (hipStream_t)stream>>>(a);
  return ZK_LAUNCH_CHECK();
}

int zk_coupling_merge(int64_t N, int D, const void* base, int64_t ldbase, const void* b, int nb, const void* add, int64_t ldadd, const int32_t* half, void* out, void* stream) {
  if (N < 0 || D <= 0 || nb <= 0 || !b || !half || !out) return ZK_EINVAL;
  if (N == 0) return 0;
  MergeArgs a{N, D, (const float*)base, ldbase, (const float*)b, nb, (const float*)add, ldadd, half, (float*)out};
  coupling_merge_kernel<<<dim3((unsigned)grid_for(N)), 256, 0, (hipStream_t)stream>>>(a);
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
