// zuko_amd — bf16 conditioner layer  Y = act(X W^T + b)  with fp32 accumulation (cfg5 of BASELINE.json:
// NSF(1024, K=16, H=[1024]^3) in bf16, where one autoregressive layer is a 1024 -> 48128 GEMM).
//
// Replaces `F.linear(x, mask * weight, bias)` + activation (zuko/nn.py:217-218, :13-15) for bf16
// modules.  The caller passes the ALREADY MASKED weight (mask * W is one elementwise pass over the
// parameters, not over the batch) and, optionally, a liveness byte per 256 x 64 weight tile: tiles
// that the mask zeroes completely are neither fetched nor multiplied.
//
// Block tile 256 x 256 x 64, 8 wavefronts (2 x 4), each owning 128 x 64 of the output as 4 x 2
// v_mfma_f32_32x32x16_bf16 accumulators (128 VGPRs).  Both operands are K-major in HBM, which is what
// the MFMA wants (a lane holds 8 consecutive k of one row): tiles go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), two stages of 64 KiB, one workgroup
// barrier per stage.  LDS rows are 128 B; the 16-byte chunk c of row r is stored at slot
// c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads conflict-free for every 16-lane
// service group of CDNA4 (MI355X_MICROARCH.md, LDS table) — the swizzle is applied on the GLOBAL
// address of the DMA, whose LDS side is fixed at lane * 16.
//
// Blocks are rasterised in super-tiles (SR row panels x SC column panels of consecutive logical ids)
// so that the panels a set of co-resident blocks touches stay in L2 / MALL: a column sweep per row
// panel would stream the 98.6 MB weight of cfg5's last layer once per 256 rows.
#include <stdio.h>
#include <stdlib.h>

#include "zk_common.h"

namespace zk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_b __attribute__((ext_vector_type(16)));
typedef float f32x4_b __attribute__((ext_vector_type(4)));

#define BBM 256
#define BBN 256
#define BBK 64
#define B_STAGE_BYTES (2 * 256 * 128)  /* A panel 256 rows x 128 B + B panel 256 rows x 128 B */

struct LinBf16Args {
  int64_t N;
  int IN, OUT;
  const __bf16* x; int64_t ldx;
  const __bf16* w;          // [OUT, IN] row-major, already masked
  const uint8_t* live;      // [ceil(OUT/256)][IN/64] or null
  const __bf16* bias;       // [OUT] or null
  int act;
  __bf16* y; int64_t ldy;
  int nbx, nby, sr, sc, nsc;  // block grid and super-tile shape (nsc super-tiles per row of super-tiles)
};

__device__ __forceinline__ float act_bf(float v, int act) {
  switch (act) {
    case 1: return v < 0.f ? 0.f : v;
    case 2: return v > 0.f ? v : expm1f(v);
    case 3: return tanhf(v);
    case 4: return v / (1.f + expf(-v));
    case 5: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case 6: return 1.f / (1.f + expf(-v));
    case 7: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char lin_bf16_lds[];

__global__ __launch_bounds__(512, 2) void linear_bf16_kernel(LinBf16Args a) {
  // ---- block -> (bx, by): XCD-contiguous logical ids, then super-tile rasterisation ---------------
  const int nwg = gridDim.x;  // a multiple of 8
  const int orig = blockIdx.x;
  const int logical = (orig % 8) * (nwg / 8) + orig / 8;
  const int per_st = a.sr * a.sc;
  const int st = logical / per_st, within = logical - st * per_st;
  const int bx = (st / a.nsc) * a.sr + within / a.sc;
  const int by = (st % a.nsc) * a.sc + within % a.sc;
  if (bx >= a.nbx || by >= a.nby) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wave tile: rows [wm*128, +128), cols [wn*64, +64)
  const int64_t row0 = (int64_t)bx * BBM;
  const int col0 = by * BBN;
  const int KT = a.IN / BBK;
  const uint8_t* live = a.live ? a.live + (size_t)by * KT : nullptr;

  // ---- DMA geometry: one wave-instruction moves 8 rows x 128 B; wave w fills rows [w*32, +32) of each panel
  const int drow = lane >> 3;                       // row inside the 8-row group
  const int dslot = lane & 7;                       // LDS slot written by this lane
  const char* gA[4];
  const char* gB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + drow;         // panel row
    const int c = dslot ^ ((r >> 1) & 7);           // logical 16-byte chunk that lives in this slot
    int64_t ra = row0 + r;
    ra = ra < a.N ? ra : a.N - 1;                   // ragged edge: re-read the last row (results discarded)
    int rb = col0 + r;
    rb = rb < a.OUT ? rb : a.OUT - 1;
    gA[i] = reinterpret_cast<const char*>(a.x + ra * a.ldx) + c * 16;
    gB[i] = reinterpret_cast<const char*>(a.w + (int64_t)rb * a.IN) + c * 16;
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* sA = lin_bf16_lds + stage * B_STAGE_BYTES + (wave * 32) * 128;
    unsigned char* sB = sA + 256 * 128;
    const int64_t koff = (int64_t)kt * (BBK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[i] + koff),
                                       (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB[i] + koff),
                                       (__attribute__((address_space(3))) void*)(sB + i * 1024), 16, 0, 0);
    }
  };
  // liveness of this column panel's k-tiles as a 64-bit wave-uniform mask (one byte load per lane, once);
  // more than 64 k-tiles (IN > 4096): every tile is treated as live
  unsigned long long lmask = ~0ull;
  if (live && KT <= 64) lmask = __builtin_amdgcn_ballot_w64(lane < KT && live[lane] != 0);
  else if (KT < 64) lmask = (1ull << KT) - 1;
  auto next_live = [&](int kt) {  // first live k-tile >= kt (KT if none)
    if (kt >= 64) return kt < KT ? kt : KT;
    const unsigned long long m = lmask >> kt;
    if (KT > 64) return kt;
    return m ? kt + (int)__builtin_ctzll(m) : KT;
  };
  // ---- fragment read offsets (bytes inside a panel), per 32-row sub-tile and k16 step -------------
  const int fr = lane & 31, kg = lane >> 5;
  f32x16_b acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int kt = next_live(0);
  int stage = 0;
  if (kt < KT) issue(kt, 0);
  while (kt < KT) {
    const int ktn = next_live(kt + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my DMAs of this stage have landed
    __syncthreads();                                   // everybody's have; the other stage is free again
    if (ktn < KT) issue(ktn, stage ^ 1);
    const unsigned char* sA = lin_bf16_lds + stage * B_STAGE_BYTES + (wm * 128) * 128;
    const unsigned char* sB = lin_bf16_lds + stage * B_STAGE_BYTES + 256 * 128 + (wn * 64) * 128;
    // fragments of k16 step kk+1 are requested before the MFMAs of step kk (register double buffer)
    bf16x8 fa[2][4], fb[2][2];
#define ZK_BF16_FRAGS(buf, kk)                                                                              \
  {                                                                                                         \
    const int c_ = (kk) * 2 + kg;                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      const int r_ = i * 32 + fr;                                                                           \
      fa[buf][i] = *reinterpret_cast<const bf16x8*>(sA + r_ * 128 + ((c_ ^ ((r_ >> 1) & 7)) << 4));        \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
      const int r_ = j * 32 + fr;                                                                           \
      fb[buf][j] = *reinterpret_cast<const bf16x8*>(sB + r_ * 128 + ((c_ ^ ((r_ >> 1) & 7)) << 4));        \
    }                                                                                                       \
  }
    ZK_BF16_FRAGS(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) ZK_BF16_FRAGS((kk + 1) & 1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs (the scheduler would sink it to its uses)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef ZK_BF16_FRAGS
    stage ^= 1;
    kt = ktn;
  }

  // ---- epilogue: + bias, activation, bf16, store.  acc[i][j][r]: row (r/4)*8 + kg*4 + r%4, col fr ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + fr;
    const bool cok = col < a.OUT;
    const float bv = (a.bias && cok) ? (float)a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wm * 128 + i * 32 + (r >> 2) * 8 + kg * 4 + (r & 3);
        if (cok && row < a.N) a.y[row * a.ldy + col] = (__bf16)act_bf(acc[i][j][r] + bv, a.act);
      }
    }
  }
}

}  // namespace zk

using namespace zk;

// super-tile shape override for experiments: ZUKO_AMD_BF16_ST="SRxSC"
static void bf16_supertile(int nbx, int nby, int& sr, int& sc) {
  sr = 8; sc = 8;
  const char* e = getenv("ZUKO_AMD_BF16_ST");
  if (e) { int a = 0, b = 0; if (sscanf(e, "%dx%d", &a, &b) == 2 && a > 0 && b > 0) { sr = a; sc = b; } }
  if (sr > nbx) sr = nbx;
  if (sc > nby) sc = nby;
}

extern "C" int zk_linear_bf16(int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* weight, const uint8_t* tile_live,
                              const void* bias, int act, void* y, int64_t ldy, void* stream) {
  if (N <= 0 || out_features <= 0) return 0;
  if (in_features <= 0 || in_features % BBK != 0 || act < 0 || act > 7) return ZK_EINVAL;
  if (ldx % 8 != 0 || (((uintptr_t)x | (uintptr_t)weight) & 15) != 0) return ZK_EINVAL;  // 16-byte DMA granules
  LinBf16Args a{};
  a.N = N; a.IN = in_features; a.OUT = out_features;
  a.x = (const __bf16*)x; a.ldx = ldx; a.w = (const __bf16*)weight; a.live = tile_live; a.bias = (const __bf16*)bias; a.act = act;
  a.y = (__bf16*)y; a.ldy = ldy;
  a.nbx = (int)((N + BBM - 1) / BBM);
  a.nby = (out_features + BBN - 1) / BBN;
  bf16_supertile(a.nbx, a.nby, a.sr, a.sc);
  a.nsc = (a.nby + a.sc - 1) / a.sc;
  const int64_t nst = (int64_t)((a.nbx + a.sr - 1) / a.sr) * a.nsc;
  int64_t grid = nst * a.sr * a.sc;
  grid = (grid + 7) / 8 * 8;
  if (grid > 0x7fffffff) return ZK_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)linear_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * B_STAGE_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(linear_bf16_kernel, dim3((unsigned)grid), dim3(512), 2 * B_STAGE_BYTES, (hipStream_t)stream, a);
  return ZK_LAUNCH_CHECK();
}
