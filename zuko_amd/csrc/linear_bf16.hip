// zuko_amd — bf16 conditioner layer  Y = act(X W^T + b)  with fp32 accumulation (cfg5 of BASELINE.json:
// NSF(1024, K=16, H=[1024]^3) in bf16, where one autoregressive layer is a 1024 -> 48128 GEMM).
//
// Replaces `F.linear(x, mask * weight, bias)` + activation (zuko/nn.py:217-218, :13-15) for bf16
// modules.  The caller passes the ALREADY MASKED weight (mask * W is one elementwise pass over the
// parameters, not over the batch) and, optionally, a liveness bit per 256 x 64 weight tile (one 64-bit word
// per panel of 256 outputs): tiles that the mask zeroes completely are neither fetched nor multiplied.
//
// Block tile 256 x 256 x 64, 8 wavefronts (2 x 4), each owning 128 x 64 of the output as 4 x 2
// v_mfma_f32_32x32x16_bf16 accumulators (128 VGPRs).  Both operands are K-major in HBM, which is what
// the MFMA wants (a lane holds 8 consecutive k of one row): tiles go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), two stages of 64 KiB, one workgroup
// barrier per stage; the second stage doubles as the epilogue's transpose image.  LDS rows are 128 B; the 16-byte chunk c of row r is stored at slot
// c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads conflict-free for every 16-lane
// service group of CDNA4 (MI355X_MICROARCH.md, LDS table) — the swizzle is applied on the GLOBAL
// address of the DMA, whose LDS side is fixed at lane * 16.
//
// One persistent block per CU walks the tiles in super-tile raster order (SR row panels x SC column
// panels of consecutive tile ids), so the panels the co-resident blocks touch stay in L2 / MALL: a
// column sweep per row panel would stream the 98.6 MB weight of cfg5's last layer once per 256 rows.
#include <stdio.h>
#include <stdlib.h>

#include "zk_univariate.h"

namespace zk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_b __attribute__((ext_vector_type(16)));

#define BBM 256
#define BBN 256
#define BBK 64
// measured variants of the k loop / epilogue (scripts/cfg5_variants.sh builds and times each; the defaults are what won)
#ifndef ZK_BF16_SPREAD
#define ZK_BF16_SPREAD 1  /* 1: the eight DMAs of the next k-tile are issued two at a time between the MFMA blocks instead of all after the barrier
                             (measured at cfg5's last layer, N = 2^19: 38.9 -> 35.6 ms; hidden layers 1.03 -> 0.95 ms) */
#endif
#ifndef ZK_BF16_XPRE
#define ZK_BF16_XPRE 0    /* 1: the x values of a thread's (sample, feature) elements are requested before the epilogue's first image write, not inside each round */
#endif
#ifndef ZK_BF16_ABLATE
#define ZK_BF16_ABLATE 0  /* probe builds only (WRONG results; scripts/cfg5_variants.sh): 1 = no spline epilogue, 2 = epilogue without the spline arithmetic,
                             3 = k loop without MFMAs, 4 = k loop without the stage DMAs, 5 = epilogue without the x loads / y stores, 6 = epilogue without
                             the image writes, 7 = k loop with fragment reads only, 8 = with MFMAs only, 9 = with DMAs only — what each part costs a tile */
#endif
#ifndef ZK_BF16_DMA1
#define ZK_BF16_DMA1 0    /* (with ZK_BF16_ROT) 1: one DMA in front of every MFMA block and one in its middle, instead of two in front */
#endif
#ifndef ZK_BF16_ROT
#define ZK_BF16_ROT 0     /* 1: rotated k loop — fragment reads from inline assembly with counted lgkmcnt waits, and the last MFMA block of k-tile t
                             issued AFTER the barrier of k-tile t + 1, behind that tile's first fragment reads (covers their latency) */
#endif
#ifndef ZK_BF16_PRIO
#define ZK_BF16_PRIO 0    /* 1: s_setprio 1 around every MFMA block */
#endif
#ifndef ZK_BF16_EPI3
#define ZK_BF16_EPI3 0    /* 1: spline epilogue in chunks of 96 / 96 / 64 samples (three rounds of up to 480 elements) instead of 2 x 128 (four rounds) */
#endif
#define B_STAGE_BYTES (2 * 256 * 128)  /* A panel 256 rows x 128 B + B panel 256 rows x 128 B */

struct LinBf16Args {
  int64_t N;
  int IN, OUT;
  const __bf16* x; int64_t ldx;
  const __bf16* w;          // [OUT, IN] row-major, already masked
  const unsigned long long* live;  // [ceil(OUT/256)]: bit k set = k-tile k (64 inputs) of the panel has non-zero weights; or null
  const __bf16* bias;       // [OUT] or null
  int act;
  __bf16* y; int64_t ldy;
  int nbx, nby, sr, sc, nsc;  // tile grid and super-tile shape (nsc super-tiles per row of super-tiles)
  int dbg;                    // ZUKO_AMD_BF16_DEBUG (ablations: 1 = no result stores, 2 = no epilogue at all)
  int ntiles;                 // raster length (super-tiles are padded: out-of-range tiles are filtered on the host)
  int map, pr, pc, xr, xc;    // map = 1: XCD-aware walk (see locate()): regions of (xr pr) x (xc pc) tiles, one pr x pc patch per XCD
  // spline epilogue (SK > 0): the layer's outputs never leave the CU as phi
  const __bf16* sx; int64_t ldsx;   // transform input  x[N, D]
  __bf16* sy; int64_t ldsy;         // transform output y[N, D]
  float* partial;                   // [panels][N]: per-panel sums of log|dy/dx|
  int D;                            // features; panel p holds features [p * FP, p * FP + FP)
  RqsLeanConst lc;
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

#define B_EPI_ROWB 144                       /* epilogue image: 64 cols of bf16 per row + 16 B pad */
#define B_EPI_WAVE (64 * B_EPI_ROWB)         /* one wave's half sub-tile: 64 rows */
#define B_LDS_BYTES (2 * B_STAGE_BYTES + 8 * B_EPI_WAVE - B_STAGE_BYTES)  /* stage 0 | stage 1 ∪ epilogue image */
#define S_ROWB 520                           /* spline image: 256 bf16 per sample + 8 B pad (130 dwords: 2-way conflicts at most) */
#define S_LDS_BYTES (B_STAGE_BYTES + 128 * S_ROWB + 11 * 128 * 4)  /* stage 0 | image of 128 samples | per-feature log-derivatives */

typedef unsigned int u32x2_b __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_b __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// Persistent: block b walks tiles b, b + G, b + 2G, ... of the super-tile raster.  The MFMA computes the
// TRANSPOSED product (A operand = weight rows, B operand = activation rows), so a lane ends up with four
// consecutive output columns of one sample per accumulator quad: bias + activation + bf16 packing happen
// on 8-byte groups, the wave transposes its 128 x 64 sub-tile through a private LDS image and leaves
// as full 128-byte row segments (dwordx4 per lane).  Between a tile's last MFMA and its epilogue the
// first k-stage of the NEXT tile is already requested, so its DMA latency hides behind the stores.
// GENERIC_ACT: activations other than none / ReLU are applied on the transposed bf16 image inside a rolled loop
// (their inline expansions, unrolled over the 128 accumulators of a lane, made the epilogue instruction-cache bound)
// SK > 0: spline epilogue.  The weight rows are laid out in PANELS of 256 that hold the 3 SK - 1 spline parameters
// of FP = 256 / (3 SK - 1) whole features (rows past FP (3 SK - 1) are zero padding), so a tile owns every
// parameter of FP features for 256 samples: bias-added outputs go to an LDS image as bf16 (the rounding the
// unfused path applies when it writes phi), one thread per (sample, feature) evaluates rqs_lean on them, y
// leaves as bf16 and the FP log-derivatives of a sample are summed into partial[panel][sample].
template <bool GENERIC_ACT, int SK> __global__ __launch_bounds__(512, 2) void linear_bf16_kernel(LinBf16Args a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wave tile: samples [wm*128, +128), outputs [wn*64, +64)
  const int KT = a.IN / BBK;
  const int ntiles = a.ntiles;
  const int drow = lane >> 3, dslot = lane & 7;   // DMA: row inside an 8-row group, LDS slot of this lane
  const int fr = lane & 31, kg = lane >> 5;

  // tile t of the super-tile raster -> (row panel, column panel); super-tiles at the right / bottom edge
  // are clipped, so the raster enumerates exactly nbx * nby tiles
  auto raster = [&](int t, int& bx, int& by) {
    const int row_tiles = a.sr * a.nby;
    const int R = t / row_tiles, rem = t - R * row_tiles;
    const int hr = (a.nbx - R * a.sr) < a.sr ? (a.nbx - R * a.sr) : a.sr;
    const int C = rem / (hr * a.sc), rem2 = rem - C * hr * a.sc;
    const int wc = (a.nby - C * a.sc) < a.sc ? (a.nby - C * a.sc) : a.sc;
    bx = R * a.sr + rem2 / wc;
    by = C * a.sc + rem2 % wc;
  };
  // XCD-aware walk (map = 1).  Workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed only), and every
  // XCD has its own L2.  Round r of the persistent grid covers one REGION of (xr pr) x (xc pc) tiles, of which XCD x owns
  // the compact pr x pc patch (x / xc, x % xc): its 32 co-resident blocks then share pr activation panels and pc weight
  // panels per k-step (12 x 32 KiB for 32 tiles at 4 x 8) instead of one weight panel and 32 activation panels, which
  // is what the id-order raster gives an XCD (its L2 hit rate on operand fetches goes from ~44 % to ~81 %).  Regions are
  // walked row-region-major, so the whole weight matrix (105 MB at cfg5, MALL-resident) is swept once per row region.
  // Ids whose tile falls outside the grid (ragged edges) are skipped.
  auto locate = [&](int t, int& bx, int& by) -> bool {
    if (a.map == 0) { raster(t, bx, by); return true; }
    const int G = (int)gridDim.x;
    const int r = t / G, b = t - r * G;
    const int xcd = b & 7, slot = b >> 3;
    const int RR = a.xr * a.pr, RC = a.xc * a.pc;
    const int nRC = (a.nby + RC - 1) / RC;
    const int Rrow = r / nRC, Rcol = r - Rrow * nRC;
    bx = Rrow * RR + (xcd / a.xc) * a.pr + slot / a.pc;
    by = Rcol * RC + (xcd % a.xc) * a.pc + slot % a.pc;
    return bx < a.nbx && by < a.nby;
  };
  auto first_valid = [&](int t, int& bx, int& by) -> int {
    while (t < ntiles && !locate(t, bx, by)) t += (int)gridDim.x;
    return t;
  };
  // liveness of a column panel's k-tiles: one 64-bit word per panel (all live without a table or beyond 64 k-tiles)
  auto load_live = [&](int by) -> unsigned long long { return (a.live && KT <= 64) ? a.live[by] : ~0ull; };
  auto mask_of = [&](unsigned long long v) -> unsigned long long {
    unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    if (KT < 64) m &= (1ull << KT) - 1;
    return m;
  };
  auto next_live = [&](unsigned long long lmask, int kt) {  // first live k-tile >= kt (KT if none)
    if (KT > 64) return kt < KT ? kt : KT;
    if (kt >= 64) return KT;
    const unsigned long long m = lmask >> kt;
    return m ? kt + (int)__builtin_ctzll(m) : KT;
  };
  const char* gA[4];
  const char* gB[4];
  auto set_panels = [&](int bx, int by) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + i * 8 + drow;       // panel row
      const int c = dslot ^ ((r >> 1) & 7);         // logical 16-byte chunk that lives in this lane's slot
      int64_t ra = (int64_t)bx * BBM + r;
      ra = ra < a.N ? ra : a.N - 1;                 // ragged edge: re-read the last row (results discarded)
      int rb = by * BBN + r;
      rb = rb < a.OUT ? rb : a.OUT - 1;
      gA[i] = reinterpret_cast<const char*>(a.x + ra * a.ldx) + c * 16;
      gB[i] = reinterpret_cast<const char*>(a.w + (int64_t)rb * a.IN) + c * 16;
    }
  };
  auto issue_pair = [&](int kt, int stage, int i) {  // one quarter of a wave's share of a stage: 8 rows of the A panel, 8 of the B panel
    unsigned char* sA = lin_bf16_lds + stage * B_STAGE_BYTES + (wave * 32) * 128;
    unsigned char* sB = sA + 256 * 128;
    const int64_t koff = (int64_t)kt * (BBK * 2);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[i] + koff), (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB[i] + koff), (__attribute__((address_space(3))) void*)(sB + i * 1024), 16, 0, 0);
  };
  auto issue_one = [&](int kt, int stage, int i, int which) {  // which = 0: 8 rows of the A panel, 1: of the B panel
    unsigned char* sA = lin_bf16_lds + stage * B_STAGE_BYTES + (wave * 32) * 128;
    const int64_t koff = (int64_t)kt * (BBK * 2);
    if (which == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[i] + koff), (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB[i] + koff), (__attribute__((address_space(3))) void*)(sA + 256 * 128 + i * 1024), 16, 0, 0);
  };
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

  int bx = 0, by = 0;
  int tile = first_valid((int)blockIdx.x, bx, by);
  if (tile >= ntiles) return;
  set_panels(bx, by);
  unsigned long long lmask = mask_of(load_live(by));
  int kt = next_live(lmask, 0);
  if (kt < KT) issue(kt, 0);

  while (true) {
    // ---- what the epilogue and the next tile will need, requested up front ------------------------
    int bxn = 0, byn = 0;
    const int tile_n = first_valid(tile + (int)gridDim.x, bxn, byn);
    const unsigned long long live_n = (tile_n < ntiles) ? load_live(byn) : 0ull;
    f32x16_b acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#if ZK_BF16_ROT
    // ---- k loop, rotated: two LDS stages, one barrier per live k-tile --------------------------------------------------------
    // Per k-tile a wave reads 4 x 6 fragments (ds_read_b128) and issues 4 x 8 MFMAs.  The reads are issued from inline assembly and
    // return RAW registers (the compiler does not know they are LDS operations and inserts no wait; with LDS-DMAs in flight every
    // wait it inserts itself is lgkmcnt(0), which serialises the prefetched fragments with the MFMAs that should cover them); a
    // block becomes usable through frag_settle<N>() = s_waitcnt lgkmcnt(N), N = the younger reads allowed to stay in flight (LDS
    // operations of a wave complete in order).  The loop is rotated by one block: MFMA block 3 of k-tile t is issued after the
    // barrier of k-tile t + 1, right behind that tile's first six reads, whose latency it covers.
    int stage = 0;
    bf16x8 fa[2][4], fb[2][2];
    bool pending = false;  // MFMA block 3 of the previous k-tile (operands in buffer 1) not issued yet
    const unsigned lane_row = (unsigned)fr * 128u, swz = (unsigned)((fr >> 1) & 7);
    unsigned offA[4], offB[4];  // LDS byte address of this lane's fragment chunk for kk = 0..3 (stage 0); rows i * 32 ride the immediate offset
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const unsigned ch = (((unsigned)(kk * 2 + kg)) ^ swz) << 4;
      const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lin_bf16_lds);
      offA[kk] = base + (unsigned)(wm * 128) * 128u + lane_row + ch;
      offB[kk] = base + 256u * 128u + (unsigned)(wn * 64) * 128u + lane_row + ch;
    }
#define ZK_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define ZK_FRAGS_RAW(buf, kk)                                                     \
  {                                                                               \
    const unsigned a_ = offA[kk] + (unsigned)stage * B_STAGE_BYTES, b_ = offB[kk] + (unsigned)stage * B_STAGE_BYTES; \
    ZK_RD(fa[buf][0], a_, 0); ZK_RD(fa[buf][1], a_, 4096); ZK_RD(fa[buf][2], a_, 8192); ZK_RD(fa[buf][3], a_, 12288); \
    ZK_RD(fb[buf][0], b_, 0); ZK_RD(fb[buf][1], b_, 4096);                        \
  }
#define ZK_SETTLE(buf, n)                                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]), "+v"(fa[buf][3]), "+v"(fb[buf][0]), "+v"(fb[buf][1]) : "n"(n)); \
  __builtin_amdgcn_sched_barrier(0)
#define ZK_MFMA_BLOCK(buf)                                                                                                        \
  {                                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][j], fa[buf][i], acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
  }
#if ZK_BF16_DMA1
#define ZK_DMA_FRONT(q) if (ktn < KT) issue_one(ktn, stage ^ 1, q, 0)
#define ZK_MFMA_BLOCK_Q(buf, q)                                                                                                   \
  {                                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][j], fa[buf][i], acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    if (ktn < KT) issue_one(ktn, stage ^ 1, q, 1);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    _Pragma("unroll") for (int i = 2; i < 4; ++i)                                                                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][j], fa[buf][i], acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
  }
#else
#define ZK_DMA_FRONT(q) if (ktn < KT) issue_pair(ktn, stage ^ 1, q)
#define ZK_MFMA_BLOCK_Q(buf, q) ZK_MFMA_BLOCK(buf)
#endif
    while (kt < KT) {
      const int ktn = next_live(lmask, kt + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my DMAs of this stage have landed (and my stores of the previous tile left)
      __syncthreads();                                   // everybody's have; the other stage / the epilogue image is free again
      ZK_FRAGS_RAW(0, 0);
      ZK_DMA_FRONT(0);
      __builtin_amdgcn_sched_barrier(0);
      if (pending) { ZK_MFMA_BLOCK_Q(1, 0); }            // block 3 of the previous k-tile (settled before the barrier) covers the reads above
#if ZK_BF16_DMA1
      else if (ktn < KT) issue_one(ktn, stage ^ 1, 0, 1);
#endif
      ZK_FRAGS_RAW(1, 1);
      ZK_DMA_FRONT(1);
      ZK_SETTLE(0, 6);
      ZK_MFMA_BLOCK_Q(0, 1);
      ZK_FRAGS_RAW(0, 2);
      ZK_DMA_FRONT(2);
      ZK_SETTLE(1, 6);
      ZK_MFMA_BLOCK_Q(1, 2);
      ZK_FRAGS_RAW(1, 3);
      ZK_DMA_FRONT(3);
      ZK_SETTLE(0, 6);
      ZK_MFMA_BLOCK_Q(0, 3);
      ZK_SETTLE(1, 0);                                   // block 3's operands are in registers: the stage may be overwritten after the next barrier
      pending = true;
      stage ^= 1;
      kt = ktn;
    }
    if (pending) ZK_MFMA_BLOCK(1);
#undef ZK_DMA_FRONT
#undef ZK_MFMA_BLOCK_Q
#undef ZK_RD
#undef ZK_FRAGS_RAW
#undef ZK_SETTLE
#undef ZK_MFMA_BLOCK
#else
    // ---- k loop: two LDS stages, one barrier per live k-tile ---------------------------------------
    int stage = 0;
    while (kt < KT) {
      const int ktn = next_live(lmask, kt + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my DMAs of this stage have landed (and my stores of the previous tile left)
      __syncthreads();                                   // everybody's have; the other stage / the epilogue image is free again
#if !ZK_BF16_SPREAD
      if (ktn < KT) issue(ktn, stage ^ 1);
#endif
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
#if ZK_BF16_ABLATE == 8 || ZK_BF16_ABLATE == 9
#pragma unroll
      for (int b_ = 0; b_ < 2; ++b_) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[b_][i] = bf16x8{}; asm volatile("" : "+v"(fa[b_][i])); }  // (opaque zeros: the MFMAs stay)
#pragma unroll
        for (int j = 0; j < 2; ++j) { fb[b_][j] = bf16x8{}; asm volatile("" : "+v"(fb[b_][j])); }
      }
#else
      ZK_BF16_FRAGS(0, 0);
#endif
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#if !(ZK_BF16_ABLATE == 8 || ZK_BF16_ABLATE == 9)
        if (kk < 3) ZK_BF16_FRAGS((kk + 1) & 1, kk + 1);
#endif
#if ZK_BF16_SPREAD
        // a wave that issues a DMA issues nothing else for ~100 cycles: two at a time, in front of its own MFMA block, so that the
        // partner wave of the SIMD has the matrix pipe meanwhile (all eight right after the barrier stall both waves at once)
        if (ZK_BF16_ABLATE != 4 && ZK_BF16_ABLATE != 7 && ZK_BF16_ABLATE != 8 && ktn < KT) issue_pair(ktn, stage ^ 1, kk);
#endif
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs (the scheduler would sink it to its uses)
#if ZK_BF16_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#if ZK_BF16_ABLATE == 3 || ZK_BF16_ABLATE == 7
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[kk & 1][i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb[kk & 1][j]));
#elif ZK_BF16_ABLATE == 9
        (void)0;
#else
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
#endif
#if ZK_BF16_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#undef ZK_BF16_FRAGS
      stage ^= 1;
      kt = ktn;
    }

#endif
    // ---- hand-over: the next tile's first stage goes out before this tile's epilogue ----------------
    __syncthreads();  // every wave is done reading the stage buffers
    const int bx_c = bx, by_c = by;
    unsigned long long lmask_n = 0;
    int kt_n = KT;
    if (tile_n < ntiles) {
      set_panels(bxn, byn);
      lmask_n = mask_of(live_n);
      kt_n = next_live(lmask_n, 0);
      if (kt_n < KT) issue(kt_n, 0);
    }

    u32x2_b bias4[2][4];  // bias of outputs j*32 + q*8 + kg*4 .. +3 as four bf16
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = by_c * BBN + wn * 64 + j * 32 + q * 8 + kg * 4;
        u32x2_b v = {0u, 0u};
        if (a.bias) {
          if (col + 4 <= a.OUT && (((uintptr_t)(a.bias + col)) & 7) == 0) v = *reinterpret_cast<const u32x2_b*>(a.bias + col);
          else {
            unsigned short e[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) e[t] = (col + t < a.OUT) ? reinterpret_cast<const unsigned short*>(a.bias)[col + t] : (unsigned short)0;
            v = u32x2_b{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
          }
        }
        bias4[j][q] = v;
      }

    // ---- epilogue.  acc[i][j][r] = C[sample i*32 + fr][output j*32 + (r/4)*8 + kg*4 + r%4] ---------------
    unsigned char* img = lin_bf16_lds + B_STAGE_BYTES + wave * B_EPI_WAVE;  // wave-private: 64 rows x 144 B
    const int64_t row_w = (int64_t)bx_c * BBM + wm * 128;
    const int col_w = by_c * BBN + wn * 64;
    const bool vec_ok = (a.ldy % 8 == 0) && ((((uintptr_t)a.y) & 15) == 0);
    const bool relu = a.act == 1;
    if constexpr (SK > 0 && ZK_BF16_ABLATE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else
    if constexpr (SK > 0) {
      constexpr int TOTAL = 3 * SK - 1, FP = 256 / TOTAL, ROUNDS = (FP + 3) / 4;
      unsigned char* simg = lin_bf16_lds + B_STAGE_BYTES;                       // [128 samples][S_ROWB]
      float* ljs = reinterpret_cast<float*>(lin_bf16_lds + B_STAGE_BYTES + 128 * S_ROWB);  // [FP][128]
      if constexpr (ZK_BF16_EPI3 != 0 && FP * 96 <= 512) {
      // chunks of 3 / 3 / 2 sample blocks of 32 (block b = wm * 4 + i): 480 / 480 / 320 (sample, feature) elements for 512 threads
#pragma unroll 1
      for (int c = 0; c < 3; ++c) {
        const int b_lo = 3 * c, b_hi = c == 2 ? 8 : 3 * c + 3, ns = (b_hi - b_lo) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int blk = wm * 4 + i;
          if (blk >= b_lo && blk < b_hi) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const u32x2_b bv = bias4[j][q];
                const float b0 = __builtin_bit_cast(float, bv.x << 16), b1 = __builtin_bit_cast(float, bv.x & 0xffff0000u);
                const float b2 = __builtin_bit_cast(float, bv.y << 16), b3 = __builtin_bit_cast(float, bv.y & 0xffff0000u);
                const u32x2_b pk = {pack_bf16x2(acc[i][j][4 * q + 0] + b0, acc[i][j][4 * q + 1] + b1),
                                    pack_bf16x2(acc[i][j][4 * q + 2] + b2, acc[i][j][4 * q + 3] + b3)};
                *reinterpret_cast<u32x2_b*>(simg + ((blk - b_lo) * 32 + fr) * S_ROWB + (wn * 64 + j * 32 + q * 8 + kg * 4) * 2) = pk;
              }
          }
        }
        __syncthreads();
        {
          const int s_ = tid % 96 < ns ? tid % 96 : 0, fl = tid / 96;  // (ns = 96 or 64; 512 threads cover 5 features x 96 samples + 32 spare)
          const bool on = fl < FP && (tid % 96) < ns;
          const int feat = by_c * FP + fl;
          const int64_t row = (int64_t)bx_c * BBM + b_lo * 32 + s_;
          if (on) {
            float lj = 0.f;
            if (feat < a.D && row < a.N) {
              float p[TOTAL];
              const unsigned short* src = reinterpret_cast<const unsigned short*>(simg + s_ * S_ROWB + fl * TOTAL * 2);
#pragma unroll
              for (int t = 0; t < TOTAL; ++t) p[t] = __builtin_bit_cast(float, (unsigned)src[t] << 16);
              const float xv = (float)a.sx[row * a.ldsx + feat];
              float yv;
              rqs_lean<SK, false>([&](int t) { return p[t]; }, [&](int t) { return p[SK + t]; }, [&](int t) { return p[2 * SK + t]; }, a.lc, xv, yv, lj);
              a.sy[row * a.ldsy + feat] = (__bf16)yv;
            }
            ljs[fl * 128 + s_] = lj;
          }
        }
        __syncthreads();
        if (tid < ns) {
          const int64_t row = (int64_t)bx_c * BBM + b_lo * 32 + tid;
          float sum = 0.f;
#pragma unroll
          for (int fl = 0; fl < FP; ++fl) sum += ljs[fl * 128 + tid];
          if (row < a.N) a.partial[(size_t)by_c * a.N + row] = sum;
        }
      }
      } else {
#if ZK_BF16_XPRE
      // x of every element this thread will evaluate (2 halves x ROUNDS rounds): scattered 2-byte loads whose latency would otherwise sit
      // inside each round with two waves per SIMD to cover it — requested here, they land during the image writes
      float xp[2 * ROUNDS];
#pragma unroll
      for (int k = 0; k < 2 * ROUNDS; ++k) {
        const int id = tid + 512 * (k % ROUNDS);
        const int s_ = id & 127, fl = id >> 7, feat = by_c * FP + fl;
        const int64_t row = (int64_t)bx_c * BBM + (k / ROUNDS) * 128 + s_;
        xp[k] = (fl < FP && feat < a.D && row < a.N) ? (float)a.sx[row * a.ldsx + feat] : 0.f;
      }
#endif
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        if (ZK_BF16_ABLATE == 6) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else
        if (wm == h) {  // the four waves that hold samples [128 h, 128 h + 128) of the tile
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const u32x2_b bv = bias4[j][q];
                const float b0 = __builtin_bit_cast(float, bv.x << 16), b1 = __builtin_bit_cast(float, bv.x & 0xffff0000u);
                const float b2 = __builtin_bit_cast(float, bv.y << 16), b3 = __builtin_bit_cast(float, bv.y & 0xffff0000u);
                const u32x2_b pk = {pack_bf16x2(acc[i][j][4 * q + 0] + b0, acc[i][j][4 * q + 1] + b1),
                                    pack_bf16x2(acc[i][j][4 * q + 2] + b2, acc[i][j][4 * q + 3] + b3)};
                *reinterpret_cast<u32x2_b*>(simg + (i * 32 + fr) * S_ROWB + (wn * 64 + j * 32 + q * 8 + kg * 4) * 2) = pk;
              }
        }
        __syncthreads();
#pragma unroll 1
        for (int rd = 0; rd < ROUNDS; ++rd) {
          const int id = tid + 512 * rd;
          const int s_ = id & 127, fl = id >> 7;
          const int feat = by_c * FP + fl;
          const int64_t row = (int64_t)bx_c * BBM + h * 128 + s_;
          if (fl < FP) {
            float lj = 0.f;
            if (feat < a.D && row < a.N) {
              float p[TOTAL];
              const unsigned short* src = reinterpret_cast<const unsigned short*>(simg + s_ * S_ROWB + fl * TOTAL * 2);
#pragma unroll
              for (int t = 0; t < TOTAL; ++t) p[t] = __builtin_bit_cast(float, (unsigned)src[t] << 16);
#if ZK_BF16_XPRE
              float xv = 0.f;
#pragma unroll
              for (int k = 0; k < 2 * ROUNDS; ++k) xv = (k == h * ROUNDS + rd) ? xp[k] : xv;
#elif ZK_BF16_ABLATE == 5
              const float xv = 0.25f;
#else
              const float xv = (float)a.sx[row * a.ldsx + feat];
#endif
              float yv;
#if ZK_BF16_ABLATE == 2
              yv = xv; lj = 0.f;
#pragma unroll
              for (int t = 0; t < TOTAL; ++t) { yv += p[t]; lj += p[t] * 0.5f; }
#else
              rqs_lean<SK, false>([&](int t) { return p[t]; }, [&](int t) { return p[SK + t]; }, [&](int t) { return p[2 * SK + t]; }, a.lc, xv, yv, lj);
#endif
#if ZK_BF16_ABLATE == 5
              lj += yv;
#else
              a.sy[row * a.ldsy + feat] = (__bf16)yv;
#endif
            }
            ljs[fl * 128 + s_] = lj;
          }
        }
        __syncthreads();
        if (tid < 128) {
          const int64_t row = (int64_t)bx_c * BBM + h * 128 + tid;
          float sum = 0.f;
#pragma unroll
          for (int fl = 0; fl < FP; ++fl) sum += ljs[fl * 128 + tid];
          if (row < a.N) a.partial[(size_t)by_c * a.N + row] = sum;
        }
      }
      }
    } else
    if (!(a.dbg & 2))
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = h * 2 + ii;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x2_b bv = bias4[j][q];
            const float b0 = __builtin_bit_cast(float, bv.x << 16), b1 = __builtin_bit_cast(float, bv.x & 0xffff0000u);
            const float b2 = __builtin_bit_cast(float, bv.y << 16), b3 = __builtin_bit_cast(float, bv.y & 0xffff0000u);
            float v0 = acc[i][j][4 * q + 0] + b0, v1 = acc[i][j][4 * q + 1] + b1, v2 = acc[i][j][4 * q + 2] + b2, v3 = acc[i][j][4 * q + 3] + b3;
            if (!GENERIC_ACT && relu) {  // NaN stays NaN, as torch.relu
              v0 = v0 < 0.f ? 0.f : v0; v1 = v1 < 0.f ? 0.f : v1; v2 = v2 < 0.f ? 0.f : v2; v3 = v3 < 0.f ? 0.f : v3;
            }
            const u32x2_b pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
            *reinterpret_cast<u32x2_b*>(img + (ii * 32 + fr) * B_EPI_ROWB + (j * 32 + q * 8 + kg * 4) * 2) = pk;
          }
      }
      // 64 rows x 128 B leave as 8 wave-stores of 8 rows: lane -> (row it*8 + lane/8, 16-byte chunk lane%8)
#pragma unroll GENERIC_ACT ? 1 : 8
      for (int it = 0; it < 8; ++it) {
        const int rl = it * 8 + drow;
        u32x4_b v = *reinterpret_cast<const u32x4_b*>(img + rl * B_EPI_ROWB + dslot * 16);
        if (GENERIC_ACT) {  // (the reference also rounds the linear output to bf16 before its activation module)
          unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float lo = act_bf(__builtin_bit_cast(float, w4[t] << 16), a.act), hi = act_bf(__builtin_bit_cast(float, w4[t] & 0xffff0000u), a.act);
            w4[t] = pack_bf16x2(lo, hi);
          }
          v = u32x4_b{w4[0], w4[1], w4[2], w4[3]};
        }
        const int64_t row = row_w + h * 64 + rl;
        const int col = col_w + dslot * 8;
        if (row < a.N && !(a.dbg & 1)) {
          __bf16* dst = a.y + row * a.ldy + col;
          if (vec_ok && col + 8 <= a.OUT) *reinterpret_cast<u32x4_b*>(dst) = v;
          else {
            const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 8; ++t)
              if (col + t < a.OUT) reinterpret_cast<unsigned short*>(dst)[t] = (unsigned short)(w4[t >> 1] >> ((t & 1) * 16));
          }
        }
      }
    }

    if (tile_n >= ntiles) break;
    tile = tile_n; bx = bxn; by = byn; lmask = lmask_n; kt = kt_n;
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

static int launch_linear_bf16(LinBf16Args a, int spline_k, hipStream_t stream) {
  a.nbx = (int)((a.N + BBM - 1) / BBM);
  a.nby = (a.OUT + BBN - 1) / BBN;
  bf16_supertile(a.nbx, a.nby, a.sr, a.sc);
  a.nsc = (a.nby + a.sc - 1) / a.sc;
  int64_t ntiles = (int64_t)a.nbx * a.nby;
  if (ntiles > 0x7fffffff) return ZK_EINVAL;
  a.ntiles = (int)ntiles;
  { const char* e = getenv("ZUKO_AMD_BF16_DEBUG"); a.dbg = e ? atoi(e) : 0; }
  const int act = a.act;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    (void)hipFuncSetAttribute((const void*)linear_bf16_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_bf16_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_bf16_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_bf16_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    n_cu = v;
  }
  // XCD-aware walk: needs the full 256-block grid (32 slots per XCD) and a tile grid at least one region large
  a.map = 0;
  if (n_cu == 256) {
    if (a.nby >= 16) { a.pr = 4; a.pc = 8; a.xr = 4; a.xc = 2; a.map = 1; }
    else if (a.nby >= 8) { a.pr = 4; a.pc = 8; a.xr = 8; a.xc = 1; a.map = 1; }
    else if (a.nby >= 4) { a.pr = 8; a.pc = 4; a.xr = 8; a.xc = 1; a.map = 1; }
    const char* e = getenv("ZUKO_AMD_BF16_MAP");  // experiments: "0" = id-order raster, "pr,pc,xr,xc" = explicit patch shape
    if (e) {
      int v[4] = {0, 0, 0, 0};
      const int n = sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]);
      if (n == 1 && v[0] == 0) a.map = 0;
      else if (n == 4 && v[0] * v[1] == 32 && v[2] * v[3] == 8) { a.pr = v[0]; a.pc = v[1]; a.xr = v[2]; a.xc = v[3]; a.map = 1; }
    }
    if (a.map && a.nbx < a.xr * a.pr) a.map = 0;
    if (a.map) {
      const int RR = a.xr * a.pr, RC = a.xc * a.pc;
      ntiles = (int64_t)((a.nbx + RR - 1) / RR) * ((a.nby + RC - 1) / RC) * 256;
      if (ntiles > 0x7fffffff) return ZK_EINVAL;
      a.ntiles = (int)ntiles;
    }
  }
  const int grid = (int)(ntiles < n_cu ? ntiles : n_cu);  // persistent: one 8-wave block per CU
  if (spline_k == 8) hipLaunchKernelGGL((linear_bf16_kernel<false, 8>), dim3((unsigned)grid), dim3(512), S_LDS_BYTES, stream, a);
  else if (spline_k == 16) hipLaunchKernelGGL((linear_bf16_kernel<false, 16>), dim3((unsigned)grid), dim3(512), S_LDS_BYTES, stream, a);
  else if (act <= 1) hipLaunchKernelGGL((linear_bf16_kernel<false, 0>), dim3((unsigned)grid), dim3(512), B_LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((linear_bf16_kernel<true, 0>), dim3((unsigned)grid), dim3(512), B_LDS_BYTES, stream, a);
  return ZK_LAUNCH_CHECK();
}

// out[n] = sum_p partial[p][n]
__global__ __launch_bounds__(256) void panel_sum_kernel(int P, int64_t N, const float* __restrict__ partial, float* __restrict__ out) {
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += partial[(size_t)p * N + n];
    out[n] = s;
  }
}

extern "C" int zk_linear_bf16(int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* weight, const uint64_t* tile_live_mask,
                              const void* bias, int act, void* y, int64_t ldy, void* stream) {
  if (N <= 0 || out_features <= 0) return 0;
  if (in_features <= 0 || in_features % BBK != 0 || act < 0 || act > 7) return ZK_EINVAL;
  if (ldx % 8 != 0 || (((uintptr_t)x | (uintptr_t)weight) & 15) != 0) return ZK_EINVAL;  // 16-byte DMA granules
  LinBf16Args a{};
  a.N = N; a.IN = in_features; a.OUT = out_features;
  a.x = (const __bf16*)x; a.ldx = ldx; a.w = (const __bf16*)weight; a.live = (const unsigned long long*)tile_live_mask; a.bias = (const __bf16*)bias; a.act = act;
  a.y = (__bf16*)y; a.ldy = ldy;
  return launch_linear_bf16(a, 0, (hipStream_t)stream);
}

extern "C" int zk_linear_bf16_rqs(int64_t N, int in_features, int panels, const void* h, int64_t ldh, const void* weight_panels, const uint64_t* tile_live_mask,
                                  const void* bias_panels, int K, int features, double bound, double slope, const void* x, int64_t ldx, void* y, int64_t ldy,
                                  float* partial, float* ladj, void* stream) {
  if (N <= 0 || features <= 0) return 0;
  if (K != 8 && K != 16) return ZK_EINVAL;
  const int FP = 256 / (3 * K - 1);
  if (in_features <= 0 || in_features % BBK != 0 || panels != (features + FP - 1) / FP) return ZK_EINVAL;
  if (ldh % 8 != 0 || (((uintptr_t)h | (uintptr_t)weight_panels) & 15) != 0) return ZK_EINVAL;
  LinBf16Args a{};
  a.N = N; a.IN = in_features; a.OUT = panels * 256;
  a.x = (const __bf16*)h; a.ldx = ldh; a.w = (const __bf16*)weight_panels; a.live = (const unsigned long long*)tile_live_mask; a.bias = (const __bf16*)bias_panels; a.act = 0;
  a.sx = (const __bf16*)x; a.ldsx = ldx; a.sy = (__bf16*)y; a.ldsy = ldy; a.partial = partial; a.D = features;
  a.lc = rqs_lean_const(bound, log(slope));
  const int rc = launch_linear_bf16(a, K, (hipStream_t)stream);
  if (rc) return rc;
  int64_t nb = (N + 255) / 256;
  hipLaunchKernelGGL(panel_sum_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, panels, N, partial, ladj);
  return ZK_LAUNCH_CHECK();
}
