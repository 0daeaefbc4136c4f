// zuko_amd — cfg5's last conditioner layer + spline, second generation (round 5): "lane-owned features".
//
// Replaces, for bf16 modules, the last `F.linear(h, mask * W, b)` of the conditioner (zuko/nn.py:217-218) TOGETHER with the
// MonotonicRQSTransform it parametrises (zuko/flows/autoregressive.py:207-218, zuko/transforms.py:469-567): phi never exists.
//
// What was wrong with the first generation (linear_bf16.hip, SK > 0; profiles/r03/cfg5_ablations*.txt): every k-tile of 64 was one DMA
// round trip (two LDS stages = a one-tile look-ahead; the "DMA only" probe build takes 18 ms of the 25.5 ms k loop), 0.75 KiB of LDS
// fragment reads per matrix instruction, and the spline needed the tile as an LDS image: bias + bf16 packing, image writes, three
// workgroup barriers per 128 samples, 47 two-byte LDS reads per element — 10 ms of a 35.5 ms launch in which the matrix pipe idles.
//
// This kernel:
//   * the weight rows are laid out (host side: zuko_amd/nn.py: _Bf16Plan.spline_lane_panels) so that a LANE's accumulator registers hold
//     ALL 3K - 1 parameters of a whole feature of its samples: in the transposed product a lane (fr, kg) owns, of every 32-output block,
//     the 16 outputs q * 8 + kg * 4 + t — 48 slots over a wave tile's three blocks = one feature at 16 bins (47 used: 98 % of the
//     multiplied rows are parameters; the 256-row panels of the first generation: 92 %, i.e. 13.5 % fewer matrix instructions here), two
//     features of 24 slots at 8 bins.  The epilogue is register arithmetic: bias, bf16 rounding (what the unfused path's phi would hold),
//     rqs_lean — no LDS image, no barrier; log-derivatives of a sample are summed over the lane's features, then with the partner lane
//     (kg ^ 1), and leave as one row of `partial` per (panel, wave column);
//   * workgroup = 8 wavefronts (two per SIMD), block tile 256 samples x 192 outputs, wave tile 64 x 96 = 2 x 3 accumulators of
//     v_mfma_f32_32x32x16_bf16 (96 registers; builtin, compiler-allocated);
//   * k-tiles of 32 in a ring of FIVE 28 KiB LDS stages filled by LDS-DMA (global_load_lds_dwordx4), requested three stream positions ahead
//     of the matrix instructions and across tile boundaries (the first stages of the next tile land during the epilogue); ONE flat loop over
//     super-steps of two k-tiles (one barrier, one round of scalar bookkeeping per 12 matrix instructions of a wavefront).  The DMA is issued
//     from inline assembly: the compiler orders every LDS read behind every LDS-DMA it knows of (s_waitcnt vmcnt(0)), which would serialise
//     the ring; hidden from it, the fragment reads stay ordinary loads with compiler-counted lgkmcnt waits and the DMA is ordered by hand (one
//     `s_waitcnt vmcnt(3)` per barrier: loads complete in order, every wavefront issues >= 3 pieces per stage);
//   * DMA addressing is `global_load_lds_dwordx4 v_offset, s[base]`: the 64-bit part (tensor, tile, k-tile) is scalar, a lane adds 32 bits;
//   * rows are 64 B in LDS; the 16-byte chunk c of row r sits at slot c ^ g(r >> 2), g(v) = (v ^ (v >> 1)) & 3, which makes every
//     16-lane service group of ds_read_b128 (MI355X_MICROARCH.md, LDS table) touch all 64 banks once (SQ_LDS_BANK_CONFLICT = 0, measured);
//     the swizzle is applied to the GLOBAL address of the DMA, whose LDS side is fixed at lane * 16;
//   * accumulation order per output = the unfused kernel's (ascending k, the same matrix instruction, bias added last in f32, then
//     rounded to bf16), so y is bit-identical to zk_linear_bf16 + the bf16 spline kernel (tests/test_gpu_flows.py) — and to the first
//     generation (same checksum of y at cfg5's shape, 2^19 rows).
// How it got here, with the measurements of every intermediate design (128 x 192 and 128 x 96 wave tiles at one wavefront per SIMD with the
// accumulators in literally named AGPRs, the 105-SALU-per-step first loop, the spilling epilogues): profiles/r05/cfg5_lanes.md.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "zk_univariate.h"

namespace zk {

typedef __bf16 ln_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ln_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int ln_u32x2 __attribute__((ext_vector_type(2)));

#define LN_TM 256
#define LN_TN 192
#define LN_NS 5
#define LN_ACT_BYTES (LN_TM * 64)
#define LN_STAGE ((LN_TM + LN_TN) * 64)  /* 28672 */
#define LN_LDS (LN_NS * LN_STAGE)        /* 143360 */
#define LN_DPS 3                         /* DMA wave-instructions per wavefront and stage that EVERY wavefront issues (28 pieces / 8 wavefronts = 3, four of them one more):
                                            the vmcnt waits count with this minimum, which only makes them stricter for the others */
#ifndef ZK_LANES_ABLATE
#define ZK_LANES_ABLATE 0 /* probe builds only (WRONG results): 1 = no epilogue (accumulators kept alive), 3 = k loop without DMAs,
                             6 = epilogue without the spline arithmetic, 7 = epilogue without the bias loads and the y stores,
                             8 = every DMA instruction issued with ONE active lane (the issue slots stay, the bytes go),
                             9 = no workgroup barrier in the k loop, 10 = no wait for the DMAs, 4 = no fragment reads (opaque zero operands) */
#endif

struct LaneArgs {
  int64_t N;
  int IN, panels;
  const __bf16* h; int64_t ldh;     // last hidden activation [N, IN]
  const __bf16* w;                  // [panels * 192, IN], masked, rows in the lane-owned order
  const unsigned long long* live;   // [panels]: bit k set = inputs [32 k, 32 k + 32) of the panel hold non-zero weights; or null
  const __bf16* bias;               // [panels * 192] or null
  const __bf16* sx; int64_t ldsx;   // transform input x [N, D]
  __bf16* sy; int64_t ldsy;         // transform output y [N, D]
  float* partial;                   // [2 * panels][N]
  int D;
  RqsLeanConst lc;
  int nbx, nby, ntiles, map, pr, pc, xr, xc;
};

extern __shared__ __attribute__((aligned(16))) unsigned char ln_lds[];

__device__ __forceinline__ unsigned ln_pack(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float ln_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float ln_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

template <int SK> __global__ __launch_bounds__(512, 2) void linear_bf16_rqs_lanes_kernel(LaneArgs a) {
  constexpr int TS = SK == 16 ? 48 : 24;     // slots per feature (3 SK - 1 parameters + one padding slot)
  constexpr int FPL = 48 / TS;               // features per lane
  constexpr int FPP = 4 * FPL;               // features per panel (two wave columns x two lane halves x FPL)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 8 wavefronts (two per SIMD); wave tile: samples [wm * 64, +64), outputs [wn * 96, +96)
  const int fr = lane & 31, kg = lane >> 5;
  const int KT = a.IN >> 5;
  const int G = (int)gridDim.x;
  const int ntiles = a.ntiles;

  // ---- tile walk (as linear_bf16.hip: XCD-aware regions when map = 1) ------------------------------------------------------------
  auto locate = [&](int t, int& bx, int& by) -> bool {
    // (integer division runs on the vector unit even for uniform operands: the quotients are pulled back into scalar registers, so that
    //  everything derived from a tile position — addresses, loop conditions — stays scalar)
    auto sdiv = [](int n, int d) { return __builtin_amdgcn_readfirstlane(n / d); };
    if (a.map == 0) { bx = sdiv(t, a.nby); by = t - bx * a.nby; return true; }
    const int r = sdiv(t, G), b = t - r * G;
    const int xcd = b & 7, slot = b >> 3;
    const int RR = a.xr * a.pr, RC = a.xc * a.pc;
    const int nRC = sdiv(a.nby + RC - 1, RC);
    const int Rrow = sdiv(r, nRC), Rcol = r - Rrow * nRC;
    const int xq = sdiv(xcd, a.xc), sq = sdiv(slot, a.pc);
    bx = Rrow * RR + xq * a.pr + sq;
    by = Rcol * RC + (xcd - xq * a.xc) * a.pc + (slot - sq * a.pc);
    return bx < a.nbx && by < a.nby;
  };
  // live word of a panel.  Read through the CONSTANT address space: a uniform address then becomes an s_load — as a global load the
  // compiler waits for it with vmcnt(0), i.e. for every DMA of the ring in flight, once per tile and cursor.  A panel without any live
  // k-tile still runs k-tiles 0 and 1 (their weights are zero: the products add +0 to zeroed accumulators), and a tile with an odd number of
  // live k-tiles one dead one more: every tile has an even number >= 2 of steps.
  const auto* live_c = (const __attribute__((address_space(4))) unsigned long long*)(uintptr_t)a.live;
  auto live_of = [&](int by) -> unsigned long long {
    unsigned long long m = a.live ? live_c[by] : ~0ull;
    if (KT < 64) m &= (1ull << KT) - 1;
    if (m == 0ull) m = 3ull;  // (no live k-tile: k-tiles 0 and 1, whose weights are zero)
    if (__builtin_popcountll(m) & 1) {  // an even number of steps per tile (a super-step multiplies two k-tiles): one dead k-tile more
      const unsigned long long dead = ~m & ((KT < 64) ? ((1ull << KT) - 1) : ~0ull);
      m |= dead & (0ull - dead);  // (lowest clear bit; KT is even, so an odd count leaves one)
    }
    return m;
  };
  struct Cur { int t, bx, by; unsigned long long rem; };
  auto seek = [&](Cur& c) {  // c.t = first valid tile id >= c.t on this block's stride (ntiles if none); loads its live word
    while (c.t < ntiles && !locate(c.t, c.bx, c.by)) c.t += G;
    c.rem = c.t < ntiles ? live_of(c.by) : 0ull;
  };

  // ---- DMA side ----------------------------------------------------------------------------------------------------------------
  // stage layout: rows [0, 256) activations of the tile's samples, rows [256, 448) weights of the panel; 64 B per row; 16 rows per DMA
  // instruction, lane -> (row l >> 2, slot l & 3).  Wavefront w fills activation rows [32 w, 32 w + 32) (pieces 0, 1), weight rows
  // [16 w, 16 w + 16) (piece 2) and, w < 4 only, weight rows [128 + 16 w, + 16) (piece 3).
  const int drow = lane >> 2, dslot = lane & 3;
  auto gsw = [](int v) { return (v ^ (v >> 1)) & 3; };
  const int c16_a = (dslot ^ gsw((drow >> 2) & 7)) << 4;        // pieces whose first row r0 has r0 % 32 == 0
  const int c16_b = (dslot ^ gsw(((drow >> 2) + 4) & 7)) << 4;  // r0 % 32 == 16
  const int c16_w = (wave & 1) ? c16_b : c16_a;                 // weight pieces: r0 = 16 w or 128 + 16 w: r0 % 32 == 16 iff w odd
  const int pitch_h = (int)(a.ldh * 2), pitch_w = a.IN * 2;  // bytes (< 2^24: checked by the entry point)
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ln_lds);
  // Addressing: `global_load_lds_dwordx4 v_offset, s[base]` — the 64-bit part of an address (tensor, tile, k-tile) is scalar arithmetic once
  // per step; per piece a lane contributes a 32-bit offset: (row inside the tile) * pitch + swizzled chunk.  Activation rows are clamped to
  // the tile's last valid row (ragged last tile: re-read, results discarded): v_min + v_mad_u32_u24 per piece; weight rows never leave the
  // padded panel: three constants.
  int rl_act[2], off_w[2], act_lds[2], w_lds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { rl_act[i] = wave * 32 + 16 * i + drow; act_lds[i] = (wave * 32 + 16 * i) * 64; }
#pragma unroll
  for (int i = 0; i < 2; ++i) { off_w[i] = (128 * i + wave * 16 + drow) * pitch_w + c16_w; w_lds[i] = LN_ACT_BYTES + (128 * i + wave * 16) * 64; }
  const bool four = wave < 4;  // this wavefront issues piece 3
  // Per step the producer's scalar work is: next live bit -> k offset, two 64-bit base additions, the ring address (+ LN_STAGE with wrap).
  // (The first build recomputed the tile's 64-bit bases, `pstep % LN_NS` and the stage multiplications every step: 105 SALU instructions per
  //  step next to 24 matrix instructions — with ONE wavefront per SIMD every instruction of any kind takes an issue slot out of the
  //  ~7 a matrix instruction's 32 cycles offer; profiles/r05/cfg5_lanes.md.)
  const char* p_tile_h = nullptr;  // a.h + (first row of the producer's tile) * pitch
  const char* p_tile_w = nullptr;  // a.w + (first row of its panel) * pitch
  const char* p_base_h = nullptr;  // ... + k-tile offset: the step being produced
  const char* p_base_w = nullptr;
  int p_nvalid1 = LN_TM - 1;       // (valid rows of the producer's tile) - 1
  unsigned p_lds = lds0;           // LDS address of the stage being produced
  auto dma_piece = [&](int i) {    // i is a literal at every call site
    if (ZK_LANES_ABLATE == 3) return;
    // (hidden from the compiler's LDS alias tracking, see the header; M0 = LDS address of the wave's 1 KiB piece)
    if (i < 2) {
      const int rl = rl_act[i] < p_nvalid1 ? rl_act[i] : p_nvalid1;
      const unsigned off = (unsigned)(rl * pitch_h + ((i & 1) ? c16_b : c16_a));
#if ZK_LANES_ABLATE == 8
      { unsigned long long sv_; asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\ts_add_i32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv_) : "v"(off), "s"(p_base_h), "s"(p_lds), "s"(act_lds[i]) : "memory"); }
#else
      asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(p_base_h), "s"(p_lds), "s"(act_lds[i]) : "memory");
#endif
    } else {  // (piece 3: wavefronts 0..3 only — the caller's instantiation)
#if ZK_LANES_ABLATE == 8
      { unsigned long long sv_; asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\ts_add_i32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv_) : "v"((unsigned)off_w[i - 2]), "s"(p_base_w), "s"(p_lds), "s"(w_lds[i - 2]) : "memory"); }
#else
      asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)off_w[i - 2]), "s"(p_base_w), "s"(p_lds), "s"(w_lds[i - 2]) : "memory");
#endif
    }
  };

  // ---- fragment side -----------------------------------------------------------------------------------------------------------
  const unsigned fsw = (unsigned)gsw((fr >> 2) & 7);
  const unsigned foff0 = (unsigned)fr * 64u + ((((unsigned)kg) ^ fsw) << 4);        // k16 step 0 of a stage: chunk kg
  const unsigned foff1 = (unsigned)fr * 64u + (((2u + (unsigned)kg) ^ fsw) << 4);   // k16 step 1: chunk 2 + kg
  const unsigned xf0 = (unsigned)(wm * (64 * 64)) + foff0, xf1 = (unsigned)(wm * (64 * 64)) + foff1;
  const unsigned wf0 = (unsigned)(LN_ACT_BYTES + wn * (96 * 64)) + foff0, wf1 = (unsigned)(LN_ACT_BYTES + wn * (96 * 64)) + foff1;
#define LN_READ(b, stage_off, xo, wo)                                                                                           \
  if (ZK_LANES_ABLATE == 4) {                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) { fx[b][i_] = ln_bf16x8{}; asm volatile("" : "+v"(fx[b][i_])); }           \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) { fw[b][j_] = ln_bf16x8{}; asm volatile("" : "+v"(fw[b][j_])); }           \
  } else {                                                                                                                      \
    const unsigned char* xp_ = ln_lds + ((stage_off) + (xo));                                                                   \
    const unsigned char* wp_ = ln_lds + ((stage_off) + (wo));                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) fx[b][i_] = *reinterpret_cast<const ln_bf16x8*>(xp_ + i_ * 2048);          \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) fw[b][j_] = *reinterpret_cast<const ln_bf16x8*>(wp_ + j_ * 2048);          \
  }
#define LN_M(b, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b][j], fx[b][i], acc[i][j], 0, 0, 0)
#define LN_SB() __builtin_amdgcn_sched_barrier(0)
  Cur C{(int)blockIdx.x, 0, 0, 0ull}, P{(int)blockIdx.x, 0, 0, 0ull};
  seek(C);
  if (C.t >= ntiles) return;
  seek(P);
  auto p_new_tile = [&]() {  // 64-bit bases of the producer's tile (once per tile)
    p_tile_h = reinterpret_cast<const char*>(a.h) + (int64_t)P.bx * LN_TM * pitch_h;
    p_tile_w = reinterpret_cast<const char*>(a.w) + (int64_t)P.by * LN_TN * pitch_w;
    const int64_t nv = a.N - (int64_t)P.bx * LN_TM;
    p_nvalid1 = nv < LN_TM ? (int)nv - 1 : LN_TM - 1;
  };
  p_new_tile();
  p_base_h = p_tile_h; p_base_w = p_tile_w;
  // next step of the producer's cursor -> p_base_*, p_lds.  Exhausted walk: the last step is re-issued (see above).
  bool p_first = true;
  auto produce_next = [&]() {
    if (P.rem == 0ull && P.t < ntiles) {  // (rare: next tile with live k-tiles — every tile has at least one)
      P.t += G;
      seek(P);
      if (P.t < ntiles) p_new_tile();
    }
    if (P.rem != 0ull) {
      const int koff = (int)__builtin_ctzll(P.rem) * 64;
      P.rem &= P.rem - 1;
      p_base_h = p_tile_h + koff;
      p_base_w = p_tile_w + koff;
    }
    if (!p_first) { p_lds += LN_STAGE; p_lds = p_lds == lds0 + LN_NS * LN_STAGE ? lds0 : p_lds; }
    p_first = false;
  };
  // prologue: three stages in flight
#pragma unroll 1
  for (int s_ = 0; s_ < 3; ++s_) {
    produce_next();
    dma_piece(0); dma_piece(1); dma_piece(2);
    if (four) dma_piece(3);
  }

  // x of the lane's (sample, feature) elements, requested a whole tile ahead from inline assembly (2 FPL registers).  x is read once, i.e. from
  // HBM: as ordinary loads in the epilogue its latency stands in front of every tile's spline with all eight wavefronts waiting.  Hidden from the
  // compiler, the loads are ordered by hand: they complete in order and every tile issues >= LN_DPS DMA pieces after them, so the epilogue's
  // `s_waitcnt vmcnt(LN_DPS)` covers them.  Addresses are clamped instead of predicated (row / feature past the edge: a valid element whose
  // result is discarded).
  int64_t row0 = 0;
  int feat0 = 0;
  unsigned xr[FPL][2];
  auto tile_loads = [&]() {
    row0 = (int64_t)C.bx * LN_TM + wm * 64 + fr;
    feat0 = C.by * FPP + wn * (2 * FPL) + kg * FPL;
    if (ZK_LANES_ABLATE == 1) return;
#pragma unroll
    for (int fi = 0; fi < FPL; ++fi)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int64_t row = row0 + i * 32 < a.N ? row0 + i * 32 : a.N - 1;
        const int feat = feat0 + fi < a.D ? feat0 + fi : a.D - 1;
        const __bf16* src = a.sx + (row * a.ldsx + feat);
        asm volatile("global_load_ushort %0, %1, off" : "=v"(xr[fi][i]) : "v"(src));
      }
  };

  // My DMAs of every stream position but the last requested one have landed (in-order completion; that one has at least LN_DPS pieces per
  // wavefront), then the workgroup barrier: everybody's have, and everybody has finished reading the stage that is overwritten next.  Stores and loads
  // of an epilogue in between only make the count stricter.
#if ZK_LANES_ABLATE == 9
#define LN_WAITBAR() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LN_DPS) : "memory")
#elif ZK_LANES_ABLATE == 10
#define LN_WAITBAR() asm volatile("s_barrier" ::: "memory")
#else
#define LN_WAITBAR() asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LN_DPS) : "memory")
#endif
  // (the barrier right behind an epilogue: the next tile's 2 FPL loads of x were issued after that position's pieces as well — without the
  //  allowance the wait would stand on the epilogue's stores)
#define LN_WAITBAR_TILE() asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LN_DPS + 2 * FPL) : "memory")

  ln_f32x16 acc[2][3];
#define LN_ZERO() _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.f
  LN_ZERO();
  tile_loads();
  unsigned long long rem = C.rem;
  unsigned c_off = 0;            // byte offset of the consumer's stage in the ring
  ln_bf16x8 fx[2][2], fw[2][3];
  LN_WAITBAR();                  // stream positions 0 and 1 have landed

  // ONE flat loop over the SUPER-STEPS (two k-tiles = stream positions 2n, 2n + 1, one barrier) of all tiles of this workgroup; the tile's end is
  // a rarely taken branch inside it.  During super-step n the positions 2n + 3 and 2n + 4 are requested, into the ring slots super-step n - 1 read
  // (everybody is past them: barrier); position 2n + 2 is the one that may still fly at the next barrier.  The scalar bookkeeping of a step and
  // the barrier are paid once per 12 matrix instructions of a wavefront (the one-k-tile version: once per 6; profiles/r05/cfg5_lanes.md).
  while (true) {
    const unsigned b_off = c_off + LN_STAGE == LN_NS * LN_STAGE ? 0u : c_off + LN_STAGE;
    const unsigned n_off = b_off + LN_STAGE == LN_NS * LN_STAGE ? 0u : b_off + LN_STAGE;
    // Everything between two LN_SB() is issued in this order.  The fragment reads of a half ride in front of the previous half's matrix
    // instructions; the first ones behind the barrier are covered by the producer's scalar work.  The two wavefronts of a SIMD (w and w + 4)
    // run this loop in lock-step — the barrier aligns them — and a DMA piece holds its wavefront's issue for ~50 cycles: with the pieces at the
    // same places in both, the matrix pipe idled through every one of them (the probe build without DMAs: 7 ms of 33 faster).  So wavefronts
    // 0..3 issue theirs in the first and third quarter of the super-step, wavefronts 4..7 in the second and fourth: a wavefront in a DMA issue
    // has a partner in a stretch of bare matrix instructions.
    // (`four` = wavefronts 0..3 = the early ones; scalar branches around single DMA instructions)
#define LN_E(i) if (four) dma_piece(i)
#define LN_L(i) if (!four) dma_piece(i)
    LN_READ(0, c_off, xf0, wf0);
    produce_next();
    LN_E(0);
    LN_READ(1, c_off, xf1, wf1);
    LN_SB();
    LN_M(0, 0, 0); LN_E(1); LN_M(0, 1, 0); LN_M(0, 0, 1); LN_E(2); LN_M(0, 1, 1); LN_M(0, 0, 2); LN_E(3); LN_M(0, 1, 2);
    LN_SB();
    LN_READ(0, b_off, xf0, wf0);
    LN_SB();
    LN_M(1, 0, 0); LN_L(0); LN_M(1, 1, 0); LN_M(1, 0, 1); LN_L(1); LN_M(1, 1, 1); LN_M(1, 0, 2); LN_L(2); LN_M(1, 1, 2);
    LN_SB();
    produce_next();
    LN_READ(1, b_off, xf1, wf1);
    LN_SB();
    LN_M(0, 0, 0); LN_E(0); LN_M(0, 1, 0); LN_E(1); LN_M(0, 0, 1); LN_M(0, 1, 1); LN_E(2); LN_M(0, 0, 2); LN_E(3); LN_M(0, 1, 2);
    LN_SB();
    LN_M(1, 0, 0); LN_L(0); LN_M(1, 1, 0); LN_M(1, 0, 1); LN_L(1); LN_M(1, 1, 1); LN_M(1, 0, 2); LN_L(2); LN_M(1, 1, 2);
    LN_SB();
#undef LN_E
#undef LN_L
    c_off = n_off;
    rem &= rem - 1;
    rem &= rem - 1;
    if (rem == 0ull) {  // ---- the tile is complete ------------------------------------------------------------------------------------
      // epilogue: a[48 i + s] = C[sample i * 32 + fr][output (s / 16) * 32 + ((s % 16) / 4) * 8 + kg * 4 + s % 4], slot s = 0..47
      if (ZK_LANES_ABLATE == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(acc[i][j]));
      } else {
        // x: requested a tile ago (above); the bias of the lane's 48 slots (groups of four consecutive outputs: 8 bytes; always in L2) by
        // ordinary loads — the compiler waits for those with vmcnt(0), i.e. also for the ring's DMAs in flight (issued during the tile's last
        // steps: mostly landed).
        float xv[FPL][2];
#pragma unroll
        for (int fi = 0; fi < FPL; ++fi) {
          asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xr[fi][0]), "+v"(xr[fi][1]) : "n"(LN_DPS));
#pragma unroll
          for (int i = 0; i < 2; ++i) xv[fi][i] = ln_lo(xr[fi][i]);
        }
        ln_u32x2 bz[12];
#pragma unroll
        for (int g = 0; g < 12; ++g) {
          const int col = C.by * LN_TN + wn * 96 + (g / 4) * 32 + (g % 4) * 8 + kg * 4;
          bz[g] = (a.bias && ZK_LANES_ABLATE != 7) ? *reinterpret_cast<const ln_u32x2*>(a.bias + col) : ln_u32x2{0u, 0u};
        }
        // Both samples' 48 slots first, as the unfused path's phi would hold them (f32 sum + bias, rounded to bf16): 2 x 24 packed registers,
        // after which the 96 accumulators and the bias are dead — the spline then has the register file to itself (evaluated per sample with
        // accumulators and bias still live, the 16-bin spline spilled to scratch memory, whose reloads wait behind the ring's DMAs: 21 ms).
        unsigned pk[2][24];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
          for (int u = 0; u < 24; ++u) {
            const unsigned bw = (u & 1) ? bz[u / 2].y : bz[u / 2].x;
            pk[I][u] = ln_pack(acc[I][(2 * u) / 16][(2 * u) % 16] + ln_lo(bw), acc[I][(2 * u) / 16][(2 * u) % 16 + 1] + ln_hi(bw));
          }
        float lsum[2];
#pragma unroll
        for (int I = 0; I < 2; ++I) {
          const int64_t row = row0 + I * 32;
          float sum = 0.f;
#pragma unroll
          for (int fi = 0; fi < FPL; ++fi) {
            const int feat = feat0 + fi;
            // (one basic block per element: the SIMD's second wavefront covers the latencies)
            float lj = 0.f;
            if (feat < a.D && row < a.N) {
              auto par = [&](int t) { const int s_ = fi * TS + t; return (s_ & 1) ? ln_hi(pk[I][s_ >> 1]) : ln_lo(pk[I][s_ >> 1]); };
              float yv;
#if ZK_LANES_ABLATE == 6
              yv = xv[fi][I];
#pragma unroll
              for (int t = 0; t < 3 * SK - 1; ++t) { yv += par(t); lj += par(t) * 0.5f; }
#else
              rqs_lean<SK, false>([&](int t) { return par(t); }, [&](int t) { return par(SK + t); }, [&](int t) { return par(2 * SK + t); }, a.lc, xv[fi][I], yv, lj);
#endif
              if (ZK_LANES_ABLATE == 7) lj += yv; else a.sy[row * a.ldsy + feat] = (__bf16)yv;
            }
            sum += lj;
          }
          lsum[I] = sum;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float tot = lsum[i] + __shfl_xor(lsum[i], 32);
          const int64_t row = row0 + i * 32;
          if (kg == 0 && row < a.N) a.partial[(size_t)(C.by * 2 + wn) * a.N + row] = tot;
        }
      }
      C.t += G;
      seek(C);
      if (C.t >= ntiles) break;
      rem = C.rem;
      LN_ZERO();
      tile_loads();
      LN_WAITBAR_TILE();
    } else {
      LN_WAITBAR();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-issued steps still write this workgroup's LDS: they must land before it is released
#undef LN_READ
#undef LN_M
#undef LN_SB
#undef LN_WAITBAR
#undef LN_WAITBAR_TILE
}

// out[n] = sum_p partial[p][n]
__global__ __launch_bounds__(256) void lanes_panel_sum_kernel(int P, int64_t N, const float* __restrict__ partial, float* __restrict__ out) {
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += partial[(size_t)p * N + n];
    out[n] = s;
  }
}

}  // namespace zk

using namespace zk;

// Lane-owned layout of the last layer (zuko_amd/nn.py: _Bf16Plan.spline_lane_panels builds it).  With TS = 48 (K = 16) or 24 (K = 8)
// slots per feature, FPL = 48 / TS features per lane and FPP = 4 FPL features per panel of 192 rows, row o of panel p holds
//   wn = o / 96, c = o % 96, j = c / 32, q = (c % 32) / 8, kg = (c % 8) / 4, t = c % 4, slot s = j * 16 + q * 4 + t,
//   feature p * FPP + wn * 2 FPL + kg * FPL + s / TS, parameter s % TS   (zero row when parameter >= 3K - 1 or feature >= features),
// parameters in the reference's order widths, heights, derivatives (zuko/flows/spline.py:55-59).  tile_live_mask[p]: bit k = inputs
// [32 k, 32 k + 32) of the panel carry a non-zero weight (NULL or in_features > 2048: every k-tile is multiplied).  partial: workspace of
// 2 * panels * N floats.  ladj[n] = sum over features of log|dy/dx| (written, not accumulated).
extern "C" int zk_linear_bf16_rqs_lanes(int64_t N, int in_features, int panels, const void* h, int64_t ldh, const void* weight_panels,
                                        const uint64_t* tile_live_mask, const void* bias_panels, int K, int features, double bound, double slope,
                                        const void* x, int64_t ldx, void* y, int64_t ldy, float* partial, float* ladj, void* stream) {
  if (N <= 0 || features <= 0) return 0;
  if (K != 8 && K != 16) return ZK_EINVAL;
  const int FPP = K == 16 ? 4 : 8;
  if (in_features <= 0 || in_features % 64 != 0 || panels != (features + FPP - 1) / FPP) return ZK_EINVAL;
  if (ldh % 8 != 0 || (((uintptr_t)h | (uintptr_t)weight_panels) & 15) != 0 || (((uintptr_t)bias_panels) & 7) != 0) return ZK_EINVAL;
  if (N > 0x7fffffff || ldh >= (1 << 22) || in_features > 2048) return ZK_EINVAL;  // (32-bit lane offsets inside a tile; the walk's k-tile words hold 64 k-tiles of 32)
  LaneArgs a{};
  a.N = N; a.IN = in_features; a.panels = panels;
  a.h = (const __bf16*)h; a.ldh = ldh; a.w = (const __bf16*)weight_panels; a.live = (const unsigned long long*)tile_live_mask; a.bias = (const __bf16*)bias_panels;
  a.sx = (const __bf16*)x; a.ldsx = ldx; a.sy = (__bf16*)y; a.ldsy = ldy; a.partial = partial; a.D = features;
  a.lc = rqs_lean_const(bound, log(slope));
  a.nbx = (int)((N + LN_TM - 1) / LN_TM);
  a.nby = panels;
  int64_t ntiles = (int64_t)a.nbx * a.nby;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    (void)hipFuncSetAttribute((const void*)linear_bf16_rqs_lanes_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS);
    (void)hipFuncSetAttribute((const void*)linear_bf16_rqs_lanes_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS);
    n_cu = v;
  }
  // XCD-aware walk (as zk_linear_bf16): regions of (xr pr) x (xc pc) tiles, one pr x pc patch of 32 tiles per XCD and round
  a.map = 0;
  if (n_cu == 256) {
    // (measured at cfg5's last layer, 2^19 rows, scripts/cfg5_lanes_map.sh: patches of 8 x 4 in regions of 64 x 4 tiles 30.6 ms; 4 x 8 in
    //  16 x 16 — the first generation's choice — 31.5; 2 x 16: 35.1; 16 x 2: 32.4; id-order raster 46.9)
    if (a.nby >= 4 && a.nbx >= 64) { a.pr = 8; a.pc = 4; a.xr = 8; a.xc = 1; a.map = 1; }
    else if (a.nby >= 16) { a.pr = 4; a.pc = 8; a.xr = 4; a.xc = 2; a.map = 1; }
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
    }
  }
  if (ntiles > 0x7fffffff) return ZK_EINVAL;
  a.ntiles = (int)ntiles;
  const int grid = (int)(ntiles < n_cu ? ntiles : n_cu);  // persistent: one 8-wave block per CU
  if (K == 8) hipLaunchKernelGGL((linear_bf16_rqs_lanes_kernel<8>), dim3((unsigned)grid), dim3(512), LN_LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((linear_bf16_rqs_lanes_kernel<16>), dim3((unsigned)grid), dim3(512), LN_LDS, (hipStream_t)stream, a);
  int rc = ZK_LAUNCH_CHECK();
  if (rc) return rc;
  const int64_t nb = (N + 255) / 256;
  hipLaunchKernelGGL(lanes_panel_sum_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, 2 * panels, N, partial, ladj);
  return ZK_LAUNCH_CHECK();
}
