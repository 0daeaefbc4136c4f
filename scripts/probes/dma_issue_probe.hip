// What does one LDS-DMA instruction cost a wavefront that is alone on its SIMD and otherwise issues back-to-back MFMAs?
// (the situation of the fused kernels' weight rings: fused_ar.hip, inc_inverse.hip, fused_coupling.hip)
//   every iteration: 16 x v_mfma_f32_16x16x4_f32 on four accumulators (512 cycles of matrix pipe) + ONE load of `MODE`
//   mode 0: no load
//   mode 1: global_load_lds_dwordx4, 64-bit per-lane address (what the compiler emits for the ring today)
//   mode 2: global_load_lds_dwordx4, scalar base + 32-bit per-lane offset
//   mode 3: buffer_load_dwordx4 ... offen lds (buffer resource + 32-bit offset)
//   mode 4: global_load_dwordx4 into registers (no LDS)
//   mode 5: global_load_lds_dword (4 bytes per lane), 64-bit address
//   mode 6: mode 1 without the M0 write (same LDS target every time)
//   mode 7: two mode-2 loads per iteration
// 256 blocks x 4 waves, one block per CU (LDS-limited), sources L2-resident (each XCD cycles through 2 MiB).
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/probes/dma_issue_probe scripts/probes/dma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) float lds[];

template <int MODE, int THREADS> __global__ __launch_bounds__(THREADS, 1) void probe(const float* __restrict__ src, int iters, float* sink, unsigned long long* ticks, int win_floats) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f4 got = {0, 0, 0, 0};
  float a = lane * 0.001f, b = wave + 1.f;
  const float* base = src + (size_t)(blockIdx.x & 7) * win_floats;  // window per XCD
  unsigned rsrc[4];
  {
    const unsigned long long p = (unsigned long long)base;
    rsrc[0] = (unsigned)p; rsrc[1] = (unsigned)(p >> 32) & 0xffff; rsrc[2] = 0x7fffffff; rsrc[3] = 0x00020000;
  }
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 rs = {rsrc[0], rsrc[1], rsrc[2], rsrc[3]};
  rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
  rs.z = __builtin_amdgcn_readfirstlane(rs.z); rs.w = __builtin_amdgcn_readfirstlane(rs.w);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const unsigned tile = (unsigned)((it * 8 + wave) & 2047);  // 2048 KiB window
    const unsigned lds_addr = (unsigned)(((it & 7) * 8 + wave) * 1024);
    const unsigned voff = tile * 1024u + lane * 16u;
    const float* g = base + tile * 256 + lane * 4;
    if (MODE == 1) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(g) : "memory");
    if (MODE == 2 || MODE == 7) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
    if (MODE == 7) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, %2" ::"s"(lds_addr + 16384u * 4u), "v"(voff + 4096u), "s"(base) : "memory");
    if (MODE == 8) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, off\n global_load_lds_dwordx4 %1, off offset:1024\n global_load_lds_dwordx4 %1, off offset:2048\n global_load_lds_dwordx4 %1, off offset:3072" ::"s"(lds_addr & 0xffffu), "v"(g) : "memory");
    if (MODE == 3) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
    if (MODE == 4) { f4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g) : "memory"); got = v; }
    if (MODE == 5) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dword %1, off" ::"s"(lds_addr), "v"(g) : "memory");
    if (MODE == 6) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(g) : "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // bounded queue, as in the rings
    if (MODE == 9) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
  f4 s = acc[0] + acc[1] + acc[2] + acc[3] + got;
  if (s.x == 123.456f) sink[0] = s.x + lds[lane];
}

int main() {
  const int win_floats = 2048 * 256;  // 2 MiB per XCD window
  float *src, *sink; unsigned long long* ticks;
  hipMalloc(&src, (size_t)8 * win_floats * 4 + 65536);
  hipMemset(src, 0, (size_t)8 * win_floats * 4 + 65536);
  hipMalloc(&sink, 16); hipMalloc(&ticks, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kernel, int threads) {
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 140 * 1024, 0, src, iters, sink, ticks, win_floats);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.3f ms  %7.1f ns/iter  %8.1f memtime ticks/iter\n", name, ms, ms * 1e6 / iters, (double)t / iters);
  };
  run("0 no load", probe<0, 256>, 256);
  run("1 global_load_lds_dwordx4 (64-bit vaddr)", probe<1, 256>, 256);
  run("2 global_load_lds_dwordx4 (saddr + voffset)", probe<2, 256>, 256);
  run("3 buffer_load_dwordx4 offen lds", probe<3, 256>, 256);
  run("4 global_load_dwordx4 -> VGPR", probe<4, 256>, 256);
  run("5 global_load_lds_dword (64-bit vaddr)", probe<5, 256>, 256);
  run("6 global_load_lds_dwordx4, M0 untouched", probe<6, 256>, 256);
  run("7 two saddr loads per iteration", probe<7, 256>, 256);
  run("8 four loads on one M0 (immediate offsets)", probe<8, 256>, 256);
  run("9 no load, s_barrier per iteration", probe<9, 256>, 256);
  printf("-- two waves per SIMD (512 threads): MFMA time per iteration doubles --\n");
  run("0 no load", probe<0, 512>, 512);
  run("1 global_load_lds_dwordx4 (64-bit vaddr)", probe<1, 512>, 512);
  run("6 global_load_lds_dwordx4, M0 untouched", probe<6, 512>, 512);
  run("4 global_load_dwordx4 -> VGPR", probe<4, 512>, 512);
  run("8 four loads on one M0 (immediate offsets)", probe<8, 512>, 512);
  run("9 no load, s_barrier per iteration", probe<9, 512>, 512);
  return 0;
}
