// Feed-rate probe: how fast can one CU pull an L2-resident panel into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), 8 waves, 64 KiB per round
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, 8 waves
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS write)
//   mode 3: mode 1 with 4 waves (256 threads)
// Every block reads its own 512 KiB window (rows of 128 B at a 2 KiB pitch, like a K=1024 bf16 panel) repeatedly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

template <int MODE, int THREADS> __global__ __launch_bounds__(THREADS) void feed(const char* __restrict__ src, int rounds, int pitch, float* sink, int nwin) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (size_t)(blockIdx.x % nwin) * 512 * pitch;   // 512 rows per window; nwin = 8: one L2-resident window per XCD
  constexpr int PIECES = 65536 / (THREADS * 16);                 // 16-byte pieces per thread per 64 KiB round
  f4 acc = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r) {
    const int kt = r & 15;                                       // k-tile: 128-byte column window
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int row = wave * (512 / (THREADS / 64)) + i * 8 + (lane >> 3);
        const char* g = base + (size_t)row * pitch + kt * 128 + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds + (r & 1) * 65536 + (wave * PIECES + i) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      f4 v[PIECES];
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int row = wave * (512 / (THREADS / 64)) + i * 8 + (lane >> 3);
        v[i] = *reinterpret_cast<const f4*>(base + (size_t)row * pitch + kt * 128 + (lane & 7) * 16);
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) acc += v[i];
      } else {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) *reinterpret_cast<f4*>(lds + (r & 1) * 65536 + ((wave * PIECES + i) * 64 + lane) * 16) = v[i];
        __syncthreads();
      }
    }
  }
  if (MODE != 2) acc = *reinterpret_cast<f4*>(lds + tid * 16);
  if (acc.x == 123.456f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const int pitch = 2048, blocks = 256;
  char* src; float* sink;
  hipMalloc(&src, (size_t)blocks * 512 * pitch + 4096);
  hipMemset(src, 1, (size_t)blocks * 512 * pitch + 4096);
  hipMalloc(&sink, 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int rounds = 2000;
  int nwin = 256;
  auto run = [&](const char* name, auto kernel, int threads) {
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 131072, 0, src, 50, pitch, sink, nwin);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 131072, 0, src, rounds, pitch, sink, nwin);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * rounds * 65536;
    printf("%-40s %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  (%s)\n", name, ms, bytes / blocks / ms / 1e6, bytes / ms / 1e9, hipGetErrorString(hipGetLastError()));
  };
  for (int pass = 0; pass < 3; ++pass) {
  nwin = pass == 0 ? 256 : (pass == 1 ? 8 : 32);
  printf("-- %d windows of 1 MiB (%s)\n", nwin, nwin == 256 ? "MALL/HBM-resident" : "L2-resident");
  run("LDS-DMA, 8 waves", feed<0, 512>, 512);
  run("LDS-DMA, 4 waves", feed<0, 256>, 256);
  run("global->VGPR->ds_write, 8 waves", feed<1, 512>, 512);
  run("global->VGPR->ds_write, 4 waves", feed<1, 256>, 256);
  run("global->VGPR only, 8 waves", feed<2, 512>, 512);
  run("global->VGPR only, 4 waves", feed<2, 256>, 256);
  }
  return 0;
}
