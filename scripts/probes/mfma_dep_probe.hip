// Issue rate of v_mfma_f32_16x16x4_f32 as a function of the distance between dependent instructions (same accumulator):
// DIST = 1: every MFMA accumulates into the one before it; 2 / 4: two / four accumulators in rotation.  One wave per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/probes/mfma_dep_probe scripts/probes/mfma_dep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DIST, int THREADS> __global__ __launch_bounds__(THREADS, 1) void probe(int iters, float* sink, unsigned long long* ticks) {
  const int lane = threadIdx.x & 63;
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = lane * 0.001f, b = 1.f + (threadIdx.x >> 6);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k % DIST] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k % DIST], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
  f4 s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s.x == 123.456f) sink[0] = s.x;
}
int main() {
  float* sink; unsigned long long* ticks;
  hipMalloc(&sink, 16); hipMalloc(&ticks, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kernel, int threads) {
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 0, 0, iters, sink, ticks); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms  %6.2f ns per MFMA per wave  %6.1f memtime ticks per MFMA\n", name, ms, ms * 1e6 / iters / 16, (double)t / iters / 16);
  };
  run("distance 1 (one accumulator), 1 wave/SIMD", probe<1, 256>, 256);
  run("distance 2, 1 wave/SIMD", probe<2, 256>, 256);
  run("distance 4, 1 wave/SIMD", probe<4, 256>, 256);
  run("distance 1, 2 waves/SIMD", probe<1, 512>, 512);
  run("distance 4, 2 waves/SIMD", probe<4, 512>, 512);
  return 0;
}
