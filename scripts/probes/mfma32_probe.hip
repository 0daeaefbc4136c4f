// Single-wavefront issue rate of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 with K independent VALU fillers behind every matrix
// instruction (one wavefront per SIMD, 256 threads per workgroup, 256 workgroups): is a 32-sample wave tile on the 32x32x16 form free of
// the ~20-cycle single-wave issue floor the 16x16x32 form shows (profiles/r04/arx2.md)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BIG, int FILL, int NACC> __global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + lane + e); b[e] = (__bf16)(0.5f * e); }
  f32x16 C[4] = {};
  f32x4 c[4] = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (BIG) C[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, C[u % NACC], 0, 0, 0);
      else c[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u % NACC], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u + j) & 7]) : "v"(v[(u + j + 3) & 7]), "v"(v[(u + j + 5) & 7]));
    }
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) t += C[i][r]; for (int r = 0; r < 4; ++r) t += c[i][r]; }
  for (int i = 0; i < 8; ++i) t += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int BIG, int FILL, int NACC> void run(float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<BIG, FILL, NACC><<<256, 256, 0>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double per = ms * 1e-3 * 2.4e9 / iters / 16;
  printf("%s, %d fillers, %d accumulators: %7.3f ms = %6.1f cycles per matrix instruction at 2.4 GHz = %5.1f %% of the bf16 peak\n", BIG ? "32x32x16" : "16x16x32", FILL, NACC, ms, per,
         100.0 * (BIG ? 32.0 : 16.0) / per);
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  run<0, 0, 4>(out); run<0, 1, 4>(out); run<0, 2, 4>(out); run<0, 3, 4>(out);
  run<1, 0, 4>(out); run<1, 0, 2>(out); run<1, 0, 1>(out); run<1, 2, 2>(out); run<1, 4, 2>(out); run<1, 5, 2>(out); run<1, 6, 2>(out); run<1, 8, 2>(out); run<1, 4, 4>(out); run<1, 6, 4>(out);
  return 0;
}
