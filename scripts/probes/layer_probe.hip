// Probe 2: the fused kernel's hidden-layer loop in isolation (dense, A tiles from LDS, no barriers).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AR_T 16
// V: 0 = as in the kernel (bit tests, out[16], r-major over 4 accs)
//    1 = no bit tests (always dense)
//    2 = as 0 but A operand splat constant (no LDS)
//    3 = as 1 but a single in-tile register set reused for all it (B operand fixed)
template <int V, int THREADS> __global__ __launch_bounds__(THREADS, THREADS / 256) void probe(float* outp, const unsigned* skip4, int layers) {
  __shared__ __attribute__((aligned(16))) float lds[24 * 256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 24 * 256; i += THREADS) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 in[AR_T], out[AR_T];
  for (int t = 0; t < AR_T; ++t) in[t] = f32x4{1.f + lane + t, 2.f, 3.f, 4.f};
  int pos = 0;
  for (int l = 0; l < layers; ++l) {
#pragma unroll
    for (int otg = 0; otg < 4; ++otg) {
      const unsigned bits = skip4[otg];
#pragma unroll
      for (int t = 0; t < 4; ++t) out[otg * 4 + t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < AR_T; ++it) {
        if (V == 1 || V == 3 || (bits & (1u << it))) {
          f32x4 a[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (V == 2) { const float v = (float)(pos + t); a[t] = f32x4{v, v, v, v}; }
            else a[t] = *reinterpret_cast<const f32x4*>(lds + (pos + t) * 256 + lane * 4);
          }
          pos = (pos + 4 == 24) ? 0 : pos + 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) out[otg * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][r], in[V == 3 ? 0 : it][r], out[otg * 4 + t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < AR_T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) in[t][r] = out[t][r] > 0.f ? out[t][r] * 1e-3f : 0.f;
  }
  f32x4 s = in[0];
  for (int t = 1; t < AR_T; ++t) s += in[t];
  outp[blockIdx.x * THREADS + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int V, int THREADS> void run(const char* name, float* out, unsigned* bits) {
  const int blocks = 256, layers = 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<V, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, bits, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, bits, layers);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * (THREADS / 64) * layers * 1024;
  printf("%-52s threads=%d  %.3f ms  %.1f%% of 157.3 TF  cyc/MFMA/SIMD@2.4GHz=%.1f\n", name, THREADS, ms, mfma * 2048.0 / (ms * 1e-3) / 1e12 / 157.3 * 100,
         ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4)));
}
int main() {
  float* out; unsigned* bits;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&bits, 16);
  unsigned h[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu}; hipMemcpy(bits, h, 16, hipMemcpyHostToDevice);
  run<0, 512>("kernel-shaped layer (bit tests, LDS A), 2 waves", out, bits);
  run<0, 256>("kernel-shaped layer (bit tests, LDS A), 1 wave", out, bits);
  run<1, 512>("no bit tests, 2 waves", out, bits);
  run<2, 512>("bit tests, constant A (no LDS), 2 waves", out, bits);
  run<3, 512>("no bit tests, fixed B tile, 2 waves", out, bits);
  run<3, 256>("no bit tests, fixed B tile, 1 wave", out, bits);
  return 0;
}
