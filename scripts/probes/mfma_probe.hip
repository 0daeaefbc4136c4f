// Micro-probe: what limits v_mfma_f32_16x16x4_f32 issue in the fused kernel's loop shape?
// hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 4096

// MODE 0: pure MFMA, 4 accumulators, 16 MFMAs per iteration
// MODE 1: + 4 ds_read_b128 then lgkmcnt(0) before the 16 MFMAs (as the fused kernel does)
// MODE 2: ds_reads for the NEXT iteration issued before this iteration's MFMAs (manual double buffer)
// MODE 3: MODE 1 + a data-dependent uniform branch per group (skip-mask test), never skipped
// MODE 4: 6 accumulators / 24 MFMAs per group, 6 ds_reads (last-layer shape)
template <int MODE, int THREADS, int UNROLL = 1> __global__ __launch_bounds__(THREADS) void probe(float* out, const unsigned* bits, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[24 * 256 * 2];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 24 * 256 * 2; i += THREADS) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[6];
  for (int t = 0; t < 6; ++t) acc[t] = f32x4{0, 0, 0, 0};
  f32x4 b = {1.f + lane, 2.f, 3.f, 4.f};
  f32x4 a[6], nxt[6];
  constexpr int G = (MODE == 4) ? 6 : 4;
  for (int t = 0; t < G; ++t) { a[t] = f32x4{1.f * t, 2.f, 3.f, 4.f}; nxt[t] = a[t]; }
  int pos = 0;
  if (MODE == 2) for (int t = 0; t < G; ++t) nxt[t] = *reinterpret_cast<f32x4*>(lds + (pos + t) * 256 + lane * 4);
  const unsigned mask = bits[0];
#pragma unroll UNROLL
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) { if (!(mask & (1u << (it & 15)))) continue; }
    if (MODE == 1 || MODE == 3 || MODE == 4) {
#pragma unroll
      for (int t = 0; t < G; ++t) a[t] = *reinterpret_cast<f32x4*>(lds + (pos + t) * 256 + lane * 4);
    }
    if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < G; ++t) a[t] = nxt[t];
      const int np = (pos + G >= 24) ? 0 : pos + G;
#pragma unroll
      for (int t = 0; t < G; ++t) nxt[t] = *reinterpret_cast<f32x4*>(lds + (np + t) * 256 + lane * 4);
    }
    pos = (pos + G >= 24) ? 0 : pos + G;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < G; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][r], b[r], acc[t], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int t = 1; t < G; ++t) s += acc[t];
  out[blockIdx.x * THREADS + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int MODE, int THREADS, int UNROLL = 1> void run(const char* name, float* out, unsigned* bits) {
  const int blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, THREADS, UNROLL>), dim3(blocks), dim3(THREADS), 0, 0, out, bits, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, THREADS, UNROLL>), dim3(blocks), dim3(THREADS), 0, 0, out, bits, ITERS);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int G = (MODE == 4) ? 6 : 4;
  const double mfma = (double)blocks * (THREADS / 64) * ITERS * 4 * G;
  const double tf = mfma * 2048.0 / (ms * 1e-3) / 1e12;
  printf("%-44s threads=%d  %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)  cyc/MFMA/SIMD@2.4GHz=%.1f\n", name, THREADS, ms, tf, tf / 157.3 * 100,
         ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4)));
}

int main() {
  float* out; unsigned* bits;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&bits, 4);
  unsigned h = 0xffffffffu; hipMemcpy(bits, &h, 4, hipMemcpyHostToDevice);
  run<0, 512, 16>("pure MFMA, 2 waves, unroll 16 (2 KB)", out, bits);
  run<0, 512, 128>("pure MFMA, 2 waves, unroll 128 (16 KB)", out, bits);
  run<0, 512, 512>("pure MFMA, 2 waves, unroll 512 (64 KB)", out, bits);
  run<0, 256, 512>("pure MFMA, 1 wave, unroll 512 (64 KB)", out, bits);
  run<1, 512, 128>("ds_read+16 MFMA, 2 waves, unroll 128", out, bits);
  run<0, 256>("pure MFMA, 1 wave/SIMD", out, bits);
  run<0, 512>("pure MFMA, 2 waves/SIMD", out, bits);
  run<1, 256>("ds_read x4 + wait + 16 MFMA, 1 wave/SIMD", out, bits);
  run<1, 512>("ds_read x4 + wait + 16 MFMA, 2 waves/SIMD", out, bits);
  run<2, 256>("ds_read one group ahead, 1 wave/SIMD", out, bits);
  run<2, 512>("ds_read one group ahead, 2 waves/SIMD", out, bits);
  run<3, 512>("+ uniform skip branch, 2 waves/SIMD", out, bits);
  run<4, 512>("6 acc / 24 MFMA groups, 2 waves/SIMD", out, bits);
  run<4, 256>("6 acc / 24 MFMA groups, 1 wave/SIMD", out, bits);
  return 0;
}
