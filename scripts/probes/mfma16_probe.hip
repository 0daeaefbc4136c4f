// Does the legacy K = 16 form v_mfma_f32_16x16x16_f16 issue at HALF the cycles of v_mfma_f32_16x16x32_f16 on gfx950?  (Round 6: a 16 x 32 weight block of which only
// one in tile is live — 13 % of the headline kernel's blocks — would then cost half.)  Same bracketing as mfma_clock_probe.hip; two wavefronts per SIMD, dependent chains.
// hipcc -O3 --offload-arch=gfx950 mfma16_probe.hip -o mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
struct Rec { unsigned long long cyc, rt; };

template <int K16, int THREADS> __global__ __launch_bounds__(THREADS) void k(float* out, Rec* rec, const float* seed, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b; f16x4 a4, b4;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed[(lane * 8 + e) & 1023]; b[e] = (_Float16)seed[(lane * 8 + e + 512) & 1023]; }
  for (int e = 0; e < 4; ++e) { a4[e] = a[e]; b4[e] = b[e]; }
  f32x4 c[4] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (K16) c[u % 4] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[u % 4], 0, 0, 0);
      else c[u % 4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u % 4], 0, 0, 0);
    }
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) t += c[i][r];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * THREADS + threadIdx.x] = t;
  if (lane == 0) rec[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = Rec{t1 - t0, r1 - r0};
}

template <int K16, int THREADS> void run(float* out, Rec* rec, const float* seed, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<K16, THREADS><<<blocks, THREADS, 0>>>(out, rec, seed, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const int nw = blocks * (THREADS / 64);
  Rec* h = (Rec*)malloc(nw * sizeof(Rec));
  hipMemcpy(h, rec, nw * sizeof(Rec), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < nw; ++i) { cyc += (double)h[i].cyc; rt += (double)h[i].rt; }
  free(h);
  const double per_simd = cyc / nw / iters / 16 / (nw / 1024.0);
  printf("%s wg %3d waves/SIMD %.0f: %7.3f ms  %5.2f GHz  %5.1f real cycles per instruction and SIMD\n", K16 ? "v_mfma_f32_16x16x16_f16" : "v_mfma_f32_16x16x32_f16", THREADS, nw / 1024.0, ms, cyc / rt * 0.1, per_simd);
}

int main() {
  float *out, *seed; Rec* rec;
  hipMalloc(&out, 2048 * 512 * 4); hipMalloc(&rec, 8192 * sizeof(Rec)); hipMalloc(&seed, 1024 * 4);
  float hs[1024];
  for (int i = 0; i < 1024; ++i) hs[i] = (float)((rand() % 2001) - 1000) / 500.f;
  hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
  run<0, 512>(out, rec, seed, 256); run<1, 512>(out, rec, seed, 256); run<0, 256>(out, rec, seed, 256); run<1, 256>(out, rec, seed, 256);
  return 0;
}
