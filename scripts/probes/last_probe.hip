// Probe 3: the fused kernel's LAST-LAYER loop in isolation: groups of 6 accumulators, triangular skip
// pattern, optional per-group VALU epilogue of EPI dependent-ish ops, optional barrier every 4 blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AR_T 16
#define NT 6
template <int EPI, bool BAR, bool TRI, int THREADS, int STAG = 0> __global__ __launch_bounds__(THREADS, THREADS / 256) void probe(float* outp, const unsigned* skipw, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[24 * 256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 24 * 256; i += THREADS) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 in[AR_T];
  for (int t = 0; t < AR_T; ++t) in[t] = f32x4{1.f + lane + t, 2.f, 3.f, 4.f};
  int pos = 0;
  float lacc = 0.f;
  const bool late = STAG && (threadIdx.x >> 6) >= 4;
  float pp[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) pp[i] = 0.f;
  auto epi = [&]() {
    float c0 = pp[0], c1 = pp[1], c2 = pp[2], c3 = pp[3];
#pragma unroll
    for (int i = 0; i < EPI / 4; ++i) {
      c0 = c0 * 1.0001f + pp[(i + 4) % 24];
      c1 = c1 * 0.9999f + pp[(i + 9) % 24];
      c2 = c2 * 1.0002f + pp[(i + 14) % 24];
      c3 = c3 * 0.9998f + pp[(i + 19) % 24];
    }
    lacc += c0 + c1 + c2 + c3;
  };
  for (int tile = 0; tile < tiles; ++tile) {
    for (int g = 0; g < 16; ++g) {
      const unsigned bits = TRI ? skipw[g] : 0xffffu;
      const int nb = __builtin_popcount(bits);
      const int mid = nb >= 16 ? 8 : nb >= 8 ? 4 : nb >= 4 ? 2 : 0;
      if (STAG == 1 && !late) epi();
      float c0 = pp[0], c1 = pp[1], c2 = pp[2], c3 = pp[3];
      int done = 0;
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < AR_T; ++it) {
        if (bits & (1u << it)) {
          if (pos == 24) { if (BAR) __syncthreads(); pos = 0; }
          f32x4 w[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const f32x4*>(lds + (pos + t) * 256 + lane * 4);
          pos += NT;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][r], in[it][r], acc[t], 0, 0, 0);
          if (STAG == 2) {
#pragma unroll
            for (int i = 0; i < EPI / 64; ++i) {
              c0 = c0 * 1.0001f + pp[(i + 4 + it) % 24];
              c1 = c1 * 0.9999f + pp[(i + 9 + it) % 24];
              c2 = c2 * 1.0002f + pp[(i + 14 + it) % 24];
              c3 = c3 * 0.9998f + pp[(i + 19 + it) % 24];
            }
            ++done;
          }
        }
        if (STAG == 1 && (it == 0 || it == 2 || it == 4 || it == 8)) { if (late && it == mid) epi(); }
      }
      if (STAG == 2) {
        for (; done < 16; ++done) {
#pragma unroll
          for (int i = 0; i < EPI / 64; ++i) {
            c0 = c0 * 1.0001f + pp[(i + 4) % 24];
            c1 = c1 * 0.9999f + pp[(i + 9) % 24];
            c2 = c2 * 1.0002f + pp[(i + 14) % 24];
            c3 = c3 * 0.9998f + pp[(i + 19) % 24];
          }
        }
        lacc += c0 + c1 + c2 + c3;
      }
      if (STAG) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) pp[4 * t + r] = acc[t][r];
        continue;
      }
      // epilogue stub: EPI VALU ops over the 24 accumulator values (4 independent chains)
      float p[24];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * t + r] = acc[t][r];
      c0 = p[0]; c1 = p[1]; c2 = p[2]; c3 = p[3];
#pragma unroll
      for (int i = 0; i < EPI / 4; ++i) {
        c0 = c0 * 1.0001f + p[(i + 4) % 24];
        c1 = c1 * 0.9999f + p[(i + 9) % 24];
        c2 = c2 * 1.0002f + p[(i + 14) % 24];
        c3 = c3 * 0.9998f + p[(i + 19) % 24];
      }
      lacc += c0 + c1 + c2 + c3;
    }
  }
  outp[blockIdx.x * THREADS + threadIdx.x] = lacc;
}
template <int EPI, bool BAR, bool TRI, int THREADS, int STAG = 0> void run(const char* name, float* out, unsigned* bits) {
  const int blocks = 256, tiles = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<EPI, BAR, TRI, THREADS, STAG>), dim3(blocks), dim3(THREADS), 0, 0, out, bits, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<EPI, BAR, TRI, THREADS, STAG>), dim3(blocks), dim3(THREADS), 0, 0, out, bits, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double blocks_per_tile = TRI ? 136.0 : 256.0;
  const double mfma = (double)blocks * (THREADS / 64) * tiles * blocks_per_tile * 24;
  printf("%-58s %.3f ms  %.1f%% of peak  cyc/MFMA/SIMD@2.4GHz=%.1f\n", name, ms, mfma * 2048.0 / (ms * 1e-3) / 1e12 / 157.3 * 100, ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4)));
}
int main() {
  float* out; unsigned* bits;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&bits, 64);
  unsigned h[16]; for (int g = 0; g < 16; ++g) h[g] = (2u << g) - 1u;
  hipMemcpy(bits, h, 64, hipMemcpyHostToDevice);
  run<0, false, false, 512>("dense, no epilogue, no barrier", out, bits);
  run<0, false, true, 512>("triangular skip, no epilogue, no barrier", out, bits);
  run<0, true, true, 512>("triangular skip, barrier / 24 tiles", out, bits);
  run<448, false, true, 512>("triangular, 448-op epilogue, no barrier", out, bits);
  run<448, true, true, 512>("triangular, 448-op epilogue, barrier", out, bits);
  run<1792, true, true, 512>("triangular, 1792-op epilogue, barrier", out, bits);
  run<448, false, true, 512, 1>("triangular, 448-op epilogue, STAGGERED, no barrier", out, bits);
  run<448, true, true, 512, 1>("triangular, 448-op epilogue, STAGGERED, barrier", out, bits);
  run<448, false, true, 512, 2>("triangular, 448-op epilogue, IN-STREAM, no barrier", out, bits);
  run<448, true, true, 512, 2>("triangular, 448-op epilogue, IN-STREAM, barrier", out, bits);
  run<448, true, true, 256, 2>("triangular, 448-op epilogue, IN-STREAM, barrier, 1 wave", out, bits);
  run<448, true, true, 256>("triangular, 448-op epilogue, barrier, 1 wave/SIMD", out, bits);
  return 0;
}
