// What does a REAL filler cost behind v_mfma_f32_32x32x16_bf16 in a one-wavefront-per-SIMD kernel?  mfma32_probe.hip used independent
// v_fma_f32; the 32-sample operand-split kernel's fillers are conversion half-units (ReLU, cvt to bf16, remainder: dependent chains,
// operands in accumulation registers).  One "duo" = two matrix instructions on two accumulators + a filler; modes:
//   0 none | 1 eight independent v_fma | 2 one conversion half-unit pair (2 values, dependent chain, source in VGPRs)
//   3 the same with the source read from AGPRs (v_accvgpr_read) | 4 two interleaved half-units (4 values: twice the work, ILP 4)
//   5 mode 2 with the results written to AGPRs (v_accvgpr_write) | 6 sixteen independent v_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float relu_i(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

template <int MODE> __global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + lane + e); b[e] = (__bf16)(0.5f * e); }
  f32x16 C[2] = {};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = lane * 0.001f + i;
  unsigned sink = 0;
  float src[4] = {1.1f + lane, 2.2f, 3.3f, 4.4f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      C[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, C[0], 0, 0, 0);
      C[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, C[1], 0, 0, 0);
      if (MODE == 1 || MODE == 6) {
#pragma unroll
        for (int j = 0; j < (MODE == 1 ? 8 : 16); ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 15]) : "v"(v[(j + 3) & 15]), "v"(v[(j + 5) & 15]));
      } else if (MODE >= 2) {
        constexpr int NV = MODE == 4 ? 4 : 2;
        float x[4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          x[i] = src[i];
          if (MODE == 3) asm volatile("v_accvgpr_write_b32 a200, %1\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a200" : "=v"(x[i]) : "v"(src[i]) : "a200");
          else asm volatile("" : "+v"(x[i]));
        }
        unsigned packed[3][2];
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
          float v0 = relu_i(x[i]), v1 = relu_i(x[i + 1]);
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          bf2 h = {(__bf16)v0, (__bf16)v1};
          float r0 = v0 - (float)h[0], r1 = v1 - (float)h[1];
          bf2 m = {(__bf16)r0, (__bf16)r1};
          float s0 = r0 - (float)m[0], s1 = r1 - (float)m[1];
          bf2 l = {(__bf16)s0, (__bf16)s1};
          packed[0][i / 2] = __builtin_bit_cast(unsigned, h); packed[1][i / 2] = __builtin_bit_cast(unsigned, m); packed[2][i / 2] = __builtin_bit_cast(unsigned, l);
        }
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
          if (MODE == 5) asm volatile("v_accvgpr_write_b32 a201, %0\n\tv_accvgpr_write_b32 a202, %1\n\tv_accvgpr_write_b32 a203, %2" ::"v"(packed[0][i]), "v"(packed[1][i]), "v"(packed[2][i]) : "a201", "a202", "a203");
          else sink ^= packed[0][i] ^ packed[1][i] ^ packed[2][i];
        }
        src[0] += 1.f; src[1] -= 1.f;
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = (float)sink;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) t += C[i][r];
  for (int i = 0; i < 16; ++i) t += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = t + src[0] + src[1];
}

template <int MODE> void run(const char* what, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    k<MODE><<<256, 256, 0>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("mode %d (%s): %7.3f ms = %6.1f nominal cycles per duo (2 matrix instructions)\n", MODE, what, ms, ms * 1e-3 * 2.4e9 / iters / 8);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  run<0>("no filler", out); run<1>("8 independent v_fma", out); run<6>("16 independent v_fma", out); run<2>("conversion half-unit pair, VGPR source", out);
  run<3>("same, source through an AGPR", out); run<4>("two interleaved half-unit pairs", out); run<5>("half-unit pair, results to AGPRs", out);
  return 0;
}
