// Can VALU work hide behind v_mfma_f32_16x16x32_bf16 on gfx950 — inside one wavefront (interleaved in program order) and across the two
// wavefronts of a SIMD?  (SQ_VALU_MFMA_COEXEC_CYCLES is 1.2 % of the matrix-busy cycles in the split headline kernel.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: MFMA only.  MODE k (1..4): k independent v_fma_f32 after every MFMA (same wavefront).
// MODE 10: wavefronts 0-3 MFMA only, wavefronts 4-7 VALU only (the partner on each SIMD).  MODE 11: only the MFMA wavefronts work.  MODE 12: only the VALU ones.
template <int MODE> __global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + lane + e); b[e] = (__bf16)(0.5f * e); }
  f32x4 c[4] = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
  const bool do_mfma = MODE < 10 || (wave < 4 && MODE != 12), do_valu = (MODE >= 1 && MODE <= 4) || (MODE >= 10 && wave >= 4 && MODE != 11);
  if (MODE < 10) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < (MODE <= 4 ? MODE : 0); ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u + j) & 7]) : "v"(v[(u + j + 3) & 7]), "v"(v[(u + j + 5) & 7]));
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 48; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(v[(u + 3) & 7]), "v"(v[(u + 5) & 7]));
    }
  }
  f32x4 s = c[0] + c[1] + c[2] + c[3];
  float t = s[0] + s[1] + s[2] + s[3];
  for (int i = 0; i < 8; ++i) t += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int MODE> void run(const char* name, float* out, int waves_per_simd) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<256, waves_per_simd * 256, 0>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-70s %7.3f ms  = %6.1f cycles per 16 MFMAs (or per 48 VALU) per wavefront at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0>("1 wavefront/SIMD, MFMA only", out, 1);
  run<1>("1 wavefront/SIMD, MFMA + 1 v_fma each", out, 1);
  run<2>("1 wavefront/SIMD, MFMA + 2 v_fma each", out, 1);
  run<3>("1 wavefront/SIMD, MFMA + 3 v_fma each", out, 1);
  run<4>("1 wavefront/SIMD, MFMA + 4 v_fma each", out, 1);
  run<0>("2 wavefronts/SIMD, MFMA only (both)", out, 2);
  run<2>("2 wavefronts/SIMD, MFMA + 2 v_fma each (both)", out, 2);
  run<4>("2 wavefronts/SIMD, MFMA + 4 v_fma each (both)", out, 2);
  run<11>("2 wavefronts/SIMD, wavefront A: 16 MFMAs per iteration, B idle", out, 2);
  run<12>("2 wavefronts/SIMD, wavefront B: 48 v_fma per iteration, A idle", out, 2);
  run<10>("2 wavefronts/SIMD, A: 16 MFMAs, B: 48 v_fma per iteration, together", out, 2);
  return 0;
}
