// What does one SIMD issue v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 at, in REAL shader cycles, and what clock does the chip hold meanwhile?
// (profiles/r04 quoted "26 cycles per instruction = 61 % of the peak" in NOMINAL 2.4 GHz cycles.)
// Every wavefront brackets its loop with s_memtime (shader-clock ticks) and s_memrealtime (100 MHz): cycles per instruction come from the former, the
// clock from their ratio.  Variants: wavefronts per SIMD (1 / 2), workgroup size (256 / 64 / 512 threads), accumulators in rotation, one shared or
// distinct A / B operand registers per instruction, s_setprio, zero or random operands (the chip clocks to its power budget: MI355X_MICROARCH.md,
// "DVFS give-back").  hipcc -O3 --offload-arch=gfx950 mfma_clock_probe.hip -o mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rec { unsigned long long cyc, rt; };

template <int BIG, int NACC, int NOPS, int PRIO, int THREADS> __global__ __launch_bounds__(THREADS) void k(float* out, Rec* rec, const float* seed, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[NOPS], b[NOPS];
  for (int o = 0; o < NOPS; ++o)
    for (int e = 0; e < 8; ++e) { a[o][e] = (__bf16)seed[(lane * 8 + e + 17 * o) & 1023]; b[o][e] = (__bf16)seed[(lane * 8 + e + 31 * o + 512) & 1023]; }
  f32x16 C[NACC] = {};
  f32x4 c[NACC] = {};
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (BIG) C[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % NOPS], b[u % NOPS], C[u % NACC], 0, 0, 0);
      else c[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % NOPS], b[u % NOPS], c[u % NACC], 0, 0, 0);
    }
  }
  float t = 0;
  for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) t += C[i][r]; for (int r = 0; r < 4; ++r) t += c[i][r]; }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * THREADS + threadIdx.x] = t;
  if (lane == 0) rec[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = Rec{t1 - t0, r1 - r0};
}

template <int BIG, int NACC, int NOPS, int PRIO, int THREADS> void run(float* out, Rec* rec, const float* seed, const char* data, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<BIG, NACC, NOPS, PRIO, THREADS><<<blocks, THREADS, 0>>>(out, rec, seed, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const int nw = blocks * (THREADS / 64);
  Rec* h = (Rec*)malloc(nw * sizeof(Rec));
  hipMemcpy(h, rec, nw * sizeof(Rec), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < nw; ++i) { cyc += (double)h[i].cyc; rt += (double)h[i].rt; }
  free(h);
  const double waves_per_simd = nw / 1024.0;
  const double per_wave = cyc / nw / iters / 16;           // shader cycles between two matrix instructions of ONE wavefront
  const double per_simd = per_wave / waves_per_simd;       // ... of one SIMD
  const double ghz = cyc / rt * 0.1;                       // s_memrealtime ticks at 100 MHz
  const double peak = BIG ? 32.0 : 16.0;
  printf("%s %-6s wg %3d  waves/SIMD %.0f  acc %d  operand sets %d  prio %d : %7.3f ms  %5.2f GHz  %5.1f cycles per instruction and SIMD (real)  = %5.1f %% of the pipe's rate;  at the event clock: %5.1f %% of 2.5 PFLOP/s\n",
         BIG ? "32x32x16" : "16x16x32", data, THREADS, waves_per_simd, NACC, NOPS, PRIO, ms, ghz, per_simd, 100.0 * peak / per_simd,
         100.0 * ((double)nw * iters * 16 * (BIG ? 32768.0 : 16384.0)) / (ms * 1e-3) / 2.5e15);
}

int main() {
  float *out, *seed; Rec* rec;
  hipMalloc(&out, 2048 * 512 * 4); hipMalloc(&rec, 8192 * sizeof(Rec)); hipMalloc(&seed, 1024 * 4);
  float hs[1024];
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 1024; ++i) hs[i] = pass ? (float)((rand() % 2001) - 1000) / 500.f : 0.f;
    hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    const char* d = pass ? "random" : "zeros";
    // one wavefront per SIMD
    run<0, 4, 1, 0, 256>(out, rec, seed, d, 256); run<0, 8, 1, 0, 256>(out, rec, seed, d, 256); run<0, 4, 4, 0, 256>(out, rec, seed, d, 256); run<0, 4, 1, 1, 256>(out, rec, seed, d, 256);
    run<0, 4, 1, 0, 64>(out, rec, seed, d, 1024); run<0, 4, 4, 0, 64>(out, rec, seed, d, 1024); run<0, 1, 1, 0, 256>(out, rec, seed, d, 256); run<0, 2, 1, 0, 256>(out, rec, seed, d, 256);
    // two wavefronts per SIMD
    run<0, 4, 1, 0, 512>(out, rec, seed, d, 256); run<0, 4, 4, 0, 512>(out, rec, seed, d, 256); run<0, 4, 1, 0, 256>(out, rec, seed, d, 512);
    // the 32 x 32 form
    run<1, 4, 1, 0, 256>(out, rec, seed, d, 256); run<1, 2, 4, 0, 256>(out, rec, seed, d, 256); run<1, 4, 1, 0, 512>(out, rec, seed, d, 256);
  }
  return 0;
}
