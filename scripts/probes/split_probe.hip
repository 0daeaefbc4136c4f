// Probe for the bf16x3 operand split on the matrix cores (16x16x32 bf16, 6 of 9 partial products): accuracy against double and
// against the f32 matrix instruction, and the issue rate of the inner block (3 LDS reads + 6 MFMAs) at two wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ inline void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v; float r = v - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}

__global__ void acc_kernel(const float* A, const float* B, float* Dsplit, float* Df32, int K) {  // A [16, K], B [K, 16]
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int e = 0; e < 8; ++e) {
      __bf16 h, m, l;
      split3(A[i * K + k0 + 8 * q + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
      split3(B[(k0 + 8 * q + e) * 16 + i], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    for (int k = k0; k < k0 + 32; k += 4) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + q], B[(k + q) * 16 + i], acc2, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) { Dsplit[(4 * q + r) * 16 + i] = acc[r]; Df32[(4 * q + r) * 16 + i] = acc2[r]; }
}

extern __shared__ __attribute__((aligned(16))) float lds[];
__device__ __forceinline__ f32x4 rd(unsigned addr, int off) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
  return v;
}
#define RD(addr, off) ({ f32x4 v_; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v_) : "v"(addr), "n"(off)); v_; })
template <int N> __device__ __forceinline__ void settle3(f32x4& a, f32x4& b, f32x4& c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N)); }
template <int N> __device__ __forceinline__ void settle6(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}
#define BF(x) __builtin_bit_cast(bf16x8, x)
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a), b, c, 0, 0, 0)
// MODE 0: one block at a time (6 dependent MFMAs); MODE 1: f32 (2 reads + 8 MFMAs); MODE 2: two blocks at a time, term-major (dependency distance 2)
template <int MODE> __global__ __launch_bounds__(512, 2) void rate_kernel(const float* src, float* out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 72 * 256; i += 512) lds[i] = src[i];
  __syncthreads();
  f32x4 acc[4] = {};
  bf16x8 bh, bm, bl;
  for (int e = 0; e < 8; ++e) { bh[e] = (__bf16)(1.0f + lane + e); bm[e] = (__bf16)(0.01f * e); bl[e] = (__bf16)(0.0001f * lane); }
  const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds) + lane * 16;
  if (MODE == 0) {
    f32x4 a[2][3];
    a[0][0] = RD(addr, 0); a[0][1] = RD(addr, 1024); a[0][2] = RD(addr, 2048);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int blk = 0; blk < 20; ++blk) {
        const int nb = (blk + 1) % 20;
        a[(blk + 1) & 1][0] = RD(addr, (nb * 3 + 0) * 1024); a[(blk + 1) & 1][1] = RD(addr, (nb * 3 + 1) * 1024); a[(blk + 1) & 1][2] = RD(addr, (nb * 3 + 2) * 1024);
        settle3<3>(a[blk & 1][0], a[blk & 1][1], a[blk & 1][2]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4& c = acc[blk & 3];
        MF(a[blk & 1][2], bh, c); MF(a[blk & 1][0], bl, c); MF(a[blk & 1][1], bm, c); MF(a[blk & 1][1], bh, c); MF(a[blk & 1][0], bm, c); MF(a[blk & 1][0], bh, c);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    settle3<0>(a[0][0], a[0][1], a[0][2]);
  } else if (MODE == 2) {
    f32x4 a[2][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[0][i] = RD(addr, i * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int st = 0; st < 10; ++st) {
        const int nb = (st + 1) % 10;
#pragma unroll
        for (int i = 0; i < 6; ++i) a[(st + 1) & 1][i] = RD(addr, (nb * 6 + i) * 1024);
        settle6<6>(a[st & 1][0], a[st & 1][1], a[st & 1][2], a[st & 1][3], a[st & 1][4], a[st & 1][5]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 &c0 = acc[(2 * st) & 3], &c1 = acc[(2 * st + 1) & 3];
        f32x4* w = a[st & 1];
        MF(w[2], bh, c0); MF(w[5], bh, c1); MF(w[0], bl, c0); MF(w[3], bl, c1); MF(w[1], bm, c0); MF(w[4], bm, c1);
        MF(w[1], bh, c0); MF(w[4], bh, c1); MF(w[0], bm, c0); MF(w[3], bm, c1); MF(w[0], bh, c0); MF(w[3], bh, c1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    settle6<0>(a[0][0], a[0][1], a[0][2], a[0][3], a[0][4], a[0][5]);
  } else {
    f32x4 a[2][3];
    a[0][0] = RD(addr, 0); a[0][1] = RD(addr, 1024); a[0][2] = a[0][0];
    const f32x4 bf = __builtin_bit_cast(f32x4, bh);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int blk = 0; blk < 20; ++blk) {
        const int nb = (blk + 1) % 20;
        a[(blk + 1) & 1][0] = RD(addr, (nb * 3 + 0) * 1024); a[(blk + 1) & 1][1] = RD(addr, (nb * 3 + 1) * 1024); a[(blk + 1) & 1][2] = a[(blk + 1) & 1][0];
        settle3<2>(a[blk & 1][0], a[blk & 1][1], a[blk & 1][2]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4& c = acc[blk & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[blk & 1][0][r], bf[r], c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[blk & 1][1][r], bf[r], c, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    settle3<0>(a[0][0], a[0][1], a[0][2]);
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

int main() {
  const int K = 256;
  std::vector<float> A(16 * K), B(K * 16);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.3f) * 3.f;
  float *dA, *dB, *dD, *dE;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024); hipMalloc(&dE, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  acc_kernel<<<1, 64>>>(dA, dB, dD, dE, K);
  float D[256], E[256];
  hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(E, dE, 1024, hipMemcpyDeviceToHost);
  double es = 0, ef = 0, ec = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double ref = 0; float c = 0.f;
    for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; c = fmaf(A[i * K + k], B[k * 16 + j], c); }
    es = fmax(es, fabs(D[i * 16 + j] - ref)); ef = fmax(ef, fabs(E[i * 16 + j] - ref)); ec = fmax(ec, fabs(c - ref));
  }
  printf("K=%d max |err| vs double: split6 %.3e   mfma_f32 %.3e   cpu fmaf chain %.3e\n", K, es, ef, ec);

  float *src, *out;
  hipMalloc(&src, 72 * 1024); hipMalloc(&out, 512 * 512 * 4);
  std::vector<float> S(72 * 256, 0.001f);
  hipMemcpy(src, S.data(), 72 * 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)rate_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipFuncSetAttribute((const void*)rate_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)rate_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  const char* names[3] = {"split6 bf16, block by block ", "f32 mfma                    ", "split6 bf16, two blocks     "};
  for (int wgs = 256; wgs <= 512; wgs += 256)
  for (int mode = 0; mode < 3; ++mode) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) rate_kernel<0><<<wgs, 512, 72 * 1024>>>(src, out, iters);
      else if (mode == 1) rate_kernel<1><<<wgs, 512, 72 * 1024>>>(src, out, iters);
      else rate_kernel<2><<<wgs, 512, 72 * 1024>>>(src, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double blocks = (double)wgs * 8 * iters * 20;  // 16 x 32 x 16 blocks
    printf("%d WGs (%d waves/SIMD) %s: %.3f ms, %.1f TFLOP/s f32-equivalent, %.0f cycles per block per SIMD at 2.4 GHz\n", wgs, wgs / 128, names[mode], ms,
           blocks * 16 * 32 * 16 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 20.0) / (wgs / 128));
  }
  return 0;
}
