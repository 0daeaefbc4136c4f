// zuko_amd — one conditioner layer  Y = act(X (W .* mask)^T + b)  as a stand-alone kernel.
//
// Replaces `F.linear(x, mask * weight, bias)` + activation (zuko/nn.py:217-218, :13-15) for
// conditioners the fused kernel (fused_ar.hip) does not cover (dense coupling MLPs, wide hidden
// layers).  The mask is applied while the weight tile is staged into LDS, so no masked copy of W
// is ever materialised (the reference builds one per call).
//
// fp32: LDS-tiled 128x128x32 block tile, 4 wavefronts (2x2), each owning a 64x64 output tile as
// 2x2 v_mfma_f32_32x32x2_f32 accumulators.  f32-input MFMA is bitwise an fmaf chain, so the result
// differs from any other fp32 GEMM only by summation order.  A = X (rows = samples), B = W^T.
// Fragments are read from LDS as ds_read_b128: lane (i, h) takes k = 8*kk + 4*h .. +3, and MFMA
// number r of the group pairs k = 8kk+r (lanes 0-31) with k = 8kk+4+r (lanes 32-63) for BOTH
// operands, which is all the instruction requires.
// fp64: plain VALU tile kernel (used only for double-precision parity runs).
#include "zk_common.h"

namespace zk {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_TANH = 3, ACT_SILU = 4, ACT_GELU = 5, ACT_SIGMOID = 6, ACT_LEAKY = 7 };

template <typename T> __device__ __forceinline__ T apply_act(T v, int act) {
  switch (act) {
    case ACT_RELU: return v < T(0) ? T(0) : v;  // NaN stays NaN, as torch.relu
    case ACT_ELU: return v > T(0) ? v : (T)expm1((double)v);
    case ACT_TANH: return (T)tanh((double)v);
    case ACT_SILU: return v / (T(1) + t_exp(-v));
    case ACT_GELU: return T(0.5) * v * (T(1) + (T)erf((double)v * 0.70710678118654752440));
    case ACT_SIGMOID: return T(1) / (T(1) + t_exp(-v));
    case ACT_LEAKY: return v > T(0) ? v : T(0.01) * v;
    default: return v;
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LinArgs {
  int64_t N;
  int IN, OUT;
  const void* x; int64_t ldx;
  const void* w;            // [OUT, IN] row-major
  const uint8_t* mask;      // [OUT, IN] or null
  const void* bias;         // [OUT] or null
  int act;
  void* y; int64_t ldy;
  int nbx, nby;
};

#define LBM 128
#define LBN 128
#define LBK 32
#define LPAD 4

__device__ __forceinline__ void xcd_remap(int nbx, int nby, int& bx, int& by) {
  // blocks that share an X row-panel (same bx) get consecutive LOGICAL ids; hardware places
  // physical id b on XCD b % 8, so give every XCD a contiguous logical range (bijective form).
  const int nwg = nbx * nby;
  const int orig = blockIdx.x;
  const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  bx = logical / nby;
  by = logical % nby;
}

template <bool VEC4> __global__ __launch_bounds__(256) void linear_f32_mfma(LinArgs a) {
  __shared__ __attribute__((aligned(16))) float As[LBM][LBK + LPAD];
  __shared__ __attribute__((aligned(16))) float Bs[LBN][LBK + LPAD];
  int bx, by;
  xcd_remap(a.nbx, a.nby, bx, by);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t row0 = (int64_t)bx * LBM;
  const int col0 = by * LBN;
  const float* X = (const float*)a.x;
  const float* W = (const float*)a.w;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ga[4], gb[4];
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // 32 rows x 8 float4 per pass
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lr + 32 * p;
      const int64_t gr = row0 + r;
      const int k = k0 + lc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < a.N) {
        const float* src = X + gr * a.ldx + k;
        if (VEC4) { if (k < a.IN) v = *reinterpret_cast<const float4*>(src); }
        else { if (k < a.IN) v.x = src[0]; if (k + 1 < a.IN) v.y = src[1]; if (k + 2 < a.IN) v.z = src[2]; if (k + 3 < a.IN) v.w = src[3]; }
      }
      ga[p] = v;
      const int gc = col0 + r;
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gc < a.OUT) {
        const int64_t o = (int64_t)gc * a.IN + k;
        const float* src = W + o;
        if (VEC4) { if (k < a.IN) u = *reinterpret_cast<const float4*>(src); }
        else { if (k < a.IN) u.x = src[0]; if (k + 1 < a.IN) u.y = src[1]; if (k + 2 < a.IN) u.z = src[2]; if (k + 3 < a.IN) u.w = src[3]; }
        if (a.mask) {
          const uint8_t* m = a.mask + o;
          if (k < a.IN && !m[0]) u.x = 0.f;
          if (k + 1 < a.IN && !m[1]) u.y = 0.f;
          if (k + 2 < a.IN && !m[2]) u.z = 0.f;
          if (k + 3 < a.IN && !m[3]) u.w = 0.f;
        }
      }
      gb[p] = u;
    }
  };

  const int nk = (a.IN + LBK - 1) / LBK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<float4*>(&As[lr + 32 * p][lc]) = ga[p];
      *reinterpret_cast<float4*>(&Bs[lr + 32 * p][lc]) = gb[p];
    }
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * LBK);
    const int fi = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < LBK / 8; ++kk) {
      typedef float f32x4_l __attribute__((ext_vector_type(4)));
      f32x4_l af[2], bf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const f32x4_l*>(&As[wr * 64 + m * 32 + fi][kk * 8 + 4 * fh]);
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const f32x4_l*>(&Bs[wc * 64 + n * 32 + fi][kk * 8 + 4 * fh]);
      // transposed product (A operand = weight rows): a lane ends up with 4 consecutive outputs of one sample.
      // k element outermost: consecutive MFMAs go to four different accumulators (a dependent one would have to
      // wait for its predecessor's 16 passes — measured: 70 % -> pipe busy with the accumulator-major order)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[n][r], af[m][r], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: acc[m][n][r] = Y[sample m*32 + (lane&31)][output n*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]: 16-byte stores of
  // four consecutive outputs.  None / ReLU are applied inline; the other activations run afterwards in a ROLLED loop
  // over the values this thread just stored (their inline expansions, unrolled 64 times, made the epilogue
  // instruction-cache bound — measured on the bf16 kernel).
  const float* bias = (const float*)a.bias;
  float* Y = (float*)a.y;
  const bool relu = a.act == ACT_RELU;
  const bool vec_ok = (a.ldy % 4 == 0) && ((((uintptr_t)Y) & 15) == 0);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = col0 + wc * 64 + n * 32 + 8 * q + 4 * (lane >> 5);
      float bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t] = (bias && col + t < a.OUT) ? bias[col + t] : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int64_t row = row0 + wr * 64 + m * 32 + (lane & 31);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v[t] = acc[m][n][4 * q + t] + bv[t];
          if (relu) v[t] = v[t] < 0.f ? 0.f : v[t];  // NaN stays NaN, as torch.relu
        }
        if (row < a.N) {
          float* dst = Y + row * a.ldy + col;
          if (vec_ok && col + 4 <= a.OUT) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (col + t < a.OUT) dst[t] = v[t];
          }
        }
      }
    }
  }
  if (a.act > ACT_RELU) {
#pragma unroll 1
    for (int e = 0; e < 64; ++e) {
      const int n = e >> 5, m = (e >> 4) & 1, r = e & 15;
      const int col = col0 + wc * 64 + n * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
      const int64_t row = row0 + wr * 64 + m * 32 + (lane & 31);
      if (col < a.OUT && row < a.N) {
        float* p = Y + row * a.ldy + col;
        *p = apply_act<float>(*p, a.act);
      }
    }
  }
}

// generic-precision tile kernel (fp64): 64x64 tile, 16-deep, 4x4 outputs per thread
template <typename T> __global__ __launch_bounds__(256) void linear_valu(LinArgs a) {
  __shared__ T As[16][64 + 1];
  __shared__ T Bs[16][64 + 1];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t row0 = (int64_t)(blockIdx.x / a.nby) * 64;
  const int col0 = (blockIdx.x % a.nby) * 64;
  const T* X = (const T*)a.x;
  const T* W = (const T*)a.w;
  T acc[4][4] = {};
  for (int k0 = 0; k0 < a.IN; k0 += 16) {
    for (int i = tid; i < 64 * 16; i += 256) {
      const int r = i >> 4, k = i & 15;
      const int64_t gr = row0 + r;
      As[k][r] = (gr < a.N && k0 + k < a.IN) ? X[gr * a.ldx + k0 + k] : T(0);
      const int gc = col0 + r;
      T w = T(0);
      if (gc < a.OUT && k0 + k < a.IN) {
        const int64_t o = (int64_t)gc * a.IN + k0 + k;
        w = W[o];
        if (a.mask && !a.mask[o]) w = T(0);
      }
      Bs[k][r] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      T av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[k][ty * 4 + i]; bv[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
  const T* bias = (const T*)a.bias;
  T* Y = (T*)a.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = row0 + ty * 4 + i;
    if (row >= a.N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + tx * 4 + j;
      if (col < a.OUT) Y[row * a.ldy + col] = apply_act<T>(acc[i][j] + (bias ? bias[col] : T(0)), a.act);
    }
  }
}

}  // namespace zk

using namespace zk;

extern "C" int zk_linear(int dtype, int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* weight, const uint8_t* mask,
                         const void* bias, int act, void* y, int64_t ldy, void* stream) {
  if (N <= 0 || out_features <= 0) return 0;
  if (in_features <= 0 || act < 0 || act > ACT_LEAKY) return ZK_EINVAL;
  LinArgs a{};
  a.N = N; a.IN = in_features; a.OUT = out_features; a.x = x; a.ldx = ldx; a.w = weight; a.mask = mask; a.bias = bias; a.act = act; a.y = y; a.ldy = ldy;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ZK_DTYPE_F32) {
    a.nbx = (int)((N + LBM - 1) / LBM);
    a.nby = (out_features + LBN - 1) / LBN;
    const bool vec4 = (in_features % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x | (uintptr_t)weight) % 16 == 0);
    dim3 grid((unsigned)(a.nbx * a.nby));
    if (vec4) hipLaunchKernelGGL((linear_f32_mfma<true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((linear_f32_mfma<false>), grid, dim3(256), 0, st, a);
  } else if (dtype == ZK_DTYPE_F64) {
    a.nbx = (int)((N + 63) / 64);
    a.nby = (out_features + 63) / 64;
    hipLaunchKernelGGL((linear_valu<double>), dim3((unsigned)(a.nbx * a.nby)), dim3(256), 0, st, a);
  } else {
    return ZK_EINVAL;
  }
  return ZK_LAUNCH_CHECK();
}
