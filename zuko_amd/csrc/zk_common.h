// zuko_amd — shared definitions for the gfx950 kernels (device + host side of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZK_DTYPE_F32 0
#define ZK_DTYPE_F64 1
#define ZK_DTYPE_BF16 2  /* storage type of x / phi / y / weights; arithmetic and ladj stay fp32 */

#define ZK_WAVE 64

// Every C-ABI entry point returns a hipError_t as int; launches are checked with hipGetLastError.
#define ZK_LAUNCH_CHECK() ((int)hipGetLastError())
#define ZK_EINVAL ((int)hipErrorInvalidValue)

namespace zk {

template <typename T> __device__ __forceinline__ T t_exp(T v);
template <> __device__ __forceinline__ float t_exp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double t_exp<double>(double v) { return exp(v); }

template <typename T> __device__ __forceinline__ T t_log(T v);
template <> __device__ __forceinline__ float t_log<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double t_log<double>(double v) { return log(v); }

template <typename T> __device__ __forceinline__ T t_log1p(T v);
template <> __device__ __forceinline__ float t_log1p<float>(float v) { return log1pf(v); }
template <> __device__ __forceinline__ double t_log1p<double>(double v) { return log1p(v); }

template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ float t_sqrt<float>(float v) { return sqrtf(v); }  // correctly rounded (no -ffast-math)
template <> __device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }

template <typename T> __device__ __forceinline__ T t_abs(T v);
template <> __device__ __forceinline__ float t_abs<float>(float v) { return fabsf(v); }
template <> __device__ __forceinline__ double t_abs<double>(double v) { return fabs(v); }

// Wave-wide (64 lanes) sum, result valid in every lane.
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Full-wave sum on the DPP cross-lane network (no LDS crossbar traffic): quad swaps, row mirrors, then
// row broadcasts; the total ends up in lane 63.
#define ZK_DPP_ADD(v, ctrl, rmask) \
  (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, true))
__device__ __forceinline__ float wave_sum_dpp_to_lane63(float v) {
  ZK_DPP_ADD(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  ZK_DPP_ADD(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  ZK_DPP_ADD(v, 0x141, 0xf);  // row_half_mirror
  ZK_DPP_ADD(v, 0x140, 0xf);  // row_mirror          -> every lane of a 16-lane row holds the row sum
  ZK_DPP_ADD(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  ZK_DPP_ADD(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return v;
}

// Sum over contiguous lane segments of length `seg` (seg <= 64, segments start at multiples of seg
// counted from lane 0).  After the call the FIRST lane of each segment holds the segment's sum.
template <typename T> __device__ __forceinline__ T segment_sum(T v, int seg, int lane_in_seg) {
  for (int off = 1; off < seg; off <<= 1) {
    T o = __shfl_down(v, off, 64);
    if (lane_in_seg + off < seg) v += o;
  }
  return v;
}

static inline int grid_for(int64_t nblocks) {
  // memory-bound kernels: cap the grid (256 CUs x 8) and grid-stride the rest
  const int64_t cap = 256 * 8;
  return (int)(nblocks < 1 ? 1 : (nblocks < cap ? nblocks : cap));
}

}  // namespace zk
