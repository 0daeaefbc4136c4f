// zuko_amd — two-part f16 operands with per-tensor power-of-two scales (csrc/gemm_half.hip, csrc/train.hip): the maxima on the device.
#pragma once

#include "zk_common.h"

namespace zk {

typedef _Float16 gh16x8 __attribute__((ext_vector_type(8)));

// A maximum is kept as ZK_AMAX_SLOTS partial maxima, one per 128-byte line (a few thousand wavefronts finishing together would otherwise
// queue on ONE address: 2048 same-address atomics cost 25 us, more than the GEMM they belong to); the readers fold the slots.
#define GH_SLOTS 64
#define GH_SLOT_STRIDE 32  /* uint32 */
__device__ __forceinline__ float gh_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ void gh_amax_put(unsigned* slots, unsigned which, float wave_max) {
  if (wave_max > 0.f) atomicMax(slots + (which % GH_SLOTS) * GH_SLOT_STRIDE, __builtin_bit_cast(unsigned, wave_max));
}
// e with amax 2^e in [2^14, 2^15) (|e| <= 90; zero / non-finite amax: 15); every lane of the wavefront calls it
__device__ __forceinline__ int gh_exp(const unsigned* slots) {
  const float amax = gh_wave_max(__builtin_bit_cast(float, slots[(threadIdx.x & 63) * GH_SLOT_STRIDE]));
  const int e = 15 - __builtin_amdgcn_frexp_expf(amax);
  return __builtin_amdgcn_readfirstlane(e > 90 ? 90 : (e < -90 ? -90 : e));
}

}  // namespace zk
