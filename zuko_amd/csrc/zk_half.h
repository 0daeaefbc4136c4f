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

// activations of the training GEMMs (codes of zuko_amd/ops.py: ACTIVATIONS; those whose derivative is a function of their OUTPUT v, as csrc/train.hip)
__device__ __forceinline__ float gh_act_fwd(float v, int act) {
  switch (act) {
    case 1: return v < 0.f ? 0.f : v;  // NaN stays NaN, as torch.relu
    case 2: return v > 0.f ? v : expm1f(v);
    case 3: return tanhf(v);
    case 6: return 1.f / (1.f + expf(-v));
    case 7: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}
__device__ __forceinline__ float gh_act_grad_out(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? 1.f : 0.f;
    case 2: return v > 0.f ? 1.f : v + 1.f;
    case 3: return 1.f - v * v;
    case 6: return v * (1.f - v);
    case 7: return v > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}

}  // namespace zk
