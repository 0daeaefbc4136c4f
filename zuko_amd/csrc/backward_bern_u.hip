// zuko_amd — the unbounded instantiation of the Bernstein adjoint (see zk_bern_bwd.h / backward_bern.hip).
#include "zk_bern_bwd.h"

namespace zk {

void bern_bwd_launch_unbounded(unsigned grid, void* stream, const PolyBwdArgs& a) {
  hipLaunchKernelGGL((bern_backward_kernel<18, 16, false>), dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
}

}  // namespace zk
