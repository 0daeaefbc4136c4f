// zuko_amd — backward (vector-Jacobian product) of the (bounded) Bernstein polynomial map (zuko/transforms.py:640-831); see backward_poly.hip
// for how the adjoints are obtained (forward-mode dual numbers through the forward kernels' own device functions).
#include "zk_bern_bwd.h"

namespace zk {

void bern_bwd_launch_bounded(unsigned grid, void* stream, const PolyBwdArgs& a) {
  hipLaunchKernelGGL((bern_backward_kernel<22, 17, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
}

}  // namespace zk

using namespace zk;

extern "C" {

// Bernstein adjoint (fp32): theta [N, D, M] unconstrained; built for the BPF defaults — bounded with M = 17 (22 coefficients)
// and unbounded with M = 16 (18 coefficients).
int zk_bernstein_backward(int64_t N, int64_t D, int M, int bounded, double bound, double eps, const void* x, const void* theta, const void* gy, const void* gl,
                          int gl_reduced, void* gx, void* gtheta, void* stream) {
  if (N <= 0 || D <= 0) return 0;
  if (!(eps > 0.0 && eps < 0.5)) return ZK_EINVAL;
  PolyBwdArgs a{};
  a.N = N; a.D = D; a.x = (const float*)x; a.p = (const float*)theta; a.gy = (const float*)gy; a.gl = (const float*)gl; a.gl_reduced = gl_reduced;
  a.gx = (float*)gx; a.gp = (float*)gtheta; a.total = M; a.bound = (float)bound; a.bounded = bounded; a.eps = (float)eps;
  const int64_t nb = (N * D + 63) / 64;
  const unsigned grid = (unsigned)(nb > 16384 ? 16384 : nb);
  if (bounded && M == 17) bern_bwd_launch_bounded(grid, stream, a);
  else if (!bounded && M == 16) bern_bwd_launch_unbounded(grid, stream, a);
  else return ZK_EINVAL;
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
