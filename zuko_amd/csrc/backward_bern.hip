// zuko_amd — backward (vector-Jacobian product) of the (bounded) Bernstein polynomial map (zuko/transforms.py:640-831); see backward_poly.hip
// for how the adjoints are obtained (forward-mode dual numbers through the forward kernels' own device functions).
#include "zk_bern_bwd.h"
#include <stdlib.h>

namespace zk {

void bern_bwd_launch_bounded(unsigned grid, void* stream, const PolyBwdArgs& a) {
  hipLaunchKernelGGL((bern_backward_kernel<22, 17, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
}
void bern_adj_launch(bool bounded, void* stream, const PolyBwdArgs& a) {
  const int64_t nb = (a.N * a.D + 127) / 128;
  const unsigned grid = (unsigned)(nb > 16384 ? 16384 : nb);
  if (bounded) hipLaunchKernelGGL((bern_adjoint_kernel<22, 17, true>), dim3(grid), dim3(128), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((bern_adjoint_kernel<18, 16, false>), dim3(grid), dim3(128), 0, (hipStream_t)stream, a);
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
  // (ZUKO_AMD_POLY_ADJOINT=dual: the forward-mode dual-number kernels the hand adjoints are checked against)
  const char* adj_env = getenv("ZUKO_AMD_POLY_ADJOINT");  // (read per call: the tests switch it)
  const bool dual = adj_env && adj_env[0] == 'd';
  if (!((bounded && M == 17) || (!bounded && M == 16))) return ZK_EINVAL;
  if (!dual) bern_adj_launch(bounded != 0, stream, a);
  else if (bounded) bern_bwd_launch_bounded(grid, stream, a);
  else bern_bwd_launch_unbounded(grid, stream, a);
  return ZK_LAUNCH_CHECK();
}

}  // extern "C"
