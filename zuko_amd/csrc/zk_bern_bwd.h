// zuko_amd — the Bernstein adjoint kernel as a template (backward_bern.hip instantiates the bounded map, backward_bern_u.hip the unbounded
// one: each instantiation differentiates 17-18 dual components through the whole forward map and takes ~90 s to compile).
#pragma once
#include "zk_dual.h"

namespace zk {

// Bernstein: variables 0 = x, 1 .. M = theta (unconstrained); NC = constrained coefficients
template <int NC, int M, bool BOUNDED> __global__ __launch_bounds__(64) void bern_backward_kernel(PolyBwdArgs a) {
  typedef DualN<M + 1> T;
  const int64_t total_e = a.N * a.D;
  for (int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x; e < total_e; e += (int64_t)gridDim.x * 64) {
    const float* pe = a.p + e * M;
    auto ld = [&](int j) { return T::var(pe[j], 1 + j); };
    T th[NC];
    const T bound = T(a.bound);
    if (BOUNDED) bern_theta_bounded<T, NC>(ld, bound, th);
    else bern_theta_unbounded<T, NC>(ld, th);
    const T eps = T(a.eps);
    const BernTails<T> tails = bern_tails<T, NC>(th, BOUNDED, bound, eps);
    T y, dydx;
    bern_fwd<T, NC>(th, tails, bound, T::var(a.x[e], 0), y, dydx, eps);
    const T l = t_log<T>(dydx);
    const float gyv = a.gy ? a.gy[e] : 0.f;
    const float glv = a.gl ? (a.gl_reduced ? a.gl[e / a.D] : a.gl[e]) : 0.f;
    a.gx[e] = gyv * y.d[0] + glv * l.d[0];
    float* ge = a.gp + e * M;
#pragma unroll
    for (int j = 0; j < M; ++j) ge[j] = gyv * y.d[1 + j] + glv * l.d[1 + j];
  }
}

// host-side launchers of the two instantiations (one per translation unit)
void bern_bwd_launch_bounded(unsigned grid, void* stream, const PolyBwdArgs& a);
void bern_bwd_launch_unbounded(unsigned grid, void* stream, const PolyBwdArgs& a);

}  // namespace zk
