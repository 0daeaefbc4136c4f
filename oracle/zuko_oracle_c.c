/* Plain-C restatement (double precision, one element at a time) of the univariate maps and flows of the hot path of
 * probabilists/zuko 1.6.0: MonotonicRQSTransform, MonotonicAffineTransform, SOSPolynomialTransform, (Bounded)BernsteinTransform,
 * masked-autoregressive flows (NSF / MAF) and coupling flows (NICE / RealNVP).
 * TEST INFRASTRUCTURE ONLY — an independent, torch-free cross-check of the golden vectors under tests/golden/ (the
 * PyTorch-ops oracle oracle/zuko_oracle.py is the one pinned bitwise against the live reference; this file shares no code
 * and no math library with it).  Built by oracle/build_c.py with gcc; exercised by tests/test_oracle_c.py.
 *
 * Reference lines (relative to /root/reference/zuko):
 *   softclip of widths / heights / derivatives      transforms.py:480-482
 *   softmax, left pad, cumsum, knots in [-B, B]     transforms.py:484-489
 *   exp(derivatives), boundary slopes 1             transforms.py:486, 490
 *   strict searchsorted, mask, k mod K, gather      transforms.py:499-523
 *   forward value and log|dy/dx|                    transforms.py:554-567
 *   inverse (stable quadratic root)                 transforms.py:534-548
 *   affine: log_scale softclip, forward, inverse    transforms.py:436-446
 */
#include <math.h>
#include <stdint.h>

#define KMAX 64

static void knots(const double* w, const double* h, const double* d, int K, double bound, double slope, double* kx, double* ky, double* kd) {
  const double ls = log(slope);
  const double* src[2] = {w, h};
  double* dst[2] = {kx, ky};
  for (int ax = 0; ax < 2; ++ax) {
    double v[KMAX], m = -INFINITY, s = 0.0, cum = 0.0;
    for (int j = 0; j < K; ++j) {
      v[j] = src[ax][j] / (1.0 + fabs(2.0 * src[ax][j] / ls));
      if (v[j] > m) m = v[j];
    }
    for (int j = 0; j < K; ++j) { v[j] = exp(v[j] - m); s += v[j]; }
    dst[ax][0] = bound * (2.0 * cum - 1.0);
    for (int j = 0; j < K; ++j) { cum += v[j] / s; dst[ax][j + 1] = bound * (2.0 * cum - 1.0); }
  }
  kd[0] = 1.0; kd[K] = 1.0;
  for (int j = 1; j < K; ++j) { const double t = d[j - 1] / (1.0 + fabs(d[j - 1] / ls)); kd[j] = exp(t); }
}

static int64_t locate(const double* ks, int K, double v, int* inside) {
  int64_t cnt = 0;
  for (int j = 0; j <= K; ++j) cnt += (ks[j] < v) ? 1 : 0;
  const int64_t k = cnt - 1;
  *inside = (k >= 0) && (k < K);
  return k;
}

/* y, ladj, k for n elements; parameters of element i at w + i*K, h + i*K, d + i*(K-1) */
void zoc_rqs_forward(int64_t n, int K, double bound, double slope, const double* x, const double* w, const double* h, const double* d, double* y,
                     double* ladj, int64_t* kout) {
  for (int64_t i = 0; i < n; ++i) {
    double kx[KMAX + 1], ky[KMAX + 1], kd[KMAX + 1];
    knots(w + i * K, h + i * K, d + i * (K - 1), K, bound, slope, kx, ky, kd);
    int inside;
    const int64_t k = locate(kx, K, x[i], &inside);
    const int64_t kw = ((k % K) + K) % K;
    const double x0 = kx[kw], x1 = kx[kw + 1], y0 = ky[kw], y1 = ky[kw + 1], d0 = kd[kw], d1 = kd[kw + 1];
    const double m = inside ? 1.0 : 0.0;
    const double s = (y1 - y0) / (x1 - x0);
    const double z = m * (x[i] - x0) / (x1 - x0);
    const double den = s + (d0 + d1 - 2.0 * s) * z * (1.0 - z);
    const double yy = y0 + (y1 - y0) * (s * z * z + d0 * z * (1.0 - z)) / den;
    const double jac = s * s * (2.0 * s * z * (1.0 - z) + d0 * (1.0 - z) * (1.0 - z) + d1 * z * z) / (den * den);
    y[i] = inside ? yy : x[i];
    ladj[i] = m * log(jac);
    kout[i] = k;
  }
}

void zoc_rqs_inverse(int64_t n, int K, double bound, double slope, const double* yin, const double* w, const double* h, const double* d, double* x,
                     int64_t* kout) {
  for (int64_t i = 0; i < n; ++i) {
    double kx[KMAX + 1], ky[KMAX + 1], kd[KMAX + 1];
    knots(w + i * K, h + i * K, d + i * (K - 1), K, bound, slope, kx, ky, kd);
    int inside;
    const int64_t k = locate(ky, K, yin[i], &inside);
    const int64_t kw = ((k % K) + K) % K;
    const double x0 = kx[kw], x1 = kx[kw + 1], y0 = ky[kw], y1 = ky[kw + 1], d0 = kd[kw], d1 = kd[kw + 1];
    const double m = inside ? 1.0 : 0.0;
    const double s = (y1 - y0) / (x1 - x0);
    const double y_ = m * (yin[i] - y0);
    const double t = d0 + d1 - 2.0 * s;
    const double a = (y1 - y0) * (s - d0) + y_ * t;
    const double b = (y1 - y0) * d0 - y_ * t;
    const double c = -s * y_;
    const double z = 2.0 * c / (-b - sqrt(b * b - 4.0 * a * c));
    x[i] = inside ? x0 + z * (x1 - x0) : yin[i];
    kout[i] = k;
  }
}

void zoc_affine(int64_t n, double slope, const double* x, const double* shift, const double* scale, double* y, double* ladj, double* x_of_y) {
  const double ls = log(slope);
  for (int64_t i = 0; i < n; ++i) {
    const double lsc = scale[i] / (1.0 + fabs(scale[i] / ls));
    y[i] = x[i] * exp(lsc) + shift[i];
    ladj[i] = lsc;
    x_of_y[i] = (y[i] - shift[i]) / exp(lsc);
  }
}

/* ---- whole-flow check: log p(x | c) of an autoregressive spline flow with explicit weights ----------------------
 * Each transform: phi = MaskedMLP(cat(x, c)) with F.linear(h, mask * W, b) + ReLU between layers
 * (nn.py:217-218, 300-313), spline over each feature from phi[f * (3K-1) ...] (flows/autoregressive.py:149,
 * 212-215), ladj summed over features (transforms.py:210-214); the flow composes the transforms and adds the
 * standard normal log-density of the result (transforms.py:141-150, distributions.py:115-119, 356).  The
 * autoregressive structure lives entirely in the masks, which the caller passes (buffers of the modules).
 * Layout: per transform t, layer l: W[t][l] is [out_l, in_l] row-major, mask likewise (bytes), b[t][l] [out_l];
 * dims[l] for l = 0..L are the layer widths (dims[0] = D + C, dims[L] = D * (3K-1)), identical for all transforms. */
/* K > 0: K-bin spline (NSF); K == 0: affine map with (shift, scale) = phi[f * 2 + (0, 1)] (MAF, transforms.py:436-446) */
void zoc_nsf_log_prob(int64_t n, int D, int C, int T, int L, const int* dims, int K, double bound, double slope, const double* x, const double* c,
                      const double* const* W, const uint8_t* const* M, const double* const* B, double* z_out, double* ladj_out, double* logp) {
  const int total = K > 0 ? 3 * K - 1 : 2;
  int wmax = 0;
  for (int l = 0; l <= L; ++l) wmax = dims[l] > wmax ? dims[l] : wmax;
  double hbuf[2][8192];
  if (wmax > 8192) return;
  for (int64_t i = 0; i < n; ++i) {
    double cur[1024], ladj = 0.0;
    for (int f = 0; f < D; ++f) cur[f] = x[i * D + f];
    for (int t = 0; t < T; ++t) {
      double* in = hbuf[0];
      double* out = hbuf[1];
      for (int f = 0; f < D; ++f) in[f] = cur[f];
      for (int j = 0; j < C; ++j) in[D + j] = c[i * C + j];
      for (int l = 0; l < L; ++l) {
        const double* w = W[t * L + l];
        const uint8_t* m = M[t * L + l];
        const double* b = B[t * L + l];
        const int ni = dims[l], no = dims[l + 1];
        for (int o = 0; o < no; ++o) {
          double acc = 0.0;
          for (int k = 0; k < ni; ++k) acc += in[k] * (m[(int64_t)o * ni + k] ? w[(int64_t)o * ni + k] : 0.0);
          acc += b[o];
          out[o] = (l + 1 < L && acc < 0.0) ? 0.0 : acc;
        }
        double* tmp = in; in = out; out = tmp;
      }
      /* `in` now holds phi[D * total] */
      for (int f = 0; f < D; ++f) {
        double y, lj;
        int64_t k;
        if (K > 0) {
          zoc_rqs_forward(1, K, bound, slope, &cur[f], in + f * total, in + f * total + K, in + f * total + 2 * K, &y, &lj, &k);
        } else {
          const double ls = log(slope), scale = in[f * 2 + 1];
          lj = scale / (1.0 + fabs(scale / ls));
          y = cur[f] * exp(lj) + in[f * 2];
        }
        cur[f] = y;
        ladj += lj;
      }
    }
    double lp = 0.0;
    for (int f = 0; f < D; ++f) {
      z_out[i * D + f] = cur[f];
      lp += -0.5 * cur[f] * cur[f] - 0.91893853320467274178;
    }
    ladj_out[i] = ladj;
    logp[i] = lp + ladj;
  }
}

/* ---- coupling flow (NICE / RealNVP) with affine univariate maps ---------------------------------------------------
 * Each transform (flows/coupling.py:79-139, transforms.py:1010-1073): x_a = x[mask] passes through, phi =
 * MLP(cat(x_a, c)) (dense Linear + ReLU, nn.py:13-22, 122-192), x_b = x[~mask] -> x_b * exp(softclip(scale)) + shift
 * with (shift, scale) = phi[j * 2 + (0, 1)] for the j-th moved feature (transforms.py:436-446), ladj = sum of the
 * soft-clipped log-scales.  masks[t] is the transform's boolean buffer (1 = pass-through). */
void zoc_coupling_affine_log_prob(int64_t n, int D, int C, int T, int L, const int* dims, double slope, const double* x, const double* c,
                                  const uint8_t* const* masks, const double* const* W, const double* const* B, double* z_out, double* ladj_out,
                                  double* logp) {
  const double ls = log(slope);
  double hbuf[2][8192];
  for (int l = 0; l < T * (L + 1); ++l)
    if (dims[l] > 8192) return;  /* dims[t * (L + 1) + l]: the layer widths differ between transforms when D is odd */
  for (int64_t i = 0; i < n; ++i) {
    double cur[4096], ladj = 0.0;
    for (int f = 0; f < D; ++f) cur[f] = x[i * D + f];
    for (int t = 0; t < T; ++t) {
      const uint8_t* mk = masks[t];
      double* in = hbuf[0];
      double* out = hbuf[1];
      int na = 0;
      for (int f = 0; f < D; ++f)
        if (mk[f]) in[na++] = cur[f];
      for (int j = 0; j < C; ++j) in[na + j] = c[i * C + j];
      for (int l = 0; l < L; ++l) {
        const double* w = W[t * L + l];
        const double* b = B[t * L + l];
        const int ni = dims[t * (L + 1) + l], no = dims[t * (L + 1) + l + 1];
        for (int o = 0; o < no; ++o) {
          double acc = 0.0;
          for (int k = 0; k < ni; ++k) acc += in[k] * w[(int64_t)o * ni + k];
          acc += b[o];
          out[o] = (l + 1 < L && acc < 0.0) ? 0.0 : acc;
        }
        double* tmp = in; in = out; out = tmp;
      }
      int j = 0;
      for (int f = 0; f < D; ++f) {
        if (mk[f]) continue;
        const double shift = in[2 * j], scale = in[2 * j + 1];
        const double lsc = scale / (1.0 + fabs(scale / ls));
        cur[f] = cur[f] * exp(lsc) + shift;
        ladj += lsc;
        ++j;
      }
    }
    double lp = 0.0;
    for (int f = 0; f < D; ++f) {
      z_out[i * D + f] = cur[f];
      lp += -0.5 * cur[f] * cur[f] - 0.91893853320467274178;
    }
    ladj_out[i] = ladj;
    logp[i] = lp + ladj;
  }
}

/* ---- sum-of-squares polynomial transform (transforms.py:927-963, 878-924; utils.py:349-363, 170-180) -------------
 * g(x) = mean_k (1 + sum_j a[k][j] (x / 10)^j)^2 + slope;  f(x) = int_0^x g by Gauss-Legendre with the caller's nodes /
 * weights on [0, 1] (n = L + 1 of them, numpy.leggauss in the reference);  ladj = log g(x);  inverse by `nbis` halvings of
 * [-10, 10] followed by the midpoint.  a: [n, P, L1] row-major. */
static double sos_g1(const double* a, int P, int L1, double slope, double x) {
  const double u = x / 10.0;
  double acc = 0.0;
  for (int k = 0; k < P; ++k) {
    double p = 1.0, pw = 1.0;
    for (int j = 0; j < L1; ++j) { p += a[k * L1 + j] * pw; pw *= u; }
    acc += p * p;
  }
  return acc / P + slope;
}
static double sos_f1(const double* a, int P, int L1, double slope, const double* nodes, const double* weights, int nn, double x) {
  double s = 0.0;
  for (int i = 0; i < nn; ++i) s += weights[i] * sos_g1(a, P, L1, slope, 0.0 + nodes[i] * (x - 0.0));
  return (x - 0.0) * s;
}
void zoc_sos(int64_t n, int P, int L1, double slope, const double* nodes, const double* weights, int nn, int nbis, const double* x, const double* a,
             const double* yin, double* y, double* ladj, double* x_inv) {
  for (int64_t i = 0; i < n; ++i) {
    const double* ai = a + i * P * L1;
    y[i] = sos_f1(ai, P, L1, slope, nodes, weights, nn, x[i]);
    ladj[i] = log(sos_g1(ai, P, L1, slope, x[i]));
    double lo = -10.0, hi = 10.0;
    for (int it = 0; it < nbis; ++it) {
      const double c = (lo + hi) / 2;
      if (sos_f1(ai, P, L1, slope, nodes, weights, nn, c) < yin[i]) lo = c; else hi = c;
    }
    x_inv[i] = (lo + hi) / 2;
  }
}

/* ---- Bernstein polynomial transforms (transforms.py:640-777 unbounded, :780-831 bounded) --------------------------
 * theta (constrained): unbounded = cumsum(cat(t_0, softplus(t_1), softplus(t_1..t_{M-1}), softplus(t_{M-1}))) - M ln2 / 2
 * (:703-727); bounded = cumsum(cat(-B, d, d, softmax(t) (2B - 4d), d, d)), d = 2B / (M + 4) (:797-818).
 * f(u) = sum_i C(n, i) u^i (1 - u)^(n - i) theta_i with u = (x + B) / 2B (:729-740; the reference's mean of Beta pdfs is
 * this sum), linear extrapolation outside [eps, 1 - eps] (:742-760; bounded: offsets -+B, slopes 2B, :820-831);
 * ladj = log f'(x) — the reference differentiates by autograd, here the closed form n sum C(n-1, i) (theta_{i+1} - theta_i)
 * u^i (1-u)^(n-1-i) / 2B inside and slope / 2B outside. */
static double binom(int n, int k) { double r = 1.0; for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i; return r; }
static double bern_eval(const double* th, int n, double u) {  /* order n: n + 1 coefficients */
  double s = 0.0;
  for (int i = 0; i <= n; ++i) s += binom(n, i) * pow(u, i) * pow(1.0 - u, n - i) * th[i];
  return s;
}
void zoc_bernstein(int64_t n, int M, int bounded, double bound, const double* x, const double* theta_unc, double* y, double* ladj, double* theta_out) {
  const double eps = 1e-6;
  for (int64_t e = 0; e < n; ++e) {
    const double* t = theta_unc + e * M;
    double th[80], dth[80];
    int nc;
    if (bounded) {
      nc = M + 5;
      const double d = 2.0 * bound / (M + 4);
      double mx = t[0], s = 0.0, cum;
      for (int i = 1; i < M; ++i) mx = t[i] > mx ? t[i] : mx;
      for (int i = 0; i < M; ++i) s += exp(t[i] - mx);
      cum = -bound; th[0] = cum;
      cum += d; th[1] = cum;
      cum += d; th[2] = cum;
      for (int i = 0; i < M; ++i) { cum += exp(t[i] - mx) / s * (2.0 * bound - 4.0 * d); th[3 + i] = cum; }
      cum += d; th[3 + M] = cum;
      cum += d; th[4 + M] = cum;
    } else {
      nc = M + 2;
      const double shift = log(2.0) * M / 2.0;
      double diffs[80];
      diffs[0] = t[0];
      diffs[1] = log1p(exp(t[1]));
      for (int i = 1; i < M; ++i) diffs[1 + i] = log1p(exp(t[i]));
      diffs[M + 1] = log1p(exp(t[M - 1]));
      double cum = 0.0;
      for (int i = 0; i < nc; ++i) { cum += diffs[i]; th[i] = cum - shift; }
    }
    const int order = nc - 1;
    for (int i = 0; i < order; ++i) dth[i] = order * (th[i + 1] - th[i]);
    for (int i = 0; i < nc; ++i) theta_out[e * nc + i] = th[i];
    double off_lo, off_hi, slp_lo, slp_hi;
    if (bounded) { off_lo = -bound; off_hi = bound; slp_lo = slp_hi = 2.0 * bound; }
    else {
      off_lo = bern_eval(th, order, eps); off_hi = bern_eval(th, order, 1.0 - eps);
      slp_lo = bern_eval(dth, order - 1, eps); slp_hi = bern_eval(dth, order - 1, 1.0 - eps);
    }
    const double u = (x[e] + bound) / (2.0 * bound);
    double f, df;
    if (u <= eps) { f = slp_lo * (u - eps) + off_lo; df = slp_lo; }
    else if (u >= 1.0 - eps) { f = slp_hi * (u - 1.0 + eps) + off_hi; df = slp_hi; }
    else { f = bern_eval(th, order, u); df = bern_eval(dth, order - 1, u); }
    y[e] = f;
    ladj[e] = log(df / (2.0 * bound));
  }
}
