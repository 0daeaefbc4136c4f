/* Plain-C restatement (double precision, one element at a time) of the two univariate maps that carry the
 * benchmark configurations: MonotonicRQSTransform and MonotonicAffineTransform of probabilists/zuko 1.6.0.
 * TEST INFRASTRUCTURE ONLY — an independent, torch-free cross-check of tests/golden/rqs_f64.npz and
 * affine_f64.npz (the PyTorch-ops oracle oracle/zuko_oracle.py is the one pinned bitwise against the live
 * reference; this file shares no code and no math library with it).  Built by oracle/build_c.py with gcc.
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
