/* zuko_amd — C ABI of the MI355X-native zuko transform hot path (libzuko_amd.so).
 *
 * The reference (probabilists/zuko v1.6.0) is pure Python on PyTorch and has NO FFI of its own;
 * its extension points are Python constructor hooks.  Each entry point below
 * therefore replaces a *sequence of ATen ops* at a named place in the reference, and is what a
 * ctypes binding in `zuko/transforms.py` / `zuko/nn.py` would call.
 *
 * Conventions (all entry points):
 *   - return value: hipError_t as int, 0 = success (hipErrorInvalidValue for unsupported arguments);
 *     the Python side raises RuntimeError on non-zero.  Value-domain problems (NaN, out-of-range x)
 *     are never errors: they propagate exactly as in the reference (Distribution._validate_args =
 *     False, zuko/distributions.py:35).
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates); the library
 *     allocates nothing, keeps no global state, never synchronises, and only enqueues work on
 *     `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - dtype: ZK_DTYPE_F32 (0) or ZK_DTYPE_F64 (1); all tensors of one call share it.
 *   - x / y are row-major contiguous [N, D].  Parameter tensors are addressed as
 *         p[n * sN + d * sD + j],   j contiguous,
 *     with element strides (sN, sD); 0 = broadcast.  When the segments of a transform form one
 *     packed buffer phi[N, D, total] — the layout the conditioner's last layer emits
 *     (zuko/flows/autoregressive.py:149,212-213; zuko/utils.py:616-622) — the LDS-staged streaming
 *     kernel is used, otherwise a strided-gather instantiation.
 *   - ladj: if `ladj_reduced` the output is ladj[N] summed over D (what
 *     DependentTransform(., 1) returns, zuko/transforms.py:210-214), else ladj[N, D]
 *     (Transform.log_abs_det_jacobian).
 */
#ifndef ZUKO_AMD_H
#define ZUKO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_DTYPE_F32 0
#define ZK_DTYPE_F64 1
#define ZK_DTYPE_BF16 2 /* zk_rqs_forward / zk_rqs_inverse only: bf16 x / phi / y in HBM, fp32 arithmetic, fp32 ladj;
                           packed 16-byte aligned phi, N*D % 64 == 0, D | 64 or 64 | D, K in {4, 8, 16} */

/* activations of zk_linear / zk_ar_* (zuko/nn.py:168-169: ReLU is the default) */
#define ZK_ACT_NONE 0
#define ZK_ACT_RELU 1
#define ZK_ACT_ELU 2
#define ZK_ACT_TANH 3
#define ZK_ACT_SILU 4
#define ZK_ACT_GELU 5
#define ZK_ACT_SIGMOID 6
#define ZK_ACT_LEAKY 7

/* ---- MonotonicRQSTransform ------------------------------------------------------------------ */

/* Replaces MonotonicRQSTransform.__init__ + call_and_ladj (zuko/transforms.py:469-490, 554-567):
 * softclip, softmax, pad, cumsum, exp, bin search (strict <), gathers, rational-quadratic value and
 * log-derivative in one pass.  widths/heights are [N, D, K], derivs [N, D, K-1].
 * bin_out (optional, int32 [N, D]) receives k = #(knots < x) - 1 in [-1, K]. */
int zk_rqs_forward(int dtype, int64_t N, int64_t D, int K, double bound, double slope, const void* x,
                   const void* widths, int64_t w_sN, int64_t w_sD, const void* heights, int64_t h_sN, int64_t h_sD,
                   const void* derivs, int64_t d_sN, int64_t d_sD, void* y, void* ladj, int ladj_reduced,
                   int32_t* bin_out, void* stream);

/* Replaces MonotonicRQSTransform.__init__ + _inverse (zuko/transforms.py:469-490, 534-548). */
int zk_rqs_inverse(int dtype, int64_t N, int64_t D, int K, double bound, double slope, const void* y,
                   const void* widths, int64_t w_sN, int64_t w_sD, const void* heights, int64_t h_sN, int64_t h_sD,
                   const void* derivs, int64_t d_sN, int64_t d_sD, void* x, int32_t* bin_out, void* stream);

/* Test entry: same evaluation from ALREADY-CONSTRAINED knots (horizontal, vertical, slopes, each
 * [N, D, K+1] with common strides) — zuko/transforms.py:521-567 only.  Used to assert the
 * bit-exact bin index on knots shared with the oracle. */
int zk_rqs_from_knots(int dtype, int inverse, int64_t N, int64_t D, int K, const void* in, const void* horizontal,
                      const void* vertical, const void* slopes, int64_t k_sN, int64_t k_sD, void* out, void* ladj,
                      int32_t* bin_out, void* stream);

/* Diagnostic twin of zk_rqs_forward / zk_rqs_inverse (fp32, K in {4, 8, 16}): the SAME per-element arithmetic the product
 * kernels run, additionally writing bin_out[N, D] (k = #(knots < v) - 1, zuko/transforms.py:521-523) and
 * knots_out[N, D, K+1], the knots of the searched axis exactly as the bin search compared them.  With these the parity
 * tests assert the product bin index on the kernel's OWN knots (exact) and bound every disagreement with the oracle's
 * index by the distance between the two sets of knots.  ladj is [N, D] (NULL for the inverse). */
int zk_rqs_diag(int inverse, int64_t N, int64_t D, int K, double bound, double slope, const void* in, const void* widths,
                int64_t w_sN, int64_t w_sD, const void* heights, int64_t h_sN, int64_t h_sD, const void* derivs, int64_t d_sN,
                int64_t d_sD, void* out, void* ladj, int32_t* bin_out, float* knots_out, void* stream);

/* ---- MonotonicAffineTransform (zuko/transforms.py:412-446) -------------------------------------- */
int zk_affine_forward(int dtype, int64_t N, int64_t D, double slope, const void* x, const void* shift, int64_t s_sN,
                      int64_t s_sD, const void* scale, int64_t c_sN, int64_t c_sD, void* y, void* ladj, int ladj_reduced,
                      void* stream);
int zk_affine_inverse(int dtype, int64_t N, int64_t D, double slope, const void* y, const void* shift, int64_t s_sN,
                      int64_t s_sD, const void* scale, int64_t c_sN, int64_t c_sD, void* x, void* stream);

/* ---- SOSPolynomialTransform (zuko/transforms.py:927-963, zuko/utils.py:349-363, :170-180) ------- *
 * a_coef is [N, D, P, L1] (L1 = degree + 1 = number of Gauss-Legendre nodes); gl_nodes01 /
 * gl_weights01 are HOST arrays of L1 nodes / weights on [0, 1] (numpy leggauss, utils.py:337-339).
 * `constant` (optional, [N, D]) is the additive shift of flows/polynomial.py:23-29. */
int zk_sos_forward(int dtype, int64_t N, int64_t D, int P, int L1, double slope, const double* gl_nodes01,
                   const double* gl_weights01, const void* x, const void* a_coef, int64_t a_sN, int64_t a_sD,
                   const void* constant, int64_t c_sN, int64_t c_sD, void* y, void* ladj, int ladj_reduced, void* stream);
int zk_sos_inverse(int dtype, int64_t N, int64_t D, int P, int L1, double slope, const double* gl_nodes01,
                   const double* gl_weights01, int n_bisect, const void* y, const void* a_coef, int64_t a_sN,
                   int64_t a_sD, const void* constant, int64_t c_sN, int64_t c_sD, void* x, void* stream);

/* ---- BernsteinTransform / BoundedBernsteinTransform (zuko/transforms.py:640-831) ---------------- *
 * theta is the UNCONSTRAINED [N, D, M]; NC = M + 2 (unbounded) or M + 5 (bounded) constrained coefficients.  NC in
 * {6, 8, 10, 13, 14, 18, 21, 22, 34, 37} run register-resident de Casteljau instantiations, any other NC <= 72 a
 * generic (slower) kernel with the same arithmetic; larger NC returns hipErrorInvalidValue. */
int zk_bernstein_forward(int dtype, int64_t N, int64_t D, int M, int bounded, double bound, double eps, const void* x,
                         const void* theta, int64_t t_sN, int64_t t_sD, void* y, void* ladj, int ladj_reduced, void* stream);
int zk_bernstein_inverse(int dtype, int64_t N, int64_t D, int M, int bounded, double bound, double eps, int n_bisect, const void* y,
                         const void* theta, int64_t t_sN, int64_t t_sD, void* x, void* stream);

/* ---- conditioner layer: F.linear(x, mask * weight, bias) + activation (zuko/nn.py:217-218, 13-15) - *
 * x [N, in] with row stride ldx, weight [out, in] contiguous, mask (uint8/bool [out, in]) and bias
 * optional, y [N, out] with row stride ldy.  fp32 runs on v_mfma_f32_32x32x2_f32. */
int zk_linear(int dtype, int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* weight,
              const uint8_t* mask, const void* bias, int act, void* y, int64_t ldy, void* stream);

/* bf16 conditioner layer (cfg5 of BASELINE.json: NSF(1024, K=16, H=[1024]^3) in bf16).  Same contract as
 * zk_linear (zuko/nn.py:217-218 + the following activation) with bf16 x / weight / bias / y and fp32
 * accumulation on v_mfma_f32_32x32x16_bf16.  `weight` is the ALREADY MASKED matrix (mask * W, one pass
 * over the parameters); `tile_live_mask` is null or one 64-bit word per panel of 256 outputs: bit k clear =
 * the [256 outputs x inputs 64k .. 64k+63] weight tile is entirely zero and is skipped (in_features <= 4096).
 * Requires in_features % 64 == 0, ldx % 8 == 0 and 16-byte aligned x / weight; returns hipErrorInvalidValue
 * otherwise. */
int zk_linear_bf16(int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* weight, const uint64_t* tile_live_mask,
                   const void* bias, int act, void* y, int64_t ldy, void* stream);

/* Last conditioner layer of a bf16 autoregressive SPLINE transform with the spline evaluated in the GEMM's
 * epilogue — replaces `phi = hyper[-1](h)` + `MonotonicRQSTransform(*unpack(phi)).call_and_ladj(x)` + the
 * feature sum (zuko/flows/autoregressive.py:211-218, zuko/transforms.py:469-567, :210-214) without writing
 * phi[N, D, 3K-1] (50 GB per transform at cfg5).  `weight_panels` / `bias_panels` hold the (masked) weight rows
 * regrouped in panels of 256: panel p = the 3K-1 parameter rows of features p*FP .. p*FP+FP-1, FP = 256 / (3K-1),
 * zero rows behind (zuko_amd/nn.py:_Bf16Plan builds them); `tile_live_mask` as in zk_linear_bf16, one word per
 * panel.  h[N, in] is the last hidden activation, x / y[N, features] the transform's input / output (bf16),
 * `partial` a caller-owned fp32 workspace [panels, N], ladj[N] fp32 = sum over features of log|dy/dx|.
 * K in {8, 16}. */
int zk_linear_bf16_rqs(int64_t N, int in_features, int panels, const void* h, int64_t ldh, const void* weight_panels, const uint64_t* tile_live_mask,
                       const void* bias_panels, int K, int features, double bound, double slope, const void* x, int64_t ldx, void* y, int64_t ldy,
                       float* partial, float* ladj, void* stream);
/* The same layer + spline on the second-generation kernel (round 5; csrc/linear_bf16_lanes.hip): the weight rows are ordered so that a LANE's
 * accumulators hold every parameter of a whole feature of its samples, and the spline runs on those registers (no LDS image of the tile, no
 * workgroup barrier in the epilogue).  Panels have 192 rows and FPP = 4 (K = 16) or 8 (K = 8) features; with TS = 48 (K = 16) or 24 (K = 8) slots
 * per feature and FPL = 48 / TS, row o of panel p holds
 *     wn = o / 96, c = o % 96, j = c / 32, q = (c % 32) / 8, kg = (c % 8) / 4, t = c % 4, slot s = 16 j + 4 q + t,
 *     feature p FPP + wn 2 FPL + kg FPL + s / TS, parameter s % TS      (a zero row when parameter >= 3K - 1 or feature >= features)
 * (zuko_amd/nn.py: _Bf16Plan.spline_lane_panels); panels = ceil(features / FPP); in_features % 32 == 0.  tile_live_mask[p]: bit k = inputs
 * [32 k, 32 k + 32) of the panel carry a non-zero weight (NULL, or in_features > 2048: everything is multiplied).  `partial`: fp32 workspace
 * [2 * panels, N].  Other arguments and results as zk_linear_bf16_rqs; y is bit-identical to it and to the unfused path. */
int zk_linear_bf16_rqs_lanes(int64_t N, int in_features, int panels, const void* h, int64_t ldh, const void* weight_panels,
                             const uint64_t* tile_live_mask, const void* bias_panels, int K, int features, double bound, double slope, const void* x,
                             int64_t ldx, void* y, int64_t ldy, float* partial, float* ladj, void* stream);

/* ---- fused masked-autoregressive layer (the dominant kernel of NSF / MAF log_prob) ----------------- *
 * Replaces, for one MaskedAutoregressiveTransform (zuko/flows/autoregressive.py:207-218 `meta` +
 * zuko/transforms.py:1005-1007 `call_and_ladj`):  phi = MaskedMLP(cat(x, c)) (zuko/nn.py:217-218 per
 * layer + activations), univariate(*unpack(phi)).call_and_ladj(x) (zuko/transforms.py:554-567 RQS /
 * :436-446 affine) and the feature sum of zuko/transforms.py:210-214 — one launch, fp32,
 * v_mfma_f32_16x16x4_f32, activations register-resident, phi never written to HBM.
 *
 *   uni_kind  0 = MonotonicAffineTransform (total 2); 1 / 2 / 3 = MonotonicRQSTransform with 8 / 4 / 16 bins;
 *             4 = circular 8-bin spline of NCSF (zuko/flows/spline.py:65-72, bound = pi).  Kinds 2-4 need
 *             D % 4 == 0 and D <= 128 (LDS-staged x / y tiles).  5 = shifted SOS polynomial with 3 polynomials of degree 4 (SOSPF's defaults,
 *             zuko/flows/polynomial.py:51-70, zuko/transforms.py:905-963; total 16: the 15 coefficients, then the shift; bound = 10, slope, gl_nodes01 /
 *             gl_weights01), 6 = bounded Bernstein polynomial of degree 16 (BPF's default, zuko/transforms.py:779-831; total 17; bound, eps): FORWARD
 *             only and only through zk_ar_forward_static on an operand-split kernel generated for the conditioner (zk_ar_forward rejects them).
 *   x         [N, DIN] row-major, row stride ldx (elements, multiple of 4), 16-byte aligned:
 *             cat(x, c) zero-padded to DIN % 4 == 0; the first D columns are the features
 *   y         [N, D] (row stride ldy); ladj [N] (may be NULL); accumulate != 0 adds to ladj
 *   wstream / bias / skip / featmap / n_layers / n_groups / n_chunks: the plan of zuko_amd/fused.py
 *             (degree-sorted, tile-skipped weight stream of 1 KiB MFMA A-operand images, bias image,
 *             per-group skip bitmasks, feature regrouping of the last layer); wstream and bias are
 *             produced on the device by zk_gather_f32 from the module's weight / mask / bias tensors.
 *   limits    DIN <= 256, every hidden width <= 256, >= 1 hidden layer.
 *
 * ARGUMENT BLOCK.  The fused entry points take ONE versioned struct instead of 20-28 positional arguments (a transposed pair of
 * ints is a silent wrong answer; a mis-named field is a compile error / a Python KeyError): set struct_size = sizeof(the struct)
 * and version = 1, fill the fields the entry point reads (listed per function), leave the rest zero.  The library rejects a
 * version it does not know, and a struct_size larger than its own, with hipErrorInvalidValue.  zk_ar_args_v1 GREW within version 1
 * (phi_packed .. eps were appended after gh3): a block of the earlier size — struct_size == offsetof(zk_ar_args_v1, phi_packed) or anything
 * between that and sizeof — is accepted and the fields it lacks read as zero, so callers built against the earlier header keep working.
 * Behaviour change that came with the growth: zk_ar_forward_train used to ignore y / ladj / bound / slope / accumulate and now honours
 * them when y != NULL (y == NULL keeps the conditioner-only launch).  zuko_amd/_C.py builds its ctypes.Structure classes by
 * parsing THESE definitions, so header and binding cannot drift apart. */
typedef struct zk_ar_args_v1 {
  uint32_t struct_size;    /* sizeof(zk_ar_args_v1) */
  uint32_t version;        /* 1 */
  int32_t uni_kind;        /* see above */
  int32_t D;               /* features */
  int32_t DIN;             /* conditioner inputs: features + context, zero-padded to a multiple of 4 */
  int32_t n_layers;        /* linear layers of the conditioner (hidden + 1) */
  int32_t n_groups;        /* feature groups of the last layer */
  int32_t n_chunks;        /* length of wstream in 24-tile chunks */
  int32_t act;             /* activation code (zuko_amd/nn.py:_act_code; 1 = ReLU) */
  int32_t bias_floats;     /* floats in the bias image */
  int32_t accumulate;      /* != 0: ladj += instead of ladj = */
  int32_t rev;             /* static-shape kernels: 1 = the alternative first-layer pattern (descending feature order) */
  int32_t n_sched;         /* partial sweeps: entries of sched */
  int32_t g0;              /* partial sweeps: last-layer groups [g0, g1) */
  int32_t g1;
  int32_t reserved;        /* 0 */
  int64_t N;               /* rows */
  int64_t ldx;             /* row stride of x (elements, multiple of 4) */
  int64_t ldy;             /* row stride of y (forward) / of y_in (inverse sweeps) */
  int64_t ldo;             /* inverse sweeps: row stride of x_out */
  int64_t ldphi;           /* training forward: row stride of phi */
  double bound;            /* spline support [-bound, bound] */
  double slope;            /* softclip slope (zuko/transforms.py:426-432, :469-477) */
  const void* x;           /* [N, DIN] = cat(x, c) zero-padded; inverse sweeps: the conditioning values x_cond */
  void* y;                 /* forward: [N, D] result */
  void* ladj;              /* [N] or NULL */
  const void* y_in;        /* inverse sweeps: [N, D] values to invert */
  void* x_out;             /* inverse sweeps: [N, D] result (may alias x) */
  const void* wstream;     /* weight stream of the plan (zuko_amd/fused.py); static-shape kernels: the PER-TILE stream */
  const void* bias;        /* bias image */
  const uint32_t* skip;    /* skip words (generic kernel) */
  const int32_t* featmap;  /* feature regrouping of the last layer */
  int32_t* bin_out;        /* diagnostic launch: [N, D] bin indices */
  float* knots_out;        /* diagnostic launch: [N, D, K + 1] search-axis knots */
  const int32_t* sched;    /* partial sweeps: DEVICE array of stream-chunk ids in consumption order */
  const int32_t* olim;     /* partial sweeps: HOST array, per hidden layer the last out-group to compute */
  const void* launcher;    /* static-shape kernels: address of the generated kernel's zk_ars_launch */
  void* h1;                /* training forward: hidden activations [N, width_l] (h2 / h3 NULL beyond n_layers - 1) */
  void* h2;
  void* h3;
  void* phi;               /* training forward: [N, D * total] */
  void* gh1;               /* dgrad chain: gradient of hidden layer l's pre-activations [N, width_l] (outputs; the last one is the input x) */
  void* gh2;
  void* gh3;
  int32_t phi_packed;      /* training launches (zk_ar_forward_train; zk_ar_backward_full always): phi / its gradient in the kernels' packed order, see there */
  int32_t pad_;
  const double* gl_nodes01;   /* uni_kind 5: HOST arrays of the 5 Gauss-Legendre nodes / weights on [0, 1] (zuko/utils.py:328-347), as zk_sos_forward takes them */
  const double* gl_weights01;
  double eps;              /* uni_kind 6: continuation margin of the Bernstein map (zuko/transforms.py:594; 0 = its default 1e-6) */
  double wdescale0;        /* zk_ar_forward_static with a TWO-PART (f16) operand-split kernel: for linear layer 0 .. 3, 2^-e = the inverse of the power */
  double wdescale1;        /* of two zk_gather_split_f16 stored that layer's weights with (0 = not given: such a kernel declines the launch) */
  double wdescale2;
  double wdescale3;
  uint32_t* amax0;         /* zk_ar_forward_train / zk_ar_backward_full (operand-split kernels), optional: DEVICE [ZK_AMAX_WORDS] each (see zk_gemm_f16x2), zeroed by the */
  uint32_t* amax1;         /* caller — forward: the maximum of h1 / h2 / h3 is folded into amax0 / 1 / 2 and of x into amax3; backward: of gh_l into amax_{l-1}... in the order the kernel */
  uint32_t* amax2;         /* writes them (amax_c for chain layer c: the gradient of hidden layer n - 1 - c) and of the packed parameter gradient x_out into amax3.  They let */
  uint32_t* amax3;         /* zk_wgrad_multi form its products from two-part f16 operands (zk_wgrad_layer_v1.g_amax / h_amax) */
} zk_ar_args_v1;

/* y, ladj of one layer on the generic tile-skipping kernel.  Reads: uni_kind, N, D, DIN, x, ldx, y, ldy, ladj, accumulate, wstream,
 * bias, bias_floats, skip, featmap, n_layers, n_groups, n_chunks, act, bound, slope. */
int zk_ar_forward(const zk_ar_args_v1* args, void* stream);
/* zk_ar_forward at the bf16 matrix rate for ANY plan zk_ar_forward covers (no kernel generated for the shape): the generic kernel's run-time
 * skip tests around the operand-split arithmetic of the static-shape split kernels (every f32 operand as three bf16 numbers, six partial
 * products on v_mfma_f32_16x16x32_bf16, f32 accumulation — csrc/fused_ar_gsplit.hip).  Fields as zk_ar_forward, except: wstream is the
 * plan's OPERAND-SPLIT stream (zuko_amd/fused.py: gsplit_gather through zk_gather_split_bf16: per hidden layer, out-group of 4 tiles and
 * live in-PAIR, 4 blocks; per last-layer group and live in-pair, NT blocks; three 1 KiB bf16 images per 16 x 32 block; every layer padded
 * to whole 24-image chunks) and n_chunks its length in 24-image chunks (>= 1).  A pair is live when either of its two skip bits is set.
 * uni_kind 0-4 (2-4: D % 4 == 0 and y 16-byte aligned rows, as zk_ar_forward); forward only.  bin_out + knots_out set (uni_kind 1-3,
 * LDS-staged rows): the diagnostic twin of the same launch, as zk_ar_forward_diag.  The result is BIT-IDENTICAL to the
 * static-shape operand-split kernel of the same conditioner (zk_ar_forward_static), and agrees with zk_ar_forward to f32 rounding. */
int zk_ar_forward_split(const zk_ar_args_v1* args, void* stream);
/* Diagnostic twin of zk_ar_forward for the spline maps (uni_kind 1-3, LDS-staged tiles: D % 4 == 0): the same kernel
 * template and arithmetic plus bin_out[N, D] (int32) and knots_out[N, D, K+1] (fp32), as zk_rqs_diag.  Reads additionally:
 * bin_out, knots_out (accumulate is ignored). */
int zk_ar_forward_diag(const zk_ar_args_v1* args, void* stream);
/* Backward of the conditioner's hidden layers under autograd (what torch.autograd does for MaskedMLP, zuko/nn.py:117-129, layer by
 * layer): for a ReLU network, g_{l} = (g_{l+1} (W_{l+1} * mask_{l+1})) * [h_l > 0] for every hidden layer and the gradient w.r.t. the
 * input, in ONE launch of a generated kernel (`launcher` = zk_ars_dgrad_launch of zuko_amd/static_ar.py:chain_kernel).
 *   x [N, DIN] (row stride ldx) = gradient of the LAST hidden layer's pre-activations, DIN = its width; D = conditioner inputs;
 *   h1.. (read) the forward's hidden activations, gh1.. (written) the gradients of the earlier hidden layers, both [N, width_l] in the
 *   sorted unit order of zk_ar_forward_train; y [N, D] (row stride ldy) = gradient w.r.t. the conditioner's input;
 *   wstream / n_chunks: the kernel's weight stream; n_layers = linear layers of the conditioner (2..4). */
int zk_ar_dgrad_chain(const zk_ar_args_v1* args, void* stream);
/* The same chain from the gradient of the packed parameters: x [N, DIN = features * total] (row stride ldx) = d loss / d phi, and the
 * dgrad of the LAST linear layer is part of the launch (its K = features * total input is streamed from global memory, each element read
 * once).  h1 .. h_{n-1} (read) / gh1 .. gh_{n-1} (written) cover all hidden layers; y [N, D] = gradient w.r.t. the conditioner's input
 * (accumulate != 0: ADDED to what y holds — the univariate map's own d/dx term of an autoregressive transform).
 * `launcher` = zk_ars_dgrad_launch of an operand-split chain kernel (zuko_amd/static_ar.py: chain_split_tables; the products run on the
 * bf16 matrix instruction with every f32 operand split three ways, as zk_gather_split_bf16 describes). */
int zk_ar_dgrad_full(const zk_ar_args_v1* args, void* stream);
/* The whole backward of ONE autoregressive transform y, ladj = univariate(net(cat(x, c))).call_and_ladj(x) up to the weight
 * gradients (what autograd derives from zuko/flows/autoregressive.py:207-218, zuko/transforms.py:436-446 / :480-490, :554-567 and
 * zuko/nn.py:217-218), in one launch of a generated kernel (`launcher` = zk_ars_dgrad_launch of zuko_amd/static_ar.py:
 * chain_split_tables(packed=...)):
 *   reads   x [N, DIN] = cat(x, c) zero-padded as for the forward (ldx; DIN % 4 == 0), phi [N, n_groups * NT * 16] (ldphi) in the PACKED order of zk_ar_forward_train(phi_packed = 1), y_in = d loss / dy [N, D] (row stride ldo),
 *           ladj = d loss / d ladj [N] (READ here), h1 .. h_{n-1} (the forward's hidden activations), wstream / n_chunks (the kernel's
 *           stream of transposed weights), featmap / n_groups (the forward plan's), uni_kind (0 affine, 1 spline with 8 bins), bound, slope;
 *   writes  x_out = d loss / d phi in the same packed order (row stride ldphi; padding slots zero; for the weight gradients, whose row table maps
 *           packed slots to rows of the last linear layer), gh1 .. gh_{n-1}, y = d loss / d cat(x, c) [N, DIN]
 *           (row stride ldy, a multiple of 4; accumulate != 0: added to what y holds) — the chain's input gradient plus the univariate
 *           map's own d/dx term. */
int zk_ar_backward_full(const zk_ar_args_v1* args, void* stream);
/* One sweep of AutoregressiveTransform._inverse (zuko/transforms.py:994-1000, the body of its loop):
 *     x_out = univariate(*unpack(MaskedMLP(x_cond))).inv(y)
 * x (= x_cond) [N, DIN] as for zk_ar_forward (features first, context after), y_in [N, D] (row stride ldy) the values to
 * invert, x_out [N, D] (row stride ldo); x_out may alias x, so `passes` sweeps over one zero-initialised buffer reproduce the
 * reference loop.  Other fields as zk_ar_forward (y, ladj, accumulate unused). */
int zk_ar_inverse_sweep(const zk_ar_args_v1* args, void* stream);
/* Partial inverse sweep (the "wavefront" form of AutoregressiveTransform._inverse, SURVEY 7 hard part 4):
 * like zk_ar_inverse_sweep, but only the features of last-layer groups [g0, g1) are updated and only
 * the prefix of the conditioner they depend on is evaluated.  The plan must be built with
 * group-aligned chunks (zuko_amd/fused.py: build_plan(align_groups=True)); `sched` is a DEVICE array of
 * n_sched stream-chunk ids in consumption order, `olim` a HOST array with, per hidden layer, the last
 * out-group (of 4 tiles) to compute (fused.partial_schedule).  Running it for s = 0..passes-1 on the
 * groups that hold the features of order s gives the same x as `passes` full sweeps. */
int zk_ar_inverse_partial(const zk_ar_args_v1* args, void* stream);
/* ---- fused coupling transform (NICE / RealNVP) ------------------------------------------------------------------------ *
 * Replaces GeneralCouplingTransform.meta + CouplingTransform.call_and_ladj (zuko/flows/coupling.py:128-136,
 * zuko/transforms.py:1040-1048, :1068-1073) with the affine univariate (zuko/transforms.py:436-446): split by index maps,
 * dense MLP (zuko/nn.py:13-15 per layer + activation) with activations register-resident (widths <= 512), affine map of the
 * moved half, merge — one launch, nothing but x / y / ladj touches HBM.
 *   x [N, D] (row stride ldx), ctx [N, C] or NULL, y [N, D], ladj [N] (accumulate != 0 adds)
 *   amap [nit * 16] (device): conditioner input i = column amap[i] of x (>= 0), context column -(2 + amap[i]), or padding (-1)
 *   fmap [n_groups * 8] (device): column of x of every transformed slot, -1 = padding
 *   tiles / widths (host, n_layers - 1 ints): 16-row output tiles and true width of every hidden layer;
 *   bias_off (host, n_layers ints); wstream / bias: zuko_amd/coupling_plan.py (zk_gather_f32 from the module's parameters);
 *   static_ok != 0 allows the shape-specialised instantiation (ReLU, 128 inputs, hidden 512s) when the shapes match; static_ok == 2:
 *   wstream is that shape's OPERAND-SPLIT stream (zuko_amd/coupling_plan.py: split_gather through zk_gather_split_bf16; three bf16 images per
 *   16 x 32 weight block) and the launch goes to the bf16-matrix-instruction kernel (six partial products per f32 product, f32 accumulate). */
typedef struct zk_coupling_args_v1 {
  uint32_t struct_size;    /* sizeof(zk_coupling_args_v1) */
  uint32_t version;        /* 1 */
  int32_t D;               /* features */
  int32_t C;               /* context columns */
  int32_t nit;             /* 16-column input tiles of the conditioner */
  int32_t n_groups;        /* groups of 8 transformed slots */
  int32_t n_layers;        /* linear layers */
  int32_t n_chunks;        /* length of wstream in chunks */
  int32_t act;             /* activation code */
  int32_t bias_floats;
  int32_t accumulate;      /* != 0: ladj += */
  int32_t static_ok;       /* != 0 allows the shape-specialised instantiation when the shapes match; 2 = wstream is the operand-split stream; 3 = the two-part stream (below) */
  int64_t N;
  int64_t ldx;             /* row stride of `in` */
  int64_t ldc;             /* row stride of ctx */
  int64_t ldy;             /* row stride of `out` */
  double slope;
  const void* in;          /* forward: x [N, D]; inverse: y [N, D] */
  const void* ctx;         /* [N, C] or NULL */
  void* out;               /* forward: y; inverse: x */
  void* ladj;              /* [N] or NULL: log|det dy/dx| of the FORWARD map (also from the inverse launch, at the solution) */
  const void* wstream;
  const void* bias;
  const int32_t* bias_off; /* HOST, n_layers ints */
  const int32_t* amap;     /* DEVICE [nit * 16] */
  const int32_t* fmap;     /* DEVICE [n_groups * 8] */
  const int32_t* tiles;    /* HOST, n_layers - 1 ints: 16-row output tiles of every hidden layer */
  const int32_t* widths;   /* HOST, n_layers - 1 ints: true width of every hidden layer */
  double wdescale0;        /* static_ok == 3: wstream is the TWO-PART stream (the operand-split stream's blocks as two f16 images of the layer's weights times a power of two, */
  double wdescale1;        /* zk_gather_split_f16; 16-image chunks) and wdescale_l the inverse of that power for linear layer l = 0 .. 3 (n_layers <= 4): three partial products on */
  double wdescale2;        /* v_mfma_f32_16x16x32_f16, activations scaled per sample inside the kernel (csrc/fused_coupling.hip: coupling_kernel_half) */
  double wdescale3;
} zk_coupling_args_v1;
int zk_coupling_forward(const zk_coupling_args_v1* args, void* stream);
/* The inverse of the same coupling transform (CouplingTransform._inverse, zuko/transforms.py:1050-1056), same plan and arguments:
 * in = y [N, D] -> out = x [N, D]; the conditioner sees the pass-through half, which both directions share, the moved half is mapped
 * back by x_b = (y_b - shift) exp(-softclip(scale)).  ladj (optional) = log|det dy/dx| of the FORWARD map at the solution (negate it
 * for the inverse transform), the convention of zk_ar_inverse_incremental. */
int zk_coupling_inverse(const zk_coupling_args_v1* args, void* stream);
/* INCREMENTAL inverse: the whole loop of AutoregressiveTransform._inverse (zuko/transforms.py:994-1000) in one launch whose
 * multiply-add count is ~1.5x ONE density evaluation (every off-diagonal weight tile is multiplied once per sample; only the
 * diagonal tiles of a 4-feature group are iterated).  Needs the aligned-tile plan of zuko_amd/incremental.py
 * (build_inc_plan returns None for conditioners that do not fit: the caller then uses zk_ar_inverse_partial / _sweep).
 *   y [N, D] values to invert, ctx [N, C] context (NULL when C = 0), x [N, D] result;
 *   ladj [N] (optional) = sum over features of log|dy/dx| of the FORWARD map at x (what rsample_and_log_prob needs,
 *   zuko/distributions.py:129-138);
 *   uni_kind 0 affine, 1 / 2 / 3 spline with 8 / 4 / 16 bins; n_hidden 1..3; D <= 68, D + C <= 256;
 *   bias_off: HOST array of n_hidden + 1 offsets into the bias image; featmap [4 n_groups], prog [n_groups][36]: device. */
typedef struct zk_ar_inc_args_v1 {
  uint32_t struct_size;    /* sizeof(zk_ar_inc_args_v1) */
  uint32_t version;        /* 1 */
  int32_t uni_kind;
  int32_t n_hidden;
  int32_t D;
  int32_t C;
  int32_t n_groups;
  int32_t n_chunks;
  int32_t act;
  int32_t bias_floats;
  int64_t N;
  int64_t ldy;
  int64_t ldc;
  int64_t ldx;
  double bound;
  double slope;
  const void* y;           /* [N, D] values to invert */
  const void* ctx;         /* [N, C] or NULL */
  void* x;                 /* [N, D] result */
  void* ladj;              /* [N] or NULL */
  const void* wstream;
  const void* bias;
  const int32_t* bias_off; /* HOST, n_hidden + 1 ints */
  const int32_t* featmap;  /* DEVICE [4 n_groups] */
  const int32_t* prog;     /* DEVICE [n_groups][36] */
  int32_t half;            /* != 0: wstream is the plan's HALF stream (zuko_amd/incremental.py: half_stream) — the pull blocks (out tile x PAIR of final in tiles) as two */
  int32_t pad_;            /* f16 images of the layer's weights times a power of two (zk_gather_split_f16), multiplied on v_mfma_f32_16x16x32_f16 with three partial */
  double wdescale1;        /* products; wdescale_l = the inverse of that power of two for linear layer l = 1 .. n_hidden.  First layer and diagonal tiles: f32 as before */
  double wdescale2;
  double wdescale3;
  const double* gl_nodes01;   /* uni_kind 5 (shifted SOS polynomial, 3 x 5 coefficients + constant): HOST arrays of the 5 Gauss-Legendre nodes / weights on [0, 1], as zk_sos_forward */
  const double* gl_weights01;
  double eps;              /* uni_kind 6 (bounded Bernstein polynomial of degree 16): continuation margin (0 = 1e-6) */
  int32_t n_bisect;        /* uni_kind 5 / 6: steps of the bisection inverse (zuko/transforms.py:608-617: ceil(log2(2 bound / 1e-6))) */
  int32_t pad2_;
} zk_ar_inc_args_v1;
int zk_ar_inverse_incremental(const zk_ar_inc_args_v1* args, void* stream);
int zk_ar_inc_lds_bytes(int bias_floats, int nit);
/* dynamic LDS bytes zk_ar_forward will request for `variant` and a bias image of `bias_floats` floats. */
int zk_ar_lds_bytes(int variant, int bias_floats);
/* zk_ar_forward through a GENERATED static-shape kernel.  With the hidden units sorted by dependency count the block pattern of
 * a MaskedMLP's masks (zuko/nn.py:270-295) is fixed per (features, context, hidden widths, order, univariate map), so the pass
 * over a wave tile can be straight-line code.  csrc/fused_ar_static_impl.h holds that kernel as a template over a struct of
 * constexpr tables; zuko_amd/static_ar.py writes the tables of a plan into a one-kernel translation unit, compiles it (hipcc,
 * gfx950) into zuko_amd/lib/ars/ars_<signature>.so — ahead of time for the BASELINE.json conditioners, on first use otherwise —
 * and passes the address of its `zk_ars_launch` symbol here as `launcher`.  rev != 0 selects the alternative first-layer pattern
 * the kernel was generated with (a descending feature order mirrors the input tiles).  The kernel consumes the plan's PER-TILE
 * stream (only the 16x16 tiles that hold non-zero weights; ArPlan.fine_gather of zuko_amd/fused.py) of n_chunks chunks, re-checks
 * D / DIN / n_layers / n_groups / n_chunks against the shape it was generated for (hipErrorInvalidValue on a mismatch or on a
 * kernel built against another ArArgs layout) and needs 16-byte addressable rows of y when it stages rows through LDS (D % 4 == 0).
 * Any fusable activation (the kernel checks `act` against the one it was generated for); uni_kind 0 / 1 / 2 / 4; hidden widths up to 512 (the generic zk_ar_forward stops at 256).  Results are
 * bit-identical to zk_ar_forward on the same plan (the tiles the per-tile stream drops hold zeros only). */
/* Reads: launcher, rev, uni_kind, N, D, DIN, x, ldx, y, ldy, ladj, accumulate, wstream, bias, bias_floats, featmap, n_layers, n_groups,
 * n_chunks, bound, slope; bin_out + knots_out (both or neither): the DIAGNOSTIC twin of an operand-split kernel — same arithmetic as the
 * product launch plus the bin index used and the knots searched, as zk_ar_forward_diag (the f32-instruction static kernels return
 * hipErrorInvalidValue for it: they are bit-identical to zk_ar_forward, whose twin serves). */
int zk_ar_forward_static(const zk_ar_args_v1* args, void* stream);
/* Conditioner-only launch of a generated static-shape kernel for the training forward (zuko_amd/train.py): phi [N, D * total] =
 * net(x) in module order (what the last MaskedLinear of zuko/nn.py:221-318 returns) and the hidden activations h_l [N, width_l]
 * (n_layers - 1 <= 3 of them; h2 / h3 may be NULL beyond that) with the units in the stream's dependency-sorted order, which the
 * mask-aware dgrad / wgrad kernels consume.  Same launcher, per-tile stream, bias image, feature map and chunk count as
 * zk_ar_forward_static.  y == NULL: the univariate map is not evaluated; y != NULL (operand-split kernels): the same launch also writes
 * y [N, D] and ladj [N] as zk_ar_forward_static does (reads y, ldy, ladj, accumulate, bound, slope as well), so that a training step
 * reads phi back only in its backward pass.  phi_packed != 0: phi is written in the kernel's PACKED order — row n (stride ldphi >=
 * n_groups * NT * 16) holds at (g NT + t) 16 + 4 q + r parameter 4 t + r of the features that lane q of group g owns (NT = tiles per
 * group, zuko_amd/fused.py: UniLayout; parameters of one feature consecutive, featmap gives the features) — which zk_ar_backward_full reads. */
int zk_ar_forward_train(const zk_ar_args_v1* args, void* stream);
/* dst[i] = idx[i] < 0 ? 0 : (mask && !mask[idx[i]] ? 0 : src[idx[i]]) — builds the weight stream
 * (mask * W gathered into tile images) and the bias image; fp32, n elements. */
int zk_gather_f32(const void* src, const uint8_t* mask, const int32_t* idx, int64_t n, void* dst, void* stream);
/* Weight stream of an operand-split static-shape kernel (zuko_amd/static_ar.py: split_tables; csrc/fused_ar_split_impl.h): every f32
 * weight as three bf16 numbers h + m + l.  idx [n_blocks * 512] int32 (lane-major, 8 per lane, -1 = zero) into src, mask as
 * zk_gather_f32; dst receives three 1 KiB images per block. */
int zk_gather_split_bf16(const void* src, const uint8_t* mask, const int32_t* idx, int64_t n_blocks, void* dst, void* stream);
/* Weight stream of a TWO-PART operand-split kernel (csrc/fused_ar_half_impl.h): as zk_gather_split_bf16, but every weight is first multiplied by
 * `scale` (a power of two chosen by the caller so that the layer's largest magnitude lands in [2^14, 2^15): exact) and written as TWO f16
 * images h = f16(w scale), l = f16(w scale - h) — 2 KiB per block.  The kernel is told 1 / scale through zk_ar_args_v1.wdescale. */
int zk_gather_split_f16(const void* src, const uint8_t* mask, const int32_t* idx, int64_t n_blocks, void* dst, double scale, void* stream);
/* Up to eight of the two gathers above in ONE launch (the streams and bias images of a conditioner are a dozen tiny gathers; training
 * re-gathers them every step).  split != 0: zk_gather_split_bf16 with count = blocks; else zk_gather_f32 with count = elements. */
typedef struct zk_gather_desc_v1 {
  uint32_t struct_size;    /* sizeof(zk_gather_desc_v1) */
  int32_t split;
  int64_t count;
  const void* src;
  const uint8_t* mask;     /* or NULL */
  const int32_t* idx;
  void* dst;
} zk_gather_desc_v1;
int zk_gather_multi(int n, const zk_gather_desc_v1* descs, void* stream);

/* ---- backward (vector-Jacobian products; fp32).  The reference has no backward code: autograd runs
 *      through the ATen ops of zuko/transforms.py:480-490,554-567 (spline), :436-446 (affine),
 *      torch Normal.log_prob (zuko/distributions.py:115-119) and the activations of zuko/nn.py. --------- */
/* kind 0 = affine (phi [N, D, 2] = [shift, scale]), 1 = RQS (phi [N, D, 3K-1], K in {4, 8, 16}).
 * phi / gphi packed and contiguous with the same 16-byte alignment; gy [N, D] (may be NULL);
 * gl [N] if gl_reduced else [N, D] (may be NULL); outputs gx [N, D], gphi [N, D, total]. */
int zk_univariate_backward(int kind, int64_t N, int64_t D, int K, double bound, double slope, const void* x, const void* phi,
                           const void* gy, const void* gl, int gl_reduced, void* gx, void* gphi, void* stream);
/* Adjoints of the polynomial maps (fp32; what autograd derives from zuko/transforms.py:927-963 + zuko/utils.py:297-326 and
 * from :640-831).  SOS: params [N, D, P*L1 (+1)] = [a | constant?] packed, built for P * L1 == 15 (SOSPF default);
 * Bernstein: theta [N, D, M] unconstrained, built for the BPF defaults (bounded M = 17, unbounded M = 16); other sizes
 * return hipErrorInvalidValue.  gy [N, D] / gl ([N] if gl_reduced else [N, D]) may be NULL. */
int zk_sos_backward(int64_t N, int64_t D, int P, int L1, double slope, const double* gl_nodes01, const double* gl_weights01,
                    int has_const, const void* x, const void* params, const void* gy, const void* gl, int gl_reduced, void* gx,
                    void* gparams, void* stream);
int zk_bernstein_backward(int64_t N, int64_t D, int M, int bounded, double bound, double eps, const void* x, const void* theta, const void* gy,
                          const void* gl, int gl_reduced, void* gx, void* gtheta, void* stream);
/* Adjoint seeds of an INVERSE univariate map x = f^{-1}(y) (gradients through rsample; inverse function theorem):
 * gy[e] = gx[e] * exp(-ladj[e]) with ladj = log f'(x), seed[e] = -gy[e] (zk_univariate_backward(x, phi, gy = seed) then
 * yields dL/dphi). */
int zk_inverse_seed(int64_t n, const void* gx, const void* ladj, void* gy, void* seed, void* stream);
/* gz[n, d] = -g[n] * (z[n, d] - loc[d]) / scale[d]^2 */
int zk_diag_normal_backward(int64_t N, int64_t D, const void* z, const void* loc, const void* scale, const void* g, void* gz,
                            void* stream);
/* gin = gout * act'(.) written in terms of the activation OUTPUT y; act in {NONE, RELU, ELU, TANH, SIGMOID, LEAKY}. */
int zk_act_backward(int64_t n, const void* y, const void* gout, int act, void* gin, void* stream);

/* ---- conditioner GEMMs of the training path (fp32, v_mfma_f32_32x32x2_f32), mask-aware by tile skipping --------------- *
 * They replace what autograd derives from `F.linear(x, mask * weight, bias)` + activation (zuko/nn.py:217-218): forward,
 * dgrad and wgrad.  The caller (zuko_amd/train.py) works in a reparametrisation with degree-sorted hidden units, in which
 * the masks are block lower-triangular, and passes skip maps / live-block lists derived from the permuted masks.
 *
 * zk_gemm_f32_skip:  y[N, OUT] = act(x[N, IN] w^T + bias) (* act'_{gate_act}(gate[N, OUT]) when gate != NULL).
 *   w [OUT, IN] row-major, ALREADY masked.  kskip: NULL or one 64-bit word per block of 128 outputs, bit kt set = the
 *   32-input k tile kt of that block holds non-zero weights (IN <= 2048).  act / gate_act in {NONE, RELU, ELU, TANH,
 *   SIGMOID, LEAKY}; act' is evaluated on the activation OUTPUT stored in `gate` (dgrad: w = Ws^T, gate = saved h).
 * zk_wgrad_f32:  dw[OUT, IN] (+)= mask .* (g[N, OUT]^T h[N, IN]) restricted to the 128 x 128 blocks listed in `pairs`
 *   (device int32 [npairs][2] = (out block, in block)); other blocks are not written.  Split over
 *   zk_wgrad_slices(N, npairs) sample slices whose partial blocks go to `partial` (>= slices * npairs * 16384 floats)
 *   and are summed in slice order (deterministic).  mask: uint8 [OUT, IN] or NULL.  rows / cols (device int32 [OUT] / [IN], or
 *   NULL = identity): the product is formed on a row / column permuted weight (units sorted by dependency count) and element (o, c)
 *   is WRITTEN to dw[rows[o], cols[c]], i.e. where the module keeps that weight — no scatter pass afterwards.  rows[o] < 0: column o of g is a
 *   padding slot (the packed gradient of zk_ar_backward_full, always zero): nothing is written for it; dw then has max(rows) + 1 rows, not OUT.
 * zk_colsum_f32:  out[C] (+)= sum_n x[n, c] (bias gradients); workspace >= zk_colsum_slices(N) * C floats. */
int zk_gemm_f32_skip(int64_t N, int in_features, int out_features, const void* x, int64_t ldx, const void* w, const uint64_t* kskip,
                     const void* bias, int act, const void* gate, int64_t ldg, int gate_act, void* y, int64_t ldy, void* stream);
int zk_wgrad_slices(int64_t N, int npairs);
int zk_wgrad_f32(int64_t N, int out_features, int in_features, const void* g, int64_t ldg, const void* h, int64_t ldh,
                 const int32_t* pairs, int npairs, float* partial, const uint8_t* mask, void* dw, int accumulate, const int32_t* rows,
                 const int32_t* cols, void* stream);
/* zk_wgrad_f32 plus the bias gradient db[rows[o]] = sum_n g[n, o] (db[o] without rows) from the same pass over g (replaces a separate zk_colsum_f32 over g;
 * zuko/nn.py:217-218 under autograd).  cs_flag [npairs] (device, uint8) marks ONE pair per 128-row out block — every out block
 * needs one; cs_partial: workspace of zk_wgrad_slices(N, npairs) * ceil(OUT / 128) * 128 floats.  Deterministic. */
int zk_wgrad_bias_f32(int64_t N, int out_features, int in_features, const void* g, int64_t ldg, const void* h, int64_t ldh,
                      const int32_t* pairs, int npairs, float* partial, const uint8_t* mask, void* dw, int accumulate,
                      const uint8_t* cs_flag, float* cs_partial, void* db, const int32_t* rows, const int32_t* cols, void* stream);
/* The same for up to four layers of one conditioner in two launches (the GEMMs of all layers, then all reductions): a layer with two or
 * three live blocks is otherwise a launch that the GPU finishes faster than the host queues the next one.  Products run on the bf16 matrix
 * instruction with three-way split f32 operands (six partial products, f32 accumulate: csrc/train.hip, wgrad_split_kernel).  dw / db are
 * written (not accumulated); partial / cs_partial sized as for zk_wgrad_bias_f32; db, cs_flag, cs_partial all NULL = no bias gradient. */
typedef struct zk_wgrad_layer_v1 {
  uint32_t struct_size;    /* sizeof(zk_wgrad_layer_v1) */
  int32_t out_features;
  int32_t in_features;
  int32_t npairs;          /* live 128 x 128 blocks */
  int64_t ldg;
  int64_t ldh;
  const void* g;           /* [N, out_features] gradient of the layer's output */
  const void* h;           /* [N, in_features] the layer's input */
  const int32_t* pairs;    /* DEVICE [npairs][2] = (out block, in block) */
  void* partial;           /* workspace, zk_wgrad_slices(N, npairs) * npairs * 128 * 128 floats */
  const uint8_t* mask;     /* [out, in] in the operands' (sorted) order, or NULL */
  void* dw;                /* [out, in], written through rows / cols */
  const uint8_t* cs_flag;  /* DEVICE [npairs]: one designated pair per out block, or NULL */
  void* cs_partial;
  void* db;                /* [out], written through rows */
  const int32_t* rows;     /* sorted row -> the module's row, or NULL */
  const int32_t* cols;
  const uint32_t* g_amax;  /* DEVICE [ZK_AMAX_WORDS] maxima of |g| and |h| (see zk_gemm_f16x2 below), or NULL; given for EVERY layer: the products */
  const uint32_t* h_amax;  /* run as two-part f16 operands with per-tensor power-of-two scales (three partial products instead of six) */
} zk_wgrad_layer_v1;
int zk_wgrad_multi(int n_layers, const zk_wgrad_layer_v1* layers, int64_t N, void* stream);
int zk_colsum_slices(int64_t N);
int zk_colsum_f32(int64_t N, int C, const void* x, int64_t ld, float* workspace, void* out, int accumulate, void* stream);

/* ---- DENSE conditioner GEMMs on the f16 matrix instruction with two-part f32 operands (csrc/gemm_half.hip) ------------------------- *
 * Forward and dgrad of the coupling conditioner `MLP` (zuko/nn.py:15 F.linear + activation, zuko/flows/coupling.py:128-136) under training:
 * every f32 operand is scaled by a power of two that puts its TENSOR's largest magnitude in [2^14, 2^15) and written as h + l in f16; a
 * product is three v_mfma_f32_16x16x32_f16 with f32 accumulation (22 significand bits for elements within 2^-18 of the maximum, an
 * absolute error of 2^-40 of the maximum below).  A maximum travels as a DEVICE array of ZK_AMAX_WORDS uint32 — 64 partial maxima, bit
 * patterns of non-negative floats, one per 128-byte line so that the atomicMax of a few thousand wavefronts do not queue on one address;
 * the caller zeroes it, the readers fold it: no host synchronisation between the layers of a chain.
 *
 * zk_amax_f32:    folds max |src[r, c]| into `out` for any number of [rows, cols] (row stride ld) tensors, eight per launch.
 * zk_wsplit_f16:  the lane images of a weight operand W'[unit u, k] = src[u * unit_stride + k * k_stride] (mask likewise, or NULL) for
 *                 `units` x `k`, scaled from *amax: dst receives ceil(units / 128) * ceil(k / 32) * 16 KiB, zero padded.  (unit_stride, k_stride) =
 *                 (in, 1) on a row-major [out, in] weight gives the forward operand, (1, in) with units = in, k = out the dgrad operand W^T.
 * zk_gemm_f16x2:  c[M, N] = act(a[M, K] W'^T + bias) (* act'_{gate_act}(gate[M, N]) when gate != NULL, the derivative in terms of the activation OUTPUT as for
 *                 zk_gemm_f32_skip); act / gate_act in {NONE, RELU, ELU, TANH, SIGMOID, LEAKY}.  a_amax >= max |a|
 *                 and w_amax (the scalar the images were made with) are read on the device; c_amax (or NULL) receives max |c| by atomicMax. */
#define ZK_AMAX_WORDS 2048
typedef struct zk_amax_desc_v1 {
  uint32_t struct_size;    /* sizeof(zk_amax_desc_v1) */
  int32_t cols;
  int64_t rows;
  int64_t ld;
  const void* src;         /* f32 */
  uint32_t* out;           /* DEVICE [ZK_AMAX_WORDS] */
} zk_amax_desc_v1;
int zk_amax_f32(int n, const zk_amax_desc_v1* descs, void* stream);
typedef struct zk_wsplit_desc_v1 {
  uint32_t struct_size;    /* sizeof(zk_wsplit_desc_v1) */
  int32_t units;
  int32_t k;
  int32_t pad_;
  int64_t unit_stride;
  int64_t k_stride;
  const void* src;         /* f32 */
  const uint8_t* mask;     /* or NULL; indexed like src */
  const uint32_t* amax;    /* DEVICE [ZK_AMAX_WORDS]: >= max |src| */
  void* dst;
} zk_wsplit_desc_v1;
int zk_wsplit_f16(int n, const zk_wsplit_desc_v1* descs, void* stream);
int zk_gemm_f16x2(int64_t M, int K, int N, const void* a, int64_t lda, const uint32_t* a_amax, const void* w_images, const uint32_t* w_amax,
                  const void* bias, int act, const void* gate, int64_t ldg, int gate_act, void* c, int64_t ldc, uint32_t* c_amax, void* stream);
/* The split / merge of a coupling transform around its conditioner under training (zuko/transforms.py:1037-1073 `x[..., idx_a]`, `x[..., idx_b]`, merge;
 * autograd's backward of the two index operations is a sort-based scatter: 1.7 ms of a RealNVP training step).  fp32, contiguous outputs.
 * zk_coupling_split:  inp[N, na + C] = [x[:, idx_a] | ctx], xb[N, nb] = x[:, idx_b]; folds max |inp| into inp_amax (the first GEMM's operand scale).
 * zk_coupling_merge:  out[n, d] = half[d] >= 0 ? b[n, half[d]] : base[n, d] (+ add[n, -1 - half[d]] if add != NULL), base == NULL reading as zero;
 *                     half [D] int32: slot of a moved feature in the b half, or -1 - (slot of a kept feature in the a half).  Forward: y = merge(x, y_b);
 *                     backward: g_x = merge(g_y, g_xb, add = the conditioner's input gradient). */
int zk_coupling_split(int64_t N, int D, int C, const void* x, int64_t ldx, const void* ctx, int64_t ldc, const int32_t* idx_a, int na, const int32_t* idx_b, int nb,
                      void* inp, void* xb, uint32_t* inp_amax, void* stream);
int zk_coupling_merge(int64_t N, int D, const void* base, int64_t ldbase, const void* b, int nb, const void* add, int64_t ldadd, const int32_t* half, void* out,
                      void* stream);

/* ---- base density + final reduction (zuko/distributions.py:115-119, 337-363) ---------------------- *
 * out[n] = sum_d Normal(loc[d], scale[d]).log_prob(z[n, d]) (+ ladj[n] if ladj != NULL). */
int zk_diag_normal_log_prob(int dtype, int64_t N, int64_t D, const void* z, const void* loc, const void* scale,
                            const void* ladj, void* out, void* stream);
/* out[0] = scale * sum_i v[i] in f64 (per-rank partial NLL for the RCCL all-reduce);
 * workspace: >= 1024 doubles of device memory. */
int zk_sum_f64(int dtype, int64_t N, const void* v, double scale, double* workspace, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZUKO_AMD_H */
