/*
 * torchsde_amd -- C ABI of the MI355X (gfx950) time-stepping hot path.
 *
 * The reference (google-research/torchsde v0.2.6) is pure Python: it has no FFI. The seams this
 * library sits behind are therefore the Python protocols of the reference, and every entry point
 * below names the reference code it replaces (paths relative to /root/reference/torchsde):
 *
 *   tsde_brownian_*      _brownian/brownian_interval.py:589-687  BrownianInterval.__call__
 *                        (+ :188-241 bridge split, :643-676 merge / H->U, :30-32 _randn)
 *   tsde_step_diag       _core/methods/euler.py:29-37, midpoint.py:29-45 (+ base_sde.py:98-99 prod_diagonal)
 *   tsde_step_prod       the same steps when the SDE supplies g_prod / f_and_g_prod (base_sde.py:115-117)
 *   tsde_step_general    euler.py / midpoint.py with base_sde.py:101-102 -> misc.py:62-63 batch_mvp (bmm)
 *   tsde_milstein_*      _core/methods/milstein.py:52-94 (+ base_sde.py:142-158)
 *   tsde_srk_diag_stage  _core/methods/srk.py:57-88 + tableaus/srid2.py
 *   tsde_step_general_w  _core/methods/srk.py:90-111 + tableaus/sra1.py (SRK for additive noise)
 *   tsde_step_shared     both of the above when g is one matrix for the whole batch (misc.py:62-63 `bmm` -> one MFMA product)
 *   tsde_rheun_*         _core/methods/reversible_heun.py:48-144 (reversible Heun and its adjoint)
 *   tsde_aug_update      _core/adjoint.py:97-119 + adjoint_sde.py:111-128,218-230 (augmented state update)
 *   tsde_linear_interp   _core/interp.py:15-18
 *   tsde_error_norm      _core/adaptive_stepping.py:42-76 (error estimate of step doubling, base_solver.py:125-128)
 *   tsde_adaptive_*      _core/base_solver.py:117-142 + adaptive_stepping.py:21-39 (accept / reject and the step-size
 *                        controller, decided on the device between attempts)
 *   tsde_trajectory_*    _core/base_solver.py:114-134 (the whole stepping loop of `integrate`) for SDEs whose
 *                        drift and diffusion are given in closed form instead of as Python callables
 *
 * Conventions
 *   - all tensors are contiguous row-major device buffers; `dtype` is TSDE_F32 or TSDE_F64;
 *   - every launch goes to `stream` (a hipStream_t; NULL = the null stream), never synchronises,
 *     never allocates; the return value is a hipError_t (0 = success) -- tsde_last_error() has text;
 *   - scalars that the reference computes as 0-d tensors in ts.dtype (dt, sqrt(dt) ...) are passed as
 *     doubles holding the already-rounded value and are cast to `dtype` inside the kernel;
 *   - arithmetic follows the reference's operation order with one rounding per operation (no FMA
 *     contraction), so that with identical increments the results are bit-identical to the
 *     reference's CPU tensors wherever the reference itself is a chain of elementwise ops.
 */
#ifndef TORCHSDE_AMD_H
#define TORCHSDE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSDE_ABI_VERSION 2
#define TSDE_F32 0
#define TSDE_F64 1

/* Where a step kernel gets the Brownian increment of its time cell from.
 *   dW == NULL : generate W (and U) in registers from the counter RNG: key = entropy,
 *                element index = elem0 + local index, cell index `cell`, cell width `h`.
 *   dW != NULL : read materialised increments (a foreign BaseBrownian, or a replay).
 * `bcast_d` > 0: the external tensors hold one value per batch row (scalar noise, m = 1). */
typedef struct tsde_noise {
  const void* dW;
  const void* dU;
  uint64_t entropy;
  uint64_t elem0;
  uint32_t cell;
  uint32_t reserved;
  double h;
  int64_t bcast_d;
  const uint64_t* entropy_dev; /* optional DEVICE word overriding `entropy` at run time: lets a captured
                                  HIP graph of a whole solve be replayed with a new Brownian seed */
} tsde_noise_t;

/* One segment of the adjoint's augmented state (y, a_y, or one parameter's a_theta). */
typedef struct tsde_seg {
  void* out;        /* may alias `s` */
  const void* s;    /* current state segment */
  const void* F;    /* drift term or NULL */
  const void* G;    /* diffusion-product term or NULL */
  const void* D;    /* Milstein gdg term or NULL */
  int64_t n;
  double sF, sG, sD; /* +1 / -1: the reference negates f and g_prod for the y segment */
} tsde_seg_t;

/* The step schedule of one whole solve, for the trajectory kernels (all DEVICE pointers).
 * Row k of `step_rows` holds, already rounded to `dtype` exactly as the per-step entry points round their
 * double arguments:  dt_k, dt_k/2, 1/dt_k, sqrt(dt_k), sqrt(h_k), sqrt(h_k/12), h_k, 0   (h_k = width of
 * Brownian cell `cells[k]`; step k must cover exactly that cell; the last slot may carry t_k, the step's start time,
 * which only tsde_trajectory_mlp_general reads). Output j (the j-th requested time after
 * t0) is written once `out_step[j]` steps are complete, as w0*y_prev + w1*y_curr with the weights of row j
 * of `out_w` ((0,1) = the step lands on the output time). `out_step` is ascending. */
typedef struct tsde_traj {
  const void* step_rows;   /* [n_steps][8], dtype */
  const uint32_t* cells;   /* [n_steps] */
  const int32_t* out_step; /* [n_out] */
  const void* out_w;       /* [n_out][2], dtype */
  int32_t n_steps;
  int32_t n_out;
} tsde_traj_t;

int tsde_abi_version(void);
const char* tsde_last_error(void);

/* Host-side (no GPU): the Philox-4x32-10 block function the kernels use; for known-answer tests. */
void tsde_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* Host-side: the 128-bit counter the kernels build for (quad, cell, node, stream). */
void tsde_noise_counter(uint64_t quad, uint32_t cell, uint64_t node, uint32_t stream, uint32_t out[4]);

/* ---- Brownian source ------------------------------------------------------------------------ */

/* Raw standard normals of one tree node: out[i] = N(entropy; elem0+i, cell, node, stream). */
int tsde_brownian_normals(void* out, int64_t n, uint64_t entropy, uint64_t elem0, uint32_t cell, uint64_t node,
                          uint32_t stream_id, int dtype, void* stream);

/* Increment over [a,b] of the Brownian path defined by (entropy, cell edges).
 *   edges : DEVICE pointer to the n_cells+1 cell edges (double); edges[ca] <= a < edges[ca+1],
 *           edges[cb] < b <= edges[cb+1].
 *   W (required), U and H (optional, need have_h): outputs of n elements each.
 *   rootW / rootH : optional user-pinned (W,H) of a single top-level cell (`W=`, `H=` arguments of
 *           BrownianInterval, brownian_interval.py:407-408); needs ca == cb == 0.
 *   max_depth / snap : in-cell dyadic descent limit and leaf rule (0 = exact split, 1 = snap).
 *   entropy_dev : optional device word overriding `entropy` (see tsde_noise_t). */
int tsde_brownian_query(void* W, void* U, void* H, int64_t n, uint64_t entropy, uint64_t elem0,
                        const double* edges, int64_t ca, int64_t cb, double a, double b, const void* rootW,
                        const void* rootH, int have_h, int max_depth, int snap, const uint64_t* entropy_dev,
                        int dtype, void* stream);

/* W (and U, optional) of ONE whole cell written to memory: the aligned fast path of a query, for
 * callers that must hand the increment to user torch code (g_prod, adjoint VJPs). `noise->dW` must be NULL. */
int tsde_cell_increment(void* W_out, void* U_out, int64_t n, const tsde_noise_t* noise, int dtype, void* stream);

/* ---- solver steps --------------------------------------------------------------------------- */

/* y1 = (y0 + cf*f) + cg*(g*dW)          diagonal noise, g:(B,d), n = B*d.
 * Euler: cf=dt, cg=1.  Midpoint stage 1: cf=dt/2, cg=0.5.  Midpoint stage 2: cf=dt, cg=1 on (f',g'). */
int tsde_step_diag(void* y1, const void* y0, const void* f, const void* g, int64_t n, double cf, double cg,
                   const tsde_noise_t* noise, int dtype, void* stream);

/* y1 = (y0 + cf*f) + cg*gp              the SDE supplied the diffusion-vector product itself. */
int tsde_step_prod(void* y1, const void* y0, const void* f, const void* gp, int64_t n, double cf, double cg,
                   int dtype, void* stream);

/* y1 = (y0 + cf*f) + cg*(g . dW)        g:(B,d,m), dW:(B,m); general / additive / scalar noise. */
int tsde_step_general(void* y1, const void* y0, const void* f, const void* g, int64_t B, int64_t d, int64_t m,
                      double cf, double cg, const tsde_noise_t* noise, int dtype, void* stream);

/* y1 = (y0 + (ca*f)*cf) + cg*(g . w)    the same contraction against a weight vector built from (W, U):
 *   weight_mode 0: w = W;  1: w = (cu*U)*rdt;  2: w = (cw*W) + (cu*U)*rdt.
 * This is every stage of SRK for additive noise (SRA1): srk.py:90-111 with tableaus/sra1.py. */
int tsde_step_general_w(void* y1, const void* y0, const void* f, const void* g, int64_t B, int64_t d, int64_t m,
                        double ca, double cf, double cg, int weight_mode, double cw, double cu, double rdt,
                        const tsde_noise_t* noise, int dtype, void* stream);

/* The same two updates when the diffusion is ONE (d, m) matrix S for every batch row -- additive noise returned as
 * `sigma.expand(B, d, m)` (base_sde.py:101-102 -> misc.py:62-63 runs `bmm` over B copies of it; SRA1's stages
 * srk.py:96-109): y1 = (y0 + (ca*f)*cf) + cg*(S . w), i.e. out(B, d) = w(B, m) . S^T as ONE dense product on the matrix
 * cores (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64), increments generated in registers as the A operand, S staged
 * once per block in LDS, epilogue fused; 12*d bytes of HBM traffic per batch row. S: contiguous (d, m) row-major.
 * Matrix cores for m % 4 == 0, m <= 64, d <= 128; other shapes run a per-output kernel (still no copies of S). */
int tsde_step_shared(void* y1, const void* y0, const void* f, const void* S, int64_t B, int64_t d, int64_t m, double ca,
                     double cf, double cg, int weight_mode, double cw, double cu, double rdt, const tsde_noise_t* noise,
                     int dtype, void* stream);

/* Milstein helper: v_out = scale * (W^2 - dt) (ito != 0) or scale * W^2; W_out optional. */
int tsde_milstein_v(void* v_out, void* W_out, int64_t n, double dt, int ito, double scale,
                    const tsde_noise_t* noise, int dtype, void* stream);
/* The cotangent of the diffusion VJP in one pass: out = g * (scale * (W^2 - dt)) (ito != 0) or g * (scale * W^2)
 * (base_sde.py:147-152 `g * v`, with v of milstein.py:56,70), W generated in registers. */
int tsde_milstein_weight(void* out, const void* g, int64_t n, double dt, int ito, double scale,
                         const tsde_noise_t* noise, int dtype, void* stream);
/* y1 = ((y0 + f*dt) + g*W) + gdg        derivative form, diagonal noise. */
int tsde_milstein_diag(void* y1, const void* y0, const void* f, const void* g, const void* gdg, int64_t n,
                       double dt, const tsde_noise_t* noise, int dtype, void* stream);
/* y' = (y0 + dt*f) + g*sqrt_dt (ito) or y0 + g*sqrt_dt (stratonovich)   derivative-free stage. */
int tsde_milstein_gf_prime(void* yp, const void* y0, const void* f, const void* g, int64_t n, double dt,
                           double sqrt_dt, int ito, int dtype, void* stream);
/* y1 = ((y0 + f*dt) + g*W) + ((g'-g)*v)/(2*sqrt_dt), v = W^2 - dt (ito) or W^2. */
int tsde_milstein_gf_diag(void* y1, const void* y0, const void* f, const void* g, const void* gprime, int64_t n,
                          double dt, double sqrt_dt, int ito, const tsde_noise_t* noise, int dtype, void* stream);

/* SRID2 stage kernels (diagonal noise), wrapped around the caller's 3 drift and 4 diffusion evaluations
 * (f0 = f(t0, y0), g0 = g(t0, y0), f1 = f(t0+dt, H0_1), g1 = g(t0+dt/4, H1_1), f2 = f(t0+dt/2, H0_2),
 * g2 = g(t0+dt, H1_2), g3 = g(t0+dt/4, H1_3); srid2.py:21-22). Partial sums travel between the kernels, so a step
 * moves 23 streams (6 + 8 + 6 + 3); every sum is formed in the reference's order (srk.py:70-87):
 *   stage 1: in = {y0, f0, g0}           out = {H0_1, H1_1, H1_2}
 *   stage 2: in = {y0, f0, g0, f1, g1}   out = {H0_2, acc, P}     acc = y1 after s = 0, 1; P = H1_3 after j = 0, 1
 *   stage 3: in = {P, acc, f2, g2}       out = {H1_3, acc'}       acc' = y1 after s = 2 (may alias acc)
 *   stage 4: in = {acc', g3}             out = {y1}
 * (H1_2's j = 1 term has zero coefficients, srid2.py:36,48, and is not added: equal for finite f1, g1.) */
int tsde_srk_diag_stage(int stage, void* const out[3], const void* const in[5], int64_t n, double dt, double rdt,
                        double sqrt_dt, const tsde_noise_t* noise, int dtype, void* stream);

/* Final stage of the Stratonovich predictor-corrector schemes, diagonal noise (or products when `prod`):
 *   mode 0, Heun       (_core/methods/heun.py:35-48):        y1 = y0 + (((dt*(f + fp)) + g*dW) + gp*dW) * 0.5
 *   mode 1, Euler-Heun (_core/methods/euler_heun.py:29-42):  y1 = (y0 + dt*f) + ((g*dW + gp*dW) * 0.5)
 * prod != 0: g / gp already are diffusion-vector products (user g_prod, or results of a contraction). */
int tsde_heun_final(void* y1, const void* y0, const void* f, const void* fp, const void* g, const void* gp, int64_t n,
                    double dt, int mode, int prod, const tsde_noise_t* noise, int dtype, void* stream);

/* Iterated integrals for the general-noise Milstein EXTENSION (the reference's Milstein rejects general noise,
 * milstein.py:25; SURVEY.md section 8 note N1): I[b,k,l] = 0.5*(W_k W_l - [k==l]*dt) + A[b,k,l] (ito != 0) or
 * 0.5*W_k W_l + A (Stratonovich); A may be NULL (commutative noise). The derivative term is then
 * ForwardSDE.dg_ga_jvp_column_sum(t, y, I) (base_sde.py:164-183). */
int tsde_iterated_integrals(void* I, const void* W, const void* A, int64_t B, int64_t m, double dt, int ito, int dtype,
                            void* stream);

/* Derivative-free form of the same EXTENSION: the reference's derivative-free Milstein (milstein.py:58-67: g' evaluated at
 * y0 + dt*f + g*sqrt_dt, gdg = (g' - g) * v / (2*sqrt_dt)) applied per Brownian channel -- the explicit order-1.0 scheme
 * of Kloeden & Platen -- instead of m Jacobian-vector products per step (base_sde.py:164-183):
 *   support:     yk[k,b,i] = (y0[b,i] + dt*f[b,i]) + g[b,i,k]*sqrt_dt   (ito != 0; else (y0 + 0) + g*sqrt_dt),
 *                laid out (m, B, d) so that ONE call of the caller's g on (m*B, d) rows evaluates every supporting state;
 *   correction:  corr[b,i] = (sum_{k,l} (gk[k,b,i,l] - g[b,i,l]) * I[b,k,l]) / sqrt_dt, gk = that call's (m, B, d, m) result,
 *                I from tsde_iterated_integrals. The step is then tsde_step_general(y0, f, g, W) + corr. */
int tsde_milstein_gf_general_support(void* yk, const void* y0, const void* f, const void* g, int64_t B, int64_t d,
                                     int64_t m, double dt, double sqrt_dt, int ito, int dtype, void* stream);
int tsde_milstein_gf_general_correction(void* corr, const void* g, const void* gk, const void* I, int64_t B, int64_t d,
                                        int64_t m, double sqrt_dt, int dtype, void* stream);

/* Davie (foster=0) / Foster (foster=1) approximation of the Levy area of one interval of width h from its
 * (W, H): A:(B,m,m) (_brownian/brownian_interval.py:78-99); antisymmetric noise keyed on (entropy, cell, node). */
int tsde_levy_area(void* A, const void* W, const void* H, int64_t B, int64_t m, double h, int foster, uint64_t entropy,
                   uint64_t elem0, uint32_t cell, uint64_t node, const uint64_t* entropy_dev, int dtype, void* stream);
/* The two calls above fused for the general-noise Milstein step: I[b,k,l] = 0.5*(W_k W_l - [k==l]*dt) + A[b,k,l] with the
 * Davie / Foster A of the same (entropy, cell, node), written directly (A never reaches memory; same bits as
 * tsde_levy_area followed by tsde_iterated_integrals). Served for even m whose m*m + 2m entries fit 12 KB (m <= 54 in float32) with elem0*m % 4 == 0; returns
 * hipErrorNotSupported (801) otherwise -- then make the two calls. */
int tsde_levy_iterated_integrals(void* I, const void* W, const void* H, int64_t B, int64_t m, double h, int foster,
                                 uint64_t entropy, uint64_t elem0, uint32_t cell, uint64_t node,
                                 const uint64_t* entropy_dev, double dt, int ito, int dtype, void* stream);

/* ---- reversible Heun (Stratonovich) and its exact-gradient adjoint: _core/methods/reversible_heun.py ---- */

/* z1 = ((2*y0 - z0) + sign*(f0*dt)) + sign*(g0*dW)    :69 forward (sign=+1), :109 reconstruction (sign=-1) */
int tsde_rheun_z_diag(void* z1, const void* y0, const void* z0, const void* f0, const void* g0, int64_t n, double dt,
                      double sign, const tsde_noise_t* noise, int dtype, void* stream);
/* y1 = (y0 + sign*((f0+f1)*half_dt)) + sign*((g0+g1)*(0.5*dW))    :71 forward, :130-131 reconstruction */
int tsde_rheun_y_diag(void* y1, const void* y0, const void* f0, const void* f1, const void* g0, const void* g1,
                      int64_t n, double half_dt, double sign, const tsde_noise_t* noise, int dtype, void* stream);
/* out = a*x + b*y   (2*y0 - z0, f0 + f1, g0 + g1 for the general-noise variants, which then use tsde_step_general_w) */
int tsde_lincomb2(void* out, const void* x, const void* y, int64_t n, double a, double b, int dtype, void* stream);
/* adjoint step, diagonal noise, before the VJP (:106-117):  af0' = af0 + ay*half_dt ; ag0' = ag0 + ay*(0.5*dW) */
int tsde_rheun_adj_a_diag(void* af0_out, void* ag0_out, const void* ay, const void* af0, const void* ag0, int64_t n,
                          double half_dt, const tsde_noise_t* noise, int dtype, void* stream);
/* adjoint step, diagonal noise, after the VJP (:127,134-137):  az0' = az0 + vjp_z ; ay1 = ay + 2*az0' ; az1 = -az0' ;
 * af1 = ay*half_dt + az0'*dt ; ag1 = ay*(0.5*dW) + az0'*dW */
int tsde_rheun_adj_b_diag(void* ay1, void* az1, void* af1, void* ag1, const void* ay, const void* az0,
                          const void* vjp_z, int64_t n, double dt, double half_dt, const tsde_noise_t* noise, int dtype,
                          void* stream);

/* ---- adjoint, output ------------------------------------------------------------------------ */

/* For each segment: out = ((s + sF*(F*cF)) + sG*(cG*G)) + sD*D   (absent terms skipped). */
int tsde_aug_update(const tsde_seg_t* segs, int nseg, double cF, double cG, int dtype, void* stream);

/* out = w0*ya + w1*yb */
int tsde_linear_interp(void* out, const void* ya, const void* yb, int64_t n, double w0, double w1, int dtype,
                       void* stream);

/* Error norm of adaptive step doubling (adaptive_stepping.py:42-76 `compute_error`, one fused reduction):
 *   out[0] = max(eps, sqrt( sum_i ((yf_i - yh_i) / max(eps, rtol*max(|yf_i|,|yh_i|) + atol))^2 / n ))   as a DOUBLE.
 * `workspace`: device scratch of >= TSDE_ERROR_NORM_WORKSPACE doubles. The summation tree is fixed (no atomics):
 * the value, and with it every accept/reject decision of an adaptive solve, is reproducible. Terms are
 * evaluated in `dtype`, accumulated in double. */
#define TSDE_ERROR_NORM_WORKSPACE 1024
int tsde_error_norm(double* out, double* workspace, const void* y_full, const void* y_half, int64_t n, double rtol,
                    double atol, double eps, int dtype, void* stream);

/* ---- adaptive stepping with the control flow on the device ------------------------------------
 * Replaces the host side of _core/base_solver.py:117-142 + _core/adaptive_stepping.py:21-39: between two attempted
 * steps a one-thread controller kernel reads the error norm, applies the PI controller, accepts or rejects, and
 * writes what the next attempt's kernels need into two device tables, so that the host can enqueue a budget of
 * attempts without synchronising and look at the state once per output time.
 *
 *   ctl  : TSDE_CTL_SIZE doubles (state + the bounds of the two half-step Brownian queries)
 *   scal : TSDE_SCAL_SIZE words of `dtype`: for sub-step s in {0: whole step, 1: first half, 2: second half}, at
 *          s * TSDE_SUB_STRIDE: dt, dt/2, sqrt(dt), 1/dt, then the stage times t0 + frac_j*dt (j < n_fracs) and the
 *          step end; then TSDE_SCAL_W0, TSDE_SCAL_W1 (interpolation weights, _core/interp.py:15-18) and
 *          TSDE_SCAL_ACCEPT.
 *
 * A step kernel reads such a word when its scalar argument is TSDE_DEV_SCALAR(address): every `double` coefficient of
 * tsde_step_diag / _prod / _general / _general_w, tsde_milstein_*, tsde_srk_diag_stage, tsde_heun_final and
 * tsde_linear_interp may be passed as a quiet NaN whose low 48 bits are the device address of the value (of the
 * launch's dtype) instead of the value itself. */
#define TSDE_DEV_SCALAR_TAG 0x7FFCull /* bits 63..48 of the double; bits 47..0 = the device address */
#define TSDE_ADAPTIVE_MAX_STAGES 6
#define TSDE_CTL_CURR_T 0
#define TSDE_CTL_PREV_T 1
#define TSDE_CTL_STEP_SIZE 2
#define TSDE_CTL_PREV_ERROR_RATIO 3 /* NaN = none yet */
#define TSDE_CTL_OUT_T 4
#define TSDE_CTL_T_END 5
#define TSDE_CTL_DT_MIN 6
#define TSDE_CTL_ATTEMPTS 7
#define TSDE_CTL_ACCEPTED 8
#define TSDE_CTL_DT_MIN_HITS 9
#define TSDE_CTL_NAN_SEEN 10
#define TSDE_CTL_ACTIVE 11 /* 1 while curr_t < out_t */
#define TSDE_CTL_BOUNDS_A 12 /* (a, b) of the first half step */
#define TSDE_CTL_BOUNDS_B 14 /* (a, b) of the second half step */
#define TSDE_CTL_WIDTHS 16   /* (ha, hb) */
#define TSDE_CTL_OUT_IDX 18    /* output times already reached (tsde_adaptive_begin_outputs) */
#define TSDE_CTL_N_OUT 19
#define TSDE_CTL_EMIT_FIRST 20 /* the output rows the last attempt's step reached: [first, first + count) */
#define TSDE_CTL_EMIT_COUNT 21
#define TSDE_CTL_SIZE 22
#define TSDE_SUB_DT 0
#define TSDE_SUB_HALF_DT 1
#define TSDE_SUB_SQRT_DT 2
#define TSDE_SUB_RDT 3
#define TSDE_SUB_TIMES 4
#define TSDE_SUB_STRIDE (TSDE_SUB_TIMES + TSDE_ADAPTIVE_MAX_STAGES)
#define TSDE_SCAL_W0 (3 * TSDE_SUB_STRIDE)
#define TSDE_SCAL_W1 (TSDE_SCAL_W0 + 1)
#define TSDE_SCAL_ACCEPT (TSDE_SCAL_W0 + 2)
#define TSDE_SCAL_SIZE (TSDE_SCAL_W0 + 3)

/* Start stepping towards output time `out_t`: sets ctl[OUT_T], ctl[ACTIVE] and the tables of the first attempt (or the
 * interpolation weights if the state is already past out_t). The caller initialises CURR_T, PREV_T, STEP_SIZE,
 * PREV_ERROR_RATIO (NaN), T_END, DT_MIN and zeroes the counters once per solve. `stage_fracs`: HOST array of the
 * solver's n_fracs stage offsets as multiples of dt (the first is 0). */
int tsde_adaptive_begin(double* ctl, void* scal, double out_t, const double* stage_fracs, int n_fracs, int dtype,
                        void* stream);
/* After an attempt (`error`: DEVICE double written by tsde_error_norm): base_solver.py:125-142. Inert once
 * ctl[ACTIVE] == 0. */
int tsde_adaptive_control(double* ctl, void* scal, const double* error, const double* stage_fracs, int n_fracs,
                          int dtype, void* stream);
/* The same two with the whole list of output times on the device (base_solver.py:117-145, both loops): `out_times` n_out
 * DEVICE doubles, ascending. The controller walks the list -- the attempt whose step carries curr_t to or past one or more
 * output times marks them (ctl[EMIT_FIRST], ctl[EMIT_COUNT]) and aims at the next -- and tsde_adaptive_emit, launched after
 * the commit of every attempt (and once after begin, for output times the start state already meets), writes the marked
 * rows  ys[j] = w0 prev_y + w1 curr_y  (interp.py:15-18; n elements per row) to the address stored in the device word
 * `ys_slot`. ctl[OUT_IDX] == ctl[N_OUT] means the solve is complete; later attempts are inert. One host
 * synchronisation per solve (to read that) instead of one per output time.
 * `accept_log` (DEVICE, 2 * log_capacity doubles, or NULL): accepted step number k (k < log_capacity) leaves its
 * (t0, t1) at [2k], [2k + 1] -- what a caller needs to run the accepted steps again, e.g. with autograd recording
 * (base_solver.py:117-142 records every attempt; only the accepted half-steps reach the result). */
int tsde_adaptive_begin_outputs(double* ctl, void* scal, const double* out_times, int32_t n_out, const double* stage_fracs,
                                int n_fracs, int dtype, void* stream);
int tsde_adaptive_control_outputs(double* ctl, void* scal, const double* error, const double* out_times,
                                  double* accept_log, int32_t log_capacity, const double* stage_fracs, int n_fracs, int dtype,
                                  void* stream);
int tsde_adaptive_emit(const void* ys_slot, const void* prev_y, const void* curr_y, int64_t n, const double* ctl,
                       const double* out_times, int dtype, void* stream);
/* prev_y <- curr_y, curr_y <- y_next if the controller accepted the attempt; moves nothing otherwise. */
int tsde_adaptive_commit(void* prev_y, void* curr_y, const void* y_next, int64_t n, const void* scal, int dtype,
                         void* stream);
/* (W, U) of the whole step from its halves (brownian_interval.py:647-672; U optional). The half widths come from
 * ctl[TSDE_CTL_WIDTHS..] (device) or, with ctl == NULL, from (ha, hb). */
int tsde_merge_halves(void* W, void* U, const void* Wa, const void* Ha, const void* Wb, const void* Hb, int64_t n,
                      const double* ctl, double ha, double hb, int dtype, void* stream);
/* tsde_brownian_query with the interval read from device memory: ab_dev[0] = a, ab_dev[1] = b (clamped to the grid;
 * the cells are located on the device; a >= b gives zeros). Exact-split leaf rule only (no `tol` snapping). */
int tsde_brownian_query_dev(void* W, void* U, void* H, int64_t n, uint64_t entropy, uint64_t elem0, const double* edges,
                            int64_t n_cells, const double* ab_dev, int have_h, int max_depth,
                            const uint64_t* entropy_dev, int dtype, void* stream);

/* ---- whole-trajectory kernels (closed-form SDEs) ---------------------------------------------- */
#define TSDE_TRAJ_EULER 0
#define TSDE_TRAJ_MILSTEIN_ITO 1
#define TSDE_TRAJ_MILSTEIN_STRAT 2
#define TSDE_TRAJ_MIDPOINT 3
#define TSDE_TRAJ_SRK 4
#define TSDE_TRAJ_HEUN 5       /* heun.py:35-48; the affine / expression / program kernels (values and sensitivities) */
#define TSDE_TRAJ_EULER_HEUN 6 /* euler_heun.py:29-42; the same kernels */
#define TSDE_TRAJ_REVERSIBLE_HEUN 7 /* reversible_heun.py:48-73; tsde_rheun_mlp_forward only */

/* All `traj->n_steps` fixed steps of a diagonal-noise SDE with per-channel affine drift and diffusion
 *   f(t, y) = drift_rate * y + drift_shift,   g(t, y) = diff_rate * y + diff_shift      (each of length d)
 * in ONE launch: y0 (rows, d) is read once, the state stays in registers, every step's increment is the
 * generated cell (entropy, elem0 + i, cells[k]) of the counter RNG, and only the requested outputs
 * ys (n_out, rows, d) are written. `method` is one of TSDE_TRAJ_*. Results are bit-identical to driving
 * tsde_step_diag / tsde_milstein_diag / tsde_srk_diag_stage step by step with f, g evaluated as
 * (rate * y) + shift. */
int tsde_trajectory_affine_diag(void* ys, const void* y0, int64_t rows, int64_t d, const void* drift_rate,
                                const void* drift_shift, const void* diff_rate, const void* diff_shift, int method,
                                const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                                const uint64_t* entropy_dev, int dtype, void* stream);

/* The same solve for coefficients that depend on TIME, f(t, y) = rate(t) * y + shift(t): each of the four arrays holds
 * S rows of d values per step, `coef_step_stride` elements apart (0 = the constant-coefficient kernel above), one row per
 * STAGE TIME of the scheme, in this order -- the times at which torchsde's step evaluates f and g:
 *   Euler, Milstein  S = 1: t_k                                             (methods/euler.py:31, milstein.py:54)
 *   midpoint         S = 2: t_k, t_k + dt/2                                 (midpoint.py:33,40)
 *   SRK (SRID2)      S = 4: t_k, t_k + dt/4, t_k + dt/2, t_k + dt           (srk.py:66-72 with tableaus/srid2.py:21-22)
 * Step k uses rows k*S ... k*S + S - 1. Values only. */
int tsde_trajectory_affine_diag_timed(void* ys, const void* y0, int64_t rows, int64_t d, const void* drift_rate,
                                      const void* drift_shift, const void* diff_rate, const void* diff_shift,
                                      int64_t coef_step_stride, int method, const tsde_traj_t* traj, uint64_t entropy,
                                      uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream);

/* The same solve with path-wise sensitivities: next to every output element its derivatives with respect to its
 * own initial value and its channel's four coefficients are written to
 *   sens (n_out, TSDE_TRAJ_SENS, rows, d)   planes in the order  d/dy0, d/d drift_rate, d/d drift_shift,
 *                                           d/d diff_rate, d/d diff_shift
 * (forward-mode derivatives carried through exactly the operations of the solve: what back-propagation through
 * torchsde.sdeint computes for this SDE). `ys` is bit-identical to tsde_trajectory_affine_diag. */
#define TSDE_TRAJ_SENS 5
int tsde_trajectory_affine_diag_sens(void* ys, void* sens, const void* y0, int64_t rows, int64_t d,
                                     const void* drift_rate, const void* drift_shift, const void* diff_rate,
                                     const void* diff_shift, int method, const tsde_traj_t* traj, uint64_t entropy,
                                     uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream);

/* All fixed steps of a diagonal-noise SDE whose drift is a two-layer perceptron shared by the batch,
 *   f(t, y) = W2 . act(W1 . y + b1) + b2,   g(t, y) = diff_rate * y + diff_shift,
 * in ONE launch (neural-SDE sampling): a wave keeps 16 batch rows in registers for the whole solve, the weights
 * live in LDS and both layers run on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 accumulation).
 *   w1 (d, hidden) and w2 (hidden, d) are stored input-major: w1[k][m] multiplies input channel k into unit m
 *   (the transpose of torch.nn.Linear.weight). d a multiple of 4 up to 128, hidden up to 128 -- up to 256 when
 *   d <= 64: both weight arrays have to fit the LDS of a CU -- (both are zero-padded
 *   to the MFMA tile sizes inside the kernel); rows * d < 2^30; ys, y0 16-byte aligned; dtype must be TSDE_F32;
 *   method in {TSDE_TRAJ_EULER, TSDE_TRAJ_MILSTEIN_ITO, TSDE_TRAJ_MILSTEIN_STRAT, TSDE_TRAJ_MIDPOINT, TSDE_TRAJ_SRK}
 *   (SRK = SRID2, torchsde/_core/methods/srk.py:57-88: three drift evaluations per step -- the tableau's alpha_3 = 0 --
 *   and, the diffusion being diagonal, all four diffusion stages elementwise; the increments' second stream H gives the
 *   space-time Levy area U = h (W/2 + H) of the cell).
 * Outputs as tsde_trajectory_affine_diag: output j is traj->out_w[2j] * y_k + traj->out_w[2j+1] * y_{k+1} of the step
 * k + 1 = traj->out_step[j] (the reference's linear interpolation, interp.py:15-18; weights (0, 1) = the step's result
 * itself); several outputs may share a step.
 * Increments: the generated cells (entropy, elem0 + i, cells[k]) of the counter RNG, i.e. the path the stepwise
 * solve of the same SDE sees; results agree with it up to the summation order of the two matrix products. */
#define TSDE_ACT_TANH 0
#define TSDE_ACT_SOFTPLUS 1
#define TSDE_ACT_SILU 2   /* act_scale * x * sigmoid(x) -- tsde_deep_mlp_t only (LipSwish: examples/sde_gan.py:44-47) */
/* diffusion of the perceptron-drift kernels, per channel:
 *   TSDE_DIFF_AFFINE   g = diff_rate * y + diff_shift
 *   TSDE_DIFF_SIGMOID  g = diff_amp * sigmoid(diff_rate * y + diff_shift)   (the elementwise diffusion nets of
 *                      examples/latent_sde_lorenz.py:137-148 reduced to one unit; diff_amp is ignored for AFFINE) */
#define TSDE_DIFF_AFFINE 0
#define TSDE_DIFF_SIGMOID 1
int tsde_trajectory_mlp_diag(void* ys, const void* y0, int64_t rows, int64_t d, int64_t hidden, const void* w1,
                             const void* b1, const void* w2, const void* b2, const void* diff_rate,
                             const void* diff_shift, int diff_kind, double diff_amp, int activation, int method,
                             const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev,
                             int dtype, void* stream);

/* tsde_trajectory_affine_diag for drift and diffusion given as expression PROGRAMS: any elementwise code -- sums and
 * products of several functions of the state, powers, quotients (the reference's ExScalar, tests/problems.py:75-103:
 * f = -p^2 sin(y) cos(y)^3, g = p cos(y)^2) -- as postfix instruction streams over a four-deep value stack.
 *   code  HOST array of 32-bit words (at most 96; they travel in the kernel arguments, so decoding is scalar work): the f
 *         program (f_len words), then g (g_len), then g' (dg_len; Milstein only, may be 0 words for the other methods).
 *         One word = opcode | source << 8 | constant row << 16.
 *         opcodes with a source: 0 LOAD (push), 1 ADD, 2 SUB, 3 RSUB, 4 MUL, 5 DIV, 6 RDIV; without: 16 NEG, 17 EXP, 18 LOG,
 *         19 SIN, 20 COS, 21 TANH, 22 SIGMOID, 23 SOFTPLUS (threshold 20), 24 SQRT, 25 ABS, 26 RELU, 27 RECIP, 28 SQUARE,
 *         29 CUBE, 30 DUP.  source: 0 the value below the top of the stack (popped), 1 constant row k (this channel's entry),
 *         2 the state y, 3 the time t at which the scheme evaluates the function (its stage time; traj->step_rows[k][7] must
 *         hold t_k, as for tsde_trajectory_mlp_general).  A binary operator computes  A op B,  A = top of stack, B = source (RSUB, RDIV: B op A); with
 *         source 0: A = the value below the top, B = the top, and the result replaces both. A program leaves its value on top.
 *   consts (n_const <= 64, d) in `dtype`, device: every number or per-channel parameter the programs use, one row each (the
 *         first 8 rows are kept in registers for the whole solve; later rows are read through the cache at each use)
 *   scalar_noise 1: noise type "scalar" -- ONE Brownian channel per row: element (row, c) takes increment elem0 + row of the
 *         (rows, 1) field; 0: diagonal noise as in tsde_trajectory_affine_diag
 * Same schedule, methods (all five), outputs and Brownian path as the affine kernel; values only. The host must make sure a
 * program never needs more than four stack slots (recognise.py orders its expression trees accordingly). */
int tsde_trajectory_prog_diag(void* ys, const void* y0, int64_t rows, int64_t d, const uint32_t* code, int32_t f_len,
                              int32_t g_len, int32_t dg_len, const void* consts, int32_t n_const, int scalar_noise, int method,
                              const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev,
                              int dtype, void* stream);

/* ... and with the path-wise sensitivities of every output (training through `sdeint`, _core/sdeint.py:27-112, for SDEs stated
 * as programs): forward-mode tangents carried through the same programs.
 *   sens (n_out, TSDE_TRAJ_SENS, rows, d): plane 0 = d out / d y0; plane s in 1 .. TSDE_TRAJ_SENS - 1 = d out / d (the
 *        constant row k with param_slot[k] == s), element-wise (a constant's entry for channel c only reaches channel c).
 *   param_slot HOST array of n_const entries: the plane of each constant row, or -1 for rows that need no gradient. */
int tsde_trajectory_prog_diag_sens(void* ys, void* sens, const void* y0, int64_t rows, int64_t d, const uint32_t* code,
                                   int32_t f_len, int32_t g_len, int32_t dg_len, const void* consts, int32_t n_const,
                                   const int8_t* param_slot, int scalar_noise, int method, const tsde_traj_t* traj,
                                   uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream);

/* Noise type "additive" (base_sde.py:101-102: g depends on t only; the reference's ExAdditive, tests/problems.py:106-132; the
 * step functions euler.py:29-37, midpoint.py:29-45 and SRK's additive_step, srk.py:90-111 with tableaus/sra1.py): the drift as
 * a program (as above, f only), the diffusion as a TABLE of the (d, m) matrix at the scheme's stage times.
 *   m        Brownian channels per row, 1 .. 16; the increments are those of the (rows, m) field: element elem0 + row * m + j
 *   g_table  `dtype`, device, 16-byte aligned. g_time_dependent == 0: (m, d), the transpose of the one matrix g.
 *            != 0: (n_steps, slots, m, d), entry [k][s][j][c] = g(t_k + frac_s * dt_k)[c, j] with
 *            frac = (0) for Euler, (0, 1/2) for midpoint, (1, 0) for SRK (sra1.py C1) -- computed by the host like the
 *            stage times of the stepwise loop (t0 + frac * dt in `dtype`).
 *   method   TSDE_TRAJ_EULER (also Milstein: with additive noise its correction is zero, base_sde.py:157-158),
 *            TSDE_TRAJ_MIDPOINT, TSDE_TRAJ_SRK (SRA1; needs the space-time Levy area, step_rows as for the affine kernel)
 * The drift program reads t from step_rows[k][7] (+ 1/2 dt for midpoint's second evaluation, + 3/4 dt for SRA1's, C0).
 * Values only. Same schedule, outputs and Brownian path as the other trajectory kernels. */
int tsde_trajectory_prog_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const uint32_t* code,
                                  int32_t f_len, const void* consts, int32_t n_const, const void* g_table,
                                  int g_time_dependent, int method, const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                                  const uint64_t* entropy_dev, int dtype, void* stream);

/* ---- neural SDEs: drift AND diffusion two-layer perceptrons of (t, y) -------------------------------------------------
 * One perceptron shared by the batch:  out = scale * final(W2 . act(W1 . y + w1t * t + b1) + b2)
 *   w1   (in, hidden)   input-major (the transpose of torch.nn.Linear.weight restricted to the STATE columns)
 *   w1t  (hidden)       the weight column of the time input, for modules that feed torch.cat([t.expand(B, 1), y], 1) to
 *                       their first layer (every Neural* problem of the reference, tests/problems.py:153-159,183-189,
 *                       215-217,246-252); NULL: the net does not see t
 *   w2   (hidden, out)  input-major;  b1 (hidden), b2 (out)
 *   final: TSDE_FINAL_NONE | TSDE_FINAL_SIGMOID (nn.Sigmoid() closing g_net);  scale: a number (the 0.1 of
 *   NeuralDiagonal.g, tests/problems.py:159); the drift uses final = NONE, scale = 1. */
#define TSDE_FINAL_NONE 0
#define TSDE_FINAL_SIGMOID 1
#define TSDE_FINAL_TANH 2    /* tsde_deep_mlp_t only (MLP(..., tanh=True), examples/sde_gan.py:50-66) */
#define TSDE_PRECISION_F32 0
#define TSDE_PRECISION_BF16X3 1
#define TSDE_NOISE_DIAGONAL 0
#define TSDE_NOISE_SCALAR 1
#define TSDE_NOISE_GENERAL 2
#define TSDE_NOISE_ADDITIVE 3   /* tsde_trajectory_mlp_additive only */
typedef struct tsde_mlp {
  const void* w1;
  const void* w1t;
  const void* b1;
  const void* w2;
  const void* b2;
  int32_t hidden;
  int32_t out;
  int32_t activation; /* TSDE_ACT_* */
  int32_t final;
  double scale;
  int32_t precision;  /* TSDE_PRECISION_F32 (exact f32 products: the reference's arithmetic, the default) or, diffusion net under
                         general noise with up to 64 hidden units only, TSDE_PRECISION_BF16X3: its second layer on split-bf16
                         products a_hi b_hi + a_hi b_lo + a_lo b_hi with f32 accumulation (~2^-16 relative per product; an
                         opt-in experiment, never benchmarked as the headline) */
  int32_t reserved;
} tsde_mlp_t;

/* All fixed steps of the SDE  dy = drift(t, y) dt + g(t, y) dW  with both functions perceptrons as above, in ONE launch
 * (replaces base_solver.py:114-134 driving methods/euler.py:29-37 -- whose f_and_g_prod is misc.batch_mvp,
 * _core/misc.py:62-63: bmm(g, dW) -- or methods/midpoint.py:29-45). BASELINE configs[2] is this with noise = GENERAL:
 *   noise = TSDE_NOISE_GENERAL   diffusion->out = d * m, read as (rows, d, m) row-major (`.view(B, d, m)`); 1 <= m <= 32 (run in
 *                                the next tile width 4 / 8 / 16 / 32: the padding is zero weights and zero increments);
 *                                increments: the (rows, m) field, element (row, j) = elem0 + row * m + j
 *   noise = TSDE_NOISE_DIAGONAL  diffusion->out = d, m = d: g[., i] dW[., i]             (NeuralDiagonal)
 *   noise = TSDE_NOISE_SCALAR    diffusion->out = d, m = 1: g[., i] dW[.]                (NeuralScalar)
 * method: TSDE_TRAJ_EULER (Ito), TSDE_TRAJ_MIDPOINT (Stratonovich; stage times t_k and t_k + dt/2), or -- diagonal and scalar
 * noise only, like the reference's (srk.py:34-35) -- TSDE_TRAJ_SRK (SRID2, srk.py:57-88: three drift and four diffusion
 * evaluations per step, every one a pass of its net; needs the increments' second stream, i.e. a Brownian motion with a
 * space-time Levy area).
 * A wave keeps 16 rows in registers for the whole solve; every weight lives in LDS; all four layers run on
 * v_mfma_f32_16x16x4_f32 (exact f32), the contraction with the increments included (csrc/mlp_general.hip).
 * traj->step_rows[k][7] must hold t_k, the time at which step k starts (the other trajectory kernels ignore that slot).
 * 1 <= d <= 64 (a multiple of 4 with an aligned field takes the 16-byte / whole-quad paths), hidden sizes up to 128 (general
 * noise: 64), and all weights must fit the 160 KiB of LDS:
 * tsde_trajectory_mlp_general_lds returns the bytes a shape needs (0: no kernel for it). dtype must be TSDE_F32;
 * any elem0; ys, y0 16-byte aligned (4-byte when d is not a multiple of 4: such rows go element by element); rows * d < 2^30. Outputs and increments as tsde_trajectory_mlp_diag. */
int tsde_trajectory_mlp_general(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                                const tsde_mlp_t* drift, const tsde_mlp_t* diffusion, int method, const tsde_traj_t* traj,
                                uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream);
int64_t tsde_trajectory_mlp_general_lds(int64_t d, int64_t m, int64_t drift_hidden, int64_t diffusion_hidden,
                                        int64_t diffusion_out, int noise);

/* The same kernel for ADDITIVE noise (base_sde.py:101-102; the reference's NeuralAdditive, tests/problems.py:195-224: the
 * drift a perceptron of cat([t, y]), the diffusion a function of t only): no diffusion net -- the (d, m) matrix arrives as
 * the table of tsde_trajectory_prog_additive (same layout, same stage times: g_time_dependent == 0: (m, d);
 * else (n_steps, slots, m, d) with slots = 1 Euler, 2 midpoint (t_k, t_k + dt/2), 2 SRK (t_k + dt, t_k)) and is contracted
 * with the row's increments on the matrix cores. method: TSDE_TRAJ_EULER (also Milstein with additive noise),
 * TSDE_TRAJ_MIDPOINT, TSDE_TRAJ_SRK (SRA1, srk.py:90-111: two drift evaluations per step, at t_k and t_k + 3/4 dt).
 * 1 <= m <= 16, any elem0; 1 <= d <= 64, hidden up to 128; dtype TSDE_F32. */
int tsde_trajectory_mlp_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const tsde_mlp_t* drift,
                                 const void* g_table, int g_time_dependent, int method, const tsde_traj_t* traj,
                                 uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream);

/* tsde_trajectory_affine_diag for drift and diffusion given as elementwise expressions per state channel:
 *     f = coef[0] * phi_f(coef[1] * y + coef[2]) + coef[3]        g = coef[4] * phi_g(coef[5] * y + coef[6]) + coef[7]
 * with phi_f, phi_g one of TSDE_FN_* (torchsde_amd.ElementwiseDiagonalSDE; e.g. the SDE the reference's own benchmark
 * integrates, f = y, g = exp(-y): benchmarks/brownian.py:131-139). coef: 8 device arrays of d values in `dtype`.
 * Same schedule, methods, outputs and Brownian path as the affine kernel; values only (no sensitivities). */
#define TSDE_FN_IDENTITY 0
#define TSDE_FN_EXP 1
#define TSDE_FN_SIGMOID 2
#define TSDE_FN_TANH 3
#define TSDE_FN_SOFTPLUS 4
#define TSDE_FN_SIN 5
#define TSDE_FN_COS 6
/* kind 7: no phi -- the four coefficients are those of a cubic, f = ((coef[0] * y + coef[1]) * y + coef[2]) * y + coef[3]
 * (likewise g from coef[4..7]): double-well and logistic drifts, quadratic diffusions. */
#define TSDE_FN_POLY3 7
int tsde_trajectory_expr_diag(void* ys, const void* y0, int64_t rows, int64_t d, const void* const coef[8], int f_kind,
                              int g_kind, int method, const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                              const uint64_t* entropy_dev, int dtype, void* stream);

/* ... and with time-dependent coefficients (see tsde_trajectory_affine_diag_timed): eight tables, S rows per step. */
int tsde_trajectory_expr_diag_timed(void* ys, const void* y0, int64_t rows, int64_t d, const void* const coef[8],
                                    int64_t coef_step_stride, int f_kind, int g_kind, int method,
                                    const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                                    const uint64_t* entropy_dev, int dtype, void* stream);

/* The stochastic adjoint of the perceptron-drift SDE of tsde_trajectory_mlp_diag, Euler-Maruyama or Milstein backwards
 * in time: what `sdeint_adjoint(..., adjoint_method="euler" | "milstein")` integrates for this module
 * (torchsde/_core/adjoint.py:64-127 driving adjoint_sde.py:177-230, 296-323, 332-377 through methods/euler.py:29-37 /
 * milstein.py:52-74), one launch per chunk of steps. `ito`: bit 0 set = Ito SDE (below: ito = 1), clear =
 * Stratonovich (ito = 0; Milstein only, like the reference); bit 1 set = Milstein backward step, clear = Euler. Processes
 * steps k_hi-1 ... k_lo of the FORWARD grid (step k walks back over Brownian cell traj->cells[k], width traj row k);
 * with dW that cell's increment, g the diagonal diffusion, everything evaluated at the current reconstructed y:
 *     f~ = f - ito * g g'                          (adjoint_sde.py:177-216)
 *     delta = (W2 a) * act'(W1^T y + b1) * dt
 *     y <- y - f~ dt - g dW ;   a <- a + W1 delta + a (g' dW - ito * dt g g'')
 *     Milstein, v = (dW^2 - ito * dt) / 2:   y <- ... + v g g' ;   a <- ... + a v (g'^2 - g g'')
 *   y, a         (rows, d)  in: the state and dL/dy at boundary k_hi; out: at boundary k_lo. The caller resets y to the
 *                stored forward state and adds the output's cotangent to a at every output time (adjoint.py:114-116).
 *   stash_a      (k_hi-k_lo, rows, d)       out: dt_k * a                      (slot k - k_lo)
 *   stash_hid    (k_hi-k_lo, rows, hidden)  out: act(W1^T y + b1)
 *   stash_delta  (k_hi-k_lo, rows, hidden)  out: delta
 *   stash_y      (k_hi-k_lo, rows, d)       out: the y the step was evaluated at
 *       => dL/dW2 += stash_a^T stash_hid, dL/dW1 += stash_delta^T stash_y (tsde_gram_partials),
 *          dL/db2 += column sums of stash_a, dL/db1 += column sums of stash_delta
 *   row_rate, row_shift (rows, d)  accumulated in place: sums over the steps AND over each aligned group of 16 rows (row
 *                16 k receives rows 16 k .. 16 k + 15; the other rows are left as they are) of
 *                a (dW dg/dc - ito dt g dg'/dc) and a (dW dg/de - ito dt g dg'/de) (Milstein: + a v (g' dg/dtheta -
 *                g dg'/dtheta)); their sums over all rows are dL/d diff_rate, dL/d diff_shift
 * Shapes, layouts and limits as tsde_trajectory_mlp_diag_backward; b2 (d) is the output bias of the drift. */
int tsde_adjoint_mlp_diag(void* y, void* a, void* stash_a, void* stash_hid, void* stash_delta, void* stash_y,
                          void* row_rate, void* row_shift, int64_t rows, int64_t d, int64_t hidden, const void* w1,
                          const void* b1, const void* w2, const void* b2, const void* diff_rate, const void* diff_shift,
                          int diff_kind, double diff_amp, int activation, int ito, const tsde_traj_t* traj, int32_t k_lo,
                          int32_t k_hi, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                          void* stream);

/* Reverse sweep of the gradient of tsde_trajectory_mlp_diag with method TSDE_TRAJ_EULER or TSDE_TRAJ_MILSTEIN_*:
 * back-propagation through the solver, i.e. what loss.backward() computes when autograd records the reference's
 * stepping loop (torchsde/_core/base_solver.py:114-134 with methods/euler.py:31-36 or milstein.py:52-74) for this SDE.
 * Processes steps k_hi-1 ... k_lo (call it on consecutive chunks, last steps first); with lam = dL/dy_{k+1}:
 *     u = W2^T lam,  delta = u * act'(W1 y_k + b1) * dt_k,   dL/dy_k = lam + W1^T delta + lam * diff_rate * dW_k
 *     (Milstein: + lam * diff_rate^2 * v_k, v = (dW^2 - dt)/2 for Ito, dW^2/2 for Stratonovich; a sigmoid
 *     diffusion -- Euler only -- replaces diff_rate by dg/dy = diff_amp s (1 - s) diff_rate).
 *   lam          (rows, d)  in: dL/dy at boundary k_hi WITHOUT the cotangent of an output at k_hi (the kernel adds
 *                           the cotangents of outputs at boundaries k_lo+1 ... k_hi itself); out: dL/dy at k_lo,
 *                           without the cotangent of an output at k_lo
 *   stash_lam    (k_hi-k_lo, rows, d)       out: dt_k * dL/dy_{k+1}       (slot k - k_lo)
 *   stash_hid    (k_hi-k_lo, rows, hidden)  out: act(W1 y_k + b1)
 *   stash_delta  (k_hi-k_lo, rows, hidden)  out: delta_k
 *       => dL/dW2 = sum stash_lam^T stash_hid, dL/dW1 = sum stash_delta^T y_k (tsde_gram_partials),
 *          dL/db2 = column sums of stash_lam,   dL/db1 = column sums of stash_delta
 *   row_rate, row_shift (rows, d)  accumulated in place: sum_k lam*y_k*dW_k and sum_k lam*dW_k per trajectory
 *                                  (their batch sums are dL/d diff_rate, dL/d diff_shift)
 *   ys_all       (.., rows, d)  the states at step boundaries ys_first, ys_first+1, ..., k_hi-1 at least (run the
 *                forward kernel with one output per step -- over the whole solve with ys_first = 0, or again over
 *                each chunk from a state kept at its start, ys_first = k_lo)
 *   grad_ys      (n_grad, rows, d), grad_step (n_grad, ascending, device): cotangent of the output at boundary
 *                grad_step[j]; grad_last = index of the last entry with grad_step <= k_hi, or -1
 * d, hidden multiples of 4 up to 128 (hidden up to 256 when d <= 64), rows * max(d, hidden) < 2^30; w1, w2 in the layout of
 * tsde_trajectory_mlp_diag; all buffers 16-byte aligned. */
int tsde_trajectory_mlp_diag_backward(void* lam, void* stash_lam, void* stash_hid, void* stash_delta, void* row_rate,
                                      void* row_shift, const void* ys_all, int32_t ys_first, const void* grad_ys,
                                      const int32_t* grad_step, int32_t grad_last, int64_t rows, int64_t d,
                                      int64_t hidden, const void* w1, const void* b1, const void* w2,
                                      const void* diff_rate, const void* diff_shift, int diff_kind, double diff_amp,
                                      int activation, int method, const tsde_traj_t* traj, int32_t k_lo, int32_t k_hi,
                                      uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                                      void* stream);

/* ---- The reversible Heun pair for neural SDEs, one launch each way (csrc/tsde_neural_rheun.h) ----------------------------
 * Replaces, for modules whose drift and diffusion are perceptrons of (t, y),
 *   forward:  base_solver.py:114-134 driving methods/reversible_heun.py:48-73 (ReversibleHeun.step; the carried (f, g, z) of
 *             init_extra_solver_state, :58-59, are formed in the kernel from z_0 = y_0)
 *   backward: adjoint.py:64-127 driving methods/reversible_heun.py:76-144 (AdjointReversibleHeun.step) with the vector-
 *             Jacobian products of adjoint_sde.py / misc.vjp evaluated on the matrix cores
 * A perceptron of up to four Linear layers:
 *   out = scale * final(W2 . a(Wm[n_mid-1] . ... a(Wm[0] . a(W1 . y + w1t * t + b1) + bm[0]) ... ) + b2),  a = act_scale * act(.)
 *   w1 (d, hidden), wm[l] (hidden, hidden), w2 (hidden, out) input-major (nn.Linear's weight transposed); w1t (hidden) or NULL;
 *   activation TSDE_ACT_TANH | _SOFTPLUS | _SILU; final TSDE_FINAL_NONE | _SIGMOID | _TANH; 0 <= n_mid <= 2. */
typedef struct tsde_deep_mlp {
  const void* w1;
  const void* w1t;
  const void* b1;
  const void* wm[2];
  const void* bm[2];
  const void* w2;
  const void* b2;
  int32_t hidden;
  int32_t out;
  int32_t activation;
  int32_t final;
  int32_t n_mid;
  int32_t reserved;
  double scale;
  double act_scale;
} tsde_deep_mlp_t;

/* ys (n_out, rows, d): the state at the outputs of traj (out_step / out_w as for the trajectory kernels); z_out (rows, d) or
 * NULL: the scheme's second state after the last step (what the backward call starts from; reversible_heun.py:73). times
 * (n_steps + 1): the step boundaries t_0 ... t_K in the state dtype -- the nets are evaluated at exactly these. noise as for
 * tsde_trajectory_mlp_general (general: 1 <= m <= 16; increments of the (rows, m) field). 1 <= d <= 64, hidden sizes <= 64,
 * weights must fit the LDS (tsde_rheun_mlp_lds: bytes, 0 = no kernel). dtype TSDE_F32; ys, z_out, y0 16-byte aligned when d is a
 * multiple of 4. */
int tsde_rheun_mlp_forward(void* ys, void* z_out, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                           const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, const tsde_traj_t* traj,
                           const void* times, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                           void* stream);
/* The same launch for the schemes that carry no state -- method = TSDE_TRAJ_EULER (euler.py:29-37), TSDE_TRAJ_MIDPOINT
 * (midpoint.py:29-45; second evaluation at t_k + dt/2), TSDE_TRAJ_HEUN (heun.py:35-48), TSDE_TRAJ_EULER_HEUN (euler_heun.py:
 * 29-42; second evaluation at t_{k+1}), or TSDE_TRAJ_REVERSIBLE_HEUN (= tsde_rheun_mlp_forward): networks deeper than the two
 * layers of tsde_trajectory_mlp_general, LipSwish, a closing tanh, under every fixed-step method the reference offers for general
 * noise. z_out (may be NULL) receives the final state for these schemes. */
int tsde_deep_mlp_forward(void* ys, void* z_out, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                          const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, int method, const tsde_traj_t* traj,
                          const void* times, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                          void* stream);
int64_t tsde_rheun_mlp_lds(int64_t d, int64_t m, int64_t drift_hidden, int64_t diffusion_hidden, int64_t diffusion_out,
                           int noise, int drift_mid, int diffusion_mid);

/* The carried state of the backward sweep, (rows, d) each, read and written by every call:
 *   y    before the first call: the last output; between calls: the state's running reconstruction
 *   z    before the first call: z_out of the forward launch
 *   a_y  before the first call: the last output's cotangent; after the call that reaches evaluation 0: dL/dy0
 *   a_z, a_f, p   zero before the first call (p: the vector of the rank-one a_g = p (x) dW, reversible_heun.py:137) */
typedef struct tsde_rheun_state {
  void* y;
  void* z;
  void* a_y;
  void* a_z;
  void* a_f;
  void* p;
} tsde_rheun_state_t;

/* What the parameter gradients are formed from: per evaluation e = j_hi - j of the call and row, (n_eval, rows, stride)
 * arrays (any may be NULL; strides are multiples of 4 floats >= the width):
 *   z (stride_d)            the point of evaluation            cf (stride_d)  cotangent of the drift net's output BEFORE `final`
 *   hf[l], df[l] (stride_hf) layer l's activations and the cotangent of its pre-activations, drift net (l = 0 .. n_mid)
 *   hg[l], dg[l] (stride_hg) the same for the diffusion net
 *   general noise: p, q (stride_d), wa, wb (stride_m): the cotangent of the diffusion net's final output is
 *                  p (x) wa + q (x) wb  (wa, wb: the two increments, scaled by the net's `scale`)
 *   diagonal / scalar noise: p (stride_d) holds the cotangent of the diffusion net's output BEFORE `final`; q, wa, wb unused
 * => dL/dW2 = sum h_top^T c, dL/dWm[l] = sum h[l]^T d[l+1], dL/dW1 = sum z^T d[0], dL/dw1t = sum t_j d[0], biases: column sums. */
typedef struct tsde_rheun_stash {
  void* z;
  void* cf;
  void* p;
  void* q;
  void* wa;
  void* wb;
  void* hf[3];
  void* df[3];
  void* hg[3];
  void* dg[3];
  int32_t stride_d;
  int32_t stride_m;
  int32_t stride_hf;
  int32_t stride_hg;
} tsde_rheun_stash_t;

/* Evaluations j_hi, j_hi - 1, ..., j_lo of the backward sweep (0 <= j_lo <= j_hi <= n_steps; call on consecutive ranges, the
 * first with j_hi = n_steps). ys_all (n_out + 1, rows, d): y0 followed by the forward outputs -- every output must sit ON a
 * step boundary (out_w = (0, 1)); grad_ys the same shape: their cotangents. At an output boundary the state is reset to the
 * stored one and its cotangent joins a_y (adjoint.py:114-116). */
int tsde_rheun_mlp_backward(const tsde_rheun_state_t* state, const tsde_rheun_stash_t* stash, const void* ys_all,
                            const void* grad_ys, int64_t rows, int64_t d, int64_t m, int noise,
                            const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, const tsde_traj_t* traj,
                            const void* times, int32_t j_hi, int32_t j_lo, uint64_t entropy, uint64_t elem0,
                            const uint64_t* entropy_dev, int dtype, void* stream);

/* The LAST layer of a general-noise diffusion net in that sweep (csrc/rheun_grad.hip): from the stash rows of
 * tsde_rheun_mlp_backward -- hid = hg[n_mid] (N, stride_h), p, q (N, stride_d), wa, wb (N, stride_m), N = evaluations x rows --
 * partial sums of  dL/dW2 = sum hid^T cot  and  dL/db2 = sum cot,  cot[., i m + j] = (p_i wa_j + q_i wb_j) final'(hid W2 + b2):
 *   gw (row_blocks, hidden, out), gb (row_blocks, out): one slice per row block (add them up in a fixed order: deterministic).
 * diffusion: w2 (hidden, out) input-major, b2, hidden <= 64, out = d * m, final as in tsde_deep_mlp_t; dtype TSDE_F32. */
int tsde_rheun_last_layer_grad(void* gw, void* gb, const void* hid, const void* p, const void* q, const void* wa, const void* wb,
                               int64_t n_rows, int64_t d, int64_t m, const tsde_deep_mlp_t* diffusion, int32_t stride_h,
                               int32_t stride_d, int32_t stride_m, int32_t row_blocks, int dtype, void* stream);

/* partials[i] = sum over the i-th contiguous range of the k rows of a[row, :m]^T b[row, :n]   (a, b row-major with
 * row strides lda >= m, ldb >= n floats -- column blocks of wider matrices are served in place --, m, n <= 128;
 * partials (blocks, m, n)) and, if colsum_partials (blocks, m) is not NULL, the column sums of a over the same range: the weight- and bias-gradient sums of the call above -- a product with a 128 x 128 result
 * and k in the tens of millions, the shape BLAS libraries serve worst. f32 MFMA; the caller adds the partials up in a
 * fixed order (deterministic). */
int tsde_gram_partials(void* partials, void* colsum_partials, const void* a, int64_t lda, const void* b, int64_t ldb,
                       int64_t k, int64_t m, int64_t n, int32_t blocks, int dtype, void* stream);

/* ---- in-library timing of one kernel family with HIP events (used by bench.py's roofline) ---- */
#define TSDE_KID_STEP_DIAG 1
#define TSDE_KID_STEP_GENERAL 2
#define TSDE_KID_MILSTEIN_DIAG 3
#define TSDE_KID_SRK_STAGE 4
#define TSDE_KID_AUG_UPDATE 5
#define TSDE_KID_BROWNIAN_QUERY 6
#define TSDE_KID_RHEUN 7
#define TSDE_KID_TRAJECTORY 8
#define TSDE_KID_MLP_BACKWARD 9
#define TSDE_KID_MLP_ADJOINT 10
#define TSDE_KID_MILSTEIN_GF_GENERAL 11
#define TSDE_KID_STEP_SHARED 12
#define TSDE_KID_RHEUN_MLP 13
/* Start timing every launch of kernel family `kid` (at most `capacity` launches). Families 1-6 and 11 are timed PER
 * DISPATCH: the launch is issued with hipExtLaunchKernel and the two events are bound to that dispatch, so their elapsed
 * time is the kernel's own start-to-end interval (what a rocprofv3 kernel trace reports) with no marker packets on the
 * stream. The other families (one long kernel per call) are bracketed by hipEventRecord calls around the launch. */
int tsde_prof_begin(int kid, int capacity);
/* A launch made OUTSIDE this library that belongs to family `kid` (a program kernel compiled at run time,
 * torchsde_amd/specialise.py): `open` records the start event of the next free slot on `stream` and returns the slot (-1: family
 * not being timed, or capacity used up), `close` records its end event. */
int tsde_prof_bracket_open(int kid, void* stream);
int tsde_prof_bracket_close(int slot, void* stream);
/* The per-launch times (ms) recorded so far, in launch order (`*used` of them, at most `capacity`); synchronises on
 * them. Call before tsde_prof_end. */
int tsde_prof_read(double* ms, int capacity, int* used);
/* Launch graphs (no counterpart in the reference, whose loop `_core/base_solver.py:114-134` launches step by step): in a
 * captured, NOT YET INSTANTIATED hipGraph_t, replaces every 1-D memset node by a kernel node that fills the same bytes,
 * with the same incoming and outgoing edges. `*n_memset` = memset nodes found, `*n_replaced` = replaced (2-D memsets are
 * left alone). Needed because recorded memset nodes -- ATen's reductions zero their semaphores with one -- stop working
 * on this runtime once eager memsets and a host synchronisation have come between two replays (csrc/graph_nodes.hip). */
int tsde_graph_memset_nodes_to_kernels(void* hip_graph, int* n_memset, int* n_replaced);
/* Occupies `stream` with a single-thread kernel for about `microseconds` (<= 2 s). bench.py queues one before
 * its event-timed pass so that the host can enqueue the whole solve first and no bracket contains queue-empty
 * time. */
int tsde_delay_us(double microseconds, void* stream);
/* What an event bracket adds to the kernel inside it, measured by bracketing `n` single-thread kernels that spin
 * for `spin_us` and report their own duration from the constant-rate wall clock. Synchronises `stream`. */
int tsde_prof_bracket_overhead(int n, double spin_us, double* overhead_ms, void* stream);
/* Synchronise the events, return the summed kernel time and launch count, stop profiling. */
int tsde_prof_end(double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* TORCHSDE_AMD_H */
