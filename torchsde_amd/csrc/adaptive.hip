// Adaptive step-doubling control ON THE DEVICE (replaces the host side of torchsde/_core/base_solver.py:117-142 and
// _core/adaptive_stepping.py:21-39 when the Brownian motion is this package's generator).
//
// The reference decides accept / reject on the host after every attempted step (`.item()` of the error estimate) and
// derives the next attempt's times from the decision. Here a one-thread controller kernel does that between the
// attempts: it reads the error norm tsde_error_norm left in device memory, applies the reference's PI controller,
// advances (curr_t, prev_t), and writes everything the NEXT attempt's kernels need into two small device tables:
//
//   * `scal` (T = the solve's dtype): per sub-step s in {whole step, first half, second half} the scalars
//     dt, dt/2, sqrt(dt), 1/dt (step kernels read them through `Coef`, tsde_common.h) and the stage times
//     t0 + frac_j dt ... t1 (the user's f(t, y), g(t, y) get them as 0-d views of this table); then the two
//     interpolation weights of the current output time and the accept flag;
//   * `ctl` (double): controller state and the (a, b) bounds of the two half-step Brownian queries, which the query
//     kernel reads from here (brownian.hip: QueryArgs::ab_dev).
//
// The host enqueues a budget of attempts without looking at any of this. With the OUTPUT TIMES on the device too
// (tsde_adaptive_begin_outputs / _control_outputs / _emit) the controller also walks the list of output times: the step that
// carries curr_t past one or more of them marks them for the emit kernel that follows the commit, which interpolates and
// writes those rows of ys (interp.py:15-18) -- so a whole solve needs no host decision and synchronises once, to learn that
// it is complete. Attempts issued after the last output time has been reached are inert: the controller leaves the state
// alone, the commit kernel copies nothing, the emit kernel writes nothing.
//
// Arithmetic: times are advanced in T exactly as the host loop does with numpy scalars of ts.dtype (t + T(step),
// min with t_end, T(0.5) * (t0 + t1)); the step size, the error ratio and the controller's powers are doubles, as
// Python floats are on the host.
#include <math.h>

#include "tsde_common.h"
#include "tsde_launch.h"

namespace tsde {

// ctl[] / scal[] layouts: include/torchsde_amd.h (TSDE_CTL_*, TSDE_SUB_*, TSDE_SCAL_*)
enum : int {
  kCurrT = TSDE_CTL_CURR_T, kPrevT = TSDE_CTL_PREV_T, kStepSize = TSDE_CTL_STEP_SIZE,
  kPrevErrRatio = TSDE_CTL_PREV_ERROR_RATIO, kOutT = TSDE_CTL_OUT_T, kTEnd = TSDE_CTL_T_END, kDtMin = TSDE_CTL_DT_MIN,
  kAttempts = TSDE_CTL_ATTEMPTS, kAccepted = TSDE_CTL_ACCEPTED, kDtMinHits = TSDE_CTL_DT_MIN_HITS,
  kNanSeen = TSDE_CTL_NAN_SEEN, kActive = TSDE_CTL_ACTIVE, kBoundsA = TSDE_CTL_BOUNDS_A, kBoundsB = TSDE_CTL_BOUNDS_B,
  kHa = TSDE_CTL_WIDTHS, kHb = TSDE_CTL_WIDTHS + 1, kOutIdx = TSDE_CTL_OUT_IDX, kNOut = TSDE_CTL_N_OUT,
  kEmitFirst = TSDE_CTL_EMIT_FIRST, kEmitCount = TSDE_CTL_EMIT_COUNT
};
constexpr int kMaxStages = TSDE_ADAPTIVE_MAX_STAGES;      // stage times per sub-step, the step end included
constexpr int kSubDt = TSDE_SUB_DT, kSubHalfDt = TSDE_SUB_HALF_DT, kSubSqrtDt = TSDE_SUB_SQRT_DT, kSubRdt = TSDE_SUB_RDT,
              kSubTimes = TSDE_SUB_TIMES, kSubStride = TSDE_SUB_STRIDE;
constexpr int kW0 = TSDE_SCAL_W0, kW1 = TSDE_SCAL_W1, kAccept = TSDE_SCAL_ACCEPT;

struct StageFracs {
  int n;                 // number of stage fractions (the step end is appended after them)
  double frac[kMaxStages - 1];
};

template <typename T>
TSDE_D void write_sub(T* s, T t0, T t1, const StageFracs& sf) {
  const T dt = t1 - t0;
  s[kSubDt] = dt;
  s[kSubHalfDt] = (T)0.5 * dt;
  s[kSubSqrtDt] = (T)sqrt(dt);               // IEEE sqrt: the bits numpy's sqrt gives the host loop
  s[kSubRdt] = (T)1 / dt;
  for (int j = 0; j < sf.n; ++j) s[kSubTimes + j] = sf.frac[j] == 0.0 ? t0 : t0 + (T)sf.frac[j] * dt;
  s[kSubTimes + sf.n] = t1;
}

// Everything the next attempt (or, once the output time is reached, the interpolation) reads.
template <typename T>
TSDE_D void refresh(double* ctl, T* scal, const StageFracs& sf) {
  const T curr = (T)ctl[kCurrT], prev = (T)ctl[kPrevT], out_t = (T)ctl[kOutT], t_end = (T)ctl[kTEnd];
  const bool active = curr < out_t;
  ctl[kActive] = active ? 1.0 : 0.0;
  if (active) {
    const T nxt = curr + (T)ctl[kStepSize];
    const T next_t = nxt <= t_end ? nxt : t_end;
    const T mid = (T)0.5 * (curr + next_t);
    write_sub<T>(scal, curr, next_t, sf);
    write_sub<T>(scal + kSubStride, curr, mid, sf);
    write_sub<T>(scal + 2 * kSubStride, mid, next_t, sf);
    ctl[kBoundsA] = (double)curr;
    ctl[kBoundsA + 1] = (double)mid;
    ctl[kBoundsB] = (double)mid;
    ctl[kBoundsB + 1] = (double)next_t;
    ctl[kHa] = (double)mid - (double)curr;
    ctl[kHb] = (double)next_t - (double)mid;
  } else if (curr > prev) {
    // interp.py:15-18 with the host's rounding: w0 = (t1 - t) / (t1 - t0), w1 = (t - t0) / (t1 - t0)
    scal[kW0] = (curr - out_t) / (curr - prev);
    scal[kW1] = (out_t - prev) / (curr - prev);
  } else {
    scal[kW0] = (T)0;
    scal[kW1] = (T)1;
  }
}

// adaptive_stepping.py:21-39 (`update_step_size`), doubles as on the host.
TSDE_D double next_step_size(double error_estimate, double prev_step_size, double& prev_error_ratio) {
  const double safety = 0.9, facmax = 1.4;
  double facmin = 0.2, pfactor, ifactor;
  if (error_estimate > 1) {
    pfactor = 0;
    ifactor = 1 / 1.5;
  } else {
    pfactor = 0.13;
    ifactor = 1 / 4.5;
  }
  const double error_ratio = safety / error_estimate;
  if (prev_error_ratio != prev_error_ratio) prev_error_ratio = error_ratio;      // None
  double factor = pow(error_ratio, ifactor) * pow(error_ratio / prev_error_ratio, pfactor);
  if (error_estimate <= 1) {
    prev_error_ratio = error_ratio;
    facmin = 1.0;
  }
  factor = fmin(facmax, fmax(facmin, factor));
  return prev_step_size * factor;
}

// The output times curr_t has reached (base_solver.py:117-145: the inner loop of an output time ends when curr_t >= out_t, and
// an output time the state is already past needs no step at all): mark them for the emit kernel and aim at the next one.
template <typename T>
TSDE_D void advance_outputs(double* ctl, const double* __restrict__ out_times) {
  const int n_out = (int)ctl[kNOut];
  int idx = (int)ctl[kOutIdx];
  const int first = idx;
  const T curr = (T)ctl[kCurrT];
  while (idx < n_out && !(curr < (T)out_times[idx])) ++idx;
  ctl[kEmitFirst] = (double)first;
  ctl[kEmitCount] = (double)(idx - first);
  ctl[kOutIdx] = (double)idx;
  if (n_out > 0) ctl[kOutT] = out_times[idx < n_out ? idx : n_out - 1];
}

// `out_times` null: one output time, handed over by the host (tsde_adaptive_begin); else the device list of n_out of them.
template <typename T>
__global__ void adaptive_begin_kernel(double* ctl, T* scal, double out_t, const double* __restrict__ out_times, int n_out,
                                      StageFracs sf) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  scal[kAccept] = (T)0;
  if (out_times == nullptr) {
    ctl[kOutT] = out_t;
  } else {
    ctl[kNOut] = (double)n_out;
    ctl[kOutIdx] = 0.0;
    advance_outputs<T>(ctl, out_times);
  }
  refresh<T>(ctl, scal, sf);
}

// After an attempt: base_solver.py:125-142.
template <typename T>
__global__ void adaptive_control_kernel(double* ctl, T* scal, const double* __restrict__ error,
                                        const double* __restrict__ out_times, double* __restrict__ accept_log,
                                        int log_capacity, StageFracs sf) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  scal[kAccept] = (T)0;
  if (out_times != nullptr) ctl[kEmitCount] = 0.0;
  if (ctl[kActive] == 0.0) return;                       // the output time was reached: this attempt is inert
  const double err = *error;
  if (err != err) ctl[kNanSeen] = 1.0;                   // the host raises the reference's AssertionError at its next read
  double per = ctl[kPrevErrRatio];
  double step = next_step_size(err, ctl[kStepSize], per);
  if (step < ctl[kDtMin]) {
    ctl[kDtMinHits] += 1.0;
    step = ctl[kDtMin];
    per = __builtin_nan("");
  }
  ctl[kAttempts] += 1.0;
  if (err <= 1 || step <= ctl[kDtMin]) {
    const T curr = (T)ctl[kCurrT];
    const T nxt = curr + (T)ctl[kStepSize];
    const T t_end = (T)ctl[kTEnd];
    ctl[kPrevT] = (double)curr;
    ctl[kCurrT] = (double)(nxt <= t_end ? nxt : t_end);
    // the accepted steps in order, for a caller that replays them (gradients: adaptive.integrate_with_grad)
    const int n_acc = (int)ctl[kAccepted];
    if (accept_log != nullptr && n_acc < log_capacity) {
      accept_log[2 * n_acc] = ctl[kPrevT];
      accept_log[2 * n_acc + 1] = ctl[kCurrT];
    }
    ctl[kAccepted] += 1.0;
    scal[kAccept] = (T)1;
  }
  ctl[kStepSize] = step;
  ctl[kPrevErrRatio] = per;
  if (out_times != nullptr) advance_outputs<T>(ctl, out_times);
  refresh<T>(ctl, scal, sf);
}

// prev_y <- curr_y ; curr_y <- y_next   iff the controller accepted the attempt (nothing moves otherwise).
template <typename T>
struct CommitOp {
  T *prev_y, *curr_y;
  const T* y_next;
  const T* accept;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    if (*accept == (T)0) return;
    store<T, W, NT>(prev_y, i, load<T, W, NT>(curr_y, i));
    store<T, W, NT>(curr_y, i, load<T, W, NT>(y_next, i));
  }
};

// ys[j] = w0 prev_y + w1 curr_y for the output times the controller has just marked (interp.py:15-18 with the host's
// rounding: w0 = (t1 - t) / (t1 - t0), w1 = (t - t0) / (t1 - t0) in T; a state that has not moved yet gives curr_y itself).
// `ys_slot`: device word holding the address of the first output row (a recorded attempt serves solves whose ys live
// elsewhere).
template <typename T>
struct EmitOp {
  T* const* ys_slot;
  const T *prev_y, *curr_y;
  const double* ctl;
  const double* out_times;
  int64_t n;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const int count = (int)ctl[kEmitCount];
    if (count == 0) return;
    const int first = (int)ctl[kEmitFirst];
    T* ys = *ys_slot;
    const T prev = (T)ctl[kPrevT], curr = (T)ctl[kCurrT];
    const Pack<T, W> a = load<T, W, NT>(prev_y, i), b = load<T, W, NT>(curr_y, i);
    for (int j = first; j < first + count; ++j) {
      const T out_t = (T)out_times[j];
      T w0 = (T)0, w1 = (T)1;
      if (curr > prev) {
        w0 = (curr - out_t) / (curr - prev);
        w1 = (out_t - prev) / (curr - prev);
      }
      Pack<T, W> o;
#pragma unroll
      for (int e = 0; e < W; ++e) o.v[e] = w0 * a.v[e] + w1 * b.v[e];
      store<T, W, NT>(ys + (int64_t)j * n, i, o);
    }
  }
};

// The whole step's (W, U) from its halves: the concatenation rule of the generator (brownian_interval.py:647-672),
//   W = Wa + Wb ;  H = (hb (Hb + Wa/2) + ha (Ha - Wb/2)) / (ha + hb) ;  U = (ha + hb) (W/2 + H)
// with the half widths read from the controller's table (the halves' own U come from the query kernel).
template <typename T>
struct MergeHalvesOp {
  T *W, *U;
  const T *Wa, *Ha, *Wb, *Hb;
  const double* widths;     // (ha, hb) in device memory, or null: the two values below
  double ha_host, hb_host;
  template <int Wd, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const Pack<T, Wd> wa = load<T, Wd, NT>(Wa, i), wb = load<T, Wd, NT>(Wb, i);
    Pack<T, Wd> w, u;
#pragma unroll
    for (int j = 0; j < Wd; ++j) w.v[j] = wa.v[j] + wb.v[j];
    store<T, Wd, NT>(W, i, w);
    if (U) {
      // the host forms ha + hb in double before it meets the tensors (solvers._step_doubling_noise)
      const double wa_ = widths ? widths[0] : ha_host, wb_ = widths ? widths[1] : hb_host;
      const T ha = (T)wa_, hb = (T)wb_, hsum = (T)(wa_ + wb_);       // as tsde_bridge.h: interval_merge
      const Pack<T, Wd> xa = load<T, Wd, NT>(Ha, i), xb = load<T, Wd, NT>(Hb, i);
#pragma unroll
      for (int j = 0; j < Wd; ++j) {
        const T h = (hb * (xb.v[j] + (T)0.5 * wa.v[j]) + ha * (xa.v[j] - (T)0.5 * wb.v[j])) / hsum;
        u.v[j] = hsum * ((T)0.5 * w.v[j] + h);
      }
      store<T, Wd, NT>(U, i, u);
    }
  }
};

template <typename T>
hipError_t launch_adaptive_begin(double* ctl, void* scal, double out_t, const double* out_times, int n_out,
                                 const double* fracs, int n_fracs, hipStream_t s) {
  StageFracs sf;
  sf.n = n_fracs;
  for (int j = 0; j < kMaxStages - 1; ++j) sf.frac[j] = j < n_fracs ? fracs[j] : 0.0;
  hipLaunchKernelGGL(adaptive_begin_kernel<T>, dim3(1), dim3(64), 0, s, ctl, (T*)scal, out_t, out_times, n_out, sf);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_adaptive_control(double* ctl, void* scal, const double* error, const double* out_times,
                                   double* accept_log, int log_capacity, const double* fracs, int n_fracs, hipStream_t s) {
  StageFracs sf;
  sf.n = n_fracs;
  for (int j = 0; j < kMaxStages - 1; ++j) sf.frac[j] = j < n_fracs ? fracs[j] : 0.0;
  hipLaunchKernelGGL(adaptive_control_kernel<T>, dim3(1), dim3(64), 0, s, ctl, (T*)scal, error, out_times, accept_log,
                     log_capacity, sf);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_adaptive_emit(const void* ys_slot, const void* prev_y, const void* curr_y, int64_t n, const double* ctl,
                                const double* out_times, hipStream_t s) {
  EmitOp<T> op{(T* const*)ys_slot, (const T*)prev_y, (const T*)curr_y, ctl, out_times, n};
  // (the rows of ys are n elements apart: 16-byte groups need n % 4 == 0; the host allocates ys 16-byte aligned)
  const bool vec = (n % 4 == 0) && aligned16(prev_y) && aligned16(curr_y);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_adaptive_commit(void* prev_y, void* curr_y, const void* y_next, int64_t n, const void* scal,
                                  hipStream_t s) {
  CommitOp<T> op{(T*)prev_y, (T*)curr_y, (const T*)y_next, (const T*)scal + kAccept};
  const bool vec = (n % 4 == 0) && aligned16(prev_y) && aligned16(curr_y) && aligned16(y_next);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_merge_halves(void* W, void* U, const void* Wa, const void* Ha, const void* Wb, const void* Hb,
                               int64_t n, const double* ctl, double ha, double hb, hipStream_t s) {
  MergeHalvesOp<T> op{(T*)W, (T*)U, (const T*)Wa, (const T*)Ha, (const T*)Wb, (const T*)Hb, ctl ? ctl + kHa : nullptr,
                      ha,    hb};
  const bool vec = (n % 4 == 0) && aligned16(W) && (!U || aligned16(U)) && aligned16(Wa) && aligned16(Wb) &&
                   (!U || (aligned16(Ha) && aligned16(Hb)));
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

#define TSDE_ADAPTIVE_INSTANTIATE(T)                                                                              \
  template hipError_t launch_adaptive_begin<T>(double*, void*, double, const double*, int, const double*, int,    \
                                               hipStream_t);                                                      \
  template hipError_t launch_adaptive_control<T>(double*, void*, const double*, const double*, double*, int,      \
                                                 const double*, int, hipStream_t);                                \
  template hipError_t launch_adaptive_emit<T>(const void*, const void*, const void*, int64_t, const double*,      \
                                              const double*, hipStream_t);                                        \
  template hipError_t launch_adaptive_commit<T>(void*, void*, const void*, int64_t, const void*, hipStream_t);    \
  template hipError_t launch_merge_halves<T>(void*, void*, const void*, const void*, const void*, const void*,    \
                                             int64_t, const double*, double, double, hipStream_t);
TSDE_ADAPTIVE_INSTANTIATE(float)
TSDE_ADAPTIVE_INSTANTIATE(double)

}  // namespace tsde
