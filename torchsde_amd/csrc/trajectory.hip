// Whole-trajectory kernel for closed-form diagonal SDEs on gfx950.
//
// When drift and diffusion are known in closed form (affine here: f = a*y + b, g = c*y + e, per state channel),
// nothing in the reference's stepping loop (torchsde/_core/base_solver.py:114-134) has to leave the chip between
// steps: every (batch row, channel) element is an independent scalar recursion. One lane owns 4 consecutive
// elements, keeps them in registers for ALL steps of the solve, draws each step's Brownian increment from the
// counter RNG (the same (entropy, element, cell) field the per-step kernels use) and only touches HBM to read
// y0 and to write the requested output times. HBM traffic drops from 4 streams per step to ~0, and the bound
// becomes the Philox/Box-Muller ALU work.
//
// The arithmetic is the per-step kernels' arithmetic (tsde_schemes.h), so the results are bit-identical to the
// stepwise path with the same SDE evaluated by torch ops:
//   Euler      _core/methods/euler.py:31-36         Midpoint  _core/methods/midpoint.py:31-43
//   Milstein   _core/methods/milstein.py:52-74      SRK       _core/methods/srk.py:57-88 (SRID2)
//   outputs    _core/base_solver.py:131-134 + _core/interp.py:19-26 (linear interpolation inside a step)
//
// Gradients: the same kernel instantiated on forward-mode dual numbers carries, next to every state element, its
// path-wise sensitivities d y / d (y0, a, b, c, e) through exactly the same operations (the recursions are
// element-wise, so each tangent is one more scalar recursion in registers). This is what back-propagation
// through the solver (ordinary autograd through torchsde.sdeint) computes for such an SDE, in the same single launch.
#include <type_traits>

#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_schemes.h"

namespace tsde {

constexpr int kSens = 5;   // tangents per element: d/dy0, d/da, d/db, d/dc, d/de

// Value + tangents. Only the operations an affine SDE needs: the state is never multiplied by the state.
template <typename T>
struct Dual {
  T v;
  T d[kSens];
  Dual() = default;
  TSDE_D explicit Dual(T value) : v(value) {
#pragma unroll
    for (int i = 0; i < kSens; ++i) d[i] = (T)0;
  }
};

// A coefficient: a plain value whose own tangent slot K is 1 (and every other 0), kept symbolic so that products and
// sums with it cost one extra add instead of a dense tangent update.
template <typename T, int K>
struct Seed {
  T v;
};

template <typename T>
TSDE_D Dual<T> operator+(const Dual<T>& x, const Dual<T>& y) {
  Dual<T> r;
  r.v = x.v + y.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = x.d[i] + y.d[i];
  return r;
}
template <typename T>
TSDE_D Dual<T> operator+(const Dual<T>& x, T s) {
  Dual<T> r = x;
  r.v = x.v + s;
  return r;
}
template <typename T>
TSDE_D Dual<T> operator*(const Dual<T>& x, T s) {
  Dual<T> r;
  r.v = x.v * s;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = x.d[i] * s;
  return r;
}
template <typename T>
TSDE_D Dual<T> operator*(T s, const Dual<T>& x) {
  Dual<T> r;
  r.v = s * x.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = s * x.d[i];
  return r;
}
template <typename T, int K>
TSDE_D Dual<T> operator*(const Seed<T, K>& p, const Dual<T>& x) {
  Dual<T> r;
  r.v = p.v * x.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = p.v * x.d[i];
  r.d[K] = r.d[K] + x.v;
  return r;
}
template <typename T, int K>
TSDE_D Dual<T> operator*(const Dual<T>& x, const Seed<T, K>& p) {
  Dual<T> r;
  r.v = x.v * p.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = x.d[i] * p.v;
  r.d[K] = r.d[K] + x.v;
  return r;
}
template <typename T, int K>
TSDE_D Dual<T> operator+(const Dual<T>& x, const Seed<T, K>& p) {
  Dual<T> r = x;
  r.v = x.v + p.v;
  r.d[K] = r.d[K] + (T)1;
  return r;
}

template <typename T>
struct TrajArgs {
  T* ys;                    // (n_out, n) outputs after t0
  T* sens;                  // (n_out, kSens, n) sensitivities of those outputs, or nullptr
  const T* y0;              // (n)
  const T *a, *b, *c, *e;   // (d) per-channel coefficients -- or (n_steps, d) tables when `cstride` = d (TIMED kernels)
  int64_t cstride;          // elements between the coefficient rows of consecutive steps; 0: the same at every step
  const T* rows;            // (n_steps, 8): dt, dt/2, 1/dt, sqrt(dt), sqrt(h), sqrt(h/12), h, 0
  const uint32_t* cells;    // (n_steps)
  const int32_t* out_step;  // (n_out) ascending: output j is due once `out_step[j]` steps are complete
  const T* out_w;           // (n_out, 2) weights on (previous state, current state)
  int64_t n, d;
  int32_t n_steps, n_out;
  NoiseKey key;
  const uint64_t* key_dev;
};

enum : int { kEuler = TSDE_TRAJ_EULER, kMilIto = TSDE_TRAJ_MILSTEIN_ITO, kMilStrat = TSDE_TRAJ_MILSTEIN_STRAT,
             kMidpoint = TSDE_TRAJ_MIDPOINT, kSrk = TSDE_TRAJ_SRK, kHeun = TSDE_TRAJ_HEUN, kEulerHeun = TSDE_TRAJ_EULER_HEUN };

// One step of one element of a diagonal SDE whose drift and diffusion the kernel can evaluate itself. `w` = W, `u` = U
// (SRK only). S is T (values only) or Dual<T>. `M` supplies f(x), g(x) and gdg(x, g, v) = (g * v) * g'(x), the
// diffusion VJP with cotangent g * v that derivative-form Milstein takes through autograd (base_sde.py:147-152).
// Stage-time slots. A scheme evaluates f and g at a few times of the step; a model whose coefficients depend on t holds
// one coefficient set per slot (TIMED kernels) and `m.f<SLOT>(x)` picks it; models with constant coefficients ignore the slot.
//   Euler, Milstein : slot 0 = t0                                         (euler.py:31, milstein.py:54)
//   midpoint        : slot 0 = t0, 1 = t0 + dt/2                          (midpoint.py:33,40)
//   SRK (SRID2)     : slot 0 = t0, 1 = t0 + dt/4, 2 = t0 + dt/2, 3 = t0 + dt;  f at C0 = (0, 1, 1/2) -> slots 0, 3, 2 and
//                     g at C1 = (0, 1/4, 1, 1/4) -> slots 0, 1, 3, 1       (srk.py:66-72, tableaus/srid2.py:21-22)
//   Heun, Euler-Heun: slot 0 = t0, 1 = t0 + dt = t1                       (heun.py:39,43, euler_heun.py:33,37)
template <int METHOD>
constexpr int stage_slots() {
  return METHOD == kSrk ? 4 : ((METHOD == kMidpoint || METHOD == kHeun || METHOD == kEulerHeun) ? 2 : 1);
}

template <typename T, int METHOD, typename S, typename M, typename N = T>
TSDE_D S scheme_step(const S y, const M& m, const N w, const N u, const T dt, const T half_dt, const T rdt,
                     const T sqrt_dt) {
  if constexpr (METHOD == kEuler) {
    return drift_diffusion_update<T, S>(y, m.template f<0>(y), m.template g<0>(y), w, dt, (T)1);
  } else if constexpr (METHOD == kMilIto || METHOD == kMilStrat) {
    const N v2 = milstein_v<T>(w, dt, (T)0.5, METHOD == kMilIto);
    const S g = m.template g<0>(y);
    const S gdg = m.template gdg<0>(y, g, v2);
    return milstein_update<T, S>(y, m.template f<0>(y), g, gdg, w, dt);
  } else if constexpr (METHOD == kMidpoint) {
    const S yp = drift_diffusion_update<T, S>(y, m.template f<0>(y), m.template g<0>(y), w, half_dt, (T)0.5);
    return drift_diffusion_update<T, S>(y, m.template f<1>(yp), m.template g<1>(yp), w, dt, (T)1);
  } else if constexpr (METHOD == kHeun || METHOD == kEulerHeun) {
    // Stratonovich predictor-corrector (heun.py:35-48, euler_heun.py:29-42) in the operation order of the stepwise route
    // (tsde_step_diag as predictor, tsde_heun_final): Heun predicts with the full Euler step, Euler-Heun with the noise alone
    constexpr bool heun = METHOD == kHeun;
    const S f0 = m.template f<0>(y), g0 = m.template g<0>(y);
    const S yp = drift_diffusion_update<T, S>(y, f0, g0, w, heun ? dt : (T)0, (T)1);
    const S g1 = m.template g<1>(yp);
    const S p0 = g0 * w, p1 = g1 * w;
    if constexpr (heun) {
      const S f1 = m.template f<1>(yp);
      return y + (((dt * (f0 + f1)) + p0) + p1) * (T)0.5;
    } else {
      return (y + dt * f0) + ((p0 + p1) * (T)0.5);
    }
  } else {
    const S zero = S((T)0);
    S f[3], g[4], h0, h1;
    S fz[3] = {zero, zero, zero};
    f[0] = m.template f<0>(y);
    g[0] = m.template g<0>(y);
    fz[0] = Srid2::need_f(1, 0) ? f[0] : zero;
    srid2_stage_states<T, 1, S>(y, fz, g, u, dt, rdt, sqrt_dt, h0, h1);
    f[1] = m.template f<3>(h0);
    g[1] = m.template g<1>(h1);
    fz[0] = Srid2::need_f(2, 0) ? f[0] : zero;
    fz[1] = Srid2::need_f(2, 1) ? f[1] : zero;
    srid2_stage_states<T, 2, S>(y, fz, g, u, dt, rdt, sqrt_dt, h0, h1);
    f[2] = m.template f<2>(h0);
    g[2] = m.template g<3>(h1);
    fz[0] = Srid2::need_f(3, 0) ? f[0] : zero;
    fz[1] = Srid2::need_f(3, 1) ? f[1] : zero;
    fz[2] = Srid2::need_f(3, 2) ? f[2] : zero;
    srid2_stage_states<T, 3, S>(y, fz, g, u, dt, rdt, sqrt_dt, h0, h1);
    g[3] = m.template g<1>(h1);
    return srid2_final<T, S>(y, f, g, w, u, dt, rdt, sqrt_dt);
  }
}

// Affine drift and diffusion, f = a*x + b, g = c*x + e; the coefficient types A..E are T or Seed<T, 1..4>.
template <typename T, typename S, typename A, typename B, typename C, typename E>
struct AffineModel {
  A a;
  B b;
  C c;
  E e;
  template <int SLOT>
  TSDE_D S f(const S& x) const { return a * x + b; }
  template <int SLOT>
  TSDE_D S g(const S& x) const { return c * x + e; }
  template <int SLOT>
  TSDE_D S gdg(const S&, const S& gv, T v2) const { return (gv * v2) * c; }   // vjp of y -> c*y + e with cotangent g*v2
};

// ... with one coefficient set per stage-time slot (coefficients that depend on t; values only)
template <typename T, int NS>
struct AffineModelTimed {
  T a[NS], b[NS], c[NS], e[NS];
  template <int SLOT>
  TSDE_D T f(const T& x) const { return a[SLOT] * x + b[SLOT]; }
  template <int SLOT>
  TSDE_D T g(const T& x) const { return c[SLOT] * x + e[SLOT]; }
  template <int SLOT>
  TSDE_D T gdg(const T&, const T& gv, T v2) const { return (gv * v2) * c[SLOT]; }
};

template <typename T, int METHOD, typename S, typename A, typename B, typename C, typename E>
TSDE_D S affine_step(const S y, const A a, const B b, const C c, const E e, const T w, const T u, const T dt,
                     const T half_dt, const T rdt, const T sqrt_dt) {
  const AffineModel<T, S, A, B, C, E> m{a, b, c, e};
  return scheme_step<T, METHOD, S>(y, m, w, u, dt, half_dt, rdt, sqrt_dt);
}

// ---- elementwise expressions ------------------------------------------------------------------------------------------
// f = fa * phi_f(fp * x + fq) + fb and g = ga * phi_g(gp * x + gq) + gb per channel, phi from a fixed set
// (closed_form.py: ElementwiseDiagonalSDE). The function codes are uniform over the launch (scalar branches). The
// evaluation mirrors what torch computes for the module's f / g on the stepwise path, operation by operation
// (`scale * phi(rate * y + shift) + offset`; sigmoid as 1 / (1 + exp(-u)), softplus with torch's threshold 20).
template <typename T>
TSDE_D T expr_phi(int kind, T u) {
  switch (kind) {
    case TSDE_FN_EXP: return exp(u);
    case TSDE_FN_SIGMOID: return (T)1 / ((T)1 + exp(-u));
    case TSDE_FN_TANH: return tanh(u);
    case TSDE_FN_SOFTPLUS: return u > (T)20 ? u : log1p(exp(u));
    case TSDE_FN_SIN: return sin(u);
    case TSDE_FN_COS: return cos(u);
    default: return u;
  }
}

// phi'(u), given phi(u) = v where that is cheaper
template <typename T>
TSDE_D T expr_dphi(int kind, T u, T v) {
  switch (kind) {
    case TSDE_FN_EXP: return v;
    case TSDE_FN_SIGMOID: return v * ((T)1 - v);
    case TSDE_FN_TANH: return (T)1 - v * v;
    case TSDE_FN_SOFTPLUS: return u > (T)20 ? (T)1 : (T)1 / ((T)1 + exp(-u));
    case TSDE_FN_SIN: return cos(u);
    case TSDE_FN_COS: return -sin(u);
    default: return (T)1;
  }
}

template <typename T>
struct ExprModel {
  T fa, fp, fq, fb, ga, gp, gq, gb;
  int fk, gk;
  // kind TSDE_FN_POLY3: the four coefficients are those of a cubic, ((c3 x + c2) x + c1) x + c0 (a drift such as y - y^3,
  // logistic growth, a quadratic diffusion): sums and products of several affine functions of the state
  template <int SLOT>
  TSDE_D T f(const T& x) const {
    if (fk == TSDE_FN_POLY3) return ((fa * x + fp) * x + fq) * x + fb;
    return fa * expr_phi<T>(fk, fp * x + fq) + fb;
  }
  template <int SLOT>
  TSDE_D T g(const T& x) const {
    if (gk == TSDE_FN_POLY3) return ((ga * x + gp) * x + gq) * x + gb;
    return ga * expr_phi<T>(gk, gp * x + gq) + gb;
  }
  // (g v) g'(x), the chain rule in the order autograd walks scale * phi(rate * x + shift) + offset backwards
  template <int SLOT>
  TSDE_D T gdg(const T& x, const T& gv, T v2) const {
    if (gk == TSDE_FN_POLY3) return (gv * v2) * ((((T)3 * ga) * x + (T)2 * gp) * x + gq);
    const T u = gp * x + gq;
    return (((gv * v2) * ga) * expr_dphi<T>(gk, u, expr_phi<T>(gk, u))) * gp;
  }
};

// ... with one set of the eight coefficients per stage-time slot (coefficients that depend on t)
template <typename T, int NS>
struct ExprModelTimed {
  T k[NS][8];          // fa, fp, fq, fb, ga, gp, gq, gb
  int fk, gk;
  template <int SLOT>
  TSDE_D T f(const T& x) const {
    if (fk == TSDE_FN_POLY3) return ((k[SLOT][0] * x + k[SLOT][1]) * x + k[SLOT][2]) * x + k[SLOT][3];
    return k[SLOT][0] * expr_phi<T>(fk, k[SLOT][1] * x + k[SLOT][2]) + k[SLOT][3];
  }
  template <int SLOT>
  TSDE_D T g(const T& x) const {
    if (gk == TSDE_FN_POLY3) return ((k[SLOT][4] * x + k[SLOT][5]) * x + k[SLOT][6]) * x + k[SLOT][7];
    return k[SLOT][4] * expr_phi<T>(gk, k[SLOT][5] * x + k[SLOT][6]) + k[SLOT][7];
  }
  template <int SLOT>
  TSDE_D T gdg(const T& x, const T& gv, T v2) const {
    if (gk == TSDE_FN_POLY3) return (gv * v2) * ((((T)3 * k[SLOT][4]) * x + (T)2 * k[SLOT][5]) * x + k[SLOT][6]);
    const T u = k[SLOT][5] * x + k[SLOT][6];
    return (((gv * v2) * k[SLOT][4]) * expr_dphi<T>(gk, u, expr_phi<T>(gk, u))) * k[SLOT][5];
  }
};

// Step count at which output j is due, as a wave-uniform (scalar) value; -1 past the last output.
TSDE_D int next_output_step(const int32_t* out_step, int j, int n_out) {
  return j < n_out ? __builtin_amdgcn_readfirstlane(out_step[j]) : -1;
}

template <typename T>
TSDE_D T primal(const T& x) { return x; }
template <typename T>
TSDE_D T primal(const Dual<T>& x) { return x.v; }

// W = 4: a lane owns one 16-byte group (needs d % 4 == 0 so the group stays inside one row).
// W = 1: a lane owns one element (any d; also used for small problems, where it exposes 4x the lanes).
// SENS : carry the kSens path-wise sensitivities of every element and write them next to the outputs.
// TIMED: the coefficients are functions of time, f(t, y) = a(t) * y + b(t), given as one row per STAGE TIME of every step
//        (`stage_slots<METHOD>()` rows per step, in slot order): re-read, through the L2, at the top of every step. A
//        separate instantiation, so the constant-coefficient kernels are untouched.
template <typename T, int METHOD, int W, bool SENS, bool TIMED = false>
__global__ void __launch_bounds__(kBlock) trajectory_kernel(const TrajArgs<T> p) {
  constexpr bool kNeedU = METHOD == kSrk;
  using S = typename std::conditional<SENS, Dual<T>, T>::type;
  using A = typename std::conditional<SENS, Seed<T, 1>, T>::type;
  using B = typename std::conditional<SENS, Seed<T, 2>, T>::type;
  using C = typename std::conditional<SENS, Seed<T, 3>, T>::type;
  using E = typename std::conditional<SENS, Seed<T, 4>, T>::type;
  const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t i = lane * W;
  if (i >= p.n) return;
  const int64_t col = i % p.d;
  const Pack<T, W> a = load<T, W>(p.a, col), b = load<T, W>(p.b, col), c = load<T, W>(p.c, col),
                   e = load<T, W>(p.e, col);
  const Pack<T, W> y_init = load<T, W>(p.y0, i);
  S y[W];
#pragma unroll
  for (int q = 0; q < W; ++q) {
    y[q] = S(y_init.v[q]);
    if constexpr (SENS) y[q].d[0] = (T)1;
  }
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const uint64_t elem = key.elem0 + (uint64_t)i;
  int j = 0;
  int next_out = next_output_step(p.out_step, 0, p.n_out);
  for (int k = 0; k < p.n_steps; ++k) {
    const T* row = p.rows + (int64_t)k * 8;   // wave-uniform
    const T dt = row[0], half_dt = row[1], rdt = row[2], sqrt_dt = row[3], sw = row[4], sh = row[5], th = row[6];
    const uint32_t cell = p.cells[k];
    constexpr int NS = stage_slots<METHOD>();
    Pack<T, W> ta[NS], tb[NS], tc[NS], te[NS];      // TIMED: the coefficient rows of this step's stage times
    if constexpr (TIMED) {
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const int64_t at = ((int64_t)k * NS + sl) * p.cstride + col;
        ta[sl] = load<T, W>(p.a, at);
        tb[sl] = load<T, W>(p.b, at);
        tc[sl] = load<T, W>(p.c, at);
        te[sl] = load<T, W>(p.e, at);
      }
    }
    Pack<T, W> w, u;
    if constexpr (W == 4) {
      T z[4];
      normal4<T>(key, elem >> 2, cell, 0, kStreamW, z);
#pragma unroll
      for (int q = 0; q < 4; ++q) w.v[q] = z[q] * sw;
      if constexpr (kNeedU) {
        normal4<T>(key, elem >> 2, cell, 0, kStreamH, z);
#pragma unroll
        for (int q = 0; q < 4; ++q) u.v[q] = th * ((T)0.5 * w.v[q] + z[q] * sh);
      }
    } else {
      w.v[0] = normal1<T>(key, elem, cell, 0, kStreamW) * sw;
      if constexpr (kNeedU) u.v[0] = th * ((T)0.5 * w.v[0] + normal1<T>(key, elem, cell, 0, kStreamH) * sh);
    }
    S y1[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
      if constexpr (TIMED) {
        AffineModelTimed<T, NS> m;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          m.a[sl] = ta[sl].v[q];
          m.b[sl] = tb[sl].v[q];
          m.c[sl] = tc[sl].v[q];
          m.e[sl] = te[sl].v[q];
        }
        y1[q] = scheme_step<T, METHOD, S>(y[q], m, w.v[q], kNeedU ? u.v[q] : (T)0, dt, half_dt, rdt, sqrt_dt);
      } else {
        y1[q] = affine_step<T, METHOD, S, A, B, C, E>(y[q], A{a.v[q]}, B{b.v[q]}, C{c.v[q]}, E{e.v[q]}, w.v[q],
                                                       kNeedU ? u.v[q] : (T)0, dt, half_dt, rdt, sqrt_dt);
      }
    }
    // (the step count of the next output lives in a scalar register: a step that is not an output time -- all but a few
    //  of them -- pays one scalar compare, and the output code is a cold block of its own)
    if (__builtin_expect(k + 1 == next_out, 0)) {
      while (j < p.n_out && p.out_step[j] == k + 1) {
        const T w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        const bool exact = (w0 == (T)0 && w1 == (T)1);
        S o[W];
#pragma unroll
        for (int q = 0; q < W; ++q) o[q] = exact ? y1[q] : (w0 * y[q] + w1 * y1[q]);
        Pack<T, W> ov;
#pragma unroll
        for (int q = 0; q < W; ++q) ov.v[q] = primal<T>(o[q]);
        store<T, W>(p.ys + (int64_t)j * p.n, i, ov);
        if constexpr (SENS) {
#pragma unroll
          for (int s = 0; s < kSens; ++s) {
#pragma unroll
            for (int q = 0; q < W; ++q) ov.v[q] = o[q].d[s];
            store<T, W>(p.sens + ((int64_t)j * kSens + s) * p.n, i, ov);
          }
        }
        ++j;
      }
      next_out = next_output_step(p.out_step, j, p.n_out);
    }
#pragma unroll
    for (int q = 0; q < W; ++q) y[q] = y1[q];
  }
}

template <typename T, int METHOD, bool SENS>
static hipError_t launch_traj_ms(const TrajArgs<T>& p, bool vec, hipStream_t s) {
  if (vec) {
    const int64_t lanes = p.n >> 2;
    hipLaunchKernelGGL((trajectory_kernel<T, METHOD, 4, SENS>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  } else {
    hipLaunchKernelGGL((trajectory_kernel<T, METHOD, 1, SENS>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

template <typename T, int METHOD>
static hipError_t launch_traj_timed(const TrajArgs<T>& p, bool vec, hipStream_t s) {
  if (vec) {
    const int64_t lanes = p.n >> 2;
    hipLaunchKernelGGL((trajectory_kernel<T, METHOD, 4, false, true>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  } else {
    hipLaunchKernelGGL((trajectory_kernel<T, METHOD, 1, false, true>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

template <typename T, int METHOD>
static hipError_t launch_traj_m(const TrajArgs<T>& p, bool vec, hipStream_t s) {
  if (p.cstride != 0) {            // coefficient tables (values only)
    if (p.sens) return hipErrorNotSupported;
    return launch_traj_timed<T, METHOD>(p, vec, s);
  }
  return p.sens ? launch_traj_ms<T, METHOD, true>(p, vec, s) : launch_traj_ms<T, METHOD, false>(p, vec, s);
}

// Below this many 16-byte groups the one-element-per-lane form is used even when the vector form is legal:
// the kernel is ALU/latency bound per lane, so a half-empty chip finishes sooner with 4x the lanes.
constexpr int64_t kTrajVecMinGroups = 256 * 8 * 64;

template <typename T>
hipError_t launch_trajectory_affine_diag(void* ys, void* sens, const void* y0, int64_t rows, int64_t d, const void* a,
                                         const void* b, const void* c, const void* e, int64_t cstride, int method,
                                         const tsde_traj_t* tr, NoiseKey key, const uint64_t* key_dev, hipStream_t s) {
  TrajArgs<T> p;
  p.cstride = cstride;
  p.ys = (T*)ys;
  p.sens = (T*)sens;
  p.y0 = (const T*)y0;
  p.a = (const T*)a;
  p.b = (const T*)b;
  p.c = (const T*)c;
  p.e = (const T*)e;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key = key;
  p.key_dev = key_dev;
  if (p.n <= 0 || p.n_steps <= 0) return hipSuccess;
  const bool can_vec = (d % 4 == 0) && (key.elem0 % 4 == 0) && aligned16(ys) && aligned16(y0) && aligned16(a) &&
                       aligned16(b) && aligned16(c) && aligned16(e) && ((p.n * sizeof(T)) % 16 == 0) &&
                       (!sens || aligned16(sens)) && (cstride % 4 == 0);
  const bool vec = can_vec && (p.n >> 2) >= kTrajVecMinGroups;
  switch (method) {
    case kEuler: return launch_traj_m<T, kEuler>(p, vec, s);
    case kMilIto: return launch_traj_m<T, kMilIto>(p, vec, s);
    case kMilStrat: return launch_traj_m<T, kMilStrat>(p, vec, s);
    case kMidpoint: return launch_traj_m<T, kMidpoint>(p, vec, s);
    case kSrk: return launch_traj_m<T, kSrk>(p, vec, s);
    case kHeun: return launch_traj_m<T, kHeun>(p, vec, s);
    case kEulerHeun: return launch_traj_m<T, kEulerHeun>(p, vec, s);
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
struct ExprArgs {
  T* ys;                    // (n_out, n) outputs after t0
  const T* y0;              // (n)
  const T* coef[8];         // (d) each: drift scale, rate, shift, offset; diffusion scale, rate, shift, offset
  int64_t cstride;          // as TrajArgs::cstride: (n_steps, d) tables for coefficients that depend on t
  int32_t f_kind, g_kind;
  const T* rows;
  const uint32_t* cells;
  const int32_t* out_step;
  const T* out_w;
  int64_t n, d;
  int32_t n_steps, n_out;
  NoiseKey key;
  const uint64_t* key_dev;
};

// The affine trajectory kernel's loop with the expression model in place of the affine one (values only).
template <typename T, int METHOD, int W, bool TIMED = false>
__global__ void __launch_bounds__(kBlock) trajectory_expr_kernel(const ExprArgs<T> p) {
  constexpr bool kNeedU = METHOD == kSrk;
  const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t i = lane * W;
  if (i >= p.n) return;
  const int64_t col = i % p.d;
  Pack<T, W> cf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) cf[c] = load<T, W>(p.coef[c], col);
  const Pack<T, W> y_init = load<T, W>(p.y0, i);
  T y[W];
#pragma unroll
  for (int q = 0; q < W; ++q) y[q] = y_init.v[q];
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const uint64_t elem = key.elem0 + (uint64_t)i;
  int j = 0;
  int next_out = next_output_step(p.out_step, 0, p.n_out);
  for (int k = 0; k < p.n_steps; ++k) {
    const T* row = p.rows + (int64_t)k * 8;   // wave-uniform
    const T dt = row[0], half_dt = row[1], rdt = row[2], sqrt_dt = row[3], sw = row[4], sh = row[5], th = row[6];
    const uint32_t cell = p.cells[k];
    constexpr int NS = stage_slots<METHOD>();
    Pack<T, W> tcf[NS][8];                          // TIMED: the coefficient rows of this step's stage times
    if constexpr (TIMED) {
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const int64_t at = ((int64_t)k * NS + sl) * p.cstride + col;
#pragma unroll
        for (int c = 0; c < 8; ++c) tcf[sl][c] = load<T, W>(p.coef[c], at);
      }
    }
    Pack<T, W> w, u;
    if constexpr (W == 4) {
      T z[4];
      normal4<T>(key, elem >> 2, cell, 0, kStreamW, z);
#pragma unroll
      for (int q = 0; q < 4; ++q) w.v[q] = z[q] * sw;
      if constexpr (kNeedU) {
        normal4<T>(key, elem >> 2, cell, 0, kStreamH, z);
#pragma unroll
        for (int q = 0; q < 4; ++q) u.v[q] = th * ((T)0.5 * w.v[q] + z[q] * sh);
      }
    } else {
      w.v[0] = normal1<T>(key, elem, cell, 0, kStreamW) * sw;
      if constexpr (kNeedU) u.v[0] = th * ((T)0.5 * w.v[0] + normal1<T>(key, elem, cell, 0, kStreamH) * sh);
    }
    T y1[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
      if constexpr (TIMED) {
        ExprModelTimed<T, NS> m;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
          for (int c = 0; c < 8; ++c) m.k[sl][c] = tcf[sl][c].v[q];
        }
        m.fk = p.f_kind;
        m.gk = p.g_kind;
        y1[q] = scheme_step<T, METHOD, T>(y[q], m, w.v[q], kNeedU ? u.v[q] : (T)0, dt, half_dt, rdt, sqrt_dt);
      } else {
        const ExprModel<T> m{cf[0].v[q], cf[1].v[q], cf[2].v[q], cf[3].v[q], cf[4].v[q], cf[5].v[q],
                             cf[6].v[q], cf[7].v[q], p.f_kind,  p.g_kind};
        y1[q] = scheme_step<T, METHOD, T>(y[q], m, w.v[q], kNeedU ? u.v[q] : (T)0, dt, half_dt, rdt, sqrt_dt);
      }
    }
    if (__builtin_expect(k + 1 == next_out, 0)) {
      while (j < p.n_out && p.out_step[j] == k + 1) {
        const T w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        const bool exact = (w0 == (T)0 && w1 == (T)1);
        Pack<T, W> ov;
#pragma unroll
        for (int q = 0; q < W; ++q) ov.v[q] = exact ? y1[q] : (w0 * y[q] + w1 * y1[q]);
        store<T, W>(p.ys + (int64_t)j * p.n, i, ov);
        ++j;
      }
      next_out = next_output_step(p.out_step, j, p.n_out);
    }
#pragma unroll
    for (int q = 0; q < W; ++q) y[q] = y1[q];
  }
}

template <typename T, int METHOD>
static hipError_t launch_expr_m(const ExprArgs<T>& p, bool vec, hipStream_t s) {
  if (p.cstride != 0) {
    // (SRK holds 4 slots x 8 coefficients per element: one element per lane keeps that in registers)
    if (vec && METHOD != kSrk) {
      const int64_t lanes = p.n >> 2;
      hipLaunchKernelGGL((trajectory_expr_kernel<T, METHOD, 4, true>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)),
                         dim3(kBlock), 0, s, p);
    } else {
      hipLaunchKernelGGL((trajectory_expr_kernel<T, METHOD, 1, true>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)),
                         dim3(kBlock), 0, s, p);
    }
    return hipGetLastError();
  }
  if (vec) {
    const int64_t lanes = p.n >> 2;
    hipLaunchKernelGGL((trajectory_expr_kernel<T, METHOD, 4>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  } else {
    hipLaunchKernelGGL((trajectory_expr_kernel<T, METHOD, 1>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t launch_trajectory_expr_diag(void* ys, const void* y0, int64_t rows, int64_t d, const void* const coef[8],
                                       int64_t cstride, int f_kind, int g_kind, int method, const tsde_traj_t* tr,
                                       NoiseKey key, const uint64_t* key_dev, hipStream_t s) {
  ExprArgs<T> p;
  p.cstride = cstride;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  bool aligned = aligned16(ys) && aligned16(y0);
  for (int c = 0; c < 8; ++c) {
    p.coef[c] = (const T*)coef[c];
    aligned = aligned && aligned16(coef[c]);
  }
  p.f_kind = f_kind;
  p.g_kind = g_kind;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key = key;
  p.key_dev = key_dev;
  if (p.n <= 0 || p.n_steps <= 0) return hipSuccess;
  const bool can_vec = (d % 4 == 0) && (key.elem0 % 4 == 0) && aligned && ((p.n * sizeof(T)) % 16 == 0) &&
                       (cstride % 4 == 0);
  const bool vec = can_vec && (p.n >> 2) >= kTrajVecMinGroups;
  switch (method) {
    case kEuler: return launch_expr_m<T, kEuler>(p, vec, s);
    case kMilIto: return launch_expr_m<T, kMilIto>(p, vec, s);
    case kMilStrat: return launch_expr_m<T, kMilStrat>(p, vec, s);
    case kMidpoint: return launch_expr_m<T, kMidpoint>(p, vec, s);
    case kSrk: return launch_expr_m<T, kSrk>(p, vec, s);
    case kHeun: return launch_expr_m<T, kHeun>(p, vec, s);
    case kEulerHeun: return launch_expr_m<T, kEulerHeun>(p, vec, s);
    default: return hipErrorInvalidValue;
  }
}

// ---- expression PROGRAMS ------------------------------------------------------------------------------------------------
// Drift, diffusion (and the diffusion's derivative, for Milstein) as small postfix programs over a four-deep value stack:
// whatever elementwise code a user's f and g consist of -- several functions of the state summed or multiplied
// (`-p**2 * sin(y) * cos(y)**3`, tests/problems.py:84-86; `tanh(y) + y`; y**4) -- which the single-function forms above
// cannot express. recognise.py builds the expression tree of the user's code while it runs on the probe, orders it so that
// four stack slots suffice, and hands over: `code` (32-bit words: opcode | source << 8 | constant row << 16) and a table of
// per-channel constants (n_const, d). One instruction = one torch operator of the user's code, evaluated in the same order
// and (for + - * /) with the same rounding; functions as in `expr_phi`. The instruction stream is wave-uniform: decoding it
// is scalar work, the vector unit sees one operation per instruction and element.
//   source: 0 = the value below the top of the stack (binary operators pop it), 1 = constant row k of this channel, 2 = the state,
//           3 = the time t at which the scheme evaluates the function (its stage time: t_k, t_k + dt/4, t_k + dt/2 or t_k + dt)
//   a binary operator computes  A op B  with A = top of stack, B = source  (R variants: B op A); source 0: A = the value
//   below the top, B = the top (R variants swapped), and the result replaces both
enum : uint32_t { kSrcStack = 0, kSrcConst = 1, kSrcState = 2, kSrcTime = 3 };
enum : uint32_t {
  kOpLoad = 0, kOpAdd, kOpSub, kOpRsub, kOpMul, kOpDiv, kOpRdiv,                       // take a source
  kOpNeg = 16, kOpExp, kOpLog, kOpSin, kOpCos, kOpTanh, kOpSigmoid, kOpSoftplus, kOpSqrt, kOpAbs, kOpRelu, kOpRecip,
  kOpSquare, kOpCube, kOpDup
};

// W elements processed together: one decoded instruction serves all of them.
template <typename T, int W>
struct Vec {
  T v[W];
  Vec() = default;
  TSDE_D explicit Vec(T s) {
#pragma unroll
    for (int q = 0; q < W; ++q) v[q] = s;
  }
};
#define TSDE_VEC_BINARY(OP)                                                                  \
  template <typename T, int W>                                                               \
  TSDE_D Vec<T, W> operator OP(const Vec<T, W>& a, const Vec<T, W>& b) {                     \
    Vec<T, W> r;                                                                             \
    _Pragma("unroll") for (int q = 0; q < W; ++q) r.v[q] = a.v[q] OP b.v[q];                 \
    return r;                                                                                \
  }                                                                                          \
  template <typename T, int W>                                                               \
  TSDE_D Vec<T, W> operator OP(const Vec<T, W>& a, T b) {                                    \
    Vec<T, W> r;                                                                             \
    _Pragma("unroll") for (int q = 0; q < W; ++q) r.v[q] = a.v[q] OP b;                      \
    return r;                                                                                \
  }                                                                                          \
  template <typename T, int W>                                                               \
  TSDE_D Vec<T, W> operator OP(T a, const Vec<T, W>& b) {                                    \
    Vec<T, W> r;                                                                             \
    _Pragma("unroll") for (int q = 0; q < W; ++q) r.v[q] = a OP b.v[q];                      \
    return r;                                                                                \
  }
TSDE_VEC_BINARY(+)
TSDE_VEC_BINARY(-)
TSDE_VEC_BINARY(*)
TSDE_VEC_BINARY(/)
#undef TSDE_VEC_BINARY

template <typename T, int W, typename F>
TSDE_D Vec<T, W> vmap(const Vec<T, W>& a, F fn) {
  Vec<T, W> r;
#pragma unroll
  for (int q = 0; q < W; ++q) r.v[q] = fn(a.v[q]);
  return r;
}

constexpr int kProgWords = 96;     // instruction words of f, g and g' together (they travel in the kernel arguments)
constexpr int kProgRegs = 8;       // constant rows kept in registers; rows beyond are read through the cache at each use

template <typename T>
struct ProgArgs;

template <typename T, int W>
struct ProgModel {
  using V = Vec<T, W>;
  TSDE_D void setup(const ProgArgs<T>& p, int64_t column);
  const uint32_t* code;            // f program, then g, then g': kernel arguments -> scalar loads, wave-uniform
  int f_len, g_len, dg_len;
  const T* consts;                 // (n_const, d)
  int64_t d, col;                  // channel of this lane's first element
  V creg[kProgRegs];               // this lane's entries of the first constant rows
  T tslot[4];                      // the scheme's stage times of the current step (slot order of `stage_slots`)

  TSDE_D V constant(uint32_t k) const {
    switch (k) {                   // (uniform: a scalar branch; a register array indexed by k would go through scratch)
      case 0: return creg[0];
      case 1: return creg[1];
      case 2: return creg[2];
      case 3: return creg[3];
      case 4: return creg[4];
      case 5: return creg[5];
      case 6: return creg[6];
      case 7: return creg[7];
      default: {
        V r;
        const Pack<T, W> pk = load<T, W>(consts, (int64_t)k * d + col);
#pragma unroll
        for (int q = 0; q < W; ++q) r.v[q] = pk.v[q];
        return r;
      }
    }
  }

  TSDE_D V run(const uint32_t* prog, int len, const V& x, const T time) const {
    V s0((T)0), s1((T)0), s2((T)0), s3((T)0);
    uint32_t fetched = prog[0];
    for (int pc = 0; pc < len; ++pc) {
      // (the next word is requested before this one executes: the scalar load's latency hides behind the vector work)
      const uint32_t ins = fetched;
      fetched = prog[pc + 1 < len ? pc + 1 : pc];
      const uint32_t op = ins & 0xFFu, src = (ins >> 8) & 0xFFu, k = ins >> 16;
      if (op < kOpNeg) {
        V a = s0, b;
        if (src == kSrcStack) {        // pop: the operands are the two top values
          b = s0;
          a = s1;
          s1 = s2;
          s2 = s3;
        } else if (src == kSrcConst) {
          b = constant(k);
        } else if (src == kSrcTime) {
          b = V(time);
        } else {
          b = x;
        }
        switch (op) {
          case kOpLoad:                // push the source (never the stack)
            s3 = s2;
            s2 = s1;
            s1 = s0;
            s0 = b;
            break;
          case kOpAdd: s0 = a + b; break;
          case kOpSub: s0 = a - b; break;
          case kOpRsub: s0 = b - a; break;
          case kOpMul: s0 = a * b; break;
          case kOpDiv: s0 = a / b; break;
          default: s0 = b / a; break;  // kOpRdiv
        }
      } else {
        switch (op) {
          case kOpNeg: s0 = vmap(s0, [](T v) { return -v; }); break;
          case kOpExp: s0 = vmap(s0, [](T v) { return exp(v); }); break;
          case kOpLog: s0 = vmap(s0, [](T v) { return log(v); }); break;
          case kOpSin: s0 = vmap(s0, [](T v) { return sin(v); }); break;
          case kOpCos: s0 = vmap(s0, [](T v) { return cos(v); }); break;
          case kOpTanh: s0 = vmap(s0, [](T v) { return tanh(v); }); break;
          case kOpSigmoid: s0 = vmap(s0, [](T v) { return (T)1 / ((T)1 + exp(-v)); }); break;
          case kOpSoftplus: s0 = vmap(s0, [](T v) { return v > (T)20 ? v : log1p(exp(v)); }); break;
          case kOpSqrt: s0 = vmap(s0, [](T v) { return sqrt(v); }); break;
          case kOpAbs: s0 = vmap(s0, [](T v) { return fabs(v); }); break;
          case kOpRelu: s0 = vmap(s0, [](T v) { return v > (T)0 ? v : (T)0; }); break;
          case kOpRecip: s0 = vmap(s0, [](T v) { return (T)1 / v; }); break;
          case kOpSquare: s0 = s0 * s0; break;
          case kOpCube: s0 = (s0 * s0) * s0; break;
          default:                     // kOpDup
            s3 = s2;
            s2 = s1;
            s1 = s0;
            break;
        }
      }
    }
    return s0;
  }
  template <int SLOT>
  TSDE_D V f(const V& x) const { return run(code, f_len, x, tslot[SLOT]); }
  template <int SLOT>
  TSDE_D V g(const V& x) const { return run(code + f_len, g_len, x, tslot[SLOT]); }
  template <int SLOT>
  TSDE_D V gdg(const V& x, const V& gv, const V& v2) const {
    return (gv * v2) * run(code + f_len + g_len, dg_len, x, tslot[SLOT]);
  }
};

// The times at which a scheme evaluates f and g within the step that starts at t0 (slot order of `stage_slots`; computed like
// the stepwise path's host code: t0 + frac * dt in the state dtype).
template <typename T, int METHOD>
TSDE_D void stage_times(T t0, T dt, T (&out)[4]) {
  out[0] = t0;
  out[1] = METHOD == kSrk ? t0 + (T)0.25 * dt : ((METHOD == kHeun || METHOD == kEulerHeun) ? t0 + dt : t0 + (T)0.5 * dt);
  out[2] = t0 + (T)0.5 * dt;
  out[3] = t0 + dt;
}

template <typename T>
struct ProgArgs {
  T* ys;
  const T* y0;
  const T* consts;
  int32_t f_len, g_len, dg_len, n_const;
  int32_t scalar_noise;     // 1: one Brownian channel per ROW (noise type "scalar"): element (row, c) meets increment `row`
  const T* rows;
  const uint32_t* cells;
  const int32_t* out_step;
  const T* out_w;
  int64_t n, d;
  int32_t n_steps, n_out;
  NoiseKey key;
  const uint64_t* key_dev;
  uint32_t code[kProgWords];
};

template <typename T, int W>
TSDE_D void ProgModel<T, W>::setup(const ProgArgs<T>& p, int64_t column) {
  code = p.code;
  f_len = p.f_len;
  g_len = p.g_len;
  dg_len = p.dg_len;
  consts = p.consts;
  d = p.d;
  col = column;
#pragma unroll
  for (int k = 0; k < kProgRegs; ++k) {
    creg[k] = V((T)0);
    if (k < p.n_const) {
      const Pack<T, W> pk = load<T, W>(p.consts, (int64_t)k * p.d + col);
#pragma unroll
      for (int q = 0; q < W; ++q) creg[k].v[q] = pk.v[q];
    }
  }
}

// The expression kernels' loop with the program model; a lane's W elements run through each program together.
// W = 4 needs d % 4 == 0 (a lane's elements share their row).
// `M`: the model of drift and diffusion -- `ProgModel` (the interpreter), or a struct GENERATED from the same programs and
// compiled at run time (torchsde_amd/specialise.py: the instruction stream becomes straight-line code, ~8x fewer cycles per
// wave-step; same operations in the same order, hence the same bits).
template <typename T, int METHOD, int W, typename M = ProgModel<T, W>>
__global__ void __launch_bounds__(kBlock) trajectory_prog_kernel(const ProgArgs<T> p) {
  constexpr bool kNeedU = METHOD == kSrk;
  using V = Vec<T, W>;
  const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t i = lane * W;
  if (i >= p.n) return;
  const int64_t col = i % p.d;
  const Pack<T, W> y_init = load<T, W>(p.y0, i);
  V y;
#pragma unroll
  for (int q = 0; q < W; ++q) y.v[q] = y_init.v[q];
  M m;
  m.setup(p, col);
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const bool scalar_noise = p.scalar_noise != 0;
  const uint64_t elem = key.elem0 + (uint64_t)(scalar_noise ? i / p.d : i);
  int j = 0;
  int next_out = next_output_step(p.out_step, 0, p.n_out);
  for (int k = 0; k < p.n_steps; ++k) {
    const T* row = p.rows + (int64_t)k * 8;   // wave-uniform
    const T dt = row[0], half_dt = row[1], rdt = row[2], sqrt_dt = row[3], sw = row[4], sh = row[5], th = row[6];
    const uint32_t cell = p.cells[k];
    V w((T)0), u((T)0);
    if (scalar_noise) {
      const T w_row = normal1<T>(key, elem, cell, 0, kStreamW) * sw;
      w = V(w_row);
      if constexpr (kNeedU) u = V(th * ((T)0.5 * w_row + normal1<T>(key, elem, cell, 0, kStreamH) * sh));
    } else if constexpr (W == 4) {
      T z[4];
      normal4<T>(key, elem >> 2, cell, 0, kStreamW, z);
#pragma unroll
      for (int q = 0; q < 4; ++q) w.v[q] = z[q] * sw;
      if constexpr (kNeedU) {
        normal4<T>(key, elem >> 2, cell, 0, kStreamH, z);
#pragma unroll
        for (int q = 0; q < 4; ++q) u.v[q] = th * ((T)0.5 * w.v[q] + z[q] * sh);
      }
    } else {
      // one element per lane -- or (W = d, a generated row model: specialise.py) a lane that owns a whole ROW of a small
      // coupled system and draws its d increments one by one
#pragma unroll
      for (int q = 0; q < W; ++q) {
        w.v[q] = normal1<T>(key, elem + (uint64_t)q, cell, 0, kStreamW) * sw;
        if constexpr (kNeedU) u.v[q] = th * ((T)0.5 * w.v[q] + normal1<T>(key, elem + (uint64_t)q, cell, 0, kStreamH) * sh);
      }
    }
    stage_times<T, METHOD>(row[7], dt, m.tslot);
    const V y1 = scheme_step<T, METHOD, V, M, V>(y, m, w, u, dt, half_dt, rdt, sqrt_dt);
    if (__builtin_expect(k + 1 == next_out, 0)) {
      while (j < p.n_out && p.out_step[j] == k + 1) {
        const T w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        const bool exact = (w0 == (T)0 && w1 == (T)1);
        Pack<T, W> ov;
#pragma unroll
        for (int q = 0; q < W; ++q) ov.v[q] = exact ? y1.v[q] : (w0 * y.v[q] + w1 * y1.v[q]);
        store<T, W>(p.ys + (int64_t)j * p.n, i, ov);
        ++j;
      }
      next_out = next_output_step(p.out_step, j, p.n_out);
    }
    y = y1;
  }
}

// ---- ... and their path-wise sensitivities ------------------------------------------------------------------------------
// Training THROUGH the solver (ordinary autograd through torchsde.sdeint, _core/sdeint.py:27-112) for an SDE stated as
// expression programs: the same programs run on forward-mode dual numbers. Tangent slot 0 is d/dy0; slots 1..4 belong to up
// to four constant rows (`param_slot[k]` = slot of row k, or -1): the per-channel parameters of the user's module. Every
// recursion is elementwise, so each tangent is one more scalar recursion in registers -- what `trajectory_kernel<.., SENS>`
// does for the affine form, for arbitrary elementwise code.
template <typename T>
TSDE_D Dual<T> operator-(const Dual<T>& x, const Dual<T>& y) {
  Dual<T> r;
  r.v = x.v - y.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = x.d[i] - y.d[i];
  return r;
}
template <typename T>
TSDE_D Dual<T> operator*(const Dual<T>& x, const Dual<T>& y) {
  Dual<T> r;
  r.v = x.v * y.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = x.d[i] * y.v + x.v * y.d[i];
  return r;
}
template <typename T>
TSDE_D Dual<T> operator/(const Dual<T>& x, const Dual<T>& y) {
  Dual<T> r;
  const T inv = (T)1 / y.v;
  r.v = x.v / y.v;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = (x.d[i] - r.v * y.d[i]) * inv;
  return r;
}
// phi(x) with derivative `slope` at x.v
template <typename T>
TSDE_D Dual<T> chain(const Dual<T>& x, T value, T slope) {
  Dual<T> r;
  r.v = value;
#pragma unroll
  for (int i = 0; i < kSens; ++i) r.d[i] = slope * x.d[i];
  return r;
}

template <typename T>
struct ProgSensArgs;

template <typename T>
struct ProgSensModel {
  using S = Dual<T>;
  TSDE_D void setup(const ProgSensArgs<T>& q, int64_t column);
  const uint32_t* code;
  int f_len, g_len, dg_len;
  const T* consts;
  const int8_t* param_slot;        // per constant row: tangent slot 1..4, or -1 (kernel arguments, uniform)
  int64_t d, col;
  T tslot[4];

  TSDE_D S constant(uint32_t k) const {
    S c(consts[(int64_t)k * d + col]);
    const int slot = param_slot[k];
    if (slot > 0) c.d[slot] = (T)1;
    return c;
  }

  TSDE_D S run(const uint32_t* prog, int len, const S& x, const T time) const {
    S s0((T)0), s1((T)0), s2((T)0), s3((T)0);
    uint32_t fetched = prog[0];
    for (int pc = 0; pc < len; ++pc) {
      const uint32_t ins = fetched;
      fetched = prog[pc + 1 < len ? pc + 1 : pc];
      const uint32_t op = ins & 0xFFu, src = (ins >> 8) & 0xFFu, k = ins >> 16;
      if (op < kOpNeg) {
        S a = s0, b;
        if (src == kSrcStack) {
          b = s0;
          a = s1;
          s1 = s2;
          s2 = s3;
        } else if (src == kSrcConst) {
          b = constant(k);
        } else if (src == kSrcTime) {
          b = S(time);
        } else {
          b = x;
        }
        switch (op) {
          case kOpLoad:
            s3 = s2;
            s2 = s1;
            s1 = s0;
            s0 = b;
            break;
          case kOpAdd: s0 = a + b; break;
          case kOpSub: s0 = a - b; break;
          case kOpRsub: s0 = b - a; break;
          case kOpMul: s0 = a * b; break;
          case kOpDiv: s0 = a / b; break;
          default: s0 = b / a; break;
        }
      } else {
        const T v = s0.v;
        switch (op) {
          case kOpNeg: s0 = chain(s0, -v, (T)-1); break;
          case kOpExp: { const T e = exp(v); s0 = chain(s0, e, e); break; }
          case kOpLog: s0 = chain(s0, log(v), (T)1 / v); break;
          case kOpSin: s0 = chain(s0, sin(v), cos(v)); break;
          case kOpCos: s0 = chain(s0, cos(v), -sin(v)); break;
          case kOpTanh: { const T t = tanh(v); s0 = chain(s0, t, (T)1 - t * t); break; }
          case kOpSigmoid: { const T g = (T)1 / ((T)1 + exp(-v)); s0 = chain(s0, g, g * ((T)1 - g)); break; }
          case kOpSoftplus:
            s0 = chain(s0, v > (T)20 ? v : log1p(exp(v)), v > (T)20 ? (T)1 : (T)1 / ((T)1 + exp(-v)));
            break;
          case kOpSqrt: { const T r = sqrt(v); s0 = chain(s0, r, (T)0.5 / r); break; }
          case kOpAbs: s0 = chain(s0, fabs(v), v > (T)0 ? (T)1 : (v < (T)0 ? (T)-1 : (T)0)); break;
          case kOpRelu: s0 = chain(s0, v > (T)0 ? v : (T)0, v > (T)0 ? (T)1 : (T)0); break;
          case kOpRecip: { const T r = (T)1 / v; s0 = chain(s0, r, -(r * r)); break; }
          case kOpSquare: s0 = chain(s0, v * v, (T)2 * v); break;
          case kOpCube: s0 = chain(s0, (v * v) * v, (T)3 * (v * v)); break;
          default:
            s3 = s2;
            s2 = s1;
            s1 = s0;
            break;
        }
      }
    }
    return s0;
  }
  template <int SLOT>
  TSDE_D S f(const S& x) const { return run(code, f_len, x, tslot[SLOT]); }
  template <int SLOT>
  TSDE_D S g(const S& x) const { return run(code + f_len, g_len, x, tslot[SLOT]); }
  template <int SLOT>
  TSDE_D S gdg(const S& x, const S& gv, T v2) const { return (gv * v2) * run(code + f_len + g_len, dg_len, x, tslot[SLOT]); }
};

constexpr int kProgParamRows = 64;

template <typename T>
struct ProgSensArgs {
  ProgArgs<T> base;
  T* sens;                          // (n_out, kSens, n)
  int8_t param_slot[kProgParamRows];
};

template <typename T>
TSDE_D void ProgSensModel<T>::setup(const ProgSensArgs<T>& q, int64_t column) {
  code = q.base.code;
  f_len = q.base.f_len;
  g_len = q.base.g_len;
  dg_len = q.base.dg_len;
  consts = q.base.consts;
  param_slot = q.param_slot;
  d = q.base.d;
  col = column;
}

// One element per lane (the duals are six values wide); values AND sensitivities of every requested output.
// (`M`: the interpreter, or the same programs as generated straight-line code on dual numbers -- specialise.py.)
template <typename T, int METHOD, typename M = ProgSensModel<T>>
__global__ void __launch_bounds__(kBlock) trajectory_prog_sens_kernel(const ProgSensArgs<T> q) {
  constexpr bool kNeedU = METHOD == kSrk;
  const ProgArgs<T>& p = q.base;
  using S = Dual<T>;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n) return;
  M m;
  m.setup(q, i % p.d);
  S y(p.y0[i]);
  y.d[0] = (T)1;
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const uint64_t elem = key.elem0 + (uint64_t)(p.scalar_noise ? i / p.d : i);
  int j = 0;
  int next_out = next_output_step(p.out_step, 0, p.n_out);
  for (int k = 0; k < p.n_steps; ++k) {
    const T* row = p.rows + (int64_t)k * 8;
    const T dt = row[0], half_dt = row[1], rdt = row[2], sqrt_dt = row[3], sw = row[4], sh = row[5], th = row[6];
    const uint32_t cell = p.cells[k];
    const T w = normal1<T>(key, elem, cell, 0, kStreamW) * sw;
    T u = (T)0;
    if constexpr (kNeedU) u = th * ((T)0.5 * w + normal1<T>(key, elem, cell, 0, kStreamH) * sh);
    stage_times<T, METHOD>(row[7], dt, m.tslot);
    const S y1 = scheme_step<T, METHOD, S, M>(y, m, w, u, dt, half_dt, rdt, sqrt_dt);
    if (__builtin_expect(k + 1 == next_out, 0)) {
      while (j < p.n_out && p.out_step[j] == k + 1) {
        const T w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        const bool exact = (w0 == (T)0 && w1 == (T)1);
        const S o = exact ? y1 : (w0 * y + w1 * y1);
        p.ys[(int64_t)j * p.n + i] = o.v;
#pragma unroll
        for (int s = 0; s < kSens; ++s) q.sens[((int64_t)j * kSens + s) * p.n + i] = o.d[s];
        ++j;
      }
      next_out = next_output_step(p.out_step, j, p.n_out);
    }
    y = y1;
  }
}

// ---- additive noise: the drift as a program, the diffusion as a table ---------------------------------------------------
// Noise type "additive" (base_sde.py:101-102; the reference's ExAdditive, tests/problems.py:106-132): g depends on t only
// and every batch row meets the same (d, m) matrix. The host evaluates the user's g at every stage time of the solve
// (recognise.RecognisedAdditive: one batched call) and hands the kernel `gtab[step][slot][j][c]` = g(t)[c, j]; a matrix that
// does not depend on t is one slot with stride 0. A lane holds W channels of its row, draws the row's m increments itself
// (the field is (rows, m): element row * m + j -- a lane of the same row draws the same numbers) and contracts them with its
// W rows of G in ascending j. Schemes: Euler (euler.py:29-37; Milstein with additive noise is the same step, milstein.py:
// base_sde.py:157-158 gdg = 0), midpoint (midpoint.py:29-45), SRK = SRA1 (srk.py:90-111, tableaus/sra1.py) with the
// operation order of the stepwise route's `tsde_step_general_w` calls (solvers.SRK._advance_additive).
template <typename T>
struct ProgAdditiveArgs {
  ProgArgs<T> base;         // g_len = dg_len = 0
  const T* gtab;
  int64_t step_stride;      // elements between the tables of consecutive steps (0: the matrix does not depend on t)
  int64_t slot_stride;      // elements between the stage-time slots of one step (m * d)
  int32_t m;                // Brownian channels per row, <= MP of the instantiation
  int32_t quads;            // != 0: m % 4 == 0 and elem0 % 4 == 0, a row's channels are whole Philox quads
};

template <typename T, int METHOD, int W, int MP, typename M = ProgModel<T, W>>
__global__ void __launch_bounds__(kBlock) trajectory_prog_additive_kernel(const ProgAdditiveArgs<T> q) {
  constexpr bool kNeedU = METHOD == kSrk;
  using V = Vec<T, W>;
  const ProgArgs<T>& p = q.base;
  const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t i = lane * W;
  if (i >= p.n) return;
  const int64_t col = i % p.d;
  const Pack<T, W> y_init = load<T, W>(p.y0, i);
  V y;
#pragma unroll
  for (int e = 0; e < W; ++e) y.v[e] = y_init.v[e];
  M m;                      // (the interpreter, or the drift program as generated code: specialise.py)
  m.setup(p, col);
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const int nm = q.m;
  const uint64_t elem = key.elem0 + (uint64_t)(i / p.d) * (uint64_t)nm;      // the row's first Brownian channel
  int j = 0;
  int next_out = next_output_step(p.out_step, 0, p.n_out);
  for (int k = 0; k < p.n_steps; ++k) {
    const T* row = p.rows + (int64_t)k * 8;   // wave-uniform
    const T dt = row[0], half_dt = row[1], rdt = row[2], sw = row[4], sh = row[5], th = row[6], t0 = row[7];
    const uint32_t cell = p.cells[k];
    T wv[MP], uv[MP];
#pragma unroll
    for (int c = 0; c < MP; ++c) wv[c] = uv[c] = (T)0;
    if (q.quads) {
#pragma unroll
      for (int c = 0; c < MP / 4; ++c) {
        if (4 * c < nm) {
          T z[4];
          normal4<T>(key, (elem >> 2) + (uint64_t)c, cell, 0, kStreamW, z);
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[4 * c + e] = z[e] * sw;
          if constexpr (kNeedU) {
            normal4<T>(key, (elem >> 2) + (uint64_t)c, cell, 0, kStreamH, z);
#pragma unroll
            for (int e = 0; e < 4; ++e) uv[4 * c + e] = th * ((T)0.5 * wv[4 * c + e] + z[e] * sh);
          }
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < MP; ++c) {
        if (c < nm) {
          wv[c] = normal1<T>(key, elem + (uint64_t)c, cell, 0, kStreamW) * sw;
          if constexpr (kNeedU) uv[c] = th * ((T)0.5 * wv[c] + normal1<T>(key, elem + (uint64_t)c, cell, 0, kStreamH) * sh);
        }
      }
    }
    const T* gk = q.gtab + (int64_t)k * q.step_stride + col;
    // sum_j G[c, j] * weight(j) for this lane's channels, ascending j
    auto contract = [&](const T* G, auto weight) {
      V acc((T)0);
#pragma unroll
      for (int c = 0; c < MP; ++c) {
        if (c < nm) {
          const Pack<T, W> gp = load<T, W>(G, (int64_t)c * p.d);
          const T wc = weight(c);
#pragma unroll
          for (int e = 0; e < W; ++e) acc.v[e] = c == 0 ? gp.v[e] * wc : acc.v[e] + gp.v[e] * wc;
        }
      }
      return acc;
    };
    m.tslot[0] = t0;
    V y1;
    if constexpr (METHOD == kEuler) {
      y1 = (y + m.template f<0>(y) * dt) + contract(gk, [&](int c) { return wv[c]; });
    } else if constexpr (METHOD == kMidpoint) {
      m.tslot[1] = t0 + (T)0.5 * dt;
      const V yp = (y + m.template f<0>(y) * half_dt) + (T)0.5 * contract(gk, [&](int c) { return wv[c]; });
      y1 = (y + m.template f<1>(yp) * dt) + contract(gk + q.slot_stride, [&](int c) { return wv[c]; });
    } else {
      // slot 0: g(t0 + C1[0] dt) = g(t0 + dt); slot 1: g(t0 + C1[1] dt) = g(t0)
      m.tslot[1] = t0 + (T)0.75 * dt;
      const V f0 = m.template f<0>(y);
      const V h = (y + ((T)0.75 * f0) * dt) + contract(gk, [&](int c) { return ((T)1.5 * uv[c]) * rdt; });
      const V acc = (y + ((T)(1.0 / 3) * f0) * dt) + contract(gk, [&](int c) { return ((T)1 * wv[c]) + ((T)-1 * uv[c]) * rdt; });
      const V f1 = m.template f<1>(h);
      y1 = (acc + ((T)(2.0 / 3) * f1) * dt) +
           contract(gk + q.slot_stride, [&](int c) { return ((T)0 * wv[c]) + ((T)1 * uv[c]) * rdt; });
    }
    if (__builtin_expect(k + 1 == next_out, 0)) {
      while (j < p.n_out && p.out_step[j] == k + 1) {
        const T w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        const bool exact = (w0 == (T)0 && w1 == (T)1);
        Pack<T, W> ov;
#pragma unroll
        for (int e = 0; e < W; ++e) ov.v[e] = exact ? y1.v[e] : (w0 * y.v[e] + w1 * y1.v[e]);
        store<T, W>(p.ys + (int64_t)j * p.n, i, ov);
        ++j;
      }
      next_out = next_output_step(p.out_step, j, p.n_out);
    }
    y = y1;
  }
}

template <typename T, int METHOD, int MP>
static hipError_t launch_additive_mp(const ProgAdditiveArgs<T>& q, bool vec, hipStream_t s) {
  const int64_t n = q.base.n;
  if (vec) {
    hipLaunchKernelGGL((trajectory_prog_additive_kernel<T, METHOD, 4, MP>), dim3((unsigned)(((n >> 2) + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, q);
  } else {
    hipLaunchKernelGGL((trajectory_prog_additive_kernel<T, METHOD, 1, MP>), dim3((unsigned)((n + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, q);
  }
  return hipGetLastError();
}

template <typename T, int METHOD>
static hipError_t launch_additive_m(const ProgAdditiveArgs<T>& q, bool vec, hipStream_t s) {
  if (q.m <= 4) return launch_additive_mp<T, METHOD, 4>(q, vec, s);
  if (q.m <= 8) return launch_additive_mp<T, METHOD, 8>(q, vec, s);
  if (q.m <= 16) return launch_additive_mp<T, METHOD, 16>(q, vec, s);
  return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_trajectory_prog_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const uint32_t* code,
                                           int f_len, const void* consts, int n_const, const void* gtab, int time_dependent,
                                           int method, const tsde_traj_t* tr, NoiseKey key, const uint64_t* key_dev,
                                           hipStream_t s) {
  ProgAdditiveArgs<T> q;
  ProgArgs<T>& p = q.base;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  if (f_len > kProgWords) return hipErrorInvalidValue;
  for (int w = 0; w < kProgWords; ++w) p.code[w] = w < f_len ? code[w] : 0u;
  p.consts = (const T*)consts;
  p.f_len = f_len;
  p.g_len = 0;
  p.dg_len = 0;
  p.n_const = n_const;
  p.scalar_noise = 0;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key = key;
  p.key_dev = key_dev;
  if (p.n <= 0 || p.n_steps <= 0) return hipSuccess;
  const int slots = method == kEuler ? 1 : 2;
  q.gtab = (const T*)gtab;
  q.slot_stride = time_dependent ? m * d : 0;
  q.step_stride = time_dependent ? (int64_t)slots * m * d : 0;
  q.m = (int32_t)m;
  q.quads = (m % 4 == 0 && key.elem0 % 4 == 0) ? 1 : 0;
  const bool can_vec = (d % 4 == 0) && aligned16(ys) && aligned16(y0) && aligned16(gtab) && ((p.n * sizeof(T)) % 16 == 0);
  const bool vec = can_vec && (p.n >> 2) >= kTrajVecMinGroups;
  switch (method) {
    case kEuler: return launch_additive_m<T, kEuler>(q, vec, s);
    case kMidpoint: return launch_additive_m<T, kMidpoint>(q, vec, s);
    case kSrk: return launch_additive_m<T, kSrk>(q, vec, s);
    default: return hipErrorInvalidValue;
  }
}

#ifndef TSDE_SPECIALISE_TU
template hipError_t launch_trajectory_prog_additive<float>(void*, const void*, int64_t, int64_t, int64_t, const uint32_t*, int,
                                                           const void*, int, const void*, int, int, const tsde_traj_t*,
                                                           NoiseKey, const uint64_t*, hipStream_t);
template hipError_t launch_trajectory_prog_additive<double>(void*, const void*, int64_t, int64_t, int64_t, const uint32_t*, int,
                                                            const void*, int, const void*, int, int, const tsde_traj_t*,
                                                            NoiseKey, const uint64_t*, hipStream_t);
#endif

template <typename T, int METHOD>
static hipError_t launch_prog_m(const ProgArgs<T>& p, bool vec, hipStream_t s) {
  if (vec) {
    const int64_t lanes = p.n >> 2;
    hipLaunchKernelGGL((trajectory_prog_kernel<T, METHOD, 4>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)), dim3(kBlock),
                       0, s, p);
  } else {
    hipLaunchKernelGGL((trajectory_prog_kernel<T, METHOD, 1>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)), dim3(kBlock),
                       0, s, p);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t launch_trajectory_prog_diag(void* ys, void* sens, const int8_t* param_slot, const void* y0, int64_t rows, int64_t d,
                                       const uint32_t* code, int f_len, int g_len, int dg_len, const void* consts, int n_const,
                                       int scalar_noise, int method, const tsde_traj_t* tr, NoiseKey key,
                                       const uint64_t* key_dev, hipStream_t s) {
  ProgArgs<T> p;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  if (f_len + g_len + dg_len > kProgWords) return hipErrorInvalidValue;
  for (int w = 0; w < kProgWords; ++w) p.code[w] = w < f_len + g_len + dg_len ? code[w] : 0u;
  p.consts = (const T*)consts;
  p.f_len = f_len;
  p.g_len = g_len;
  p.dg_len = dg_len;
  p.n_const = n_const;
  p.scalar_noise = scalar_noise;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key = key;
  p.key_dev = key_dev;
  if (p.n <= 0 || p.n_steps <= 0) return hipSuccess;
  if (sens != nullptr) {
    ProgSensArgs<T> q;
    q.base = p;
    q.sens = (T*)sens;
    for (int k = 0; k < kProgParamRows; ++k) q.param_slot[k] = (param_slot && k < n_const) ? param_slot[k] : (int8_t)-1;
    const dim3 grid((unsigned)((p.n + kBlock - 1) / kBlock));
    switch (method) {
      case kEuler: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kEuler>), grid, dim3(kBlock), 0, s, q); break;
      case kMilIto: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kMilIto>), grid, dim3(kBlock), 0, s, q); break;
      case kMilStrat: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kMilStrat>), grid, dim3(kBlock), 0, s, q); break;
      case kMidpoint: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kMidpoint>), grid, dim3(kBlock), 0, s, q); break;
      case kSrk: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kSrk>), grid, dim3(kBlock), 0, s, q); break;
      case kHeun: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kHeun>), grid, dim3(kBlock), 0, s, q); break;
      case kEulerHeun: hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, kEulerHeun>), grid, dim3(kBlock), 0, s, q); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  // (scalar noise addresses the field by row: no alignment condition on elem0)
  const bool can_vec = (d % 4 == 0) && (scalar_noise || key.elem0 % 4 == 0) && aligned16(ys) && aligned16(y0) &&
                       ((p.n * sizeof(T)) % 16 == 0);
  const bool vec = can_vec && (p.n >> 2) >= kTrajVecMinGroups;
  switch (method) {
    case kEuler: return launch_prog_m<T, kEuler>(p, vec, s);
    case kMilIto: return launch_prog_m<T, kMilIto>(p, vec, s);
    case kMilStrat: return launch_prog_m<T, kMilStrat>(p, vec, s);
    case kMidpoint: return launch_prog_m<T, kMidpoint>(p, vec, s);
    case kSrk: return launch_prog_m<T, kSrk>(p, vec, s);
    case kHeun: return launch_prog_m<T, kHeun>(p, vec, s);
    case kEulerHeun: return launch_prog_m<T, kEulerHeun>(p, vec, s);
    default: return hipErrorInvalidValue;
  }
}

#ifndef TSDE_SPECIALISE_TU
template hipError_t launch_trajectory_prog_diag<float>(void*, void*, const int8_t*, const void*, int64_t, int64_t,
                                                       const uint32_t*, int, int, int, const void*, int, int, int,
                                                       const tsde_traj_t*, NoiseKey, const uint64_t*, hipStream_t);
template hipError_t launch_trajectory_prog_diag<double>(void*, void*, const int8_t*, const void*, int64_t, int64_t,
                                                        const uint32_t*, int, int, int, const void*, int, int, int,
                                                        const tsde_traj_t*, NoiseKey, const uint64_t*, hipStream_t);
#endif

#ifndef TSDE_SPECIALISE_TU
template hipError_t launch_trajectory_expr_diag<float>(void*, const void*, int64_t, int64_t, const void* const[8],
                                                       int64_t, int, int, int, const tsde_traj_t*, NoiseKey,
                                                       const uint64_t*, hipStream_t);
template hipError_t launch_trajectory_expr_diag<double>(void*, const void*, int64_t, int64_t, const void* const[8],
                                                        int64_t, int, int, int, const tsde_traj_t*, NoiseKey,
                                                        const uint64_t*, hipStream_t);
#endif

#ifndef TSDE_SPECIALISE_TU
template hipError_t launch_trajectory_affine_diag<float>(void*, void*, const void*, int64_t, int64_t, const void*,
                                                         const void*, const void*, const void*, int64_t, int,
                                                         const tsde_traj_t*, NoiseKey, const uint64_t*, hipStream_t);
template hipError_t launch_trajectory_affine_diag<double>(void*, void*, const void*, int64_t, int64_t, const void*,
                                                          const void*, const void*, const void*, int64_t, int,
                                                          const tsde_traj_t*, NoiseKey, const uint64_t*, hipStream_t);
#endif

}  // namespace tsde
