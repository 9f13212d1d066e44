// Per-element arithmetic of the solver schemes, shared by the per-step kernels (steps.hip: operands streamed
// from HBM) and the whole-trajectory kernel (trajectory.hip: operands live in registers). One rounding per
// operation, operation order of the reference (build with -ffp-contract=off):
//   Euler      torchsde/_core/methods/euler.py:31-36
//   Midpoint   torchsde/_core/methods/midpoint.py:31-43
//   Milstein   torchsde/_core/methods/milstein.py:52-74  (derivative form)
//   SRK/SRID2  torchsde/_core/methods/srk.py:57-88, tableaus/srid2.py:19-54
#pragma once
#include "tsde_rng.h"

namespace tsde {

// The functions are generic in the STATE type S (default: the scalar type T). The trajectory kernel instantiates
// them with a forward-mode dual number (value + tangents, trajectory.hip) to carry path-wise sensitivities through
// exactly the same sequence of operations; time steps and Brownian increments stay plain T.

// y1 = (y0 + cf*f) + cg*(g*dW)      (Euler: cf = dt, cg = 1; midpoint predictor: cf = dt/2, cg = 1/2)
// `N` is the type of the Brownian increment handed in: T (one element), or a vector of elements processed together (the
// expression-program kernel, S = N = Vec<T, 4>: one instruction stream for a lane's four elements).
template <typename T, typename S = T, typename N = T>
TSDE_D S drift_diffusion_update(S y, S f, S g, N w, T cf, T cg) {
  return (y + f * cf) + cg * (g * w);
}

// v/2 of Milstein: scale * (W^2 - dt) for Ito, scale * W^2 for Stratonovich (milstein.py:54-57 with the 0.5 of :69)
template <typename T, typename N = T>
TSDE_D N milstein_v(N w, T dt, T scale, int ito) {
  const N sq = w * w;
  return scale * (ito ? (sq - dt) : sq);
}

// y1 = ((y0 + f*dt) + g*W) + gdg
template <typename T, typename S = T, typename N = T>
TSDE_D S milstein_update(S y, S f, S g, S gdg, N w, T dt) {
  return ((y + f * dt) + g * w) + gdg;
}

// Tableau (srid2.py:21-54); entries are cast to T at use, exactly like `python_float * tensor`.
struct Srid2 {
  static TSDE_HD constexpr double A0(int s, int j) {
    constexpr double t[4][3] = {{0, 0, 0}, {1, 0, 0}, {0.25, 0.25, 0}, {0, 0, 0}};
    return t[s][j];
  }
  static TSDE_HD constexpr double A1(int s, int j) {
    constexpr double t[4][3] = {{0, 0, 0}, {0.25, 0, 0}, {1, 0, 0}, {0, 0, 0.25}};
    return t[s][j];
  }
  static TSDE_HD constexpr double B0(int s, int j) {
    constexpr double t[4][3] = {{0, 0, 0}, {0, 0, 0}, {1, 0.5, 0}, {0, 0, 0}};
    return t[s][j];
  }
  static TSDE_HD constexpr double B1(int s, int j) {
    constexpr double t[4][3] = {{0, 0, 0}, {-0.5, 0, 0}, {1, 0, 0}, {2, -1, 0.5}};
    return t[s][j];
  }
  static TSDE_HD constexpr double alpha(int s) {
    constexpr double t[4] = {1.0 / 6, 1.0 / 6, 2.0 / 3, 0};
    return t[s];
  }
  static TSDE_HD constexpr double beta1(int s) {
    constexpr double t[4] = {-1, 4.0 / 3, 2.0 / 3, 0};
    return t[s];
  }
  static TSDE_HD constexpr double beta2(int s) {
    constexpr double t[4] = {1, -4.0 / 3, 1.0 / 3, 0};
    return t[s];
  }
  static TSDE_HD constexpr double beta3(int s) {
    constexpr double t[4] = {2, -4.0 / 3, -2.0 / 3, 0};
    return t[s];
  }
  static TSDE_HD constexpr double beta4(int s) {
    constexpr double t[4] = {-2, 5.0 / 3, -2.0 / 3, 1};
    return t[s];
  }
  // f_j enters stage s only through A0/A1; when both are zero the term is +0 and f_j is not needed.
  static TSDE_HD constexpr bool need_f(int s, int j) { return A0(s, j) != 0.0 || A1(s, j) != 0.0; }
};

// One term of the stage-state recursion (srk.py:74-75), stage STAGE, earlier stage J. `f` must be 0 where
// !need_f(STAGE, J) (the reference multiplies the real f by a zero coefficient: +0 for finite f).
template <typename T, int STAGE, int J, typename S = T, typename N = T>
TSDE_D S srid2_h0_term(S h0, S f, S g, N u, T dt, T rdt) {
  return (h0 + ((T)Srid2::A0(STAGE, J) * f) * dt) + (((T)Srid2::B0(STAGE, J) * g) * u) * rdt;
}
template <typename T, int STAGE, int J, typename S = T>
TSDE_D S srid2_h1_term(S h1, S f, S g, T dt, T sqrt_dt) {
  return (h1 + ((T)Srid2::A1(STAGE, J) * f) * dt) + ((T)Srid2::B1(STAGE, J) * g) * sqrt_dt;
}

// Stage states H0_s, H1_s (srk.py:69-77) from the s = STAGE earlier stages; f[j] must be 0 where !need_f(s, j).
template <typename T, int STAGE, typename S = T, typename N = T>
TSDE_D void srid2_stage_states(S y, const S* f, const S* g, N u, T dt, T rdt, T sqrt_dt, S& h0, S& h1) {
  h0 = y;
  h1 = y;
  if constexpr (STAGE > 0) { h0 = srid2_h0_term<T, STAGE, 0, S, N>(h0, f[0], g[0], u, dt, rdt); h1 = srid2_h1_term<T, STAGE, 0, S>(h1, f[0], g[0], dt, sqrt_dt); }
  if constexpr (STAGE > 1) { h0 = srid2_h0_term<T, STAGE, 1, S, N>(h0, f[1], g[1], u, dt, rdt); h1 = srid2_h1_term<T, STAGE, 1, S>(h1, f[1], g[1], dt, sqrt_dt); }
  if constexpr (STAGE > 2) { h0 = srid2_h0_term<T, STAGE, 2, S, N>(h0, f[2], g[2], u, dt, rdt); h1 = srid2_h1_term<T, STAGE, 2, S>(h1, f[2], g[2], dt, sqrt_dt); }
}

// The weight g_s is multiplied with in the final sum (srk.py:80-85), stage S_.
template <typename T, int S_, typename N = T>
TSDE_D N srid2_g_weight(N Ik, N u, T dt, T rdt, T sqrt_dt) {
  const N Ikk = (Ik * Ik - dt) * (T)0.5;
  const N Ikkk = ((Ik * Ik) * Ik - ((T)3 * dt) * Ik) * (T)(1.0 / 6);
  return ((((T)Srid2::beta1(S_) * Ik) + ((T)Srid2::beta2(S_) * Ikk) / sqrt_dt) + ((T)Srid2::beta3(S_) * u) * rdt) +
         ((T)Srid2::beta4(S_) * Ikkk) * rdt;
}

// One term of the final sum (srk.py:87): acc + alpha_s f_s dt + g_s * gw_s. alpha_3 = 0: f_3 does not exist and the
// reference still adds the zero drift term.
template <typename T, int S_, typename S = T, typename N = T>
TSDE_D S srid2_final_term(S acc, S f, S g, N Ik, N u, T dt, T rdt, T sqrt_dt) {
  const N gw = srid2_g_weight<T, S_, N>(Ik, u, dt, rdt, sqrt_dt);
  if constexpr (S_ < 3) return (acc + ((T)Srid2::alpha(S_) * f) * dt) + g * gw;
  else return (acc + (T)0) + g * gw;
}

// y1 = y0 + sum_s [alpha_s f_s dt + g_s * g_weight_s]   (srk.py:79-87), summed in stage order like the reference.
template <typename T, typename S = T, typename N = T>
TSDE_D S srid2_final(S y, const S* f, const S* g, N Ik, N u, T dt, T rdt, T sqrt_dt) {
  S acc = y;
  acc = srid2_final_term<T, 0, S, N>(acc, f[0], g[0], Ik, u, dt, rdt, sqrt_dt);
  acc = srid2_final_term<T, 1, S, N>(acc, f[1], g[1], Ik, u, dt, rdt, sqrt_dt);
  acc = srid2_final_term<T, 2, S, N>(acc, f[2], g[2], Ik, u, dt, rdt, sqrt_dt);
  acc = srid2_final_term<T, 3, S, N>(acc, g[3], g[3], Ik, u, dt, rdt, sqrt_dt);   // (f slot unused for s = 3)
  return acc;
}

}  // namespace tsde
