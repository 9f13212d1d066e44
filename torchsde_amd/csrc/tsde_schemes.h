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
template <typename T, typename S = T>
TSDE_D S drift_diffusion_update(S y, S f, S g, T w, T cf, T cg) {
  return (y + f * cf) + cg * (g * w);
}

// v/2 of Milstein: scale * (W^2 - dt) for Ito, scale * W^2 for Stratonovich (milstein.py:54-57 with the 0.5 of :69)
template <typename T>
TSDE_D T milstein_v(T w, T dt, T scale, int ito) {
  const T sq = w * w;
  return scale * (ito ? (sq - dt) : sq);
}

// y1 = ((y0 + f*dt) + g*W) + gdg
template <typename T, typename S = T>
TSDE_D S milstein_update(S y, S f, S g, S gdg, T w, T dt) {
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

// Stage states H0_s, H1_s (srk.py:69-77) from the s = STAGE earlier stages; f[j] must be 0 where !need_f(s, j).
template <typename T, int STAGE, typename S = T>
TSDE_D void srid2_stage_states(S y, const S* f, const S* g, T u, T dt, T rdt, T sqrt_dt, S& h0, S& h1) {
  h0 = y;
  h1 = y;
#pragma unroll
  for (int j = 0; j < STAGE; ++j) {
    h0 = (h0 + ((T)Srid2::A0(STAGE, j) * f[j]) * dt) + (((T)Srid2::B0(STAGE, j) * g[j]) * u) * rdt;
    h1 = (h1 + ((T)Srid2::A1(STAGE, j) * f[j]) * dt) + ((T)Srid2::B1(STAGE, j) * g[j]) * sqrt_dt;
  }
}

// y1 = y0 + sum_s [alpha_s f_s dt + g_s * g_weight_s]   (srk.py:79-87); alpha_3 = 0 so f_3 does not exist.
template <typename T, typename S = T>
TSDE_D S srid2_final(S y, const S* f, const S* g, T Ik, T u, T dt, T rdt, T sqrt_dt) {
  S acc = y;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const T Ikk = (Ik * Ik - dt) * (T)0.5;
    const T Ikkk = ((Ik * Ik) * Ik - ((T)3 * dt) * Ik) * (T)(1.0 / 6);
    const T gw = ((((T)Srid2::beta1(s) * Ik) + ((T)Srid2::beta2(s) * Ikk) / sqrt_dt) + ((T)Srid2::beta3(s) * u) * rdt) +
                 ((T)Srid2::beta4(s) * Ikkk) * rdt;
    if (s < 3) {
      acc = (acc + ((T)Srid2::alpha(s) * f[s]) * dt) + g[s] * gw;
    } else {
      acc = (acc + (T)0) + g[s] * gw;       // alpha_3 = 0: the reference still adds the zero drift term
    }
  }
  return acc;
}

}  // namespace tsde
