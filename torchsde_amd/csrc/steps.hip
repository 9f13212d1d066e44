// Fused per-step solver updates for gfx950: one kernel per solver stage over the flat (batch, state) axis,
// 16 B per lane, Brownian increment generated in registers (tsde_common.h: cell_noise).
//
// Reference arithmetic being reproduced (operation order kept, one rounding per op):
//   Euler      torchsde/_core/methods/euler.py:31-36        y1 = y0 + f*dt + g_prod
//   Midpoint   torchsde/_core/methods/midpoint.py:31-43
//   Milstein   torchsde/_core/methods/milstein.py:52-94
//   SRK/SRID2  torchsde/_core/methods/srk.py:57-88, tableaus/srid2.py:19-54
//   prod       torchsde/_core/base_sde.py:98-102 (diagonal: g*v; otherwise bmm(g, v))
#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_schemes.h"

namespace tsde {

// ---- y1 = (y0 + cf*f) + cg*(g*dW) ------------------------------------------------------------------
template <typename T>
struct StepDiagOp {
  T* y1;
  const T *y0, *f, *g;
  Coef<T> cf_;
  T cg;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T cf = cf_.get();
    const Pack<T, W> a = load<T, W, NT>(y0, i), b = load<T, W, NT>(f, i), c = load<T, W, NT>(g, i);
    Pack<T, W> w, u, o;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = drift_diffusion_update<T>(a.v[j], b.v[j], c.v[j], w.v[j], cf, cg);
    store<T, W, NT>(y1, i, o);
  }
};

// ---- y1 = (y0 + cf*f) + cg*gp ----------------------------------------------------------------------
template <typename T>
struct StepProdOp {
  T* y1;
  const T *y0, *f, *gp;
  Coef<T> cf_;
  T cg;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T cf = cf_.get();
    const Pack<T, W> a = load<T, W, NT>(y0, i), b = load<T, W, NT>(f, i), c = load<T, W, NT>(gp, i);
    Pack<T, W> o;
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = (a.v[j] + b.v[j] * cf) + cg * c.v[j];
    store<T, W, NT>(y1, i, o);
  }
};

// ---- materialise the increment of one cell (for user code that needs dW as a tensor) ---------------
template <typename T>
struct CellIncrementOp {
  T *W_out, *U_out;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    Pack<T, W> w, u;
    if (U_out) {
      cell_noise<T, W, true>(nz, i, w, u);
      store<T, W, NT>(U_out, i, u);
    } else {
      cell_noise<T, W, false>(nz, i, w, u);
    }
    store<T, W, NT>(W_out, i, w);
  }
};

// ---- Milstein ------------------------------------------------------------------------------------
template <typename T>
struct MilsteinVOp {
  T *v_out, *W_out;
  const T* g;      // optional: write the cotangent g * v of the diffusion VJP instead of v (base_sde.py:147-152)
  Coef<T> dt_;
  T scale;
  int ito;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T dt = dt_.get();
    Pack<T, W> w, u, o;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = milstein_v<T>(w.v[j], dt, scale, ito);
    if (g) {
      const Pack<T, W> gg = load<T, W, NT>(g, i);
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = gg.v[j] * o.v[j];
    }
    store<T, W, NT>(v_out, i, o);
    if (W_out) store<T, W, NT>(W_out, i, w);
  }
};

template <typename T>
struct MilsteinDiagOp {
  T* y1;
  const T *y0, *f, *g, *gdg;
  Coef<T> dt_;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T dt = dt_.get();
    const Pack<T, W> a = load<T, W, NT>(y0, i), b = load<T, W, NT>(f, i), c = load<T, W, NT>(g, i), d = load<T, W, NT>(gdg, i);
    Pack<T, W> w, u, o;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = milstein_update<T>(a.v[j], b.v[j], c.v[j], d.v[j], w.v[j], dt);
    store<T, W, NT>(y1, i, o);
  }
};

template <typename T>
struct MilsteinGfPrimeOp {
  T* yp;
  const T *y0, *f, *g;
  Coef<T> dt_, sqrt_dt_;
  int ito;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T dt = dt_.get(), sqrt_dt = sqrt_dt_.get();
    const Pack<T, W> a = load<T, W, NT>(y0, i), c = load<T, W, NT>(g, i);
    Pack<T, W> o;
    if (ito) {
      const Pack<T, W> b = load<T, W, NT>(f, i);
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = (a.v[j] + dt * b.v[j]) + c.v[j] * sqrt_dt;
    } else {
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = (a.v[j] + (T)0) + c.v[j] * sqrt_dt;
    }
    store<T, W, NT>(yp, i, o);
  }
};

template <typename T>
struct MilsteinGfDiagOp {
  T* y1;
  const T *y0, *f, *g, *gp;
  Coef<T> dt_, sqrt_dt_;
  int ito;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    // `2 * sqrt_dt` is a 0-d tensor product in the reference (milstein.py:67): rounded in T (exact: a power of two)
    const T dt = dt_.get(), two_sqrt_dt = (T)2 * sqrt_dt_.get();
    const Pack<T, W> a = load<T, W, NT>(y0, i), b = load<T, W, NT>(f, i), c = load<T, W, NT>(g, i), d = load<T, W, NT>(gp, i);
    Pack<T, W> w, u, o;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const T sq = w.v[j] * w.v[j];
      const T v = ito ? (sq - dt) : sq;
      const T gdg = ((d.v[j] - c.v[j]) * v) / two_sqrt_dt;
      o.v[j] = ((a.v[j] + b.v[j] * dt) + c.v[j] * w.v[j]) + gdg;
    }
    store<T, W, NT>(y1, i, o);
  }
};

// ---- SRK (SRID2, diagonal noise): tableau and per-element arithmetic live in tsde_schemes.h -----------
// Four kernels around the user's f, g evaluations (3 f + 4 g per step). Partial sums travel between them so that a
// later kernel never re-reads what an earlier one already folded in -- 23 streams per step (6 + 8 + 6 + 3):
//   stage 1: in  y0, f0, g0             out H0_1, H1_1, H1_2
//   stage 2: in  y0, f0, g0, f1, g1     out H0_2, acc = y1 after s = 0, 1 (srk.py:87), P = H1_3 after j = 0, 1
//   stage 3: in  P, acc, f2, g2         out H1_3, acc = y1 after s = 2
//   stage 4: in  acc, g3                out y1
// Every sum is formed in the reference's own order (H_s over j ascending, y1 over s ascending, srk.py:70-87), so the
// bits are the reference's. One liberty, as for the drift terms `Srid2::need_f` already drops: H1_2's j = 1 term has
// A1 = B1 = 0 (srid2.py:36,48) and is not added -- `x + 0*f1*dt + 0*g1*sqrt_dt == x` for finite f1, g1 (torch.equal;
// only the sign of an exact zero could differ) -- which is what lets stage 1 emit H1_2 before f1, g1 exist.
template <typename T, int STAGE>
struct SrkDiagOp {
  T* out[3];
  const T* in[5];
  Coef<T> dt_, rdt_, sqrt_dt_;
  CellNoise<T> nz;

  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T dt = dt_.get(), rdt = rdt_.get(), sqrt_dt = sqrt_dt_.get();
    constexpr T zero = (T)0;
    if constexpr (STAGE == 1) {
      const Pack<T, W> y = load<T, W, NT>(in[0], i), f0 = load<T, W, NT>(in[1], i), g0 = load<T, W, NT>(in[2], i);
      Pack<T, W> w, u, h01, h11, h12;
      cell_noise<T, W, true>(nz, i, w, u);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        h01.v[k] = srid2_h0_term<T, 1, 0>(y.v[k], f0.v[k], g0.v[k], u.v[k], dt, rdt);
        h11.v[k] = srid2_h1_term<T, 1, 0>(y.v[k], f0.v[k], g0.v[k], dt, sqrt_dt);
        h12.v[k] = srid2_h1_term<T, 2, 0>(y.v[k], f0.v[k], g0.v[k], dt, sqrt_dt);
      }
      store<T, W, NT>(out[0], i, h01);
      store<T, W, NT>(out[1], i, h11);
      store<T, W, NT>(out[2], i, h12);
    } else if constexpr (STAGE == 2) {
      const Pack<T, W> y = load<T, W, NT>(in[0], i), f0 = load<T, W, NT>(in[1], i), g0 = load<T, W, NT>(in[2], i),
                       f1 = load<T, W, NT>(in[3], i), g1 = load<T, W, NT>(in[4], i);
      Pack<T, W> w, u, h02, acc, p13;
      cell_noise<T, W, true>(nz, i, w, u);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        T h = srid2_h0_term<T, 2, 0>(y.v[k], f0.v[k], g0.v[k], u.v[k], dt, rdt);
        h02.v[k] = srid2_h0_term<T, 2, 1>(h, f1.v[k], g1.v[k], u.v[k], dt, rdt);
        T a = srid2_final_term<T, 0>(y.v[k], f0.v[k], g0.v[k], w.v[k], u.v[k], dt, rdt, sqrt_dt);
        acc.v[k] = srid2_final_term<T, 1>(a, f1.v[k], g1.v[k], w.v[k], u.v[k], dt, rdt, sqrt_dt);
        T p = srid2_h1_term<T, 3, 0>(y.v[k], zero, g0.v[k], dt, sqrt_dt);      // A1(3,0) = A1(3,1) = 0: need_f is false
        p13.v[k] = srid2_h1_term<T, 3, 1>(p, zero, g1.v[k], dt, sqrt_dt);
      }
      store<T, W, NT>(out[0], i, h02);
      store<T, W, NT>(out[1], i, acc);
      store<T, W, NT>(out[2], i, p13);
    } else if constexpr (STAGE == 3) {
      const Pack<T, W> p = load<T, W, NT>(in[0], i), a = load<T, W, NT>(in[1], i), f2 = load<T, W, NT>(in[2], i),
                       g2 = load<T, W, NT>(in[3], i);
      Pack<T, W> w, u, h13, acc;
      cell_noise<T, W, true>(nz, i, w, u);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        h13.v[k] = srid2_h1_term<T, 3, 2>(p.v[k], f2.v[k], g2.v[k], dt, sqrt_dt);
        acc.v[k] = srid2_final_term<T, 2>(a.v[k], f2.v[k], g2.v[k], w.v[k], u.v[k], dt, rdt, sqrt_dt);
      }
      store<T, W, NT>(out[0], i, h13);
      store<T, W, NT>(out[1], i, acc);
    } else {
      const Pack<T, W> a = load<T, W, NT>(in[0], i), g3 = load<T, W, NT>(in[1], i);
      Pack<T, W> w, u, y1;
      // beta3(3) = 0 (srid2.py:52): the last stage's weight does not depend on U -- (0*U)*rdt is a zero for every
      // finite U -- so U is not generated here (one Philox call and two Box-Muller pairs per 16 bytes less in a kernel
      // that only moves three streams); 0 stands in for it.
      cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
      for (int k = 0; k < W; ++k)
        y1.v[k] = srid2_final_term<T, 3>(a.v[k], zero, g3.v[k], w.v[k], zero, dt, rdt, sqrt_dt);
      store<T, W, NT>(out[0], i, y1);
    }
  }
};


// ---- adjoint augmented-state update ----------------------------------------------------------------
template <typename T>
struct AugSegOp {
  T* out;
  const T *s, *F, *G, *D;
  T cF, cG, sF, sG, sD;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    Pack<T, W> o = load<T, W, NT>(s, i);
    if (F) {
      const Pack<T, W> x = load<T, W, NT>(F, i);
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = o.v[j] + sF * (x.v[j] * cF);
    }
    if (G) {
      const Pack<T, W> x = load<T, W, NT>(G, i);
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = o.v[j] + sG * (cG * x.v[j]);
    }
    if (D) {
      const Pack<T, W> x = load<T, W, NT>(D, i);
#pragma unroll
      for (int j = 0; j < W; ++j) o.v[j] = o.v[j] + sD * x.v[j];
    }
    store<T, W, NT>(out, i, o);
  }
};

// All segments of the augmented state (y, a_y, one a_theta per parameter tensor) in ONE launch:
// a block walks 4096-element chunks; the chunk -> segment map is a small table in kernel arguments.
constexpr int kAugMaxSeg = 24;
constexpr int kAugChunk = 4096;

template <typename T>
struct AugTable {
  AugSegOp<T> seg[kAugMaxSeg];
  int64_t n[kAugMaxSeg];
  int32_t chunk_begin[kAugMaxSeg + 1];
  int32_t vec[kAugMaxSeg];
  int32_t nseg;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) aug_multi_kernel(const AugTable<T> tb) {
  const int total = tb.chunk_begin[tb.nseg];
  for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
    int sidx = 0;
    while (sidx + 1 < tb.nseg && chunk >= tb.chunk_begin[sidx + 1]) ++sidx;
    const AugSegOp<T>& op = tb.seg[sidx];
    const int64_t base = (int64_t)(chunk - tb.chunk_begin[sidx]) * kAugChunk;
    const int64_t end = (base + kAugChunk < tb.n[sidx]) ? base + kAugChunk : tb.n[sidx];
    if (tb.vec[sidx]) {
      for (int64_t i = base + (int64_t)threadIdx.x * 4; i < end; i += kBlock * 4) op.template run<4>(i);
    } else {
      for (int64_t i = base + threadIdx.x; i < end; i += kBlock) op.template run<1>(i);
    }
  }
}

template <typename T>
struct InterpOp {
  T* out;
  const T *ya, *yb;
  Coef<T> w0_, w1_;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T w0 = w0_.get(), w1 = w1_.get();
    const Pack<T, W> a = load<T, W, NT>(ya, i), b = load<T, W, NT>(yb, i);
    Pack<T, W> o;
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = w0 * a.v[j] + w1 * b.v[j];
    store<T, W, NT>(out, i, o);
  }
};

// ---- general-noise contraction: y1 = (y0 + cf*f) + cg * sum_j g[b,i,j] dW[b,j] ---------------------
// HBM-bound on g (d*m*4 B per row vs 12*d for y0,f,y1). The two vector kernels below (general_fast_kernel,
// general_rows_kernel) stream g with coalesced 16-B loads, keep the row's increments in REGISTERS (lane L sits on
// channel quad L % (m/4) of its row: one Philox call per row per lane, issued under the loads) and finish the m-long
// dot products with an xor-shuffle reduction over the m/4 neighbouring lanes: no LDS, no barrier. The LDS-staged
// design (increments generated once per block into LDS, two barriers per tile) measured 2x slower at the C3 shape,
// because g has no reuse across rows (tools/microbench_general.hip, profiles/r2_c3_lds_vs_registers.txt); only the
// generic fallback for odd shapes (general_generic_kernel: any d, m, scalar loads) still stages them.
constexpr int kGenMaxNoise = 2048;  // LDS floats/doubles of staged increments per block (generic kernel only)

template <typename T>
struct GeneralArgs {
  T* y1;
  const T *y0, *f, *g;
  int64_t B, d, m;
  T ca, cg;            // drift term is (ca*f)*cf: the reference's `alpha * f * dt` rounding order
  Coef<T> cf_, rdt_;   // the step size and its reciprocal may live in device memory (tsde_common.h: Coef)
  // weight vector the diffusion row is contracted with (SRA1, srk.py:96-109):
  //   mode 0: W      mode 1: (cu*U)*rdt      mode 2: (cw*W) + (cu*U)*rdt
  int weight_mode;
  T cw, cu;
  CellNoise<T> nz;
  int rows_per_tile;
  int vec_io;          // (unused)
  int shared;          // g is ONE (d, m) matrix for every batch row (generic kernel; the MFMA kernel always reads it so)
};

template <typename T>
TSDE_D T row_weight(const GeneralArgs<T>& a, T W, T U) {
  if (a.weight_mode == 0) return W;
  const T rdt = a.rdt_.get();
  if (a.weight_mode == 1) return (a.cu * U) * rdt;
  return (a.cw * W) + (a.cu * U) * rdt;
}

template <typename T>
TSDE_D void stage_noise(const GeneralArgs<T>& a, T* lds, int64_t row0, int rows) {
  // lds[r*m + j] = weight(dW[row0 + r, j], dU[row0 + r, j])
  const CellNoise<T>& nz = a.nz;
  const int64_t m = a.m;
  const int64_t cnt = (int64_t)rows * m;
  const int64_t base = row0 * m;
  const bool need_u = a.weight_mode != 0;
  if (nz.dW != nullptr) {
    for (int64_t t = threadIdx.x; t < cnt; t += kBlock) {
      const T W = nz.dW[base + t];
      const T U = need_u ? nz.dU[base + t] : (T)0;
      lds[t] = row_weight<T>(a, W, U);
    }
  } else {
    const T sw = nz.sw, sh = nz.sh, th = nz.th;
    const NoiseKey key = live_key(nz);
    const uint64_t e0 = key.elem0 + (uint64_t)base;
    const uint64_t q0 = e0 >> 2, q1 = (e0 + (uint64_t)cnt + 3) >> 2;
    for (uint64_t q = q0 + threadIdx.x; q < q1; q += kBlock) {
      T n[4], hn[4] = {(T)0, (T)0, (T)0, (T)0};
      normal4<T>(key, q, nz.cell, 0, kStreamW, n);
      if (need_u) normal4<T>(key, q, nz.cell, 0, kStreamH, hn);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t t = (int64_t)(q * 4 + j) - (int64_t)e0;
        if (t >= 0 && t < cnt) {
          const T W = n[j] * sw;
          const T U = th * ((T)0.5 * W + hn[j] * sh);
          lds[t] = row_weight<T>(a, W, U);
        }
      }
    }
  }
}

// Fast path: m % 4 == 0, G = m/4 a power of two <= 64, everything 16-B aligned. No LDS, no barriers:
// a wave streams a contiguous span of g with 16-B loads (U independent loads in flight per lane); lane L
// always sits on channel quad L % G of its row, so it regenerates its 4 increments only when the row changes
// (one Philox call per row per lane), multiplies, and the m-long dot product is finished by an xor-shuffle
// reduction over the G neighbouring lanes.
template <typename T>
TSDE_D void lane_weights(const GeneralArgs<T>& a, int64_t row, int lp, T (&wq)[4]) {
  const CellNoise<T>& nz = a.nz;
  const bool need_u = a.weight_mode != 0;
  const int64_t off = row * a.m + (int64_t)lp * 4;
  T W[4], Uv[4] = {(T)0, (T)0, (T)0, (T)0};
  if (nz.dW != nullptr) {
    const Pack<T, 4> w = load<T, 4>(nz.dW, off);
#pragma unroll
    for (int j = 0; j < 4; ++j) W[j] = w.v[j];
    if (need_u) {
      const Pack<T, 4> u = load<T, 4>(nz.dU, off);
#pragma unroll
      for (int j = 0; j < 4; ++j) Uv[j] = u.v[j];
    }
  } else {
    const T sw = nz.sw, sh = nz.sh, th = nz.th;
    const NoiseKey key = live_key(nz);
    const uint64_t quad = (key.elem0 + (uint64_t)off) >> 2;
    T n[4];
    normal4<T>(key, quad, nz.cell, 0, kStreamW, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) W[j] = n[j] * sw;
    if (need_u) {
      normal4<T>(key, quad, nz.cell, 0, kStreamH, n);
#pragma unroll
      for (int j = 0; j < 4; ++j) Uv[j] = th * ((T)0.5 * W[j] + n[j] * sh);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) wq[j] = row_weight<T>(a, W[j], Uv[j]);
}

constexpr int kGenUnroll = 4;

template <typename T>
__global__ void __launch_bounds__(kBlock) general_fast_kernel(const GeneralArgs<T> a) {
  const T cf = a.cf_.get();
  const int G = (int)(a.m >> 2);
  const int logG = __builtin_ctz(G);
  const int64_t row4 = a.d * G;            // 16-B groups of g per batch row
  const int64_t total = a.B * row4;
  const int lane = threadIdx.x & 63;
  const int lp = lane & (G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  constexpr int64_t kSpan = 64 * kGenUnroll;
  int64_t cached_row = -1;
  T wq[4] = {(T)0, (T)0, (T)0, (T)0};
  for (int64_t s0 = wave * kSpan; s0 < total; s0 += n_waves * kSpan) {
    Pack<T, 4> gq[kGenUnroll];
    int64_t row[kGenUnroll], rem[kGenUnroll];
#pragma unroll
    for (int u = 0; u < kGenUnroll; ++u) {
      const int64_t v = s0 + u * 64 + lane;
      row[u] = -1;
      if (v < total) {
        gq[u] = load<T, 4>(a.g, v * 4);
        row[u] = v / row4;
        rem[u] = v - row[u] * row4;
      }
    }
#pragma unroll
    for (int u = 0; u < kGenUnroll; ++u) {
      T part = (T)0;
      const bool live = row[u] >= 0;
      if (live) {
        if (row[u] != cached_row) {
          lane_weights<T>(a, row[u], lp, wq);
          cached_row = row[u];
        }
        part = ((gq[u].v[0] * wq[0] + gq[u].v[1] * wq[1]) + gq[u].v[2] * wq[2]) + gq[u].v[3] * wq[3];
      }
      for (int off = 1; off < G; off <<= 1) part += __shfl_xor(part, off, 64);
      if (live && lp == 0) {
        const int64_t o = row[u] * a.d + (rem[u] >> logG);
        a.y1[o] = (a.y0[o] + (a.ca * a.f[o]) * cf) + a.cg * part;
      }
    }
  }
}

// Rows that are a whole number (NC = 1, 2, 4 or 8) of 64-lane, 16-B vector loads (C3: d*m/4 = 128 -> NC = 2): no index
// division, and ONE Philox call of the wave serves several rows. A row needs only G = m/4 quads of increments, so the 64
// lanes of one call generate the quads of 64/G rows at once (lane L: quad L%G of row L/G); each row then picks its four
// weights up from the lane that made them with a wave shuffle. Round 3's form made one call per row -- 64 lanes
// computing G distinct quads -- and at the C3 shape that VALU work (~5 us per launch per SIMD) was as long as the HBM
// time it was supposed to hide under (rocprofv3: 0.54 of the HBM peak with traffic = 1.00 x algorithmic). `rows_per_wave`
// (1 .. 64/G, chosen by the launcher so that the launch still has a few waves per SIMD) rows share a call; their loads
// go out in sub-batches of 8 vector loads per lane, the first one BEFORE the RNG. The weights, the order of the four
// products and the xor-shuffle reduction are unchanged, hence the bits.
template <typename T, int NC>
__global__ void __launch_bounds__(kBlock) general_rows_kernel(const GeneralArgs<T> a) {
  constexpr int SB = NC >= 8 ? 1 : 8 / NC;      // rows per sub-batch: SB * NC = 8 loads of g in flight per lane
  const T cf = a.cf_.get();
  const int G = (int)(a.m >> 2);
  const int logG = __builtin_ctz(G);
  const int lane = threadIdx.x & 63;
  const int lp = lane & (G - 1);
  const int RW = a.rows_per_tile;               // rows per wave iteration (a power of two, <= 64 / G)
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  const int64_t n_groups = (a.B + RW - 1) / RW;
  const int outs_per_chunk = 64 >> logG;        // outputs (b, i) finished by one 64-lane load
  for (int64_t group = wave; group < n_groups; group += n_waves) {
    const int64_t row0 = group * RW;
    const int rows_here = (int)((a.B - row0 < RW) ? (a.B - row0) : RW);
    Pack<T, 4> gq[SB][NC];
    T y0v[SB][NC], fv[SB][NC];
    T wq[4] = {(T)0, (T)0, (T)0, (T)0};
    for (int sb0 = 0; sb0 < rows_here; sb0 += SB) {
#pragma unroll
      for (int s = 0; s < SB; ++s) {
        if (sb0 + s < rows_here) {
          const int64_t row = row0 + sb0 + s;
          const T* grow = a.g + row * (int64_t)(NC * 64 * 4);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            gq[s][c] = load<T, 4>(grow, (int64_t)(c * 64 + lane) * 4);
            if (lp == 0) {
              const int64_t o = row * a.d + c * outs_per_chunk + (lane >> logG);
              y0v[s][c] = a.y0[o];
              fv[s][c] = a.f[o];
            }
          }
        }
      }
      if (sb0 == 0) {      // the increments of all rows of this group, after the first loads have been requested
        const int mine = lane >> logG;
        if (mine < rows_here) lane_weights<T>(a, row0 + mine, lp, wq);
      }
#pragma unroll
      for (int s = 0; s < SB; ++s) {
        if (sb0 + s < rows_here) {
          const int64_t row = row0 + sb0 + s;
          const int src = ((sb0 + s) << logG) + lp;
          T w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = __shfl(wq[j], src, 64);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            T part = ((gq[s][c].v[0] * w[0] + gq[s][c].v[1] * w[1]) + gq[s][c].v[2] * w[2]) + gq[s][c].v[3] * w[3];
            for (int off = 1; off < G; off <<= 1) part += __shfl_xor(part, off, 64);
            if (lp == 0) {
              const int64_t o = row * a.d + c * outs_per_chunk + (lane >> logG);
              a.y1[o] = (y0v[s][c] + (a.ca * fv[s][c]) * cf) + a.cg * part;
            }
          }
        }
      }
    }
  }
}

// ---- shared (batch-broadcast) diffusion on the matrix cores -----------------------------------------------------
// Additive noise returned as `sigma.expand(B, d, m)`: every batch row is contracted with the SAME (d, m) matrix S, so
// g . w is ONE dense product  out(B, d) = w(B, m) . S^T  -- the only matrix-core-shaped product on this path (the
// reference: base_sde.py:101-102 -> misc.py:62-63 `bmm` over B copies of S; SRA1's stages srk.py:96-109). One launch:
//   * S is staged once per block in LDS (rows padded to m16 + 4 so that the 16 lanes of a quarter-wave read distinct
//     16-B slots), zero-padded to whole 16 x 16 tiles;
//   * a wave owns 16 batch rows at a time. MFMA operands (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64):
//       A (16 x 4)  = w[batch row][k]           lane L supplies w[row0 + L%16][k(L/16)]
//       B (4 x 16)  = S^T[k][channel]           lane L supplies S[16t + L%16][k(L/16)]     (one ds_read_b128 per 4 steps)
//     The sum over k does not care in which ORDER the k's are fed as long as A and B agree, so step (j, e) feeds
//     k = 4*(L/16 + 4j) + e: lane (r, q) then needs exactly the four increments of Philox quad (row0 + r, q + 4j) --
//     each quad of each row is generated ONCE, by one lane, in registers (`lane_weights`), never written anywhere;
//   * accumulator register v of lane L is out[row0 + rowof(L/16, v)][16t + L%16]: the 16 lanes of a quarter-wave sit on
//     16 CONSECUTIVE channels of one batch row, so every y0 / f / y1 access of a quarter-wave is one contiguous 64-B
//     (f64: 128-B) segment -- no layout change, no LDS round trip, three VALU operations per output;
//     (two earlier forms, timed in profiles/r4_shared_mfma_forms.txt: channels on the M side -- a lane owns 4 channels of
//     one row, 16-B accesses -- makes every load instruction touch 16 cache lines for 16 B each (0.50 of the HBM peak
//     at d = 32..64, 0.28 at d = 128); the same with a coalesced fetch and an LDS patch to change layout costs ~300
//     VALU operations and 8*DT registers per tile (0.65 at d = 32, 0.18 at d = 128));
//   * epilogue fused: y1 = (y0 + (ca*f)*cf) + cg*acc, the rounding order of the other step kernels.
// HBM-bound (12*d bytes per row against 2*d*m flops): the matrix cores are there to keep the VALU free for the RNG.
typedef float shared_f4 __attribute__((ext_vector_type(4)));
typedef double shared_d4 __attribute__((ext_vector_type(4)));

template <typename T>
struct SharedMfma;
template <>
struct SharedMfma<float> {
  using acc_t = shared_f4;
  TSDE_D static acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  TSDE_D static int rowof(int part, int v) { return 4 * part + v; }     // M index of accumulator register v
};
template <>
struct SharedMfma<double> {
  using acc_t = shared_d4;
  TSDE_D static acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  TSDE_D static int rowof(int part, int v) { return 4 * v + part; }
};

template <typename T, int DT>      // DT = ceil(d / 16): channel tiles of a batch row, all held by one wave
__global__ void __launch_bounds__(kBlock) shared_mfma_kernel(const GeneralArgs<T> a) {
  using M = SharedMfma<T>;
  extern __shared__ __align__(16) unsigned char shared_raw[];
  T* S = reinterpret_cast<T*>(shared_raw);
  const int d = (int)a.d, m = (int)a.m;
  const int m16 = (m + 15) & ~15, ld = m16 + 4, MJ = m16 >> 4;
  // 16 threads per row of S, 16 rows per pass, 16-B groups (m % 4 == 0 and S is 16-B aligned: the launcher checked)
  for (int i = threadIdx.x >> 4; i < DT * 16; i += kBlock / 16) {
    for (int k = (threadIdx.x & 15) * 4; k < m16; k += 64) {
      Pack<T, 4> z = {{(T)0, (T)0, (T)0, (T)0}};
      if (i < d && k < m) z = load<T, 4>(a.g, (int64_t)i * m + k);
      store<T, 4>(S, i * ld + k, z);
    }
  }
  __syncthreads();
  const T cf = a.cf_.get();
  const int lane = threadIdx.x & 63, r = lane & 15, part = lane >> 4;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  const int64_t n_tiles = (a.B + 15) >> 4;
  for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
    const int64_t row0 = tile * 16;
    const bool full = row0 + 16 <= a.B && d == 16 * DT;      // (wave-uniform) no guards, pointer arithmetic only
    // all of this tile's operands are requested before the RNG runs, so the Philox rounds hide the memory latency
    T y0v[DT][4], fv[DT][4];
    int64_t at[4];                                            // element offset of (row of register v, channel r)
#pragma unroll
    for (int v = 0; v < 4; ++v) at[v] = (row0 + M::rowof(part, v)) * d + r;
    if (full) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          y0v[t][v] = a.y0[at[v] + 16 * t];
          fv[t][v] = a.f[at[v] + 16 * t];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const bool ok = row0 + M::rowof(part, v) < a.B && 16 * t + r < d;
          y0v[t][v] = ok ? a.y0[at[v] + 16 * t] : (T)0;
          fv[t][v] = ok ? a.f[at[v] + 16 * t] : (T)0;
        }
      }
    }
    typename M::acc_t acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = typename M::acc_t{(T)0, (T)0, (T)0, (T)0};
    const int64_t wrow = row0 + r;                // the row whose increments this lane supplies (A operand)
    for (int j = 0; j < MJ; ++j) {
      const int q = part + 4 * j;                 // Philox quad of this lane within that row
      T wq[4] = {(T)0, (T)0, (T)0, (T)0};
      if (wrow < a.B && 4 * q < m) lane_weights<T>(a, wrow, q, wq);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const Pack<T, 4> sv = load<T, 4>(S + (16 * t + r) * ld + 4 * q, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = M::mfma(wq[e], sv.v[e], acc[t]);
      }
    }
    if (full) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a.y1[at[v] + 16 * t] = (y0v[t][v] + (a.ca * fv[t][v]) * cf) + a.cg * acc[t][v];
      }
    } else {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (row0 + M::rowof(part, v) < a.B && 16 * t + r < d)
            a.y1[at[v] + 16 * t] = (y0v[t][v] + (a.ca * fv[t][v]) * cf) + a.cg * acc[t][v];
        }
      }
    }
  }
}

// Generic path: any d, m (m*rows_per_tile <= kGenMaxNoise): one thread per output, scalar loads of g.
template <typename T>
__global__ void __launch_bounds__(kBlock) general_generic_kernel(const GeneralArgs<T> a) {
  const T cf = a.cf_.get();
  __shared__ T lds[kGenMaxNoise];
  const int64_t n_tiles = (a.B + a.rows_per_tile - 1) / a.rows_per_tile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * a.rows_per_tile;
    const int rows = (int)((a.B - row0 < a.rows_per_tile) ? (a.B - row0) : a.rows_per_tile);
    __syncthreads();
    stage_noise<T>(a, lds, row0, rows);
    __syncthreads();
    const int64_t nout = (int64_t)rows * a.d;
    for (int64_t o = threadIdx.x; o < nout; o += kBlock) {
      const int64_t r = o / a.d;
      const int64_t idx = row0 * a.d + o;
      const T* grow = a.g + (a.shared ? (o - r * a.d) : idx) * a.m;
      const T* wrow = lds + r * a.m;
      T acc = (T)0;
      for (int64_t j = 0; j < a.m; ++j) acc += grow[j] * wrow[j];
      a.y1[idx] = (a.y0[idx] + (a.ca * a.f[idx]) * cf) + a.cg * acc;
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------
template <typename T>
static CellNoise<T> make_noise(const tsde_noise_t* nz) {
  CellNoise<T> c;
  c.dW = (const T*)nz->dW;
  c.dU = (const T*)nz->dU;
  c.key.k0 = (uint32_t)nz->entropy;
  c.key.k1 = (uint32_t)(nz->entropy >> 32);
  c.key.elem0 = nz->elem0;
  c.cell = nz->cell;
  set_width<T>(c, nz->h);
  c.bcast_d = nz->bcast_d;
  c.key_dev = nz->entropy_dev;
  return c;
}

static bool noise_vec_ok(const tsde_noise_t* nz, bool need_u) {
  if (nz->dW == nullptr) return (nz->elem0 % 4) == 0;
  if (nz->bcast_d > 0) return (nz->bcast_d % 4) == 0;  // 4 consecutive elements share a row only if d % 4 == 0
  return aligned16(nz->dW) && (!need_u || aligned16(nz->dU));
}

template <typename T>
hipError_t launch_step_diag(void* y1, const void* y0, const void* f, const void* g, int64_t n, double cf, double cg,
                            const tsde_noise_t* nz, hipStream_t s) {
  StepDiagOp<T> op{(T*)y1, (const T*)y0, (const T*)f, (const T*)g, coef<T>(cf), (T)cg, make_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(y1) && aligned16(y0) && aligned16(f) && aligned16(g) &&
                   noise_vec_ok(nz, false);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_cell_increment(void* W_out, void* U_out, int64_t n, const tsde_noise_t* nz, hipStream_t s) {
  CellIncrementOp<T> op{(T*)W_out, (T*)U_out, make_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(W_out) && (!U_out || aligned16(U_out)) && (nz->elem0 % 4 == 0);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_step_prod(void* y1, const void* y0, const void* f, const void* gp, int64_t n, double cf, double cg,
                            hipStream_t s) {
  StepProdOp<T> op{(T*)y1, (const T*)y0, (const T*)f, (const T*)gp, coef<T>(cf), (T)cg};
  const bool vec = (n % 4 == 0) && aligned16(y1) && aligned16(y0) && aligned16(f) && aligned16(gp);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_milstein_v(void* v_out, void* W_out, const void* g, int64_t n, double dt, int ito, double scale,
                             const tsde_noise_t* nz, hipStream_t s) {
  MilsteinVOp<T> op{(T*)v_out, (T*)W_out, (const T*)g, coef<T>(dt), (T)scale, ito, make_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(v_out) && (!W_out || aligned16(W_out)) && (!g || aligned16(g)) &&
                   noise_vec_ok(nz, false);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_milstein_diag(void* y1, const void* y0, const void* f, const void* g, const void* gdg, int64_t n,
                                double dt, const tsde_noise_t* nz, hipStream_t s) {
  MilsteinDiagOp<T> op{(T*)y1, (const T*)y0, (const T*)f, (const T*)g, (const T*)gdg, coef<T>(dt), make_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(y1) && aligned16(y0) && aligned16(f) && aligned16(g) && aligned16(gdg) &&
                   noise_vec_ok(nz, false);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_milstein_gf_prime(void* yp, const void* y0, const void* f, const void* g, int64_t n, double dt,
                                    double sqrt_dt, int ito, hipStream_t s) {
  MilsteinGfPrimeOp<T> op{(T*)yp, (const T*)y0, (const T*)f, (const T*)g, coef<T>(dt), coef<T>(sqrt_dt), ito};
  const bool vec = (n % 4 == 0) && aligned16(yp) && aligned16(y0) && aligned16(f) && aligned16(g);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_milstein_gf_diag(void* y1, const void* y0, const void* f, const void* g, const void* gp, int64_t n,
                                   double dt, double sqrt_dt, int ito, const tsde_noise_t* nz, hipStream_t s) {
  MilsteinGfDiagOp<T> op{(T*)y1, (const T*)y0, (const T*)f, (const T*)g, (const T*)gp, coef<T>(dt), coef<T>(sqrt_dt),
                         ito,          make_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(y1) && aligned16(y0) && aligned16(f) && aligned16(g) && aligned16(gp) &&
                   noise_vec_ok(nz, false);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

constexpr int kSrkIn[5] = {0, 3, 5, 4, 2}, kSrkOut[5] = {0, 3, 3, 2, 1};   // operand counts of stages 1..4

template <typename T, int STAGE>
static hipError_t launch_srk_stage_t(void* const out[3], const void* const in[5], int64_t n, double dt, double rdt,
                                     double sqrt_dt, const tsde_noise_t* nz, hipStream_t s) {
  SrkDiagOp<T, STAGE> op;
  bool vec = (n % 4 == 0) && noise_vec_ok(nz, true);
  for (int j = 0; j < 3; ++j) {
    op.out[j] = j < kSrkOut[STAGE] ? (T*)out[j] : nullptr;
    vec = vec && aligned16(op.out[j]);
  }
  for (int j = 0; j < 5; ++j) {
    op.in[j] = j < kSrkIn[STAGE] ? (const T*)in[j] : nullptr;
    vec = vec && aligned16(op.in[j]);
  }
  op.dt_ = coef<T>(dt);
  op.rdt_ = coef<T>(rdt);
  op.sqrt_dt_ = coef<T>(sqrt_dt);
  op.nz = make_noise<T>(nz);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_srk_stage(int stage, void* const out[3], const void* const in[5], int64_t n, double dt, double rdt,
                            double sqrt_dt, const tsde_noise_t* nz, hipStream_t s) {
  switch (stage) {
    case 1: return launch_srk_stage_t<T, 1>(out, in, n, dt, rdt, sqrt_dt, nz, s);
    case 2: return launch_srk_stage_t<T, 2>(out, in, n, dt, rdt, sqrt_dt, nz, s);
    case 3: return launch_srk_stage_t<T, 3>(out, in, n, dt, rdt, sqrt_dt, nz, s);
    case 4: return launch_srk_stage_t<T, 4>(out, in, n, dt, rdt, sqrt_dt, nz, s);
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
hipError_t launch_aug_segments(const tsde_seg_t* segs, int nseg, double cF, double cG, hipStream_t s) {
  int done = 0;
  while (done < nseg) {
    AugTable<T> tb;
    tb.nseg = 0;
    tb.chunk_begin[0] = 0;
    while (done < nseg && tb.nseg < kAugMaxSeg) {
      const tsde_seg_t& sg = segs[done++];
      if (sg.n <= 0) continue;
      const int k = tb.nseg++;
      tb.seg[k] = AugSegOp<T>{(T*)sg.out, (const T*)sg.s, (const T*)sg.F, (const T*)sg.G, (const T*)sg.D,
                              (T)cF,      (T)cG,          (T)sg.sF,       (T)sg.sG,       (T)sg.sD};
      tb.n[k] = sg.n;
      // chunks start at multiples of 4096 elements, so pointer alignment decides the vector path
      tb.vec[k] = ((sg.n % 4 == 0) && aligned16(sg.out) && aligned16(sg.s) && (!sg.F || aligned16(sg.F)) &&
                   (!sg.G || aligned16(sg.G)) && (!sg.D || aligned16(sg.D)))
                      ? 1
                      : 0;
      tb.chunk_begin[k + 1] = tb.chunk_begin[k] + (int32_t)((sg.n + kAugChunk - 1) / kAugChunk);
    }
    if (tb.nseg == 0) break;
    const int total = tb.chunk_begin[tb.nseg];
    const int grid = total < kMaxGrid ? total : kMaxGrid;
    TSDE_LAUNCH(aug_multi_kernel<T>, dim3(grid), dim3(kBlock), 0, s, tb);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <typename T>
hipError_t launch_interp(void* out, const void* ya, const void* yb, int64_t n, double w0, double w1, hipStream_t s) {
  InterpOp<T> op{(T*)out, (const T*)ya, (const T*)yb, coef<T>(w0), coef<T>(w1)};
  const bool vec = (n % 4 == 0) && aligned16(out) && aligned16(ya) && aligned16(yb);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_step_general(void* y1, const void* y0, const void* f, const void* g, int64_t B, int64_t d, int64_t m,
                               double ca, double cf, double cg, int weight_mode, double cw, double cu, double rdt,
                               const tsde_noise_t* nz, hipStream_t s) {
  if (B <= 0 || d <= 0) return hipSuccess;
  if (m <= 0 || m > kGenMaxNoise) return hipErrorInvalidValue;
  GeneralArgs<T> a;
  a.y1 = (T*)y1;
  a.y0 = (const T*)y0;
  a.f = (const T*)f;
  a.g = (const T*)g;
  a.B = B;
  a.d = d;
  a.m = m;
  a.ca = (T)ca;
  a.cf_ = coef<T>(cf);
  a.cg = (T)cg;
  a.weight_mode = weight_mode;
  a.cw = (T)cw;
  a.cu = (T)cu;
  a.rdt_ = coef<T>(rdt);
  a.nz = make_noise<T>(nz);
  a.shared = 0;
  a.vec_io = 0;
  const int64_t G = m / 4;
  const bool pow2 = (m % 4 == 0) && G >= 1 && G <= 64 && ((G & (G - 1)) == 0);
  // the fast path loads increments (external) or forms Philox quads at row*m + 4*lane: needs 16-B alignment there
  const bool noise_ok = nz->dW ? (aligned16(nz->dW) && (!nz->dU || aligned16(nz->dU))) : (nz->elem0 % 4 == 0);
  const bool fast = pow2 && aligned16(g) && noise_ok;
  if (fast && (d * G) % 64 == 0 && ((d * G) / 64 == 1 || (d * G) / 64 == 2 || (d * G) / 64 == 4 || (d * G) / 64 == 8)) {
    const int nc = (int)((d * G) / 64);
    // rows sharing one Philox call of their wave: as many as the call covers (64 / G), less while the launch would
    // otherwise have fewer than ~4 waves per SIMD (1024 SIMDs) to overlap their loads with
    static const int64_t min_waves = [] {
      const char* e = getenv("TSDE_GENERAL_MIN_WAVES");      // (tuning knob of tools/bench_kernels.py; default below)
      return e ? (int64_t)atoll(e) : (int64_t)4096;
    }();
    int64_t rw = 64 / G;
    while (rw > 1 && (B + rw - 1) / rw < min_waves) rw >>= 1;
    a.rows_per_tile = (int)rw;
    const int64_t groups = (B + rw - 1) / rw;
    int64_t blocks = (groups + (kBlock / 64) - 1) / (kBlock / 64);   // one wave per group of rows
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    if (nc == 1) TSDE_LAUNCH((general_rows_kernel<T, 1>), dim3((int)blocks), dim3(kBlock), 0, s, a);
    else if (nc == 2) TSDE_LAUNCH((general_rows_kernel<T, 2>), dim3((int)blocks), dim3(kBlock), 0, s, a);
    else if (nc == 4) TSDE_LAUNCH((general_rows_kernel<T, 4>), dim3((int)blocks), dim3(kBlock), 0, s, a);
    else TSDE_LAUNCH((general_rows_kernel<T, 8>), dim3((int)blocks), dim3(kBlock), 0, s, a);
    return hipGetLastError();
  }
  if (fast) {
    const int64_t total4 = B * d * G;
    const int64_t spans = (total4 + 64 * kGenUnroll - 1) / (64 * kGenUnroll);   // one wave per span
    int64_t blocks = (spans + (kBlock / 64) - 1) / (kBlock / 64);
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    a.rows_per_tile = 0;
    TSDE_LAUNCH(general_fast_kernel<T>, dim3((int)blocks), dim3(kBlock), 0, s, a);
    return hipGetLastError();
  }
  // generic path: rows per tile bounded by the LDS staging buffer
  int64_t rows = kGenMaxNoise / m;
  const int64_t want = (2 * kBlock + d - 1) / d;
  if (rows > want) rows = want;
  if (rows < 1) rows = 1;
  a.rows_per_tile = (int)rows;
  const int64_t n_tiles = (B + rows - 1) / rows;
  const int grid = (int)(n_tiles < kMaxGrid ? n_tiles : kMaxGrid);
  TSDE_LAUNCH(general_generic_kernel<T>, dim3(grid), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

// S: (d, m) row-major, contiguous. Matrix cores for m % 4 == 0 (the counter RNG and external increments are addressed in
// quads of one row), m <= 64, d <= 128; other shapes take the generic per-output kernel with S shared.
template <typename T>
hipError_t launch_step_shared(void* y1, const void* y0, const void* f, const void* S, int64_t B, int64_t d, int64_t m,
                              double ca, double cf, double cg, int weight_mode, double cw, double cu, double rdt,
                              const tsde_noise_t* nz, hipStream_t s) {
  if (B <= 0 || d <= 0) return hipSuccess;
  if (m <= 0 || m > kGenMaxNoise) return hipErrorInvalidValue;
  const bool noise_ok = nz->dW ? (aligned16(nz->dW) && (!nz->dU || aligned16(nz->dU))) : (nz->elem0 % 4 == 0);
  // S^T staged in LDS: (d rounded to 16) rows of (m rounded to 16) + 4 elements; float64 at d > 112, m > 48 needs 68 KiB, which
  // a device with 64 KiB per workgroup cannot give (queried once): such shapes take the generic kernel, like the odd ones
  static const size_t lds_limit = [] {
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (size_t)(64 * 1024);
    // (the opt-in maximum -- what hipFuncSetAttribute can raise a kernel to: 160 KiB on gfx950 -- else the default one)
    if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && bytes > 0)
      return (size_t)bytes;
    if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && bytes > 0)
      return (size_t)bytes;
    return (size_t)(64 * 1024);
  }();
  const size_t lds_needed = (size_t)((d + 15) / 16) * 16 * (((m + 15) & ~(int64_t)15) + 4) * sizeof(T);
  const bool mfma = m <= 64 && m % 4 == 0 && d <= 128 && noise_ok && aligned16(S) && lds_needed <= lds_limit;
  GeneralArgs<T> a;
  a.y1 = (T*)y1;
  a.y0 = (const T*)y0;
  a.f = (const T*)f;
  a.g = (const T*)S;
  a.B = B;
  a.d = d;
  a.m = m;
  a.ca = (T)ca;
  a.cf_ = coef<T>(cf);
  a.cg = (T)cg;
  a.weight_mode = weight_mode;
  a.cw = (T)cw;
  a.cu = (T)cu;
  a.rdt_ = coef<T>(rdt);
  a.nz = make_noise<T>(nz);
  a.rows_per_tile = 0;
  a.shared = 1;
  a.vec_io = 0;
  if (!mfma) {
    // shapes the tiles do not cover (m not a multiple of 4, m > 64, d > 128): one thread per output, the increments of
    // a tile of rows staged in LDS, S read through the cache -- still without materialising B copies of it
    int64_t rows = kGenMaxNoise / m;
    const int64_t want = (2 * kBlock + d - 1) / d;
    if (rows > want) rows = want;
    if (rows < 1) rows = 1;
    a.rows_per_tile = (int)rows;
    const int64_t n_tiles = (B + rows - 1) / rows;
    const int grid = (int)(n_tiles < kMaxGrid ? n_tiles : kMaxGrid);
    TSDE_LAUNCH(general_generic_kernel<T>, dim3(grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
  }
  const int dt = (int)((d + 15) / 16);
  const int m16 = (int)((m + 15) & ~(int64_t)15);
  const size_t lds = (size_t)dt * 16 * (m16 + 4) * sizeof(T);
  const int64_t tiles = (B + 15) / 16;
  // one 16-row tile per wave, then grid-stride over at most as many blocks as are resident at once (4 per CU at the
  // register count of the wider tiles): a block stages S once, however many tiles its waves go on to process
  int64_t blocks = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
  if (blocks > 256 * 4) blocks = 256 * 4;
#define TSDE_SHARED_CASE(N)                                                                                          \
  case N: {                                                                                                          \
    if (lds > 64 * 1024) {                                                                                           \
      const hipError_t attr_ = hipFuncSetAttribute((const void*)shared_mfma_kernel<T, N>,                            \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
      if (attr_ != hipSuccess) return attr_;                                                                         \
    }                                                                                                                \
    TSDE_LAUNCH((shared_mfma_kernel<T, N>), dim3((int)blocks), dim3(kBlock), lds, s, a);                             \
    break;                                                                                                           \
  }
  switch (dt) {
    TSDE_SHARED_CASE(1)
    TSDE_SHARED_CASE(2)
    TSDE_SHARED_CASE(3)
    TSDE_SHARED_CASE(4)
    TSDE_SHARED_CASE(5)
    TSDE_SHARED_CASE(6)
    TSDE_SHARED_CASE(7)
    TSDE_SHARED_CASE(8)
    default: return hipErrorNotSupported;
  }
#undef TSDE_SHARED_CASE
  return hipGetLastError();
}

#define TSDE_INSTANTIATE(T)                                                                                          \
  template hipError_t launch_cell_increment<T>(void*, void*, int64_t, const tsde_noise_t*, hipStream_t);             \
  template hipError_t launch_step_diag<T>(void*, const void*, const void*, const void*, int64_t, double, double,     \
                                          const tsde_noise_t*, hipStream_t);                                         \
  template hipError_t launch_step_prod<T>(void*, const void*, const void*, const void*, int64_t, double, double,     \
                                          hipStream_t);                                                              \
  template hipError_t launch_step_general<T>(void*, const void*, const void*, const void*, int64_t, int64_t, int64_t, \
                                             double, double, double, int, double, double, double,                    \
                                             const tsde_noise_t*, hipStream_t);                                      \
  template hipError_t launch_step_shared<T>(void*, const void*, const void*, const void*, int64_t, int64_t, int64_t, \
                                            double, double, double, int, double, double, double,                     \
                                            const tsde_noise_t*, hipStream_t);                                       \
  template hipError_t launch_milstein_v<T>(void*, void*, const void*, int64_t, double, int, double,                  \
                                           const tsde_noise_t*, hipStream_t);                                        \
  template hipError_t launch_milstein_diag<T>(void*, const void*, const void*, const void*, const void*, int64_t,    \
                                              double, const tsde_noise_t*, hipStream_t);                             \
  template hipError_t launch_milstein_gf_prime<T>(void*, const void*, const void*, const void*, int64_t, double,     \
                                                  double, int, hipStream_t);                                         \
  template hipError_t launch_milstein_gf_diag<T>(void*, const void*, const void*, const void*, const void*, int64_t, \
                                                 double, double, int, const tsde_noise_t*, hipStream_t);             \
  template hipError_t launch_srk_stage<T>(int, void* const[3], const void* const[5], int64_t, double, double, double, \
                                          const tsde_noise_t*, hipStream_t);                                         \
  template hipError_t launch_aug_segments<T>(const tsde_seg_t*, int, double, double, hipStream_t);                   \
  template hipError_t launch_interp<T>(void*, const void*, const void*, int64_t, double, double, hipStream_t);

TSDE_INSTANTIATE(float)
TSDE_INSTANTIATE(double)

}  // namespace tsde
