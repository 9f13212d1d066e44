// The stochastic adjoint of the perceptron-drift SDE (closed_form.py: MLPDriftDiagonalSDE) on the matrix cores:
// `sdeint_adjoint(..., adjoint_method="euler")` for that module as ONE kernel per chunk of steps instead of the
// reference's ~60 torch kernels per step (adjoint.py:64-127 driving adjoint_sde.py:177-230, 296-323 through
// methods/euler.py:29-37).
//
// The reference integrates, backwards in time and on the same Brownian path, the augmented state (y, a, a_theta):
//
//   f~ = f - g g'            (Ito; the double Stratonovich correction of adjoint_sde.py:177-216. Stratonovich: f~ = f)
//   y      <- y - f~ dt - g dW                                     (the forward state, RECONSTRUCTED, not stored)
//   a      <- a + dt J_f^T a + a (g' dW - dt g g'')                (the g'^2 terms of the correction cancel)
//   a_th_f <- a_th_f + dt a^T df/dtheta_f                          (perceptron weights and biases)
//   a_th_g <- a_th_g + a (dW dg/dtheta_g - dt g dg'/dtheta_g)      (diffusion rate c and shift e; Stratonovich: first term)
//
// and with `adjoint_method="milstein"` (methods/milstein.py:52-74 on adjoint_sde.py:332-377; the default for diagonal
// Ito noise) the step gains, with v = (dW^2 - dt) / 2 (Ito) or dW^2 / 2 (Stratonovich), the ELEMENTWISE terms
//
//   y      <- ... + v g g'            a <- ... + a v (g'^2 - g g'')
//   a_th_g <- ... + a v (g' dg/dtheta_g - g dg'/dtheta_g)
//
// (g is diagonal: Milstein's correction never touches the perceptron, so the matrix products are Euler's)
//
// everything evaluated at the current reconstructed y, dW the increment of the forward cell the step walks back over.
// With f = W2^T act(W1^T y + b1) + b2 that is four matrix products per step, on v_mfma_f32_16x16x4_f32 against the
// same two LDS weight arrays the sampling and reverse-sweep kernels use (mlp_trajectory.hip, mlp_backward.hip):
//
//   z = W1^T y          (as stored)          -> h = act(z), act'(z)
//   f = W2^T h          (as stored)          accumulated tile by tile as the h tiles complete
//   u = W2 a            (16-byte rows)       -> delta = u * act'(z) * dt
//   a += W1 delta       (16-byte rows)
//
// A wave owns 16 batch rows for all steps of the launch; y and a stay in registers in the MFMA accumulator layout; dW is
// regenerated from the counter RNG; the per-step factors of the weight gradients (dt a, h, delta, y) go to a stash in
// HBM for tsde_gram_partials (mlp_backward.hip); the diffusion-parameter terms are summed over the wave's 16 rows with
// DPP row shifts and accumulated per wave in LDS. Memory is O(chunk),
// independent of the number of steps: nothing of the forward pass is kept but the states at the output times.
#include <type_traits>

#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_mlp.h"

namespace tsde {

struct MlpAdjArgs {
  float* y;                 // (B, d)  in: state at boundary k_hi; out: reconstructed state at k_lo
  float* a;                 // (B, d)  in: dL/dy at boundary k_hi; out: at k_lo
  float* stash_a;           // (k_hi - k_lo, B, d)   dt_k * a
  float* stash_hid;         // (k_hi - k_lo, B, h)   act(W1^T y + b1)
  float* stash_delta;       // (k_hi - k_lo, B, h)   delta
  float* stash_y;           // (k_hi - k_lo, B, d)   the y the step was evaluated at
  float* row_rate;          // (B, d)  row 16 k += the sums of rows 16 k .. 16 k + 15 for dL/dc (the caller sums over the batch)
  float* row_shift;         // (B, d)  ... for dL/de
  const float* W1;          // (d, h) input-major, as MlpArgs
  const float* b1;          // (h)
  const float* W2;          // (h, d)
  const float* b2;          // (d)
  const float *c, *e;       // (d) diffusion coefficients
  int32_t diff_kind;        // TSDE_DIFF_AFFINE / TSDE_DIFF_SIGMOID
  float diff_amp;
  int32_t ito;              // bit 0: Ito (corrected drift) / Stratonovich; bit 1: Milstein backward step / Euler
  const float* rows;        // (n_steps, 8) schedule rows of the FORWARD cells: [0] = dt, [4] = sqrt(h)
  const uint32_t* cells;
  int64_t B;
  int32_t d, h;
  int32_t k_lo, k_hi;
  NoiseKey key;
  const uint64_t* key_dev;
};

// Sum of `v` over the 16 lanes of a DPP row, valid in the row's last lane (inclusive scan by row_shr 1, 2, 4, 8; lanes
// shifted in from outside the row read 0).
TSDE_D float row_total(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}

template <int D, int H, int ACT, int NW, bool FULL, bool MILSTEIN>
__global__ void __launch_bounds__(NW * 64) mlp_adjoint_kernel(const MlpAdjArgs p) {
  constexpr int R = 16;
  using TL = Tile<R>;
  constexpr int TD = D / R, TH = H / R, kThreads = NW * 64;
  constexpr int S1 = H + MlpLds<R>::kPad, S2 = D + MlpLds<R>::kPad;
  extern __shared__ float lds[];
  float* W1s = lds;                 // D rows of S1: W1s[channel][hidden]
  float* W2s = W1s + D * S1;        // H rows of S2: W2s[hidden][channel]
  float* b1s = W2s + H * S2;        // H
  float* b2s = b1s + H;             // D
  float* cs = b2s + D;              // D
  float* es = cs + D;               // D
  float* sums = es + D;             // NW x 2 x D: per wave, the diffusion-parameter sums (rate, shift) of its 16 rows
  const int dT = p.d, hT = p.h;
  for (int i = threadIdx.x; i < D * H; i += kThreads) {
    const int k1 = i / H, m1 = i % H, k2 = i / D, m2 = i % D;
    W1s[k1 * S1 + m1] = (k1 < dT && m1 < hT) ? p.W1[k1 * hT + m1] : 0.0f;
    W2s[k2 * S2 + m2] = (k2 < hT && m2 < dT) ? p.W2[k2 * dT + m2] : 0.0f;
  }
  for (int i = threadIdx.x; i < H; i += kThreads) b1s[i] = i < hT ? p.b1[i] : 0.0f;
  for (int i = threadIdx.x; i < D; i += kThreads) {
    b2s[i] = i < dT ? p.b2[i] : 0.0f;
    cs[i] = i < dT ? p.c[i] : 0.0f;
    es[i] = i < dT ? p.e[i] : 0.0f;
  }
  for (int i = threadIdx.x; i < NW * 2 * D; i += kThreads) sums[i] = 0.0f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane / R, n = lane % R;
  // (as in the reverse sweep: a wave without rows is done; surplus lanes of the last partial wave shadow its last row)
  const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * R;
  if (row0 >= p.B) return;
  const int64_t row = row0 + n < p.B ? row0 + n : p.B - 1;
  const float own_row = row0 + n < p.B ? 1.0f : 0.0f;     // (a shadow lane of the last partial wave adds nothing to the sums)
  float* wave_sums = sums + wave * 2 * D;
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const uint32_t off_d = (uint32_t)(row * dT) + 4 * part, off_h = (uint32_t)(row * hT) + 4 * part;
  const uint64_t quad0 = (key.elem0 + (uint64_t)(row * dT) + 4 * part) >> 2;
  auto real_d = [&](int t) { return FULL || R * t + 4 * part < dT; };
  auto real_h = [&](int th) { return FULL || R * th + 4 * part < hT; };
  auto load_tile = [&](const float* base, int t) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (real_d(t)) v = *reinterpret_cast<const f32x4*>(base + off_d + R * t);
    return v;
  };
  auto store_tile = [&](float* base, int t, const f32x4& v) {
    if (real_d(t)) *reinterpret_cast<f32x4*>(base + off_d + R * t) = v;
  };
  auto store_hidden = [&](float* base, int th, const f32x4& v) {
    if (real_h(th)) *reinterpret_cast<f32x4*>(base + off_h + R * th) = v;
  };

  f32x4 y[TD], a[TD];
#pragma unroll
  for (int t = 0; t < TD; ++t) {
    y[t] = load_tile(p.y, t);
    a[t] = load_tile(p.a, t);
  }
  const bool sigmoid = p.diff_kind == TSDE_DIFF_SIGMOID;
  const float ito = (p.ito & 1) ? 1.0f : 0.0f;

  for (int k = p.k_hi - 1; k >= p.k_lo; --k) {
    const float* srow = p.rows + (int64_t)k * 8;
    const float dt = srow[0], sw = srow[4];
    const uint32_t cell = p.cells[k];
    const int64_t slot = k - p.k_lo;

#pragma unroll
    for (int t = 0; t < TD; ++t) {
      store_tile(p.stash_a + slot * p.B * dT, t, a[t] * dt);
      store_tile(p.stash_y + slot * p.B * dT, t, y[t]);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- z^T = W1^T y^T -> h = act(z) (stashed), act'(z) (kept); f^T += W2^T h^T as each h tile completes ------------
    f32x4 hid[TH], f[TD];
#pragma unroll
    for (int t = 0; t < TD; ++t) f[t] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int th = 0; th < TH; ++th) {
      f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < TD; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) z = TL::mfma(W1s[(R * t + 4 * part + r) * S1 + R * th + n], y[t][r], z);
      }
      // (as in the reverse sweep: the A operands arrive as 2 TD two-address reads, each feeding two MFMAs; keep four of
      //  them in flight ahead of the MFMAs instead of letting every read of the tile be hoisted to the top)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 2 * TD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (i < 2 * TD - 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 value;
      const f32x4 bias = lds_quad(b1s, R * th + 4 * part);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v, slope;
        activate_with_slope<ACT>(z[r] + bias[r], v, slope);
        value[r] = v;
        hid[th][r] = slope;
      }
      // padded hidden units: act(0 + 0) is not 0 (softplus), but their W2 rows are zero in LDS, so they add nothing
      store_hidden(p.stash_hid + slot * p.B * hT, th, value);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) f[t] = TL::mfma(W2s[(R * th + 4 * part + r) * S2 + R * t + n], value[r], f[t]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 2 * TD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (i < 2 * TD - 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- u^T = W2 a^T (rows of W2s, four consecutive channels per lane); delta = u * act'(z) * dt --------------------
#pragma unroll
    for (int th = 0; th < TH; ++th) {
      f32x4 u = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&W2s[(R * th + n) * S2 + R * t + 4 * part]);
#pragma unroll
        for (int r = 0; r < 4; ++r) u = TL::mfma(w[r], a[t][r], u);
        if ((t + 1) % 4 == 0) __builtin_amdgcn_sched_barrier(0);
      }
      hid[th] = (u * hid[th]) * dt;
      store_hidden(p.stash_delta + slot * p.B * hT, th, hid[th]);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- elementwise: dW of the forward cell again, the diffusion terms, the reconstruction of y --------------------
    // (the diffusion kind is uniform over the launch: ONE branch around the whole phase, not one per element -- with
    //  the test inside the loops the phase was 70 basic blocks and the register allocator gave up on it)
    // (the backward scheme is a template parameter, not a second uniform branch: with four instances of this phase in
    //  one kernel the 8-wave 128 x 128 variant went from 15 to 210 spilled dwords)
    auto elementwise = [&](auto is_sigmoid) {
      constexpr bool kSigmoid = decltype(is_sigmoid)::value, kMilstein = MILSTEIN;
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const int ch = R * t + 4 * part;
        float zn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint64_t quad = quad0 + 4 * t;
        asm volatile("" : "+v"(quad));          // (keeps the step-invariant first Philox round inside the loop)
        if (real_d(t)) normal4<float>(key, quad, cell, 0, kStreamW, zn);
        const f32x4 cq = lds_quad(cs, ch);
        const f32x4 eq = lds_quad(es, ch);
        const f32x4 bq = lds_quad(b2s, ch);
        float rate_terms[4], shift_terms[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yt = y[t][r], at = a[t][r], cc = cq[r];
          const float w = zn[r] * sw;
          // g, q = dg/de; then dg/dc = q y, g' = q c; for the sigmoid q' := dq/du = q (1 - 2 s): g'' = q' c^2
          // (affine: q = 1, q' = 0)
          const float u = cc * yt + eq[r];
          float g = u, q = 1.0f, qp = 0.0f;
          if constexpr (kSigmoid) {
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
            g = p.diff_amp * sg;
            q = g * (1.0f - sg);
            qp = q * (1.0f - 2.0f * sg);
          }
          const float gp = q * cc;
          const float drift = (f[t][r] + bq[r]) - ito * (g * gp);           // f~
          const float aw = at * w;                                            // a dW
          const float adg = ito * ((at * dt) * g);                           // dt a g   (Ito correction terms)
          const float gpp = qp * (cc * cc);                                  // g''
          float a1 = at + (aw * gp - adg * gpp), y1 = (yt - drift * dt) - g * w;
          // the terms of dL/de, dL/dc: a (dW dg/dtheta - ito dt g dg'/dtheta), with dg/de = q, dg'/de = q' c, dg/dc = q y,
          // dg'/dc = q + q' c y (Milstein: + a v (g' dg/dtheta - g dg'/dtheta))
          const float dgp_de = qp * cc, dgp_dc = dgp_de * yt + q;
          float to_shift = aw * q - adg * dgp_de, to_rate = (aw * q) * yt - adg * dgp_dc;
          if constexpr (kMilstein) {
            const float v = 0.5f * (w * w - ito * dt), av = at * v;
            a1 += av * (gp * gp - g * gpp);
            y1 += v * (g * gp);
            to_shift += av * (gp * q - g * dgp_de);
            to_rate += av * (gp * (q * yt) - g * dgp_dc);
          }
          rate_terms[r] = row_total(to_rate * own_row);
          shift_terms[r] = row_total(to_shift * own_row);
          a[t][r] = a1;
          y[t][r] = y1;
          // (materialised HERE: left alone, the compiler sinks this arithmetic past the last product of the step and
          //  keeps its inputs alive until then -- 226 spilled dwords in the 8-wave 128 x 128 Milstein variant, 10 with)
          asm volatile("" : "+v"(a[t][r]), "+v"(y[t][r]));
        }
        // summed over the 16 rows of the wave (the 16 lanes of a DPP row hold one channel quad of 16 rows); the row's last
        // lane adds the totals to the wave's LDS accumulators (ds_add_f32: one lane per address, in order per wave)
        if (n == R - 1 && real_d(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(&wave_sums[ch + r], rate_terms[r]);
            atomicAdd(&wave_sums[D + ch + r], shift_terms[r]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (sigmoid) elementwise(std::true_type{});
    else elementwise(std::false_type{});

    // ---- a^T += W1 delta^T (rows of W1s, four consecutive hidden units per lane) --------------------------------------
#pragma unroll
    for (int t = 0; t < TD; ++t) {
      f32x4 back = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int th = 0; th < TH; ++th) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&W1s[(R * t + n) * S1 + R * th + 4 * part]);
#pragma unroll
        for (int r = 0; r < 4; ++r) back = TL::mfma(w[r], hid[th][r], back);
        if ((th + 1) % 4 == 0) __builtin_amdgcn_sched_barrier(0);
      }
      a[t] += back;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

#pragma unroll
  for (int t = 0; t < TD; ++t) {
    store_tile(p.y, t, y[t]);
    store_tile(p.a, t, a[t]);
  }
  // the wave's sums join row `row0` of (row_rate, row_shift)
  for (int i = lane; i < dT; i += 64) {
    p.row_rate[row0 * dT + i] += wave_sums[i];
    p.row_shift[row0 * dT + i] += wave_sums[D + i];
  }
}

template <int D, int H, int ACT, int NW, bool FULL, bool MILSTEIN>
static hipError_t launch_adj_scheme(const MlpAdjArgs& p, hipStream_t s) {
  constexpr int R = 16;
  const size_t lds_bytes =
      (size_t)(D * (H + MlpLds<R>::kPad) + H * (D + MlpLds<R>::kPad) + H + 3 * D + NW * 2 * D) * sizeof(float);
  static bool configured = false;   // per instantiation
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_adjoint_kernel<D, H, ACT, NW, FULL, MILSTEIN>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int64_t rows_per_block = NW * R;
  const int64_t blocks = (p.B + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((mlp_adjoint_kernel<D, H, ACT, NW, FULL, MILSTEIN>), dim3((unsigned)blocks), dim3(NW * 64), lds_bytes,
                     s, p);
  return hipGetLastError();
}

template <int D, int H, int ACT, int NW, bool FULL>
static hipError_t launch_adj_variant(const MlpAdjArgs& p, hipStream_t s) {
  return (p.ito & 2) ? launch_adj_scheme<D, H, ACT, NW, FULL, true>(p, s)
                     : launch_adj_scheme<D, H, ACT, NW, FULL, false>(p, s);
}

// y, a, the f accumulators and act'(z) are 3 x D/4 + H/4 live registers per lane (128 at d = hidden = 128). The large
// shapes exist as 8-wave blocks (two waves per SIMD, 256 registers: at most 10 spilled dwords) and as 4-wave blocks (one
// wave per SIMD, up to 512 registers: no spills); TSDE_ADJ_WAVES=4 picks the latter (default: 8).
template <int D, int H, int ACT>
static hipError_t launch_adj_shape(const MlpAdjArgs& p, hipStream_t s) {
  const bool full = p.d == D && p.h == H;
  if constexpr (D >= 128 || H >= 256) {
    static const int waves = [] {
      const char* e = getenv("TSDE_ADJ_WAVES");
      return e ? atoi(e) : 0;
    }();
    const bool eight = waves != 4;
    if (eight) return full ? launch_adj_variant<D, H, ACT, 8, true>(p, s) : launch_adj_variant<D, H, ACT, 8, false>(p, s);
    return full ? launch_adj_variant<D, H, ACT, 4, true>(p, s) : launch_adj_variant<D, H, ACT, 4, false>(p, s);
  } else {
    return full ? launch_adj_variant<D, H, ACT, 8, true>(p, s) : launch_adj_variant<D, H, ACT, 8, false>(p, s);
  }
}

template <int D, int H>
static hipError_t launch_adj_act(const MlpAdjArgs& p, int act, hipStream_t s) {
  if (act == TSDE_ACT_TANH) return launch_adj_shape<D, H, TSDE_ACT_TANH>(p, s);
  if (act == TSDE_ACT_SOFTPLUS) return launch_adj_shape<D, H, TSDE_ACT_SOFTPLUS>(p, s);
  return hipErrorInvalidValue;
}

template <int D>
static hipError_t launch_adj_h(const MlpAdjArgs& p, int act, hipStream_t s) {
  if (p.h <= 32) return launch_adj_act<D, 32>(p, act, s);
  if (p.h <= 64) return launch_adj_act<D, 64>(p, act, s);
  if (p.h <= 128) return launch_adj_act<D, 128>(p, act, s);
  if constexpr (D <= 64) {                 // both weight arrays must fit the LDS of a CU: d * hidden <= 16384
    if (p.h <= 256) return launch_adj_act<D, 256>(p, act, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_adjoint_mlp_diag(void* y, void* a, void* stash_a, void* stash_hid, void* stash_delta, void* stash_y,
                                   void* row_rate, void* row_shift, int64_t rows, int64_t d, int64_t h, const void* W1,
                                   const void* b1, const void* W2, const void* b2, const void* c, const void* e,
                                   int diff_kind, double diff_amp, int act, int ito, const tsde_traj_t* tr, int32_t k_lo,
                                   int32_t k_hi, NoiseKey key, const uint64_t* key_dev, hipStream_t s) {
  MlpAdjArgs p;
  p.y = (float*)y;
  p.a = (float*)a;
  p.stash_a = (float*)stash_a;
  p.stash_hid = (float*)stash_hid;
  p.stash_delta = (float*)stash_delta;
  p.stash_y = (float*)stash_y;
  p.row_rate = (float*)row_rate;
  p.row_shift = (float*)row_shift;
  p.W1 = (const float*)W1;
  p.b1 = (const float*)b1;
  p.W2 = (const float*)W2;
  p.b2 = (const float*)b2;
  p.c = (const float*)c;
  p.e = (const float*)e;
  p.diff_kind = diff_kind;
  p.diff_amp = (float)diff_amp;
  p.ito = ito;
  p.rows = (const float*)tr->step_rows;
  p.cells = tr->cells;
  p.B = rows;
  p.d = (int32_t)d;
  p.h = (int32_t)h;
  p.k_lo = k_lo;
  p.k_hi = k_hi;
  p.key = key;
  p.key_dev = key_dev;
  if (rows <= 0 || k_hi <= k_lo) return hipSuccess;
  hipError_t r = hipErrorInvalidValue;
  if (d <= 32) r = launch_adj_h<32>(p, act, s);
  else if (d <= 64) r = launch_adj_h<64>(p, act, s);
  else if (d <= 128) r = launch_adj_h<128>(p, act, s);
  return r;
}

}  // namespace tsde
