// Whole-trajectory kernel for diagonal-noise SDEs with a two-layer perceptron drift (neural-SDE sampling) on gfx950:
//
//     f(t, y) = W2 . act(W1 . y + b1) + b2          g(t, y) = c * y + e   (per channel)
//
// The drift is the one GEMM-shaped object on the torchsde hot path that is shared by the whole batch, so it is the one
// place where the matrix cores apply: a wave owns 32 batch rows for ALL steps of the solve, the weights live in LDS,
// and both layers run on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact f32, the reference's precision).
//
// Everything between the two GEMMs stays in registers, with no transpose and no LDS round trip, by computing the
// TRANSPOSED products  H^T = W1^T Y^T  and  F^T = W2^T H^T:
//   * the MFMA accumulator holds, in lane l and register r, the element (row m, column n) with
//         n = l & 31                      (the batch row)
//         m = 32*tile + (r & 3) + 8*(r >> 2) + 4*(l >> 5)        (the channel / hidden unit)
//   * the B operand of the 32x32x2 instruction wants, in lane l, the element (k = l >> 5, n = l & 31): for a fixed
//     register r the two lane halves hold the channels m(r, 0) and m(r, 1) = m(r, 0) + 4 of batch row n, which is
//     exactly a legal (k0, k1) pair -- if the A operand (the weights, read from LDS) is addressed with the same
//     pairing. So the state y, kept in accumulator layout, IS the B operand of layer 1; the activated accumulators
//     of layer 1 ARE the B operand of layer 2; and the accumulators of layer 2 come out in the layout of y.
//   * four consecutive registers (r & 3 = 0..3) of a lane are four consecutive channels of one batch row: one Philox
//     quad of the counter RNG, so the Brownian increments are the same field every other kernel draws from
//     (same path as the stepwise solve, sharding-invariant).
//
// Reference being replaced: the stepping loop torchsde/_core/base_solver.py:114-134 with euler.py:31-36 /
// milstein.py:52-74 as the step, evaluated for an SDE whose f, g are the torch modules above.
#include <type_traits>

#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_schemes.h"
#include "tsde_mlp.h"

namespace tsde {


struct MlpArgs {
  float* ys;                // (n_out, B, D)
  const float* y0;          // (B, D)
  const float* W1;          // (D, H): W1[k][m] = weight of input channel k into hidden unit m
  const float* b1;          // (H)
  const float* W2;          // (H, D)
  const float* b2;          // (D)
  const float *c, *e;       // (D) diffusion g = c*y + e
  const float* rows;        // (n_steps, 8) schedule rows as in tsde_traj_t
  const uint32_t* cells;
  const int32_t* out_step;
  const float* out_w;
  int64_t B;
  int32_t d, h;             // true state / hidden sizes (d % 4 == 0); the kernel pads them to its tile sizes D, H
  int32_t n_steps, n_out;
  int32_t method;           // TSDE_TRAJ_EULER / _MILSTEIN_ITO / _MILSTEIN_STRAT / _MIDPOINT / _SRK
  int32_t diff_kind;        // TSDE_DIFF_AFFINE: g = c*y + e;  TSDE_DIFF_SIGMOID: g = diff_amp * sigmoid(c*y + e)
  float diff_amp;
  NoiseKey key;
  const uint64_t* key_dev;
};


// NW = waves per block: all of them share one copy of the weights in LDS.
// SCHEME: 0 = the one-stage Euler / Milstein step, 1 = the two-stage Stratonovich midpoint scheme (midpoint.py:31-43),
// 2 = the stochastic Runge-Kutta scheme SRID2 (srk.py:57-88, tableaus/srid2.py; Ito, needs the space-time Levy area).
// FULL: d == D and h == H (no channel padding inside the kernel), which removes every per-tile bounds test.
// IL: explicitly scheduled step (operand reads ahead of use, noise generation between the matrix instructions) for
// one-stage schemes, unpadded shapes, 16-row waves: see the step loop. TSDE_MLP_INTERLEAVE=0 selects the plain form.
template <int D, int H, int ACT, int R, int NW, int SCHEME, bool FULL, bool IL = false>
__global__ void __launch_bounds__(NW * 64) mlp_trajectory_kernel(const MlpArgs p) {
  constexpr bool MID = SCHEME == 1, SRK = SCHEME == 2;
  using TL = Tile<R>;
  using acc_t = typename TL::acc_t;
  constexpr int TD = D / R, TH = H / R, kRegs = TL::kRegs, kThreads = NW * 64;
  // LDS row strides. The parts of a wave read weight rows 4 apart; with a stride that is a multiple of 32 floats all
  // of them hit the same banks. Padding the stride to 4 (mod 8) floats spreads the four quarters of a 16-row wave
  // over both halves of the 32 banks (2 lanes per bank: the minimum for 64 lanes).
  constexpr int S1 = H + MlpLds<R>::kPad, S2 = D + MlpLds<R>::kPad;
  extern __shared__ float lds[];
  float* W1s = lds;                 // D rows of S1
  float* W2s = W1s + D * S1;        // H rows of S2
  float* b1s = W2s + H * S2;        // H
  float* b2s = b1s + H;             // D
  float* cs = b2s + D;              // D
  float* es = cs + D;               // D
  // weights into LDS, zero-padded from (d, h) to the tile sizes (D, H): padded hidden units feed zero rows of W2 and
  // padded state channels have zero drift, zero diffusion and no noise, so they stay exactly 0
  const int dT = p.d, hT = p.h;
  for (int i = threadIdx.x; i < D * H; i += kThreads) {
    const int k1 = i / H, m1 = i % H, k2 = i / D, m2 = i % D;
    const float w1 = (k1 < dT && m1 < hT) ? p.W1[k1 * hT + m1] : 0.0f;
    const float w2 = (k2 < hT && m2 < dT) ? p.W2[k2 * dT + m2] : 0.0f;
    if constexpr (IL) {
      // K-contiguous rows (the transposes of the arrays above, same footprint): W1s[unit][channel] in H rows of S2,
      // W2s[channel][unit] in D rows of S1 -- a lane's four K values of a quad are one 16-byte read
      W1s[m1 * S2 + k1] = w1;
      (W1s + H * S2)[m2 * S1 + k2] = w2;
    } else {
      W1s[k1 * S1 + m1] = w1;
      W2s[k2 * S2 + m2] = w2;
    }
  }
  for (int i = threadIdx.x; i < H; i += kThreads) b1s[i] = i < hT ? p.b1[i] : 0.0f;
  for (int i = threadIdx.x; i < D; i += kThreads) {
    b2s[i] = i < dT ? p.b2[i] : 0.0f;
    cs[i] = i < dT ? p.c[i] : 0.0f;
    es[i] = i < dT ? p.e[i] : 0.0f;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane / R, n = lane % R;
  // A wave with no batch row left is done; in the last partial wave the surplus lanes shadow the last row: they read
  // what its lane reads, draw the same noise and write the same values to the same addresses -- no lane masks.
  const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * R;
  if (row0 >= p.B) return;
  const int64_t row = row0 + n < p.B ? row0 + n : p.B - 1;
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  // Addresses are a wave-uniform base (SGPRs) + ONE 32-bit lane offset + a constant per quad, so that no per-tile
  // 64-bit address lives in vector registers across the solve (the C ABI keeps rows * d < 2^30).
  const uint32_t off_d = (uint32_t)(row * dT);
  const uint64_t quad_row = (key.elem0 + (uint64_t)(row * dT)) >> 2;      // RNG quad of channel 0 of this row
  auto real = [&](int ch) { return FULL || ch < dT; };                    // a quad of channels is real or padding

  // state in accumulator layout: y[t][r] = y(row, R t + TL::row(r, part))
  acc_t y[TD];
#pragma unroll
  for (int t = 0; t < TD; ++t) {
#pragma unroll
    for (int q = 0; q < TL::kQuads; ++q) {
      const int ch = R * t + TL::quad_base(q, part);      // channels ch .. ch+3 = registers 4q .. 4q+3
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (real(ch)) v = *reinterpret_cast<const f32x4*>(p.y0 + off_d + ch);
#pragma unroll
      for (int s = 0; s < 4; ++s) y[t][4 * q + s] = v[s];
    }
  }

  const bool sigmoid_diffusion = p.diff_kind == TSDE_DIFF_SIGMOID;
  int jout = 0;
  for (int k = 0; k < p.n_steps; ++k) {
    const float* srow = p.rows + (int64_t)k * 8;
    const float dt = srow[0], sw = srow[4];
    const uint32_t cell = p.cells[k];

    const float half_dt = srow[1];
    const bool due = jout < p.n_out && p.out_step[jout] == k + 1;
    // Outputs that fall INSIDE this step are the reference's linear interpolation w0 y_k + w1 y_{k+1}
    // (base_solver.py:147, interp.py:15-18; weights from the host). y_k is gone by the time y_{k+1} exists (the schemes
    // update in place), so its share w0 y_k goes to the output's place in `ys` now and the end of the step adds w1 y_{k+1}
    // to it -- the same lane, in program order, (w0 y_k) + (w1 y_{k+1}) rounded like the reference's expression.
    if (due) {
      for (int j = jout; j < p.n_out && p.out_step[j] == k + 1; ++j) {
        const float w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        if (w0 == 0.0f && w1 == 1.0f) continue;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
#pragma unroll
          for (int q = 0; q < TL::kQuads; ++q) {
            const int ch = R * t + TL::quad_base(q, part);
            Pack<float, 4> o;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) o.v[c4] = w0 * y[t][4 * q + c4];
            if (real(ch)) store<float, 4>(p.ys + (int64_t)j * p.B * dT, off_d + ch, o);
          }
        }
      }
    }
    // ... and the end of the step: every output of this step receives y_{k+1}, or its weighted share
    auto emit = [&](int ch, const Pack<float, 4>& o) {
      for (int j = jout; j < p.n_out && p.out_step[j] == k + 1; ++j) {
        const float w0 = p.out_w[2 * j], w1 = p.out_w[2 * j + 1];
        float* dst = p.ys + (int64_t)j * p.B * dT;
        if (w0 == 0.0f && w1 == 1.0f) {
          store<float, 4>(dst, off_d + ch, o);
        } else {
          const f32x4 prev = *reinterpret_cast<const f32x4*>(dst + off_d + ch);
          Pack<float, 4> mixed;
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) mixed.v[c4] = prev[c4] + w1 * o.v[c4];
          store<float, 4>(dst, off_d + ch, mixed);
        }
      }
    };

    // ---- layer 1: hid^T = W1^T x^T ------------------------------------------------------------------------
    // (the scheduling barriers keep the compiler from hoisting hundreds of LDS reads ahead of the MFMAs that use
    //  them: without them the unrolled body needs > 512 registers and spills)
    // 16-row waves: the A operands of a tile arrive as `pairs` two-address reads (ds_read2_b32: weight rows r, r+1), each
    // feeding two MFMAs; four of them are kept in flight ahead of the MFMAs (hipcc alone: read -> wait -> two MFMAs)
    auto reads_ahead = [](int pairs) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (i < pairs) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (i < pairs - 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    acc_t hid[TH];
    auto hidden_layer = [&](const acc_t* x) {
      // (an opaque zero in every weight address: a multi-stage scheme evaluates the drift several times per step, and
      //  without it the compiler merges the identical LDS reads of all evaluations -- and keeps thousands of weights)
      int o1 = 0;
      asm volatile("" : "+v"(o1));
#pragma unroll
      for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int r = 0; r < kRegs; ++r) hid[th][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
#pragma unroll
          for (int r = 0; r < kRegs; ++r) {
            const float a = W1s[o1 + (R * t + TL::row(r, part)) * S1 + R * th + n];
            hid[th] = TL::mfma(a, x[t][r], hid[th]);
          }
          if (R != 16 && (t + 1) * kRegs % 16 == 0) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (R == 16) reads_ahead(2 * TD);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const f32x4 bias = lds_quad(b1s, R * th + TL::quad_base(q, part));
#pragma unroll
          for (int s = 0; s < 4; ++s) hid[th][4 * q + s] = activate<ACT>(hid[th][4 * q + s] + bias[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // ---- layer 2, one tile of channels: f^T tile = W2^T hid^T ---------------------------------------------------
    auto drift_tile = [&](int t, int o2 = 0) {
      acc_t acc;
#pragma unroll
      for (int r = 0; r < kRegs; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int r = 0; r < kRegs; ++r) {
          const float a = W2s[o2 + (R * th + TL::row(r, part)) * S2 + R * t + n];
          acc = TL::mfma(a, hid[th][r], acc);
        }
        if (R != 16 && (th + 1) * kRegs % 16 == 0) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (R == 16) reads_ahead(2 * TH);
      return acc;
    };

    if constexpr (IL) {
      // One stage, laid out as straight-line regions whose instruction order is spelled out to the scheduler with
      // scheduling-group barriers. Weights sit K-contiguous in LDS, so the four K values a lane feeds to the four
      // MFMAs of a quad are ONE 16-byte read; reads are issued two ahead of the MFMAs that consume them (left to
      // itself hipcc emits read -> wait -> two MFMAs, exposing the LDS latency 128 times per step), and the vector
      // work of tile t (Philox + Box-Muller, ~130 VALU instructions) is issued between the dependent MFMAs of the
      // same tile's drift instead of in a phase of its own. d = hidden = 128: 11.6 -> 10.6 ms per 500-step solve.
      static_assert(SCHEME == 0 && FULL && R == 16, "interleaved path: one-stage schemes, unpadded shapes, 16-row waves");
      const float* W1t = W1s;                 // H rows of S2: [unit][channel]
      const float* W2t = W1s + H * S2;        // D rows of S1: [channel][unit]
      // layer 1: per tile of hidden units, TD 16-byte operand reads feed 4 TD MFMAs; the reads run two ahead of the
      // MFMAs that consume them (the LDS latency hides behind matrix instructions instead of a wait before each pair)
#pragma unroll
      for (int th = 0; th < TH; ++th) {
        __builtin_amdgcn_sched_barrier(0);
        acc_t h4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&W1t[(R * th + n) * S2 + R * t + 4 * part]);
#pragma unroll
          for (int r = 0; r < 4; ++r) h4 = TL::mfma(a[r], y[t][r], h4);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < TD - 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 bias = lds_quad(b1s, R * th + 4 * part);
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[th][r] = activate<ACT>(h4[r] + bias[r]);
      }
      // (the diffusion kind is uniform over the launch: one branch around the tile loop, two straight-line copies)
      auto drift_noise_update = [&](auto is_sigmoid) {
      constexpr bool kSigmoid = decltype(is_sigmoid)::value;
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const int ch = R * t + 4 * part;
        uint64_t quad = quad_row + (ch >> 2);
        asm volatile("" : "+v"(quad));
        __builtin_amdgcn_sched_barrier(0);
        acc_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int th = 0; th < TH; ++th) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&W2t[(R * t + n) * S1 + R * th + 4 * part]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = TL::mfma(a[r], hid[th][r], acc);
        }
        float z[4];
        normal4<float>(key, quad, cell, 0, kStreamW, z);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // operand reads two ahead
#pragma unroll
        for (int i = 0; i < TH; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    // the four MFMAs of one operand read
          __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);   // sixteen VALU of the noise generation
          if (i < TH - 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 b2q = lds_quad(b2s, ch), cq = lds_quad(cs, ch), eq = lds_quad(es, ch);
        Pack<float, 4> o;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float yy = y[t][s];
          const float f = acc[s] + b2q[s];
          const float cc = cq[s];
          const DiffusionValue dv = diffusion_value(kSigmoid, p.diff_amp, cc, eq[s], yy);
          const float g = dv.g;
          const float w = z[s] * sw;
          float yn;
          if (p.method == TSDE_TRAJ_EULER) {
            yn = drift_diffusion_update<float>(yy, f, g, w, dt, 1.0f);
          } else {
            const float v2 = milstein_v<float>(w, dt, 0.5f, p.method == TSDE_TRAJ_MILSTEIN_ITO);
            yn = milstein_update<float>(yy, f, g, (g * v2) * (dv.q * cc), w, dt);
          }
          y[t][s] = yn;
          o.v[s] = yn;
        }
        if (due) emit(ch, o);
        __builtin_amdgcn_sched_barrier(0);
      }
      };
      if (sigmoid_diffusion) drift_noise_update(std::true_type{});
      else drift_noise_update(std::false_type{});
    } else if constexpr (SRK) {
      // SRID2 (srk.py:57-88). The diffusion is diagonal, so every stage state H1_s is elementwise in the state and the
      // drifts already known: only the three drift evaluations f(H0_0 = y), f(H0_1), f(H0_2) are matrix products (the
      // tableau's alpha_3 = 0: f(H0_3) never contributes). After stage 0 of a tile everything that depends on (y, f0,
      // W, U) is folded into four running arrays, so y, W, U and the g_s never live across a drift evaluation:
      //   acc   = y + alpha_0 f0 dt + sum_{s<3} g_s gw_s          (the step's result once alpha_1 f1, alpha_2 f2, g_3 gw_3 join)
      //   h02   = y + A0_20 f0 dt + (B0_20 g0 + B0_21 g1) U/dt     (H0_2 once A0_21 f1 dt joins)
      //   h13   = y + (B1_30 g0 + B1_31 g1 + B1_32 g2) sqrt(dt)    (H1_3 once A1_32 f2 dt joins)
      //   gw3   = beta4_3 I_kkk / dt                               (the weight of g_3 = g(H1_3))
      // and y itself is overwritten by H0_1 = y + f0 dt, the next drift evaluation's input.
      const float rdt = srow[2], sqrt_dt = srow[3], sh = srow[5], th = srow[6];
      acc_t accum[TD], h02[TD], h13[TD], gw3[TD];
      int oA = 0, oB = 0, oC = 0;               // (see hidden_layer: one opaque zero per drift evaluation)
      asm volatile("" : "+v"(oA), "+v"(oB), "+v"(oC));
      auto g_of = [&](float cc, float ee, float x) { return diffusion_value(sigmoid_diffusion, p.diff_amp, cc, ee, x).g; };
      hidden_layer(y);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const acc_t f0 = drift_tile(t, oA);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const int ch = R * t + TL::quad_base(q, part);
          float zw[4] = {0.0f, 0.0f, 0.0f, 0.0f}, zh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          uint64_t quad = quad_row + (ch >> 2);
          asm volatile("" : "+v"(quad));
          if (real(ch)) {
            normal4<float>(key, quad, cell, 0, kStreamW, zw);
            normal4<float>(key, quad, cell, 0, kStreamH, zh);
          }
          const f32x4 b2q = lds_quad(b2s, ch), cq = lds_quad(cs, ch), eq = lds_quad(es, ch);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 4 * q + s;
            const float yy = y[t][r], cc = cq[s], ee = eq[s];
            const float f = f0[r] + b2q[s];
            const float w = zw[s] * sw;
            const float u = th * (0.5f * w + zh[s] * sh);
            const float g0 = g_of(cc, ee, yy);
            // H1_1, H1_2 need f0 only (A1 rows (1/4), (1, 0)); their diffusions are known right here
            const float h11 = (yy + ((float)Srid2::A1(1, 0) * f) * dt) + ((float)Srid2::B1(1, 0) * g0) * sqrt_dt;
            const float g1 = g_of(cc, ee, h11);
            const float h12 = (yy + ((float)Srid2::A1(2, 0) * f) * dt) + ((float)Srid2::B1(2, 0) * g0) * sqrt_dt;
            const float g2 = g_of(cc, ee, h12);
            const float Ikk = (w * w - dt) * 0.5f;
            const float Ikkk = ((w * w) * w - (3.0f * dt) * w) * (float)(1.0 / 6);
            auto weight = [&](int st) {
              return ((((float)Srid2::beta1(st) * w) + ((float)Srid2::beta2(st) * Ikk) / sqrt_dt) +
                      ((float)Srid2::beta3(st) * u) * rdt) + ((float)Srid2::beta4(st) * Ikkk) * rdt;
            };
            float a0 = (yy + ((float)Srid2::alpha(0) * f) * dt) + g0 * weight(0);
            a0 = a0 + g1 * weight(1);
            a0 = a0 + g2 * weight(2);
            accum[t][r] = a0;
            gw3[t][r] = weight(3);
            h02[t][r] = ((yy + ((float)Srid2::A0(2, 0) * f) * dt) + (((float)Srid2::B0(2, 0) * g0) * u) * rdt) +
                        (((float)Srid2::B0(2, 1) * g1) * u) * rdt;
            h13[t][r] = ((yy + ((float)Srid2::B1(3, 0) * g0) * sqrt_dt) + ((float)Srid2::B1(3, 1) * g1) * sqrt_dt) +
                        ((float)Srid2::B1(3, 2) * g2) * sqrt_dt;
            y[t][r] = yy + ((float)Srid2::A0(1, 0) * f) * dt;                       // H0_1
            // (materialise the running values HERE: left alone, the compiler sinks their arithmetic to the first use,
            //  past the next drift evaluation, and keeps its six inputs per element alive until then -- 480 spilled
            //  dwords at d = hidden = 128 instead of none)
            asm volatile("" : "+v"(accum[t][r]), "+v"(gw3[t][r]), "+v"(h02[t][r]), "+v"(h13[t][r]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // f1 = f(H0_1): joins the result and H0_2 (which replaces H0_1 as the drift's input, tile by tile)
      hidden_layer(y);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const acc_t f1 = drift_tile(t, oB);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const int ch = R * t + TL::quad_base(q, part);
          const f32x4 b2q = lds_quad(b2s, ch);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 4 * q + s;
            const float f = f1[r] + b2q[s];
            accum[t][r] = accum[t][r] + ((float)Srid2::alpha(1) * f) * dt;
            y[t][r] = h02[t][r] + ((float)Srid2::A0(2, 1) * f) * dt;              // H0_2
            asm volatile("" : "+v"(accum[t][r]), "+v"(y[t][r]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // f2 = f(H0_2): joins the result and completes H1_3, whose diffusion is the last term
      hidden_layer(y);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const acc_t f2 = drift_tile(t, oC);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const int ch = R * t + TL::quad_base(q, part);
          const f32x4 b2q = lds_quad(b2s, ch), cq = lds_quad(cs, ch), eq = lds_quad(es, ch);
          Pack<float, 4> o;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 4 * q + s;
            const float f = f2[r] + b2q[s];
            const float g3 = g_of(cq[s], eq[s], h13[t][r] + ((float)Srid2::A1(3, 2) * f) * dt);
            const float yn = (accum[t][r] + ((float)Srid2::alpha(2) * f) * dt) + g3 * gw3[t][r];
            y[t][r] = yn;
            o.v[s] = yn;
          }
          if (due && real(ch)) emit(ch, o);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (!MID) {
      // one stage; in place: tile t of the state is only read by its own update
      hidden_layer(y);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const acc_t acc = drift_tile(t);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const int ch = R * t + TL::quad_base(q, part);
          float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          // (opaque to the optimiser: otherwise the step-invariant first Philox round of every quad is hoisted out
          //  of the step loop and pinned in ~4 registers per quad)
          uint64_t quad = quad_row + (ch >> 2);
          asm volatile("" : "+v"(quad));
          if (real(ch)) normal4<float>(key, quad, cell, 0, kStreamW, z);
          const f32x4 b2q = lds_quad(b2s, ch), cq = lds_quad(cs, ch), eq = lds_quad(es, ch);
          Pack<float, 4> o;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 4 * q + s;
            const float yy = y[t][r];
            const float f = acc[r] + b2q[s];
            const float cc = cq[s];
            const DiffusionValue dv = diffusion_value(sigmoid_diffusion, p.diff_amp, cc, eq[s], yy);
            const float g = dv.g;
            const float w = z[s] * sw;
            float yn;
            if (p.method == TSDE_TRAJ_EULER) {
              yn = drift_diffusion_update<float>(yy, f, g, w, dt, 1.0f);
            } else {
              const float v2 = milstein_v<float>(w, dt, 0.5f, p.method == TSDE_TRAJ_MILSTEIN_ITO);
              yn = milstein_update<float>(yy, f, g, (g * v2) * (dv.q * cc), w, dt);
            }
            y[t][r] = yn;
            o.v[s] = yn;
          }
          if (due && real(ch)) emit(ch, o);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // stage 1: y' = (y + f(y) dt/2) + (g(y) W)/2, keeping W for stage 2
      acc_t yp[TD], wk[TD];
      hidden_layer(y);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const acc_t acc = drift_tile(t);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const int ch = R * t + TL::quad_base(q, part);
          float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          // (opaque to the optimiser: otherwise the step-invariant first Philox round of every quad is hoisted out
          //  of the step loop and pinned in ~4 registers per quad)
          uint64_t quad = quad_row + (ch >> 2);
          asm volatile("" : "+v"(quad));
          if (real(ch)) normal4<float>(key, quad, cell, 0, kStreamW, z);
          const f32x4 b2q = lds_quad(b2s, ch), cq = lds_quad(cs, ch), eq = lds_quad(es, ch);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 4 * q + s;
            const float yy = y[t][r];
            const float w = z[s] * sw;
            wk[t][r] = w;
            yp[t][r] = drift_diffusion_update<float>(yy, acc[r] + b2q[s],
                                                     diffusion_value(sigmoid_diffusion, p.diff_amp, cq[s], eq[s], yy).g, w,
                                                     half_dt, 0.5f);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // stage 2: y1 = (y + f(y') dt) + g(y') W
      hidden_layer(yp);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const acc_t acc = drift_tile(t);
#pragma unroll
        for (int q = 0; q < TL::kQuads; ++q) {
          const int ch = R * t + TL::quad_base(q, part);
          const f32x4 b2q = lds_quad(b2s, ch), cq = lds_quad(cs, ch), eq = lds_quad(es, ch);
          Pack<float, 4> o;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int r = 4 * q + s;
            const float yn = drift_diffusion_update<float>(
                y[t][r], acc[r] + b2q[s], diffusion_value(sigmoid_diffusion, p.diff_amp, cq[s], eq[s], yp[t][r]).g, wk[t][r],
                dt, 1.0f);
            y[t][r] = yn;
            o.v[s] = yn;
          }
          if (due && real(ch)) emit(ch, o);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    while (jout < p.n_out && p.out_step[jout] == k + 1) ++jout;
  }
}

template <int D, int H, int ACT, int R, int NW, int SCHEME, bool FULL, bool IL = false>
static hipError_t launch_mlp_full(const MlpArgs& p, hipStream_t s) {
  const size_t lds_bytes = MlpLds<R>::bytes(D, H);
  static bool configured = false;   // per instantiation
  if (!configured) {
    const hipError_t e =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_trajectory_kernel<D, H, ACT, R, NW, SCHEME, FULL, IL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int64_t rows_per_block = NW * R;
  const int64_t blocks = (p.B + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((mlp_trajectory_kernel<D, H, ACT, R, NW, SCHEME, FULL, IL>), dim3((unsigned)blocks), dim3(NW * 64),
                     lds_bytes, s, p);
  return hipGetLastError();
}

template <int D, int H, int ACT, int R, int NW, int SCHEME>
static hipError_t launch_mlp_variant(const MlpArgs& p, hipStream_t s) {
  // (the multi-stage schemes keep the bounds tests even for unpadded shapes: without them hipcc's schedule of the wider
  //  basic blocks needs MORE registers -- 390 spilled dwords instead of 62 for the midpoint scheme at d = hidden = 128)
  if constexpr (SCHEME == 0) {
    if (p.d == D && p.h == H) {
      if constexpr (R == 16) {
        static const bool interleave = [] {
          const char* e = getenv("TSDE_MLP_INTERLEAVE");
          return e == nullptr || atoi(e) != 0;
        }();
        if (interleave) return launch_mlp_full<D, H, ACT, R, NW, SCHEME, true, true>(p, s);
      }
      return launch_mlp_full<D, H, ACT, R, NW, SCHEME, true>(p, s);
    }
  }
  return launch_mlp_full<D, H, ACT, R, NW, SCHEME, false>(p, s);
}

// Variant choice (tools/bench_mlp_trajectory.py, MI355X): 16-row waves in 8-wave blocks everywhere. The weights of a
// d = hidden = 128 net fill the LDS of a CU, so all waves of a CU share one block; 16-row waves then put two waves
// on every SIMD (one's RNG / activation / update overlaps the other's MFMAs) at the same 128 rows per CU: 10.9 ms
// vs 12.1 ms for 32-row waves at B 32768 x 500 steps, 6.6 vs 6.9 ms at d = hidden = 64. TSDE_MLP_VARIANT=32
// selects the 32x32x2 form (kept for the tests: both layouts must agree).
template <int D, int H, int ACT>
static hipError_t launch_mlp_dh(const MlpArgs& p, hipStream_t s) {
  static const int forced = [] {
    const char* e = getenv("TSDE_MLP_VARIANT");
    return e ? atoi(e) : 0;
  }();
  if (p.method == TSDE_TRAJ_MIDPOINT) {
    if (forced == 32) return launch_mlp_variant<D, H, ACT, 32, 4, 1>(p, s);
    return launch_mlp_variant<D, H, ACT, 16, 8, 1>(p, s);
  }
  if (p.method == TSDE_TRAJ_SRK) {
    // five state-sized arrays live across the drift evaluations. d = hidden = 128, 32768 x 500 steps: 8-wave blocks (two
    // waves per SIMD, 70 spilled dwords) 43.0 ms, 4-wave blocks (one wave per SIMD, no spills) 49.0 ms
    if constexpr (D >= 128) {
      static const int waves = [] {
        const char* e = getenv("TSDE_MLP_SRK_WAVES");
        return e ? atoi(e) : 0;
      }();
      if (waves == 4) return launch_mlp_variant<D, H, ACT, 16, 4, 2>(p, s);
      return launch_mlp_variant<D, H, ACT, 16, 8, 2>(p, s);
    } else {
      return launch_mlp_variant<D, H, ACT, 16, 8, 2>(p, s);
    }
  }
  if (forced == 32) return launch_mlp_variant<D, H, ACT, 32, 4, 0>(p, s);
  return launch_mlp_variant<D, H, ACT, 16, 8, 0>(p, s);
}

template <int D, int H>
static hipError_t launch_mlp_act(const MlpArgs& p, int act, hipStream_t s) {
  if (act == TSDE_ACT_TANH) return launch_mlp_dh<D, H, TSDE_ACT_TANH>(p, s);
  if (act == TSDE_ACT_SOFTPLUS) return launch_mlp_dh<D, H, TSDE_ACT_SOFTPLUS>(p, s);
  return hipErrorInvalidValue;
}

template <int D>
static hipError_t launch_mlp_h(const MlpArgs& p, int act, hipStream_t s) {
  if (p.h <= 32) return launch_mlp_act<D, 32>(p, act, s);
  if (p.h <= 64) return launch_mlp_act<D, 64>(p, act, s);
  if (p.h <= 128) return launch_mlp_act<D, 128>(p, act, s);
  if constexpr (D <= 64) {                 // both weight arrays must fit the LDS of a CU: d * hidden <= 16384
    if (p.h <= 256) return launch_mlp_act<D, 256>(p, act, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_trajectory_mlp_diag(void* ys, const void* y0, int64_t rows, int64_t d, int64_t h, const void* W1,
                                      const void* b1, const void* W2, const void* b2, const void* c, const void* e,
                                      int diff_kind, double diff_amp, int act, int method, const tsde_traj_t* tr,
                                      NoiseKey key, const uint64_t* key_dev, hipStream_t s) {
  MlpArgs p;
  p.ys = (float*)ys;
  p.y0 = (const float*)y0;
  p.W1 = (const float*)W1;
  p.b1 = (const float*)b1;
  p.W2 = (const float*)W2;
  p.b2 = (const float*)b2;
  p.c = (const float*)c;
  p.e = (const float*)e;
  p.rows = (const float*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const float*)tr->out_w;
  p.B = rows;
  p.d = (int32_t)d;
  p.h = (int32_t)h;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.method = method;
  p.diff_kind = diff_kind;
  p.diff_amp = (float)diff_amp;
  p.key = key;
  p.key_dev = key_dev;
  if (rows <= 0 || tr->n_steps <= 0) return hipSuccess;
  if (d <= 32) return launch_mlp_h<32>(p, act, s);
  if (d <= 64) return launch_mlp_h<64>(p, act, s);
  if (d <= 128) return launch_mlp_h<128>(p, act, s);
  return hipErrorInvalidValue;
}

}  // namespace tsde
