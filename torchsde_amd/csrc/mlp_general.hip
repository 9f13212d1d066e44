// Whole-trajectory kernel for neural SDEs whose drift AND diffusion are two-layer perceptrons of (t, y) shared by the
// batch, on gfx950 -- the reference's `Neural*` problems (tests/problems.py:135-252) and BASELINE configs[2]:
//
//     f(t, y) = W2f . act(W1f . y + w1tf * t + b1f) + b2f                                       (rows, d)
//     g(t, y) = scale * final(W2g . act(W1g . y + w1tg * t + b1g) + b2g)                         final: identity | sigmoid
//         general noise : (rows, d * m) read as (rows, d, m)  -> the step adds  sum_j g[., i, j] dW[., j]
//         diagonal noise: (rows, d)                            -> g[., i] dW[., i]
//         scalar noise  : (rows, d), one Brownian channel/row  -> g[., i] dW[.]
//         additive noise: no diffusion net -- g depends on t only and arrives as a TABLE of its (d, m) matrix at the
//                         scheme's stage times (the reference's NeuralAdditive, tests/problems.py:195-224, whose g is a
//                         network of t alone: the host evaluates it for all stage times in one batched call)
//
// `w1t` is the first layer's weight column of the TIME input, torch.cat([t.expand(B, 1), y], 1) in the reference's
// modules: per stage time it is one more bias, b1 + w1t * t, so t never becomes a matrix operand.
//
// Replaces, for such modules, the whole stepping loop torchsde/_core/base_solver.py:114-134 with
//   methods/euler.py:29-37 (f_and_g_prod -> misc.batch_mvp, _core/misc.py:62-63: bmm(g, dW))     TSDE_TRAJ_EULER
//   methods/midpoint.py:29-45 (two evaluations, the second at t + dt/2 and the predicted state)   TSDE_TRAJ_MIDPOINT
//   methods/srk.py:57-88 (SRID2, diagonal / scalar noise: 3 drift + 4 diffusion evaluations)       TSDE_TRAJ_SRK
// in ONE launch. Stepwise, this SDE costs a (rows, d, m) diffusion tensor through HBM and four library GEMMs per step
// (the user's two nets are 93 % of the solve at the configs[2] shape); here nothing but y0 and the outputs touches HBM.
//
// A wave owns 16 batch rows; state, hidden activations and results live in the accumulator layout of
// v_mfma_f32_16x16x4_f32 exactly as in mlp_trajectory.hip (lane (part, n): batch row n, channels 4 part + r of a tile),
// all weights live in LDS for the whole solve (hidden-major, rows padded by 4 floats: conflict-free ds_read_b32).
// The diffusion's second layer produces G^T tile by tile -- 16 consecutive outputs o = i * m + j of 16 batch rows -- and
// each tile is consumed on the spot:
//   * lane (part, n) holds G[n][o = 16 tile + 4 part + r], r = 0..3: four consecutive Brownian channels j of ONE state
//     channel i (m % 4 == 0), i.e. the four normals of ONE Philox quad of the (rows, m) increment field every other
//     kernel draws from -- generated in registers once per step and lane;
//   * the lane's partial sum s = sum_r G[r] dW[r] still has to be added over the lanes that hold the other j of the same
//     (n, i): that reduction is one more MFMA, D[i][n] += sum_k A[i][k] s_k[n] with a 0/1 selector as A operand, which
//     lands the sum in register r', lane (part', n) with i = 4 part' + r' -- the layout of the state. No shuffle, no
//     LDS round trip, no select: the contraction's result is born where the update needs it.
#include <type_traits>

#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_mlp.h"
#include "tsde_schemes.h"

namespace tsde {

struct NeuralNet {          // device view of tsde_mlp_t (pointers as given: input-major weights)
  const float *w1, *w1t, *b1, *w2, *b2;
  int32_t hidden, out, act, final;
  float scale;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct NeuralArgs {
  int32_t split;            // 1: the diffusion net's second layer on split-bf16 products (opt-in; see the kernel)
  float* ys;                // (n_out, B, d)
  const float* y0;          // (B, d)
  NeuralNet f, g;
  const float* rows;        // (n_steps, 8): dt, dt/2, 1/dt, sqrt(dt), sqrt(h), sqrt(h/12), h, t_k
  const uint32_t* cells;
  const int32_t* out_step;
  const float* out_w;
  int64_t B;
  int32_t d, m;
  int32_t n_steps, n_out;
  int32_t method;           // TSDE_TRAJ_EULER | TSDE_TRAJ_MIDPOINT | TSDE_TRAJ_SRK (diagonal / scalar noise)
  NoiseKey key;
  const uint64_t* key_dev;
  const float* gtab;        // additive noise: (m, d) or (n_steps, slots, m, d), the diffusion matrix transposed
  int64_t g_step_stride, g_slot_stride;     // floats; 0 for a matrix that does not depend on t
};

// MODE: 0 = diagonal noise, 1 = scalar noise, 2 = additive noise (table), else general noise with m = MODE Brownian
// channels (4, 8, 16, 32).
template <int MODE>
struct NoiseShape {
  static constexpr bool kGeneral = MODE >= 4;
  static constexpr bool kTable = MODE == 2;
  static constexpr int M = kGeneral ? MODE : 1;
  // tiles of G^T handled together (independent accumulator chains: a dependent f32 MFMA waits 40 cycles, issue is 32)
  static constexpr int G = (M >= 32) ? M / 16 : 2;
  static constexpr int kQuads = (M >= 16) ? M / 16 : 1;      // Philox quads of a row's increments one lane needs
};

// One normal of the field, out of line: the element-by-element paths (d % 4 != 0, m not a tile width, an unaligned field) are
// rare and must not cost the common path registers.
__device__ __noinline__ float draw_one(NoiseKey key, uint64_t elem, uint32_t cell, uint32_t stream) {
  return normal1<float>(key, elem, cell, 0, stream);
}

// the diffusion net's output function (uniform over the launch: callers branch once, outside their loops)
template <bool SIGMOID>
TSDE_D float finalise(float z) {
  if constexpr (SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
  return z;
}

// Schedule of a straight-line region of READS LDS operand reads, each feeding PER matrix instructions: the first few reads
// go out ahead, then every group of PER MFMAs is followed by one more read -- left alone, hipcc emits read -> wait -> PER
// MFMAs and the wave (there is ONE per SIMD at the configs[2] shape, nothing else to switch to) sits out the LDS latency
// once per read: 18.7 ms per 1000-step solve at 16384 x 32 x 16 before, see DESIGN.md for after.
template <int READS, int PER>
TSDE_D void reads_ahead() {
  constexpr int AHEAD = READS < 4 ? READS : 4;
  __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
  for (int i = 0; i < READS; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
    if (i < READS - AHEAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// LDS footprint in floats (rows padded by 4: the four lane quarters of a wave read rows 4 apart, see mlp_trajectory.hip)
template <int D, int H>
struct NeuralLds {
  static constexpr int S1 = H + 4, S2F = D + 4;
  static constexpr int out_padded(int out, int group) { return (out + 16 * group - 1) / (16 * group) * (16 * group); }
};

// General noise, exact f32: the diffusion net's second layer sits in LDS with the two tiles of a PAIR interleaved -- element
// (unit u, output o = 16 tile + c) at u * stride + 32 (tile / 2) + 2 c + (tile & 1) -- so that a lane's two A operands of a
// unit are ONE ds_read_b64, and with a row stride of 8 (mod 16) floats: a b64 read is served in two halves of 32 lanes, the two
// lane quarters of a half read units 4 apart, 4 * stride = 32 (mod 64) banks puts them on the two halves of the banks.
// For D <= 32 the stride is D * M + 8 whatever the real width: every row offset of the products is then an immediate of the
// read (the address arithmetic between the matrix instructions cost more than the reads: profiles/r6_microbench_mfma_fillers.txt).
template <int D, int MODE, bool SPLIT>
struct PairLayout {
  static constexpr bool kOn = MODE >= 4 && !SPLIT;
  static constexpr bool kFixed = kOn && D <= 32;
  static constexpr int kPad = kOn ? 8 : 4;
};

// SPLIT (opt-in, `options={"matrix_precision": "bf16x3"}`; general noise, H = 64): the diffusion net's SECOND layer -- 512 of
// the 640 f32 MFMAs of a configs[2] step -- runs on v_mfma_f32_16x16x32_bf16 with both operands split into a bf16 head and
// a bf16 tail, a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi (f32 accumulation; the dropped a_lo b_lo and the tails' own rounding
// are ~2^-16 relative per product): three instructions of ~17 cycles per 32 hidden units instead of eight of 32. NOT the
// reference's arithmetic: the default and every benchmarked configuration stay exact f32; the mode's error against float64 is
// reported beside its speed (tests/test_gpu_neural.py, bench_also.json). The weights are split once per launch into two bf16
// arrays in LDS, K-contiguous per output (a lane's eight k of a 32-block: the four channels 4 part + r of the block's two
// 16-unit tiles -- the order its own activations have in the accumulator layout), 16-byte chunks XOR-swizzled by the output
// row so that the ds_read_b128 of a 16-lane group hits 16 different bank quads.
// GENERIC (diagonal noise only): the state width is not a multiple of 4 or the field is unaligned, so the increments are
// drawn element by element; a separate instantiation, so that the common one carries no call and no second path (the SRK
// body keeps ~300 registers live across the draws).
template <int D, int H, int MODE, bool SPLIT = false, bool GENERIC = false>
__global__ void __launch_bounds__(256, (MODE >= 4 || (H <= 64 && !GENERIC)) ? 2 : 1) neural_trajectory_kernel(const NeuralArgs p, const int outp) {
  // (up to 64 hidden units the compiler is asked for at most 256 registers: it then keeps the accumulators in ordinary
  //  registers -- with the 512 of one wave per SIMD it parks them in the accumulation file and every tile pays eight copies
  //  out and back -- and the 64-channel shapes, whose LDS footprint admits two blocks per CU, get their second wave per SIMD
  //  (they asked for ~300 registers before: one wave per SIMD whatever the LDS allowed). Not the element-by-element
  //  instantiation, which would spill ~100 registers.)
  using NS = NoiseShape<MODE>;
  static_assert(!SPLIT || (NS::kGeneral && H == 64), "split mode: general noise, 64 hidden units");
  using L = NeuralLds<D, H>;
  using PL = PairLayout<D, MODE, SPLIT>;
  static_assert(!PL::kOn || NS::G == 2, "pairs of tiles");
  constexpr int TD = D / 16, TH = H / 16, S1 = L::S1, S2F = L::S2F, M = NS::M, G = NS::G;
  const int S2G = PL::kFixed ? D * M + PL::kPad : outp + PL::kPad;
  extern __shared__ float lds[];
  float* W1f = lds;                     // D rows of S1:  [input channel][hidden unit]
  float* W1g = W1f + D * S1;
  float* W2f = W1g + D * S1;            // H rows of S2F: [hidden unit][state channel]
  float* W2g = W2f + H * S2F;           // H rows of S2G: [hidden unit][output o]
  float* b1f = W2g + H * S2G;           // H each: b1f, wtf, b1g, wtg
  float* wtf = b1f + H;
  float* b1g = wtf + H;
  float* wtg = b1g + H;
  float* b2f = wtg + H;                 // D
  float* b2g = b2f + D;                 // outp
  const int dT = p.d, hf = p.f.hidden, hg = p.g.hidden, outT = p.g.out;
  // weights into LDS, zero-padded to the tile sizes: padded hidden units see zero weights both ways, padded state channels
  // and padded outputs are never read back (their selectors are 0 / their channels are skipped)
  for (int i = threadIdx.x; i < D * H; i += 256) {
    const int k = i / H, u = i % H;
    W1f[k * S1 + u] = (k < dT && u < hf) ? p.f.w1[k * hf + u] : 0.0f;
    W1g[k * S1 + u] = (k < dT && u < hg) ? p.g.w1[k * hg + u] : 0.0f;
    const int u2 = i / D, c = i % D;
    W2f[u2 * S2F + c] = (u2 < hf && c < dT) ? p.f.w2[u2 * dT + c] : 0.0f;
  }
  __bf16* W2hi = reinterpret_cast<__bf16*>(W2g);         // SPLIT: [outp][H] heads, then [outp][H] tails (same bytes as f32)
  __bf16* W2lo = W2hi + (size_t)outp * H;
  for (int i = threadIdx.x; i < H * outp; i += 256) {
    const int u = i / outp, o = i % outp;
    // general noise: the net's outputs are (i, j) row-major with m REAL Brownian channels; the tiles want i * M + j with M the
    // channel count padded to 4 / 8 / 16 / 32 (padded channels: zero weights, zero bias, zero increments)
    int src = o;
    bool have = o < outT;
    if constexpr (NS::kGeneral) {
      const int ci = o / M, cj = o % M;
      have = ci < dT && cj < p.m;
      src = ci * p.m + cj;
    }
    const float w = (u < hg && have) ? p.g.w2[(int64_t)u * outT + src] : 0.0f;
    if constexpr (SPLIT) {
      // unit u = 32 b + 16 tt + 4 part + r  ->  chunk 4 b + part (swizzled by the row), position 4 tt + r
      const int b = u >> 5, tt = (u >> 4) & 1, pq = (u >> 2) & 3, r = u & 3;
      const int chunk = (4 * b + pq) ^ ((o >> 1) & 7);
      const __bf16 hi = (__bf16)w;
      W2hi[(size_t)o * H + 8 * chunk + 4 * tt + r] = hi;
      W2lo[(size_t)o * H + 8 * chunk + 4 * tt + r] = (__bf16)(w - (float)hi);
    } else if constexpr (PL::kOn) {
      W2g[u * S2G + 32 * (o >> 5) + 2 * (o & 15) + ((o >> 4) & 1)] = w;
    } else {
      W2g[u * S2G + o] = w;
    }
  }
  for (int i = threadIdx.x; i < H; i += 256) {
    b1f[i] = i < hf ? p.f.b1[i] : 0.0f;
    wtf[i] = (i < hf && p.f.w1t) ? p.f.w1t[i] : 0.0f;
    b1g[i] = i < hg ? p.g.b1[i] : 0.0f;
    wtg[i] = (i < hg && p.g.w1t) ? p.g.w1t[i] : 0.0f;
  }
  for (int i = threadIdx.x; i < D; i += 256) b2f[i] = i < dT ? p.f.b2[i] : 0.0f;
  // (general noise with a closing sigmoid: the bias is staged as -log2(e) * b2, so that the epilogue's sigmoid argument is one
  //  fma of the accumulator, 1 / (1 + 2^(-log2(e) acc - log2(e) b2)))
  constexpr float kNegLog2e = -1.4426950408889634f;
  const bool prescaled = NS::kGeneral && p.g.final == TSDE_FINAL_SIGMOID;
  for (int i = threadIdx.x; i < outp; i += 256) {
    int src = i;
    bool have = i < outT;
    if constexpr (NS::kGeneral) {
      const int ci = i / M, cj = i % M;
      have = ci < dT && cj < p.m;
      src = ci * p.m + cj;
    }
    const float b = have ? p.g.b2[src] : 0.0f;
    b2g[i] = prescaled ? b * kNegLog2e : b;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane >> 4, n = lane & 15;
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }
  const bool midpoint = p.method == TSDE_TRAJ_MIDPOINT;
  const float g_scale = p.g.scale;
  const bool sigmoid_out = p.g.final == TSDE_FINAL_SIGMOID;
  const int64_t n_groups = (p.B + 15) / 16;
  // a block stages the weights once and its four waves walk over groups of 16 rows (grid = resident blocks)
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row0 = grp * 16;
    // in the last partial group the surplus lanes shadow the last row (same reads, same noise, same writes)
    const int64_t row = row0 + n < p.B ? row0 + n : p.B - 1;
    const uint32_t off_d = (uint32_t)(row * dT);
    auto real = [&](int ch) { return ch < dT; };
    // d a multiple of 4 (and the field aligned): rows are 16-byte groups and a lane's four channels one Philox quad; any other d:
    // element by element (the start, the outputs and the diagonal-noise draws only -- everything else lives in padded tiles)
    const bool row_quads = MODE == 0 ? !GENERIC : (dT & 3) == 0;      // (diagonal noise: decided with the instantiation)
    constexpr bool noise_quads = !GENERIC;      // (the launcher picks GENERIC unless d % 4 == 0 and elem0 % 4 == 0)

    f32x4 y[TD];
#pragma unroll
    for (int t = 0; t < TD; ++t) {
      const int ch = 16 * t + 4 * part;
      y[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (row_quads) {
        if (real(ch)) y[t] = *reinterpret_cast<const f32x4*>(p.y0 + off_d + ch);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (ch + r < dT) y[t][r] = p.y0[off_d + ch + r];
        }
      }
    }

    // hid^T = act(W1^T x^T + (b1 + wt * t)): loop order (t, r) outer, th inner -> consecutive MFMAs hit different accumulators
    auto hidden_layer = [&](const float* W1, const float* b1, const float* wt, int act, float time, const f32x4* x,
                            f32x4* hid) {
      int o1 = 0;
      asm volatile("" : "+v"(o1));        // (the two stages of the midpoint scheme must not share their operand reads)
#pragma unroll
      for (int th = 0; th < TH; ++th) hid[th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TD; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int th = 0; th < TH; ++th) {
            const float a = W1[o1 + (16 * t + 4 * part + r) * S1 + 16 * th + n];
            hid[th] = Tile<16>::mfma(a, x[t][r], hid[th]);
          }
        }
      }
      reads_ahead<TD * 4 * TH / 2, 2>();      // (operands of two neighbouring unit tiles arrive as one ds_read2_b32)
      // (the activation kind is uniform over the launch: one scalar branch around two straight-line copies)
      auto finish = [&](auto kind) {
        constexpr int ACT = decltype(kind)::value;
#pragma unroll
        for (int th = 0; th < TH; ++th) {
          const f32x4 bias = lds_quad(b1, 16 * th + 4 * part), slope = lds_quad(wt, 16 * th + 4 * part);
#pragma unroll
          for (int r = 0; r < 4; ++r) hid[th][r] = activate<ACT>(hid[th][r] + (bias[r] + slope[r] * time));
        }
      };
      if (act == TSDE_ACT_TANH) finish(std::integral_constant<int, TSDE_ACT_TANH>{});
      else finish(std::integral_constant<int, TSDE_ACT_SOFTPLUS>{});
    };

    // drift tiles f^T = W2f^T hid^T + b2f (TD independent chains)
    auto drift = [&](const f32x4* hid, f32x4* f) {
      int o2 = 0;
      asm volatile("" : "+v"(o2));
#pragma unroll
      for (int t = 0; t < TD; ++t) f[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int t = 0; t < TD; ++t) {
            const float a = W2f[o2 + (16 * th + 4 * part + r) * S2F + 16 * t + n];
            f[t] = Tile<16>::mfma(a, hid[th][r], f[t]);
          }
        }
      }
      if constexpr (TD == 1) reads_ahead<TH * 4, 1>();
      else reads_ahead<TH * 4 * TD / 2, 2>();
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const f32x4 bias = lds_quad(b2f, 16 * t + 4 * part);
#pragma unroll
        for (int r = 0; r < 4; ++r) f[t][r] += bias[r];
      }
    };

    // (g dW)^T in the state's layout, from the diffusion net's hidden activations
    // (the closing function is uniform over the launch: `closing` carries it as a type, one scalar branch per evaluation)
    auto diffusion_product_as = [&](const f32x4* hid, uint32_t cell, float sw, f32x4* gdw, auto closing) {
      constexpr bool sigmoid_out = decltype(closing)::value;
#pragma unroll
      for (int t = 0; t < TD; ++t) gdw[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (NS::kGeneral) {
        // this lane's increments: kQuads Philox quads of row `row`, scaled by sqrt(h) and the net's output scale
        const uint64_t quad_row = (key.elem0 + (uint64_t)row * (uint64_t)M) >> 2;
        float dw[NS::kQuads][4];
#pragma unroll
        for (int q = 0; q < NS::kQuads; ++q) {
          const int quad_of_row = M >= 16 ? 4 * q + part : (M == 8 ? (part & 1) : 0);
          uint64_t quad = quad_row + quad_of_row;
          asm volatile("" : "+v"(quad));
          float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if (p.m == M && (key.elem0 & 3) == 0) {
            normal4<float>(key, quad, cell, 0, kStreamW, z);
          } else {          // m not one of the tile widths (or an unaligned field): the row's REAL channels one by one
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int cj = 4 * quad_of_row + r;
              if (cj < p.m) z[r] = draw_one(key, key.elem0 + (uint64_t)row * (uint64_t)p.m + (uint64_t)cj, cell, kStreamW);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) dw[q][r] = (z[r] * sw) * g_scale;
        }
        // SPLIT: the activations of this evaluation as bf16 heads and tails, in B-operand order (see the kernel's comment)
        bf16x8 hid_hi[SPLIT ? H / 32 : 1], hid_lo[SPLIT ? H / 32 : 1];
        if constexpr (SPLIT) {
#pragma unroll
          for (int b = 0; b < H / 32; ++b) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float v = hid[2 * b + (j >> 2)][j & 3];
              const __bf16 hi = (__bf16)v;
              hid_hi[b][j] = hi;
              hid_lo[b][j] = (__bf16)(v - (float)hi);
            }
          }
        }
        // what one lane does with a finished tile: its four outputs are four Brownian channels of ONE state channel of its row
        auto close_tile = [&](const f32x4 acc, const f32x4 bias, int tl, int g, int tiles, float& s, float& sel) {
          int target, q;
          if constexpr (M >= 16) {
            target = tl / (M / 16);
            q = (M == 16) ? 0 : g;                                           // (G = M / 16 tiles per channel for M >= 32)
          } else if constexpr (M == 8) {
            target = 2 * tl + (part >> 1);
            q = 0;
          } else {
            target = 4 * tl + part;
            q = 0;
          }
          s = 0.0f;
          if constexpr (sigmoid_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __builtin_amdgcn_exp2f(fmaf(acc[r], kNegLog2e, bias[r]));      // exp(-(acc + b2))
              s = fmaf(__builtin_amdgcn_rcpf(1.0f + e), dw[q][r], s);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(acc[r] + bias[r], dw[q][r], s);
          }
          sel = (n == target && tl < tiles) ? 1.0f : 0.0f;
        };
#pragma unroll
        for (int ty = 0; ty < TD; ++ty) {
          const int channels = dT - 16 * ty < 16 ? dT - 16 * ty : 16;        // real state channels of this tile (wave-uniform)
          if (channels <= 0) continue;
          const int tiles = (channels * M + 15) / 16;                        // G^T tiles that feed them
          if constexpr (SPLIT) {
            for (int p0 = 0; p0 < tiles; p0 += G) {
              f32x4 acc[G];
#pragma unroll
              for (int g = 0; g < G; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
              // the tiles' output biases are requested BEFORE the matrix products (they used to be read, and waited for, in
              // the epilogue: two exposed LDS round trips per group of tiles)
              f32x4 bias[G];
#pragma unroll
              for (int g = 0; g < G; ++g) bias[g] = lds_quad(b2g, 16 * (ty * M + p0 + g) + 4 * part);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int b = 0; b < H / 32; ++b) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                  const int o = 16 * (ty * M + p0 + g) + n;
                  const int chunk = (4 * b + part) ^ ((o >> 1) & 7);
                  const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(W2hi + (size_t)o * H + 8 * chunk);
                  const bf16x8 alo = *reinterpret_cast<const bf16x8*>(W2lo + (size_t)o * H + 8 * chunk);
                  acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, hid_hi[b], acc[g], 0, 0, 0);
                  acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, hid_lo[b], acc[g], 0, 0, 0);
                  acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, hid_hi[b], acc[g], 0, 0, 0);
                }
              }
#pragma unroll
              for (int g = 0; g < G; ++g) {
                float s, sel;
                close_tile(acc[g], bias[g], p0 + g, g, tiles, s, sel);
                gdw[ty] = Tile<16>::mfma(sel, s, gdw[ty]);
              }
            }
          } else {
            // Exact f32, pair by pair, one stage of software pipelining (one wave per SIMD: nothing else hides a wait, and a
            // vector instruction between two matrix instructions costs three times its own issue --
            // profiles/r6_microbench_mfma_fillers.txt). Per pair: [the PREVIOUS pair's two selector products + 2 NR products,
            // the operand reads (ds_read_b64, immediate row offsets) running kAhead ahead] [the NEXT pair's first kAhead operands
            // and its output biases requested] [this pair's closing arithmetic, unbroken].
            constexpr int NR = TH * 4, kAhead = 4;
            // (the operands' LDS addresses as two opaque 32-bit bases -- units 0..31 and 32..63 of this lane's pair -- so that
            //  every row offset is an immediate of its read: left to itself hipcc folds the array's own offset into the
            //  immediates, overflows their 16 bits and repairs that with additions between the matrix instructions)
            typedef const __attribute__((address_space(3))) f32x2* lds_pair_t;
            typedef const __attribute__((address_space(3))) float* lds_float_t;
            uint32_t lo = (uint32_t)(uintptr_t)(lds_float_t)(W2g + (4 * part) * S2G + 32 * ((ty * M) >> 1) + 2 * n);
            uint32_t up = lo + (uint32_t)(32 * S2G * sizeof(float));
            asm volatile("" : "+v"(lo), "+v"(up));
            const float* bq = b2g + 16 * (ty * M) + 4 * part;
            auto operand = [&](int i) {
              const int th = i >> 2, r = i & 3;
              const uint32_t at = th < 2 ? lo + (uint32_t)((16 * th + r) * S2G * sizeof(float))
                                         : up + (uint32_t)((16 * (th - 2) + r) * S2G * sizeof(float));
              return *(lds_pair_t)(uintptr_t)at;
            };
            f32x2 ahead[kAhead];
            f32x4 bias_next[G];
#pragma unroll
            for (int i = 0; i < kAhead; ++i) ahead[i] = operand(i);
#pragma unroll
            for (int g = 0; g < G; ++g) bias_next[g] = *reinterpret_cast<const f32x4*>(bq + 16 * g);
            float s_prev[G], sel_prev[G];
#pragma unroll
            for (int g = 0; g < G; ++g) s_prev[g] = sel_prev[g] = 0.0f;
            for (int p0 = 0; p0 < tiles; p0 += G) {
              f32x4 acc[G], bias[G];
              f32x2 a[NR];
#pragma unroll
              for (int g = 0; g < G; ++g) {
                acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                bias[g] = bias_next[g];
              }
#pragma unroll
              for (int i = 0; i < kAhead; ++i) a[i] = ahead[i];
              __builtin_amdgcn_sched_barrier(0);
              gdw[ty] = Tile<16>::mfma(sel_prev[0], s_prev[0], gdw[ty]);
#pragma unroll
              for (int i = kAhead; i < NR; ++i) a[i] = operand(i);
#pragma unroll
              for (int i = 0; i < NR; ++i) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = Tile<16>::mfma(a[i][g], hid[i >> 2][i & 3], acc[g]);
              }
              gdw[ty] = Tile<16>::mfma(sel_prev[1], s_prev[1], gdw[ty]);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
              for (int i = kAhead; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, kAhead * G + 1, 0);
              __builtin_amdgcn_sched_barrier(0);
              lo += 32 * sizeof(float);
              up += 32 * sizeof(float);
              bq += 32;
              asm volatile("" : "+v"(lo), "+v"(up));
#pragma unroll
              for (int i = 0; i < kAhead; ++i) ahead[i] = operand(i);
#pragma unroll
              for (int g = 0; g < G; ++g) bias_next[g] = *reinterpret_cast<const f32x4*>(bq + 16 * g);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int g = 0; g < G; ++g) close_tile(acc[g], bias[g], p0 + g, g, tiles, s_prev[g], sel_prev[g]);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) gdw[ty] = Tile<16>::mfma(sel_prev[g], s_prev[g], gdw[ty]);
          }
        }
      } else {
        // diagonal / scalar noise: the net's output IS one diffusion value per state channel
        float w_row = 0.0f;
        if constexpr (MODE == 1) w_row = normal1<float>(key, key.elem0 + (uint64_t)row, cell, 0, kStreamW) * sw;
        int o2 = 0;
        asm volatile("" : "+v"(o2));
        f32x4 acc[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int th = 0; th < TH; ++th) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int t = 0; t < TD; ++t) {
              const float a = W2g[o2 + (16 * th + 4 * part + r) * S2G + 16 * t + n];
              acc[t] = Tile<16>::mfma(a, hid[th][r], acc[t]);
            }
          }
        }
        if constexpr (TD == 1) reads_ahead<TH * 4, 1>();
        else reads_ahead<TH * 4 * TD / 2, 2>();
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const int ch = 16 * t + 4 * part;
          const f32x4 bias = lds_quad(b2g, ch);
          float z[4] = {w_row, w_row, w_row, w_row};
          if constexpr (MODE == 0) {
            uint64_t quad = (key.elem0 + (uint64_t)off_d + (uint64_t)ch) >> 2;
            asm volatile("" : "+v"(quad));
            if constexpr (noise_quads) {
              if (real(ch)) normal4<float>(key, quad, cell, 0, kStreamW, z);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                z[r] = ch + r < dT ? draw_one(key, key.elem0 + (uint64_t)off_d + (uint64_t)(ch + r), cell, kStreamW) : 0.0f;
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] *= sw;
          }
          if (sigmoid_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gdw[t][r] = (g_scale * finalise<true>(acc[t][r] + bias[r])) * z[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) gdw[t][r] = (g_scale * (acc[t][r] + bias[r])) * z[r];
          }
        }
      }
    };

    auto diffusion_product = [&](const f32x4* hid, uint32_t cell, float sw, f32x4* gdw) {
      if (sigmoid_out) diffusion_product_as(hid, cell, sw, gdw, std::true_type{});
      else diffusion_product_as(hid, cell, sw, gdw, std::false_type{});
    };

    // diagonal / scalar noise: the diffusion VALUES g (one per state channel) of a net evaluation, and the step's increments
    // W (and U = h (W/2 + H), SRK) in the state's layout -- what the stochastic Runge-Kutta scheme combines stage by stage
    auto diffusion_values = [&](const f32x4* hid, f32x4* g) {
      int o2 = 0;
      asm volatile("" : "+v"(o2));
#pragma unroll
      for (int t = 0; t < TD; ++t) g[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int t = 0; t < TD; ++t) {
            const float a = W2g[o2 + (16 * th + 4 * part + r) * S2G + 16 * t + n];
            g[t] = Tile<16>::mfma(a, hid[th][r], g[t]);
          }
        }
      }
      if constexpr (TD == 1) reads_ahead<TH * 4, 1>();
      else reads_ahead<TH * 4 * TD / 2, 2>();
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const f32x4 bias = lds_quad(b2g, 16 * t + 4 * part);
        if (sigmoid_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r) g[t][r] = g_scale * finalise<true>(g[t][r] + bias[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) g[t][r] = g_scale * (g[t][r] + bias[r]);
        }
      }
    };
    auto increments = [&](uint32_t cell, float sw, float sh, float th_, f32x4* w, f32x4* u) {
      if constexpr (MODE == 1) {
        const uint64_t e = key.elem0 + (uint64_t)row;
        const float wr = normal1<float>(key, e, cell, 0, kStreamW) * sw;
        const float ur = th_ * (0.5f * wr + normal1<float>(key, e, cell, 0, kStreamH) * sh);
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          w[t] = f32x4{wr, wr, wr, wr};
          u[t] = f32x4{ur, ur, ur, ur};
        }
      } else {
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const int ch = 16 * t + 4 * part;
          uint64_t quad = (key.elem0 + (uint64_t)off_d + (uint64_t)ch) >> 2;
          asm volatile("" : "+v"(quad));
          float zw[4] = {0.0f, 0.0f, 0.0f, 0.0f}, zh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if constexpr (noise_quads) {
            if (real(ch)) {
              normal4<float>(key, quad, cell, 0, kStreamW, zw);
              normal4<float>(key, quad, cell, 0, kStreamH, zh);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (ch + r < dT) {
                const uint64_t e = key.elem0 + (uint64_t)off_d + (uint64_t)(ch + r);
                zw[r] = draw_one(key, e, cell, kStreamW);
                zh[r] = draw_one(key, e, cell, kStreamH);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            w[t][r] = zw[r] * sw;
            u[t][r] = th_ * (0.5f * w[t][r] + zh[r] * sh);
          }
        }
      }
    };

    // additive noise: (G w)^T in the state's layout, D[i][n] = sum_j G[i][j] w[n][j] on the matrix cores -- lane (part, n)
    // supplies A[i = n][k = part] = G[16 t + n][4 kk + part] (read from the table: a few KB, cache-resident) and
    // B[k = part][n] = the weight of ITS batch row for Brownian channel j = 4 kk + part, drawn by the lane itself
    // (element row * m + j of the (rows, m) field; every channel count and alignment). m <= 16: four k-blocks.
    auto table_increments = [&](uint32_t cell, float sw, float sh, float th_, bool need_u, float* wj, float* uj) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        wj[kk] = 0.0f;
        uj[kk] = 0.0f;
        const int j = 4 * kk + part;
        if (4 * kk < p.m && j < p.m) {
          const uint64_t e = key.elem0 + (uint64_t)row * (uint64_t)p.m + (uint64_t)j;
          wj[kk] = normal1<float>(key, e, cell, 0, kStreamW) * sw;
          if (need_u) uj[kk] = th_ * (0.5f * wj[kk] + normal1<float>(key, e, cell, 0, kStreamH) * sh);
        }
      }
    };
    auto table_product = [&](const float* Gt, const float* wsel, f32x4* gdw) {
#pragma unroll
      for (int t = 0; t < TD; ++t) gdw[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (4 * kk < p.m) {
          const int j = 4 * kk + part;
#pragma unroll
          for (int t = 0; t < TD; ++t) {
            const int ch = 16 * t + n;
            const float a = (j < p.m && ch < dT) ? Gt[j * dT + ch] : 0.0f;
            gdw[t] = Tile<16>::mfma(a, wsel[kk], gdw[t]);
          }
        }
      }
    };

    int jout = 0;
    for (int k = 0; k < p.n_steps; ++k) {
      const float* srow = p.rows + (int64_t)k * 8;
      const float dt = srow[0], half_dt = srow[1], sw = srow[4], t0 = srow[7];
      const uint32_t cell = p.cells[k];
      f32x4 hid[TH], f[TD], gdw[TD], yn[TD];
      bool stepped = false;
      if constexpr (NS::kTable) {
        // additive noise (base_sde.py:101-102): Euler (euler.py:29-37), midpoint (midpoint.py:29-45), SRK = SRA1
        // (srk.py:90-111, tableaus/sra1.py: C0 = (0, 3/4), C1 = (1, 0)) in the stepwise route's operation order
        const float rdt = srow[2], sh = srow[5], th_ = srow[6];
        const float* gk = p.gtab + (int64_t)k * p.g_step_stride;
        float wj[4], uj[4], ws[4];
        table_increments(cell, sw, sh, th_, p.method == TSDE_TRAJ_SRK, wj, uj);
        hidden_layer(W1f, b1f, wtf, p.f.act, t0, y, hid);
        drift(hid, f);
        if (p.method == TSDE_TRAJ_SRK) {
          f32x4 h[TD], acc[TD];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) ws[kk] = (1.5f * uj[kk]) * rdt;
          table_product(gk, ws, gdw);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h[t][r] = (y[t][r] + (0.75f * f[t][r]) * dt) + gdw[t][r];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) ws[kk] = (1.0f * wj[kk]) + (-1.0f * uj[kk]) * rdt;
          table_product(gk, ws, gdw);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = (y[t][r] + ((float)(1.0 / 3) * f[t][r]) * dt) + gdw[t][r];
          }
          hidden_layer(W1f, b1f, wtf, p.f.act, t0 + 0.75f * dt, h, hid);
          drift(hid, f);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) ws[kk] = (0.0f * wj[kk]) + (1.0f * uj[kk]) * rdt;
          table_product(gk + p.g_slot_stride, ws, gdw);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yn[t][r] = (acc[t][r] + ((float)(2.0 / 3) * f[t][r]) * dt) + gdw[t][r];
          }
        } else {
          table_product(gk, wj, gdw);
          if (midpoint) {
            f32x4 yp[TD];
#pragma unroll
            for (int t = 0; t < TD; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) yp[t][r] = (y[t][r] + f[t][r] * half_dt) + 0.5f * gdw[t][r];
            }
            hidden_layer(W1f, b1f, wtf, p.f.act, t0 + half_dt, yp, hid);
            drift(hid, f);
            table_product(gk + p.g_slot_stride, wj, gdw);
          }
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yn[t][r] = (y[t][r] + f[t][r] * dt) + gdw[t][r];
          }
        }
        stepped = true;
      }
      if constexpr (MODE <= 1) {
        if (p.method == TSDE_TRAJ_SRK) {
          // SRID2 (srk.py:57-88, tableaus/srid2.py) for diagonal / scalar noise with BOTH functions networks: three drift
          // evaluations f(t, y), f(t + dt, H0_1), f(t + dt/2, H0_2) (the tableau's alpha_3 = 0) and four diffusion
          // evaluations g(t, y), g(t + dt/4, H1_1), g(t + dt, H1_2), g(t + dt/4, H1_3), every one a pass of its net on the
          // matrix cores; the running sums hold everything a later stage needs (operation order of tsde_schemes.h).
          const float rdt = srow[2], sqrt_dt = srow[3], sh = srow[5], th_ = srow[6];
          f32x4 w[TD], u[TD], g0[TD], g1[TD], g2[TD], f0[TD], f1[TD], acc[TD], hs[TD];
          increments(cell, sw, sh, th_, w, u);
          auto weight = [&](int st, float wv, float uv) {
            const float Ikk = (wv * wv - dt) * 0.5f;
            const float Ikkk = ((wv * wv) * wv - (3.0f * dt) * wv) * (float)(1.0 / 6);
            return ((((float)Srid2::beta1(st) * wv) + ((float)Srid2::beta2(st) * Ikk) / sqrt_dt) +
                    ((float)Srid2::beta3(st) * uv) * rdt) + ((float)Srid2::beta4(st) * Ikkk) * rdt;
          };
          hidden_layer(W1f, b1f, wtf, p.f.act, t0, y, hid);
          drift(hid, f0);
          hidden_layer(W1g, b1g, wtg, p.g.act, t0, y, hid);
          diffusion_values(hid, g0);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[t][r] = (y[t][r] + ((float)Srid2::alpha(0) * f0[t][r]) * dt) + g0[t][r] * weight(0, w[t][r], u[t][r]);
              hs[t][r] = (y[t][r] + ((float)Srid2::A1(1, 0) * f0[t][r]) * dt) + ((float)Srid2::B1(1, 0) * g0[t][r]) * sqrt_dt;
            }
          }
          hidden_layer(W1g, b1g, wtg, p.g.act, t0 + 0.25f * dt, hs, hid);          // g1 = g(t + dt/4, H1_1)
          diffusion_values(hid, g1);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[t][r] = acc[t][r] + g1[t][r] * weight(1, w[t][r], u[t][r]);
              hs[t][r] = (y[t][r] + ((float)Srid2::A1(2, 0) * f0[t][r]) * dt) + ((float)Srid2::B1(2, 0) * g0[t][r]) * sqrt_dt;
            }
          }
          hidden_layer(W1g, b1g, wtg, p.g.act, t0 + dt, hs, hid);                  // g2 = g(t + dt, H1_2)
          diffusion_values(hid, g2);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[t][r] = acc[t][r] + g2[t][r] * weight(2, w[t][r], u[t][r]);
              hs[t][r] = y[t][r] + ((float)Srid2::A0(1, 0) * f0[t][r]) * dt;      // H0_1
            }
          }
          hidden_layer(W1f, b1f, wtf, p.f.act, t0 + dt, hs, hid);                  // f1 = f(t + dt, H0_1)
          drift(hid, f1);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[t][r] = acc[t][r] + ((float)Srid2::alpha(1) * f1[t][r]) * dt;
              hs[t][r] = (((y[t][r] + ((float)Srid2::A0(2, 0) * f0[t][r]) * dt) +
                           (((float)Srid2::B0(2, 0) * g0[t][r]) * u[t][r]) * rdt) +
                          ((float)Srid2::A0(2, 1) * f1[t][r]) * dt) + (((float)Srid2::B0(2, 1) * g1[t][r]) * u[t][r]) * rdt;
            }
          }
          hidden_layer(W1f, b1f, wtf, p.f.act, t0 + 0.5f * dt, hs, hid);           // f2 = f(t + dt/2, H0_2)
          drift(hid, f);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[t][r] = acc[t][r] + ((float)Srid2::alpha(2) * f[t][r]) * dt;
              hs[t][r] = (((y[t][r] + ((float)Srid2::B1(3, 0) * g0[t][r]) * sqrt_dt) +
                           ((float)Srid2::B1(3, 1) * g1[t][r]) * sqrt_dt) + ((float)Srid2::A1(3, 2) * f[t][r]) * dt) +
                         ((float)Srid2::B1(3, 2) * g2[t][r]) * sqrt_dt;             // H1_3
            }
          }
          hidden_layer(W1g, b1g, wtg, p.g.act, t0 + 0.25f * dt, hs, hid);          // g3 = g(t + dt/4, H1_3)
          diffusion_values(hid, g0);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yn[t][r] = acc[t][r] + g0[t][r] * weight(3, w[t][r], u[t][r]);
          }
          stepped = true;
        }
      }
      if constexpr (!NS::kTable) {
      if (!stepped) {
      hidden_layer(W1f, b1f, wtf, p.f.act, t0, y, hid);
      drift(hid, f);
      hidden_layer(W1g, b1g, wtg, p.g.act, t0, y, hid);
      diffusion_product(hid, cell, sw, gdw);
      if (midpoint) {
        // y' = (y + f dt/2) + (g dW)/2, then everything again at (t + dt/2, y')          (midpoint.py:31-43)
        f32x4 yp[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) yp[t][r] = (y[t][r] + f[t][r] * half_dt) + 0.5f * gdw[t][r];
        }
        const float tm = t0 + half_dt;
        hidden_layer(W1f, b1f, wtf, p.f.act, tm, yp, hid);
        drift(hid, f);
        hidden_layer(W1g, b1g, wtg, p.g.act, tm, yp, hid);
        diffusion_product(hid, cell, sw, gdw);
      }
#pragma unroll
      for (int t = 0; t < TD; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) yn[t][r] = (y[t][r] + f[t][r] * dt) + gdw[t][r];          // euler.py:36
      }
      }
      }
      // outputs due at the end of this step: w0 y_k + w1 y_{k+1} (base_solver.py:147, interp.py:15-18)
      while (jout < p.n_out && p.out_step[jout] == k + 1) {
        const float w0 = p.out_w[2 * jout], w1 = p.out_w[2 * jout + 1];
        const bool exact = w0 == 0.0f && w1 == 1.0f;
        float* dst = p.ys + (int64_t)jout * p.B * dT;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const int ch = 16 * t + 4 * part;
          Pack<float, 4> o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o.v[r] = exact ? yn[t][r] : (w0 * y[t][r] + w1 * yn[t][r]);
          if (row_quads) {
            if (real(ch)) store<float, 4>(dst, off_d + ch, o);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (ch + r < dT) dst[off_d + ch + r] = o.v[r];
            }
          }
        }
        ++jout;
      }
#pragma unroll
      for (int t = 0; t < TD; ++t) y[t] = yn[t];
    }
  }
}

size_t neural_lds_bytes(int D, int H, int outp, int pad) {
  // (+ 32: the general-noise pipeline requests the output biases of the pair after the last one)
  return ((size_t)2 * D * (H + 4) + (size_t)H * (D + 4) + (size_t)H * (outp + pad) + 4 * H + D + outp + 32) * sizeof(float);
}

static size_t neural_lds_limit() {
  static const size_t limit = [] {
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&bytes, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && bytes > 0)
      return (size_t)bytes;
    return (size_t)(64 * 1024);
  }();
  return limit;
}

template <int D, int H, int MODE, bool SPLIT = false, bool GENERIC = false>
static hipError_t launch_neural_mode(const NeuralArgs& p, hipStream_t s) {
  if constexpr (!SPLIT && NoiseShape<MODE>::kGeneral && H == 64) {
    if (p.split) return launch_neural_mode<D, H, MODE, true>(p, s);
  }
  using PL = PairLayout<D, MODE, SPLIT>;
  const int outp = PL::kFixed ? D * MODE
                   : NoiseShape<MODE>::kGeneral ? NeuralLds<D, H>::out_padded(p.d * MODE, NoiseShape<MODE>::G)
                                                : NeuralLds<D, H>::out_padded(p.g.out, 1);
  const size_t lds_bytes = neural_lds_bytes(D, H, outp, PL::kPad);
  if (lds_bytes > neural_lds_limit()) return hipErrorInvalidValue;
  static bool configured = false;   // per instantiation
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&neural_trajectory_kernel<D, H, MODE, SPLIT, GENERIC>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int64_t groups = (p.B + 15) / 16;
  int64_t blocks = (groups + 3) / 4;
  // resident blocks: the LDS of a CU holds floor(160 KiB / footprint) of them
  const int64_t per_cu = (int64_t)((160 * 1024) / lds_bytes) < 1 ? 1 : (int64_t)((160 * 1024) / lds_bytes);
  const int64_t resident = 256 * (per_cu > 8 ? 8 : per_cu);
  if (blocks > resident) blocks = resident;
  hipLaunchKernelGGL((neural_trajectory_kernel<D, H, MODE, SPLIT, GENERIC>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, p,
                     outp);
  return hipGetLastError();
}

template <int D, int H>
static hipError_t launch_neural_dh(const NeuralArgs& p, int noise, hipStream_t s) {
  if (noise == TSDE_NOISE_DIAGONAL) {
    const bool quads = (p.d % 4 == 0) && (p.key.elem0 % 4 == 0);
    return quads ? launch_neural_mode<D, H, 0, false, false>(p, s) : launch_neural_mode<D, H, 0, false, true>(p, s);
  }
  if (noise == TSDE_NOISE_SCALAR) return launch_neural_mode<D, H, 1>(p, s);
  if (noise == TSDE_NOISE_ADDITIVE) return launch_neural_mode<D, H, 2>(p, s);
  if constexpr (H <= 64) {      // (general noise: H x d*m weights of the diffusion's second layer; 128 units do not fit the LDS)
    // (m real Brownian channels run in the next tile width up; the padding is zero weights and zero increments)
    if (p.m < 1 || p.m > 32) return hipErrorInvalidValue;
    if (p.m <= 4) return launch_neural_mode<D, H, 4>(p, s);
    if (p.m <= 8) return launch_neural_mode<D, H, 8>(p, s);
    if (p.m <= 16) return launch_neural_mode<D, H, 16>(p, s);
    return launch_neural_mode<D, H, 32>(p, s);
  }
  return hipErrorInvalidValue;
}

template <int D>
static hipError_t launch_neural_d(const NeuralArgs& p, int noise, hipStream_t s) {
  const int h = p.f.hidden > p.g.hidden ? p.f.hidden : p.g.hidden;
  if (h <= 32) return launch_neural_dh<D, 32>(p, noise, s);
  if (h <= 64) return launch_neural_dh<D, 64>(p, noise, s);
  if (h <= 128 && noise != TSDE_NOISE_GENERAL) return launch_neural_dh<D, 128>(p, noise, s);
  return hipErrorInvalidValue;
}

static NeuralNet device_view(const tsde_mlp_t* n) {
  NeuralNet v;
  v.w1 = (const float*)n->w1;
  v.w1t = (const float*)n->w1t;
  v.b1 = (const float*)n->b1;
  v.w2 = (const float*)n->w2;
  v.b2 = (const float*)n->b2;
  v.hidden = n->hidden;
  v.out = n->out;
  v.act = n->activation;
  v.final = n->final;
  v.scale = (float)n->scale;
  return v;
}

// The LDS footprint of a shape in bytes, or 0 if no instantiation covers it (the C ABI's tsde_trajectory_mlp_general_lds:
// the host asks before it routes a module here).
size_t neural_footprint(int64_t d, int64_t m, int64_t hf, int64_t hg, int64_t out, int noise) {
  const int D = d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : 0;
  const int64_t h = hf > hg ? hf : hg;
  const int H = h <= 32 ? 32 : h <= 64 ? 64 : (h <= 128 && noise != TSDE_NOISE_GENERAL) ? 128 : 0;
  if (D == 0 || H == 0) return 0;
  int group = 1, pad = 4;
  if (noise == TSDE_NOISE_GENERAL) {
    if (m < 1 || m > 32) return 0;
    const int64_t M = m <= 4 ? 4 : m <= 8 ? 8 : m <= 16 ? 16 : 32;
    group = M >= 32 ? (int)M / 16 : 2;
    out = (D <= 32 ? D : d) * M;          // (PairLayout: a fixed stride up to 32 state channels; the split mode is no larger)
    pad = 8;
  }
  const int outp = (int)((out + 16 * group - 1) / (16 * group) * (16 * group));
  return neural_lds_bytes(D, H, outp, pad);
}

hipError_t launch_trajectory_mlp_general(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                                         const tsde_mlp_t* drift, const tsde_mlp_t* diffusion, int method,
                                         const tsde_traj_t* tr, NoiseKey key, const uint64_t* key_dev, hipStream_t s) {
  NeuralArgs p;
  p.split = diffusion->precision == TSDE_PRECISION_BF16X3 ? 1 : 0;
  p.ys = (float*)ys;
  p.y0 = (const float*)y0;
  p.f = device_view(drift);
  p.g = device_view(diffusion);
  p.rows = (const float*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const float*)tr->out_w;
  p.B = rows;
  p.d = (int32_t)d;
  p.m = (int32_t)m;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.method = method;
  p.key = key;
  p.key_dev = key_dev;
  p.gtab = nullptr;
  p.g_step_stride = p.g_slot_stride = 0;
  if (rows <= 0 || tr->n_steps <= 0) return hipSuccess;
  if (d <= 16) return launch_neural_d<16>(p, noise, s);
  if (d <= 32) return launch_neural_d<32>(p, noise, s);
  if (d <= 64) return launch_neural_d<64>(p, noise, s);
  return hipErrorInvalidValue;
}

// Additive noise: the drift a perceptron of (t, y), the diffusion a table (see the kernel's MODE 2).
hipError_t launch_trajectory_mlp_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const tsde_mlp_t* drift,
                                          const void* gtab, int time_dependent, int method, const tsde_traj_t* tr,
                                          NoiseKey key, const uint64_t* key_dev, hipStream_t s) {
  NeuralArgs p;
  p.split = 0;
  p.ys = (float*)ys;
  p.y0 = (const float*)y0;
  p.f = device_view(drift);
  p.g = NeuralNet{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, TSDE_ACT_TANH, TSDE_FINAL_NONE, 1.0f};
  p.rows = (const float*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const float*)tr->out_w;
  p.B = rows;
  p.d = (int32_t)d;
  p.m = (int32_t)m;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.method = method;
  p.key = key;
  p.key_dev = key_dev;
  const int slots = method == TSDE_TRAJ_EULER ? 1 : 2;
  p.gtab = (const float*)gtab;
  p.g_slot_stride = time_dependent ? m * d : 0;
  p.g_step_stride = time_dependent ? (int64_t)slots * m * d : 0;
  if (rows <= 0 || tr->n_steps <= 0) return hipSuccess;
  if (d <= 16) return launch_neural_d<16>(p, TSDE_NOISE_ADDITIVE, s);
  if (d <= 32) return launch_neural_d<32>(p, TSDE_NOISE_ADDITIVE, s);
  if (d <= 64) return launch_neural_d<64>(p, TSDE_NOISE_ADDITIVE, s);
  return hipErrorInvalidValue;
}

}  // namespace tsde
