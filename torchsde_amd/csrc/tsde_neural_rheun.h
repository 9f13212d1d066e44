// The reversible Heun pair -- the reference's recommended method for adjoint training (DOCUMENTATION.md:97,118) -- for neural
// SDEs whose drift AND diffusion are perceptrons of (t, y) of any depth up to four Linear layers, in ONE launch each way:
//
//   forward   methods/reversible_heun.py:48-73   (ReversibleHeun.step driven by base_solver.py:114-134)
//   backward  methods/reversible_heun.py:76-144  (AdjointReversibleHeun.step driven by adjoint.py:64-127): the forward state is
//             reconstructed algebraically, NOTHING of the trajectory is stored -- that is the point of the method.
//
//     f(t, y) = scale_f * final_f(W2f . a(... a(W1f . y + w1tf * t + b1f) ...) + b2f)                   (rows, d)
//     g(t, y) = scale_g * final_g(W2g . a(... a(W1g . y + w1tg * t + b1g) ...) + b2g)
//         general noise : (rows, d * m) read as (rows, d, m);  diagonal: (rows, d);  scalar: (rows, d), one channel per row
//     a: tanh | softplus | act_scale * silu (the LipSwish of examples/sde_gan.py:44-47);  final: none | sigmoid | tanh
//     (examples/sde_gan.py:50-66, 93-94: MLP(1 + hidden, hidden [* noise], mlp_size, num_layers, tanh=True) for both)
//
// The scheme carries (y, z, f, g). With g a (rows, d, m) tensor per row that does not fit registers, so the kernels carry the
// PRODUCTS instead: the evaluation at z_j contracts G_j with BOTH increments it will ever meet -- dW_{j-1} (it closes step j-1:
// y_j = y_{j-1} + (f_{j-1} + f_j) dt/2 + (G_{j-1} + G_j) dW_{j-1} / 2) and dW_j (it opens step j: z_{j+1} = 2 y_j - z_j +
// f_j dt + G_j dW_j) -- which the counter generator can draw in any order. One pass of both nets per step, K + 1 for K steps.
//
// Backward, per evaluation j = K ... 0 (state and adjoints (y, z, a_y, a_z, a_f, a_g) of reversible_heun.py:98-144): a_g is
// never materialised either -- after a step it is the rank-one (a_y / 2 + a_z') (x) dW_j, and entering the next step it gains
// (a_y / 2) (x) dW_{j-1} (:113-117, :137), so the cotangent of G_j is p (x) dW_j + q (x) dW_{j-1} with two state-sized vectors;
// the SAME pass of the nets that yields f_j, G_j dW_j, G_j dW_{j-1} back-propagates that cotangent (and a_f's) to z_j on the
// matrix cores (transposed weight reads from the same LDS copy). Parameter gradients: the pass stashes, per evaluation and row,
// the layer inputs and pre-activation cotangents; the host forms the weight gradients as tall matrix products over the stash
// (kernels.py: rheun_mlp_weight_gradients), chunk by chunk.
//
// Layout as in mlp_general.hip: a wave owns 16 batch rows; state, activations and cotangents live in the accumulator layout
// of v_mfma_f32_16x16x4_f32 (lane (part, n): row n, channels 4 part + r of a 16-channel tile); all weights in LDS, input-major,
// rows padded by 4 floats. The contraction with the increments and its reduction over lanes are MFMAs with a 0/1 selector.
#pragma once
#include <type_traits>

#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_mlp.h"

namespace tsde {

constexpr int kMaxMid = 2;      // hidden-to-hidden layers beyond the first: up to four Linear layers per net

struct DeepNet {                // device view of tsde_deep_mlp_t
  const float *w1, *w1t, *b1, *wm[kMaxMid], *bm[kMaxMid], *w2, *b2;
  int32_t hidden, out, act, final, n_mid;
  float scale, act_scale;
};

struct RheunStash {             // (evaluations of this launch, rows, stride) float arrays; all null in the forward kernel
  float *z, *cf, *p, *q, *wa, *wb;
  float *hf[kMaxMid + 1], *df[kMaxMid + 1], *hg[kMaxMid + 1], *dg[kMaxMid + 1];
  int32_t sd, sm, shf, shg;     // row strides: state, Brownian channels, hidden units of the drift / diffusion net (multiples of 4)
};

struct RheunArgs {
  float* ys;                    // forward: (n_out, B, d)
  float* z_out;                 // forward: (B, d), z after the last step
  const float* y0;              // forward: (B, d)
  float *s_y, *s_z, *s_ay, *s_az, *s_af, *s_p;      // backward: the carried state, (B, d) each, read and written
  const float* ys_all;          // backward: (n_out + 1, B, d): y0, then the forward outputs
  const float* gys;             // backward: (n_out + 1, B, d): their cotangents
  DeepNet f, g;
  const float* rows;            // (n_steps, 8): dt, dt/2, 1/dt, sqrt(dt), sqrt(h), sqrt(h/12), h, t_k
  const float* times;           // (n_steps + 1): t_0 ... t_K, the times the nets are evaluated at
  const uint32_t* cells;
  const int32_t* out_step;
  const float* out_w;
  int64_t B;
  int32_t d, m, n_steps, n_out;
  int32_t j_hi, j_lo;           // backward: this launch runs the evaluations j_hi, j_hi - 1, ..., j_lo
  int32_t method;               // forward: TSDE_TRAJ_REVERSIBLE_HEUN, or one of the stateless schemes for the same nets
  NoiseKey key;
  const uint64_t* key_dev;
  RheunStash st;
};

__device__ __noinline__ float rheun_draw_one(NoiseKey key, uint64_t elem, uint32_t cell) {
  return normal1<float>(key, elem, cell, 0, kStreamW);
}

// Schedule of a straight-line region of READS LDS operand reads, each feeding PER matrix instructions: the first few reads go out
// ahead, then every group of PER MFMAs is followed by one more read (cf. mlp_general.hip: left alone, hipcc emits read -> wait ->
// MFMAs, and with one wave per SIMD nothing else hides the LDS latency).
template <int READS, int PER>
TSDE_D void rheun_reads_ahead() {
  constexpr int AHEAD = READS < 4 ? READS : 4;
  __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
  for (int i = 0; i < READS; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
    if (i < READS - AHEAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// value and slope of a hidden activation / of a net's closing function
template <int ACT>
TSDE_D void hidden_act(float x, float c, float& value, float& slope) {
  if constexpr (ACT == TSDE_ACT_SILU) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    value = c * (x * s);
    slope = c * (s * (1.0f + x * (1.0f - s)));
  } else {
    activate_with_slope<ACT>(x, value, slope);
    value *= c;
    slope *= c;
  }
}

template <int FINAL>
TSDE_D void final_act(float x, float& value, float& slope) {
  if constexpr (FINAL == TSDE_FINAL_SIGMOID) {
    value = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    slope = value * (1.0f - value);
  } else if constexpr (FINAL == TSDE_FINAL_TANH) {
    const float e2x = __builtin_amdgcn_exp2f(x * (2.0f * 1.4426950408889634f));
    value = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e2x + 1.0f);
    slope = 1.0f - value * value;
  } else {
    value = x;
    slope = 1.0f;
  }
}

// LDS floats of a shape: both first layers, the hidden-to-hidden layers, both last layers, every bias
// (the diffusion's last layer: rows padded by 8 and 32 floats of slack behind the output biases -- what the forward kernel's
//  paired layout and its one-pair-ahead requests need, see `kPaired`; the backward kernel pads by 4 and leaves the rest unused)
inline size_t rheun_lds_floats(int D, int H, int outp, int nmf, int nmg) {
  const size_t S1 = H + 4;
  return (size_t)2 * D * S1 + (size_t)(nmf + nmg) * H * S1 + (size_t)H * (D + 4) + (size_t)H * (outp + 8) + (size_t)4 * H +
         (size_t)(nmf + nmg) * H + D + outp + 32;
}

// MODE: 0 = diagonal noise, 1 = scalar noise, else general noise with the Brownian channels padded to MODE (4 or 16).
template <int D, int H, int MODE, bool BACKWARD>
__global__ void __launch_bounds__(256, (!BACKWARD && D <= 32) ? 2 : 1) neural_rheun_kernel(const RheunArgs p, const int outp) {
  // (forward, up to 32 state channels: at most 256 registers asked of the compiler, which then keeps the accumulators in
  //  ordinary registers instead of copying every tile out of the accumulation file -- cf. mlp_general.hip)
  constexpr bool kGeneral = MODE >= 4;
  static_assert(MODE == 0 || MODE == 1 || MODE == 4 || MODE == 16, "diagonal, scalar, or general noise in tiles of 4 / 16 channels");
  constexpr int M = kGeneral ? MODE : 1;
  constexpr int kQuads = M >= 16 ? M / 16 : 1;
  constexpr int TD = D / 16, TH = H / 16, S1 = H + 4, S2F = D + 4;
  // general noise, forward: the two tiles of a PAIR interleaved in the diffusion's last layer -- element (unit u, output
  // o = 16 tile + c) at u * S2G + 32 (tile / 2) + 2 c + (tile & 1), rows padded by 8 -- one ds_read_b64 per unit and pair,
  // conflict-free (mlp_general.hip, PairLayout); the backward kernel reads the same weights transposed, four consecutive
  // outputs at a time, and keeps them in output order. Up to 32 state channels the stride is a compile-time constant
  // (the launcher passes outp = D * M), so every row offset is an immediate of its read.
  constexpr bool kPaired = kGeneral && !BACKWARD;
  constexpr int kPad = kPaired ? 8 : 4;
  const int S2G = (kGeneral && D <= 32) ? D * M + kPad : outp + kPad;
  const int nmf = p.f.n_mid, nmg = p.g.n_mid;
  extern __shared__ float lds[];
  float* cur = lds;
  auto take = [&](int n) {
    float* r = cur;
    cur += n;
    return r;
  };
  float* W1f = take(D * S1);            // [input channel][hidden unit]
  float* W1g = take(D * S1);
  float *Wmf[kMaxMid], *Wmg[kMaxMid];   // [hidden unit in][hidden unit out]
#pragma unroll
  for (int l = 0; l < kMaxMid; ++l) Wmf[l] = l < nmf ? take(H * S1) : nullptr;
#pragma unroll
  for (int l = 0; l < kMaxMid; ++l) Wmg[l] = l < nmg ? take(H * S1) : nullptr;
  float* W2f = take(H * S2F);           // [hidden unit][state channel]
  float* W2g = take(H * S2G);           // [hidden unit][output o]
  float* b1f = take(H);
  float* wtf = take(H);
  float* b1g = take(H);
  float* wtg = take(H);
  float *bmf[kMaxMid], *bmg[kMaxMid];
#pragma unroll
  for (int l = 0; l < kMaxMid; ++l) bmf[l] = l < nmf ? take(H) : nullptr;
#pragma unroll
  for (int l = 0; l < kMaxMid; ++l) bmg[l] = l < nmg ? take(H) : nullptr;
  float* b2f = take(D);
  float* b2g = take(outp);

  const int dT = p.d, hf = p.f.hidden, hg = p.g.hidden, outT = p.g.out;
  // weights into LDS, zero-padded to the tile sizes (padded units see zero weights both ways; padded outputs are never used)
  for (int i = threadIdx.x; i < D * H; i += 256) {
    const int k = i / H, u = i % H;
    W1f[k * S1 + u] = (k < dT && u < hf) ? p.f.w1[k * hf + u] : 0.0f;
    W1g[k * S1 + u] = (k < dT && u < hg) ? p.g.w1[k * hg + u] : 0.0f;
    const int u2 = i / D, c = i % D;
    W2f[u2 * S2F + c] = (u2 < hf && c < dT) ? p.f.w2[u2 * dT + c] : 0.0f;
  }
#pragma unroll
  for (int l = 0; l < kMaxMid; ++l) {
    for (int i = threadIdx.x; i < H * H; i += 256) {
      const int a = i / H, b = i % H;
      if (l < nmf) Wmf[l][a * S1 + b] = (a < hf && b < hf) ? p.f.wm[l][a * hf + b] : 0.0f;
      if (l < nmg) Wmg[l][a * S1 + b] = (a < hg && b < hg) ? p.g.wm[l][a * hg + b] : 0.0f;
    }
    for (int i = threadIdx.x; i < H; i += 256) {
      if (l < nmf) bmf[l][i] = i < hf ? p.f.bm[l][i] : 0.0f;
      if (l < nmg) bmg[l][i] = i < hg ? p.g.bm[l][i] : 0.0f;
    }
  }
  for (int i = threadIdx.x; i < H * outp; i += 256) {
    const int u = i / outp, o = i % outp;
    // general noise: the net's outputs are (i, j) row-major with m REAL channels; the tiles want i * M + j
    int src = o;
    bool have = o < outT;
    if constexpr (kGeneral) {
      const int ci = o / M, cj = o % M;
      have = ci < dT && cj < p.m;
      src = ci * p.m + cj;
    }
    const int at = kPaired ? 32 * (o >> 5) + 2 * (o & 15) + ((o >> 4) & 1) : o;
    W2g[u * S2G + at] = (u < hg && have) ? p.g.w2[(int64_t)u * outT + src] : 0.0f;
  }
  for (int i = threadIdx.x; i < H; i += 256) {
    b1f[i] = i < hf ? p.f.b1[i] : 0.0f;
    wtf[i] = (i < hf && p.f.w1t) ? p.f.w1t[i] : 0.0f;
    b1g[i] = i < hg ? p.g.b1[i] : 0.0f;
    wtg[i] = (i < hg && p.g.w1t) ? p.g.w1t[i] : 0.0f;
  }
  for (int i = threadIdx.x; i < D; i += 256) b2f[i] = i < dT ? p.f.b2[i] : 0.0f;
  for (int i = threadIdx.x; i < outp; i += 256) {
    int src = i;
    bool have = i < outT;
    if constexpr (kGeneral) {
      const int ci = i / M, cj = i % M;
      have = ci < dT && cj < p.m;
      src = ci * p.m + cj;
    }
    b2g[i] = have ? p.g.b2[src] : 0.0f;
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
  const int K = p.n_steps;
  const float f_scale = p.f.scale, g_scale = p.g.scale;
  const int64_t n_groups = (p.B + 15) / 16;
  const bool row_quads = (dT & 3) == 0;
  const bool noise_quads = kGeneral ? (p.m == M && (key.elem0 & 3) == 0) : (row_quads && (key.elem0 & 3) == 0);

  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row0 = grp * 16;
    const int64_t row = row0 + n < p.B ? row0 + n : p.B - 1;     // surplus lanes of the last group shadow the last row
    const uint32_t off_d = (uint32_t)(row * dT);

    auto load_state = [&](const float* src, f32x4* v) {
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const int ch = 16 * t + 4 * part;
        v[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (row_quads) {
          if (ch < dT) v[t] = *reinterpret_cast<const f32x4*>(src + off_d + ch);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (ch + r < dT) v[t][r] = src[off_d + ch + r];
          }
        }
      }
    };
    auto store_state = [&](float* dst, const f32x4* v) {
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const int ch = 16 * t + 4 * part;
        if (row_quads) {
          if (ch < dT) *reinterpret_cast<f32x4*>(dst + off_d + ch) = v[t];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (ch + r < dT) dst[off_d + ch + r] = v[t][r];
          }
        }
      }
    };
    // one 16-byte group per tile of a stash row (strides are multiples of 4; columns past the stride do not exist)
    auto stash = [&](float* base, int stride, int e, const f32x4* v, int tiles) {
      if (base == nullptr) return;
      float* dst = base + ((int64_t)e * p.B + row) * stride;
#pragma unroll
      for (int t = 0; t < tiles; ++t) {
        const int ch = 16 * t + 4 * part;
        if (ch < stride) *reinterpret_cast<f32x4*>(dst + ch) = v[t];
      }
    };

    // ---- the increments of one cell, as this lane meets them --------------------------------------------------------------
    // general: kQuads quads of row `row` (channels 4 quad + r), scaled by sqrt(h) and the net's output scale;
    // diagonal: the quad of the lane's four state channels per tile; scalar: the row's one increment in every slot
    constexpr int NW = kGeneral ? kQuads : TD;
    auto draw = [&](uint32_t cell, float sw, f32x4* dw) {
      if constexpr (kGeneral) {
        const uint64_t quad_row = (key.elem0 + (uint64_t)row * (uint64_t)M) >> 2;
#pragma unroll
        for (int q = 0; q < kQuads; ++q) {
          const int quad_of_row = M >= 16 ? 4 * q + part : 0;
          float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if (noise_quads) {
            normal4<float>(key, quad_row + quad_of_row, cell, 0, kStreamW, z);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int cj = 4 * quad_of_row + r;
              if (cj < p.m) z[r] = rheun_draw_one(key, key.elem0 + (uint64_t)row * (uint64_t)p.m + (uint64_t)cj, cell);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) dw[q][r] = (z[r] * sw) * g_scale;
        }
      } else if constexpr (MODE == 1) {
        const float w = (normal1<float>(key, key.elem0 + (uint64_t)row, cell, 0, kStreamW) * sw) * g_scale;
#pragma unroll
        for (int t = 0; t < TD; ++t) dw[t] = f32x4{w, w, w, w};
      } else {
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const int ch = 16 * t + 4 * part;
          float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if (noise_quads) {
            if (ch < dT) normal4<float>(key, (key.elem0 + (uint64_t)off_d + (uint64_t)ch) >> 2, cell, 0, kStreamW, z);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (ch + r < dT) z[r] = rheun_draw_one(key, key.elem0 + (uint64_t)off_d + (uint64_t)(ch + r), cell);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) dw[t][r] = (z[r] * sw) * g_scale;
        }
      }
    };
    auto zero_w = [&](f32x4* dw) {
#pragma unroll
      for (int i = 0; i < NW; ++i) dw[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };

    // ---- matrix products ----------------------------------------------------------------------------------------------------
    // out^T (TOUT tiles) = W^T x^T for W [in][out] with row stride `stride`
    auto product = [&](const float* W, int stride, const f32x4* x, int tin, f32x4* out, int tout) {
#pragma unroll
      for (int to = 0; to < tout; ++to) out[to] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ti = 0; ti < tin; ++ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int to = 0; to < tout; ++to) {
            const float a = W[(16 * ti + 4 * part + r) * stride + 16 * to + n];
            out[to] = Tile<16>::mfma(a, x[ti][r], out[to]);
          }
        }
      }
    };
    // cin^T (TIN tiles) += W cout^T: the transposed product from the same copy (one 16-byte read per four MFMAs)
    auto product_t = [&](const float* W, int stride, const f32x4* cout, int tout, f32x4* cin, int tin) {
#pragma unroll
      for (int to = 0; to < tout; ++to) {
#pragma unroll
        for (int ti = 0; ti < tin; ++ti) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(W + (16 * ti + n) * stride + 16 * to + 4 * part);
#pragma unroll
          for (int r = 0; r < 4; ++r) cin[ti] = Tile<16>::mfma(a[r], cout[to][r], cin[ti]);
        }
      }
    };
    // hid = act(hid + bias (+ slope_t * time)), slope kept for the way back
    auto activate_tiles = [&](int act, float c, const float* bias, const float* wt, float time, f32x4* hid, f32x4* slope) {
      auto run = [&](auto kind) {
        constexpr int ACT = decltype(kind)::value;
#pragma unroll
        for (int th = 0; th < TH; ++th) {
          const f32x4 b = lds_quad(bias, 16 * th + 4 * part);
          f32x4 w = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (wt != nullptr) w = lds_quad(wt, 16 * th + 4 * part);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v, s;
            hidden_act<ACT>(hid[th][r] + (b[r] + w[r] * time), c, v, s);
            hid[th][r] = v;
            slope[th][r] = s;
          }
        }
      };
      if (act == TSDE_ACT_TANH) run(std::integral_constant<int, TSDE_ACT_TANH>{});
      else if (act == TSDE_ACT_SOFTPLUS) run(std::integral_constant<int, TSDE_ACT_SOFTPLUS>{});
      else run(std::integral_constant<int, TSDE_ACT_SILU>{});
    };
    auto final_value = [&](int kind, float x, float& v, float& s) {
      if (kind == TSDE_FINAL_SIGMOID) final_act<TSDE_FINAL_SIGMOID>(x, v, s);
      else if (kind == TSDE_FINAL_TANH) final_act<TSDE_FINAL_TANH>(x, v, s);
      else final_act<TSDE_FINAL_NONE>(x, v, s);
    };

    // ---- one evaluation of both nets at (time, z) ---------------------------------------------------------------------------
    //   f            drift value
    //   sa, sb       (G dW_a)^T, (G dW_b)^T in the state's layout (dwa, dwb: `draw` of the step this evaluation opens / closes)
    //   BACKWARD:    cot_f (a_f'), pv, qv: cotangent of G is pv (x) dwa + qv (x) dwb;  vjp^T = d/dz of <cot_f, f> + <cot_G, G>;
    //                the stash rows of evaluation e of this launch
    auto evaluate = [&](float time, const f32x4* z, const f32x4* dwa, const f32x4* dwb, f32x4* f, f32x4* sa, f32x4* sb,
                        const f32x4* cot_f, const f32x4* pv, const f32x4* qv, f32x4* vjp, int e) {
      f32x4 hid[kMaxMid + 1][TH], slope[kMaxMid + 1][TH];
      f32x4 delta[TH], back[TH];
      // -- drift net (`top`: the activations of the last hidden layer so far; `hid[l]`: every layer's, for the way back)
      f32x4 top[TH], nxt[TH];
      product(W1f, S1, z, TD, top, TH);
      activate_tiles(p.f.act, p.f.act_scale, b1f, wtf, time, top, slope[0]);
#pragma unroll
      for (int th = 0; th < TH; ++th) hid[0][th] = top[th];
#pragma unroll
      for (int l = 0; l < kMaxMid; ++l) {
        if (l < nmf) {
          product(Wmf[l], S1, top, TH, nxt, TH);
          activate_tiles(p.f.act, p.f.act_scale, bmf[l], nullptr, 0.0f, nxt, slope[l + 1]);
#pragma unroll
          for (int th = 0; th < TH; ++th) hid[l + 1][th] = top[th] = nxt[th];
        }
      }
      f32x4 cpre[TD];
      {
        f32x4 pre[TD];
        product(W2f, S2F, top, TH, pre, TD);
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const f32x4 b = lds_quad(b2f, 16 * t + 4 * part);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v, s;
            final_value(p.f.final, pre[t][r] + b[r], v, s);
            f[t][r] = f_scale * v;
            if constexpr (BACKWARD) cpre[t][r] = (cot_f[t][r] * f_scale) * s;
          }
        }
      }
      if constexpr (BACKWARD) {
        stash(p.st.z, p.st.sd, e, z, TD);
        stash(p.st.cf, p.st.sd, e, cpre, TD);
#pragma unroll
        for (int t = 0; t < TD; ++t) vjp[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // back through the drift net: delta_l = (W_{l+1} delta_{l+1}) * slope_l
#pragma unroll
        for (int th = 0; th < TH; ++th) back[th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        product_t(W2f, S2F, cpre, TD, back, TH);
#pragma unroll
        for (int l = kMaxMid; l >= 0; --l) {
          if (l <= nmf) {
            stash(p.st.hf[l], p.st.shf, e, hid[l], TH);
#pragma unroll
            for (int th = 0; th < TH; ++th) {
#pragma unroll
              for (int r = 0; r < 4; ++r) delta[th][r] = back[th][r] * slope[l][th][r];
            }
            stash(p.st.df[l], p.st.shf, e, delta, TH);
#pragma unroll
            for (int th = 0; th < TH; ++th) back[th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (l > 0) product_t(Wmf[l - 1], S1, delta, TH, back, TH);
            else product_t(W1f, S1, delta, TH, vjp, TD);
          }
        }
      }
      // -- diffusion net
      product(W1g, S1, z, TD, top, TH);
      activate_tiles(p.g.act, p.g.act_scale, b1g, wtg, time, top, slope[0]);
#pragma unroll
      for (int th = 0; th < TH; ++th) hid[0][th] = top[th];
#pragma unroll
      for (int l = 0; l < kMaxMid; ++l) {
        if (l < nmg) {
          product(Wmg[l], S1, top, TH, nxt, TH);
          activate_tiles(p.g.act, p.g.act_scale, bmg[l], nullptr, 0.0f, nxt, slope[l + 1]);
#pragma unroll
          for (int th = 0; th < TH; ++th) hid[l + 1][th] = top[th] = nxt[th];
        }
      }
#pragma unroll
      for (int t = 0; t < TD; ++t) sa[t] = sb[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int th = 0; th < TH; ++th) back[th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (kGeneral) {
        if constexpr (BACKWARD) {
          stash(p.st.p, p.st.sd, e, pv, TD);
          stash(p.st.q, p.st.sd, e, qv, TD);
          stash(p.st.wa, p.st.sm, e, dwa, kQuads);          // (tile q of the stash row = channels 16 q + 4 part + r)
          stash(p.st.wb, p.st.sm, e, dwb, kQuads);
        }
        if constexpr (!BACKWARD) {
          // Forward: pair by pair with one stage of software pipelining, as in mlp_general.hip -- per pair [the PREVIOUS pair's
          // four selector products and 2 NR products, the operand reads kAhead ahead] [the NEXT pair's first operands and output
          // biases requested] [this pair's closing arithmetic, unbroken]; the closing function is a type here (one scalar
          // branch per evaluation instead of one per output).
          auto pairs = [&](auto kind) {
            constexpr int FINAL = decltype(kind)::value;
            constexpr int NR = TH * 4, kAhead = 4;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(3))) f32x2* lds_pair_t;
            typedef const __attribute__((address_space(3))) float* lds_float_t;
#pragma unroll
            for (int ty = 0; ty < TD; ++ty) {
              const int channels = dT - 16 * ty < 16 ? dT - 16 * ty : 16;
              if (channels <= 0) continue;
              const int tiles = (channels * M + 15) / 16;
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
              f32x4 bias_next[2];
#pragma unroll
              for (int i = 0; i < kAhead; ++i) ahead[i] = operand(i);
#pragma unroll
              for (int g = 0; g < 2; ++g) bias_next[g] = *reinterpret_cast<const f32x4*>(bq + 16 * g);
              float a_prev[2], b_prev[2], sel_prev[2];
#pragma unroll
              for (int g = 0; g < 2; ++g) a_prev[g] = b_prev[g] = sel_prev[g] = 0.0f;
              for (int tl = 0; tl < tiles; tl += 2) {
                f32x4 acc[2], bias[2];
                f32x2 a[NR];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                  bias[g] = bias_next[g];
                }
#pragma unroll
                for (int i = 0; i < kAhead; ++i) a[i] = ahead[i];
                __builtin_amdgcn_sched_barrier(0);
                sa[ty] = Tile<16>::mfma(sel_prev[0], a_prev[0], sa[ty]);
                sb[ty] = Tile<16>::mfma(sel_prev[0], b_prev[0], sb[ty]);
#pragma unroll
                for (int i = kAhead; i < NR; ++i) a[i] = operand(i);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
#pragma unroll
                  for (int g = 0; g < 2; ++g) acc[g] = Tile<16>::mfma(a[i][g], top[i >> 2][i & 3], acc[g]);
                }
                sa[ty] = Tile<16>::mfma(sel_prev[1], a_prev[1], sa[ty]);
                sb[ty] = Tile<16>::mfma(sel_prev[1], b_prev[1], sb[ty]);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int i = kAhead; i < NR; ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, kAhead * 2 + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                lo += 32 * sizeof(float);
                up += 32 * sizeof(float);
                bq += 32;
                asm volatile("" : "+v"(lo), "+v"(up));
#pragma unroll
                for (int i = 0; i < kAhead; ++i) ahead[i] = operand(i);
#pragma unroll
                for (int g = 0; g < 2; ++g) bias_next[g] = *reinterpret_cast<const f32x4*>(bq + 16 * g);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  const bool live = g == 0 || tl + 1 < tiles;
                  const int target = M >= 16 ? tl + g : 4 * (tl + g) + part;
                  float s_a = 0.0f, s_b = 0.0f;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    float v, s;
                    final_act<FINAL>(acc[g][r] + bias[g][r], v, s);
                    s_a = fmaf(v, dwa[0][r], s_a);
                    s_b = fmaf(v, dwb[0][r], s_b);
                  }
                  a_prev[g] = s_a;
                  b_prev[g] = s_b;
                  sel_prev[g] = (live && n == target) ? 1.0f : 0.0f;
                }
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                sa[ty] = Tile<16>::mfma(sel_prev[g], a_prev[g], sa[ty]);
                sb[ty] = Tile<16>::mfma(sel_prev[g], b_prev[g], sb[ty]);
              }
            }
          };
          if (p.g.final == TSDE_FINAL_SIGMOID) pairs(std::integral_constant<int, TSDE_FINAL_SIGMOID>{});
          else if (p.g.final == TSDE_FINAL_TANH) pairs(std::integral_constant<int, TSDE_FINAL_TANH>{});
          else pairs(std::integral_constant<int, TSDE_FINAL_NONE>{});
        }
        if constexpr (BACKWARD) {
          // Backward: two tiles per turn -- two independent accumulator chains (a dependent f32 MFMA waits 40 cycles, issue is
          // 32); a turn's second tile may lie past the real outputs -- `outp` is a multiple of 32, its weights and bias are zero,
          // and it is masked out of the selector and the cotangent. The weights' LDS addresses are four opaque 32-bit bases
          // (forward reads / transposed reads, units 0..31 / 32..63) advanced by one pair per turn, so that with a
          // compile-time stride every row offset is an immediate of its read; the closing function is a type.
          auto pairs_back = [&](auto kind) {
            constexpr int FINAL = decltype(kind)::value;
            typedef const __attribute__((address_space(3))) float* lds_float_t;
            typedef const __attribute__((address_space(3))) f32x4* lds_quad_t;
#pragma unroll
            for (int ty = 0; ty < TD; ++ty) {
              const int channels = dT - 16 * ty < 16 ? dT - 16 * ty : 16;
              if (channels <= 0) continue;
              const int tiles = (channels * M + 15) / 16;
              uint32_t f_lo = (uint32_t)(uintptr_t)(lds_float_t)(W2g + (4 * part) * S2G + 16 * (ty * M) + n);
              uint32_t t_lo = (uint32_t)(uintptr_t)(lds_float_t)(W2g + n * S2G + 16 * (ty * M) + 4 * part);
              uint32_t f_up = f_lo + (uint32_t)(32 * S2G * sizeof(float)), t_up = t_lo + (uint32_t)(32 * S2G * sizeof(float));
              asm volatile("" : "+v"(f_lo), "+v"(f_up), "+v"(t_lo), "+v"(t_up));
              const float* bq = b2g + 16 * (ty * M) + 4 * part;
              for (int tl = 0; tl < tiles; tl += 2) {
                const bool two = tl + 1 < tiles;
                f32x4 bias[2], acc[2];
                // p, q of this lane's four outputs, asked for BEFORE the products (their LDS round trip -- a ds_bpermute each --
                // used to sit exposed in the closing arithmetic)
                f32x4 pl[2], ql[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  // the state channel (within tile ty) this lane's four outputs belong to (their Brownian channels: quad 0 of
                  // the lane's draws -- 4 part + r for 16 channels per state channel, r for 4)
                  const int target = M >= 16 ? tl + g : 4 * (tl + g) + part;
                  pl[g] = ql[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                  const f32x4 pt = pv[ty], qt = qv[ty];
                  if constexpr (M >= 16) {
                    // one channel per tile: p, q of (row n, channel target) sit in lane (target / 4, n), register target % 4
                    const int src = ((((target >> 2) & 3) << 4) + n) << 2;
                    const int reg = target & 3;
                    const float psel = reg == 0 ? pt[0] : reg == 1 ? pt[1] : reg == 2 ? pt[2] : pt[3];
                    const float qsel = reg == 0 ? qt[0] : reg == 1 ? qt[1] : reg == 2 ? qt[2] : qt[3];
                    const float pb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, psel)));
                    const float qb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, qsel)));
                    pl[g] = f32x4{pb, pb, pb, pb};
                    ql[g] = f32x4{qb, qb, qb, qb};
                  } else {
                    // four channels per tile, one per lane quarter: the asking lanes want DIFFERENT registers of the holder, and a
                    // chain of ds_bpermutes with a per-lane pick afterwards is folded by hipcc into ONE bpermute of a value picked
                    // in the SOURCE lane (seen in the ISA; wrong gradients). The matrix cores do the broadcast instead:
                    // D[o][row] = sum_c Sel[o][c] p[c][row], Sel[o][c] = 1 iff output o of the tile belongs to channel c -- and the
                    // result is born in the layout of the tile's outputs.
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                      const float a = (part == ((tl + g) & 3) && r == (n >> 2) && tl + g < 4) ? 1.0f : 0.0f;   // A[o = n][c = 4 part + r]
                      pl[g] = Tile<16>::mfma(a, pt[r], pl[g]);
                      ql[g] = Tile<16>::mfma(a, qt[r], ql[g]);
                    }
                  }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  bias[g] = *reinterpret_cast<const f32x4*>(bq + 16 * g);
                  acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int th = 0; th < TH; ++th) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const uint32_t at = (th < 2 ? f_lo : f_up) + (uint32_t)((16 * (th & 1) + r) * S2G * sizeof(float));
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                      const float a = *(lds_float_t)(uintptr_t)(at + 64 * g);
                      acc[g] = Tile<16>::mfma(a, top[th][r], acc[g]);
                    }
                  }
                }
                {   // (eight operand reads ahead, then one more per matrix instruction)
                  constexpr int kReads = TH * 8, kAhead = 8;
                  __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
                  for (int i = 0; i < kReads; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < kReads - kAhead) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                  }
                  __builtin_amdgcn_sched_barrier(0);
                }
                f32x4 cot[2];
                float s_a[2], s_b[2], sel[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  const bool live = g == 0 || two;
                  const int target = M >= 16 ? tl + g : 4 * (tl + g) + part;
                  s_a[g] = s_b[g] = 0.0f;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    float v, s;
                    final_act<FINAL>(acc[g][r] + bias[g][r], v, s);
                    s_a[g] = fmaf(v, dwa[0][r], s_a[g]);
                    s_b[g] = fmaf(v, dwb[0][r], s_b[g]);
                    cot[g][r] = live ? (pl[g][r] * dwa[0][r] + ql[g][r] * dwb[0][r]) * s : 0.0f;
                  }
                  sel[g] = (live && n == target) ? 1.0f : 0.0f;
                }
                __builtin_amdgcn_sched_barrier(0);
                // (the four selector products open the block of transposed products: no vector instruction between them)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  sa[ty] = Tile<16>::mfma(sel[g], s_a[g], sa[ty]);
                  sb[ty] = Tile<16>::mfma(sel[g], s_b[g], sb[ty]);
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
#pragma unroll
                  for (int th = 0; th < TH; ++th) {      // (th inner: consecutive MFMAs hit different accumulators)
                    const uint32_t at = (th < 2 ? t_lo : t_up) + (uint32_t)((16 * (th & 1)) * S2G * sizeof(float)) + 64 * g;
                    const f32x4 a = *(lds_quad_t)(uintptr_t)at;
#pragma unroll
                    for (int r = 0; r < 4; ++r) back[th] = Tile<16>::mfma(a[r], cot[g][r], back[th]);
                  }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                rheun_reads_ahead<TH * 2, 4>();          // (one 16-byte read feeds four MFMAs)
                f_lo += 128;
                f_up += 128;
                t_lo += 128;
                t_up += 128;
                bq += 32;
                asm volatile("" : "+v"(f_lo), "+v"(f_up), "+v"(t_lo), "+v"(t_up));
              }
            }
          };
          if (p.g.final == TSDE_FINAL_SIGMOID) pairs_back(std::integral_constant<int, TSDE_FINAL_SIGMOID>{});
          else if (p.g.final == TSDE_FINAL_TANH) pairs_back(std::integral_constant<int, TSDE_FINAL_TANH>{});
          else pairs_back(std::integral_constant<int, TSDE_FINAL_NONE>{});
        }
      } else {
        // diagonal / scalar noise: one diffusion value per state channel
        f32x4 pre[TD], cg[TD];
        product(W2g, S2G, top, TH, pre, TD);
#pragma unroll
        for (int t = 0; t < TD; ++t) {
          const f32x4 b = lds_quad(b2g, 16 * t + 4 * part);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v, s;
            final_value(p.g.final, pre[t][r] + b[r], v, s);
            sa[t][r] = v * dwa[t][r];
            sb[t][r] = v * dwb[t][r];
            if constexpr (BACKWARD) cg[t][r] = (pv[t][r] * dwa[t][r] + qv[t][r] * dwb[t][r]) * s;
          }
        }
        if constexpr (BACKWARD) {
          stash(p.st.p, p.st.sd, e, cg, TD);               // (diagonal / scalar: the slot holds the cotangent before `final`)
          product_t(W2g, S2G, cg, TD, back, TH);
        }
      }
      if constexpr (BACKWARD) {
#pragma unroll
        for (int l = kMaxMid; l >= 0; --l) {
          if (l <= nmg) {
            stash(p.st.hg[l], p.st.shg, e, hid[l], TH);
#pragma unroll
            for (int th = 0; th < TH; ++th) {
#pragma unroll
              for (int r = 0; r < 4; ++r) delta[th][r] = back[th][r] * slope[l][th][r];
            }
            stash(p.st.dg[l], p.st.shg, e, delta, TH);
#pragma unroll
            for (int th = 0; th < TH; ++th) back[th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (l > 0) product_t(Wmg[l - 1], S1, delta, TH, back, TH);
            else product_t(W1g, S1, delta, TH, vjp, TD);
          }
        }
      }
    };

    f32x4 dwa[NW], dwb[NW], f[TD], sa[TD], sb[TD];
    if constexpr (!BACKWARD) {
      f32x4 y[TD], z[TD], yp[TD], yprev[TD];
      load_state(p.y0, y);
#pragma unroll
      for (int t = 0; t < TD; ++t) z[t] = yp[t] = yprev[t] = y[t];
      zero_w(dwa);
      int jout = 0;
      if (p.method != TSDE_TRAJ_REVERSIBLE_HEUN) {
        // ---- the schemes that carry no state, for the same (deep) nets: Euler (euler.py:29-37), midpoint (midpoint.py:29-45),
        //      Heun (heun.py:35-48), Euler-Heun (euler_heun.py:29-42) in the stepwise route's operation order ------------------
        f32x4 f2[TD], sa2[TD], x2[TD], y1[TD];
        zero_w(dwb);
        for (int k = 0; k < K; ++k) {
          const float dt = p.rows[(int64_t)k * 8], hdt = p.rows[(int64_t)k * 8 + 1];
          draw(p.cells[k], p.rows[(int64_t)k * 8 + 4], dwa);
          evaluate(p.times[k], y, dwa, dwb, f, sa, sb, nullptr, nullptr, nullptr, nullptr, 0);
          if (p.method == TSDE_TRAJ_EULER) {
#pragma unroll
            for (int t = 0; t < TD; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) y1[t][r] = (y[t][r] + f[t][r] * dt) + sa[t][r];
            }
          } else {
            float t2 = p.times[k + 1];
#pragma unroll
            for (int t = 0; t < TD; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                x2[t][r] = p.method == TSDE_TRAJ_MIDPOINT ? (y[t][r] + f[t][r] * hdt) + 0.5f * sa[t][r]
                           : p.method == TSDE_TRAJ_HEUN   ? (y[t][r] + dt * f[t][r]) + sa[t][r]
                                                          : y[t][r] + sa[t][r];
              }
            }
            if (p.method == TSDE_TRAJ_MIDPOINT) t2 = p.times[k] + 0.5f * dt;
            evaluate(t2, x2, dwa, dwb, f2, sa2, sb, nullptr, nullptr, nullptr, nullptr, 0);
#pragma unroll
            for (int t = 0; t < TD; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                y1[t][r] = p.method == TSDE_TRAJ_MIDPOINT ? (y[t][r] + f2[t][r] * dt) + sa2[t][r]
                           : p.method == TSDE_TRAJ_HEUN   ? y[t][r] + (((dt * (f[t][r] + f2[t][r])) + sa[t][r]) + sa2[t][r]) * 0.5f
                                                          : (y[t][r] + dt * f[t][r]) + ((sa[t][r] + sa2[t][r]) * 0.5f);
              }
            }
          }
          while (jout < p.n_out && p.out_step[jout] == k + 1) {
            const float w0 = p.out_w[2 * jout], w1 = p.out_w[2 * jout + 1];
            const bool exact = w0 == 0.0f && w1 == 1.0f;
            f32x4 o[TD];
#pragma unroll
            for (int t = 0; t < TD; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) o[t][r] = exact ? y1[t][r] : (w0 * y[t][r] + w1 * y1[t][r]);
            }
            store_state(p.ys + (int64_t)jout * p.B * dT, o);
            ++jout;
          }
#pragma unroll
          for (int t = 0; t < TD; ++t) y[t] = y1[t];
        }
        if (p.z_out != nullptr) store_state(p.z_out, y);
        continue;
      }
      // ---- forward: reversible_heun.py:61-73 ---------------------------------------------------------------------------------
      for (int j = 0; j <= K; ++j) {
        const bool later = j < K, earlier = j > 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) dwb[i] = dwa[i];
        if (later) draw(p.cells[j], p.rows[(int64_t)j * 8 + 4], dwa);
        else zero_w(dwa);
        evaluate(p.times[j], z, dwa, dwb, f, sa, sb, nullptr, nullptr, nullptr, nullptr, 0);
        if (earlier) {
          const float hdt = p.rows[(int64_t)(j - 1) * 8 + 1];
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][r] = (yp[t][r] + f[t][r] * hdt) + 0.5f * sb[t][r];
          }
          while (jout < p.n_out && p.out_step[jout] == j) {
            const float w0 = p.out_w[2 * jout], w1 = p.out_w[2 * jout + 1];
            const bool exact = w0 == 0.0f && w1 == 1.0f;
            f32x4 o[TD];
#pragma unroll
            for (int t = 0; t < TD; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) o[t][r] = exact ? y[t][r] : (w0 * yprev[t][r] + w1 * y[t][r]);
            }
            store_state(p.ys + (int64_t)jout * p.B * dT, o);
            ++jout;
          }
        }
        if (later) {
          const float dt = p.rows[(int64_t)j * 8], hdt = p.rows[(int64_t)j * 8 + 1];
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float zn = ((2.0f * y[t][r] - z[t][r]) + f[t][r] * dt) + sa[t][r];
              yp[t][r] = (y[t][r] + f[t][r] * hdt) + 0.5f * sa[t][r];
              yprev[t][r] = y[t][r];
              z[t][r] = zn;
            }
          }
        }
      }
      if (p.z_out != nullptr) store_state(p.z_out, z);
    } else {
      // ---- backward: reversible_heun.py:98-144, one evaluation per iteration (see the head of this file) -------------------------
      f32x4 yp[TD], z[TD], ay[TD], az[TD], af[TD], pv[TD], qv[TD], cf[TD], vjp[TD], y[TD];
      load_state(p.s_y, yp);
      load_state(p.s_z, z);
      load_state(p.s_ay, ay);
      load_state(p.s_az, az);
      load_state(p.s_af, af);
      load_state(p.s_p, pv);
      int jout = p.n_out - 1;
      zero_w(dwb);
      if (p.j_hi > 0) draw(p.cells[p.j_hi - 1], p.rows[(int64_t)(p.j_hi - 1) * 8 + 4], dwb);
      zero_w(dwa);
      if (p.j_hi < K) draw(p.cells[p.j_hi], p.rows[(int64_t)p.j_hi * 8 + 4], dwa);
      for (int j = p.j_hi; j >= p.j_lo; --j) {
        const bool later = j < K, earlier = j > 0;
        while (jout >= 0 && p.out_step[jout] > j) --jout;
        const int boundary = j == 0 ? 0 : (jout >= 0 && p.out_step[jout] == j) ? jout + 1 : -1;
        if (boundary >= 0 && later) {                       // adjoint.py:114-116: the output's cotangent joins a_y
          f32x4 gy[TD];
          load_state(p.gys + (int64_t)boundary * p.B * dT, gy);
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ay[t][r] += gy[t][r];
          }
        }
        const float dt_e = earlier ? p.rows[(int64_t)(j - 1) * 8] : 0.0f;
        const float hdt_e = earlier ? p.rows[(int64_t)(j - 1) * 8 + 1] : 0.0f;
        const float hdt_l = later ? p.rows[(int64_t)j * 8 + 1] : 0.0f;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            cf[t][r] = af[t][r] + ay[t][r] * hdt_e;         // reversible_heun.py:112
            qv[t][r] = earlier ? 0.5f * ay[t][r] : 0.0f;    // :113 -- a_y (x) dW / 2, kept as the vector
          }
        }
        evaluate(p.times[j], z, dwa, dwb, f, sa, sb, cf, pv, qv, vjp, p.j_hi - j);
#pragma unroll
        for (int t = 0; t < TD; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) y[t][r] = later ? (yp[t][r] - f[t][r] * hdt_l) - 0.5f * sa[t][r] : yp[t][r];
        }
        if (boundary >= 0 && later) load_state(p.ys_all + (int64_t)boundary * p.B * dT, y);     // the stored state (adjoint.py:114)
        if (!earlier) {
          // z_0 = y_0 and (f_0, g_0) are functions of y_0 (init_extra_solver_state, reversible_heun.py:58-59)
#pragma unroll
          for (int t = 0; t < TD; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ay[t][r] = (ay[t][r] + az[t][r]) + vjp[t][r];
          }
          break;
        }
#pragma unroll
        for (int t = 0; t < TD; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float zn = ((2.0f * y[t][r] - z[t][r]) - f[t][r] * dt_e) - sb[t][r];            // :109
            const float azp = az[t][r] + vjp[t][r];                                              // :127
            af[t][r] = ay[t][r] * hdt_e + azp * dt_e;                                            // :112, :136
            pv[t][r] = 0.5f * ay[t][r] + azp;                                                    // :113, :137
            ay[t][r] = ay[t][r] + 2.0f * azp;                                                    // :134
            az[t][r] = -azp;                                                                     // :135
            yp[t][r] = (y[t][r] - f[t][r] * hdt_e) - 0.5f * sb[t][r];                            // :130-131, its first half
            z[t][r] = zn;
          }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) dwa[i] = dwb[i];
        if (j - 2 >= 0) draw(p.cells[j - 2], p.rows[(int64_t)(j - 2) * 8 + 4], dwb);
        else zero_w(dwb);
      }
      store_state(p.s_y, yp);
      store_state(p.s_z, z);
      store_state(p.s_ay, ay);
      store_state(p.s_az, az);
      store_state(p.s_af, af);
      store_state(p.s_p, pv);
    }
  }
}

}  // namespace tsde
