// Gradient of the perceptron-drift trajectory kernel (mlp_trajectory.hip), Euler-Maruyama: back-propagation through
// the solver, the gradient `loss.backward()` yields when autograd records torchsde's stepping loop
// (base_solver.py:114-134 with euler.py:31-36) -- as two kernels instead of a tape of ~20 torch ops per step.
//
//   step k (forward):   z = W1 y_k + b1,  h = act(z),  f = W2 h + b2,  g = c*y_k + e,
//                       y_{k+1} = y_k + f dt_k + g dW_k
//   step k (reverse):   lam = dL/dy_{k+1}
//                       u     = W2^T lam                 delta = u * act'(z) * dt_k
//                       dL/dy_k = lam + W1^T delta + lam * c * dW_k
//                       dL/dW2 += (dt_k lam) h^T         dL/dW1 += delta y_k^T
//                       dL/db2 += dt_k lam               dL/db1 += delta
//                       dL/dc  += lam * y_k * dW_k       dL/de  += lam * dW_k
//   Milstein (milstein.py:52-74 for this g: gdg = g c) adds (g v) c with v = (dW^2 - dt)/2 (Ito) or dW^2/2:
//                       dL/dy_k += lam c^2 v             dL/dc += lam v (g + c y_k)       dL/de += lam c v
//
// 1. mlp_backward_kernel -- the reverse sweep. A wave owns 16 batch rows for all steps; lam stays in registers in the
//    MFMA accumulator layout (the layout of the sampling kernel: it is the B operand of the next product as it
//    stands); the three products per step run on v_mfma_f32_16x16x4_f32 against the SAME two LDS weight arrays the
//    sampling kernel uses -- layer 1 reads W1 as stored, the two transposed products read 16-byte rows (four
//    consecutive K per lane, one ds_read_b128 feeding four MFMAs; conflict-free with the +4 row padding). The states
//    y_k come from the forward launch (which wrote every step: 288 GB of HBM is what pays for this), dW_k is
//    regenerated from the counter RNG, and the per-step factors of the weight gradients (dt lam, h, delta) are
//    stashed in HBM for:
// 2. gram_kernel -- C = A^T B and the column sums of A over a very tall K (K = steps x batch rows, M, N <= 128): the
//    weight- and bias-gradient sums. Each block owns a contiguous K range and accumulates the whole M x N result in
//    MFMA accumulators across its 8 waves, streaming both operands through a double-buffered LDS tile; per-block
//    partials are written out and summed in a fixed order by the caller (deterministic, unlike atomics).
#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_mlp.h"
#include "tsde_schemes.h"

namespace tsde {

struct MlpBackArgs {
  float* lam;               // (B, d)  in: dL/dy at boundary k_hi (before that boundary's own cotangent); out: at k_lo
  float* stash_lam;         // (k_hi - k_lo, B, d)   dt_k * dL/dy_{k+1}
  float* stash_hid;         // (k_hi - k_lo, B, h)   act(W1 y_k + b1)
  float* stash_delta;       // (k_hi - k_lo, B, h)   delta_k
  float* row_rate;          // (B, d)  += sum_k lam * y_k * dW_k   (per trajectory; the caller sums over the batch)
  float* row_shift;         // (B, d)  += sum_k lam * dW_k
  const float* ys_all;      // (.., B, d) states at step boundaries ys_first, ys_first + 1, ... (at least up to k_hi - 1)
  int32_t ys_first;
  const float* grad_ys;     // (n_grad, B, d) cotangents of the outputs
  const int32_t* grad_step; // (n_grad) ascending boundary index of each output
  int32_t grad_last;        // index of the last output at a boundary <= k_hi (-1: none)
  const float* W1;          // (d, h) as in MlpArgs
  const float* b1;          // (h)
  const float* W2;          // (h, d)
  const float *c, *e;       // (d) diffusion g = c*y + e (e is only read by the Milstein terms)
  int32_t method;           // TSDE_TRAJ_EULER / _MILSTEIN_ITO / _MILSTEIN_STRAT
  int32_t diff_kind;        // TSDE_DIFF_AFFINE / TSDE_DIFF_SIGMOID (the latter with Euler only)
  float diff_amp;
  const float* rows;        // (n_steps, 8)
  const uint32_t* cells;
  int64_t B;
  int32_t d, h;
  int32_t k_lo, k_hi;
  NoiseKey key;
  const uint64_t* key_dev;
};

// FULL: d == D and h == H (no channel padding inside the kernel), which removes every per-tile bounds test.
template <int D, int H, int ACT, int NW, bool FULL>
__global__ void __launch_bounds__(NW * 64) mlp_backward_kernel(const MlpBackArgs p) {
  constexpr int R = 16;
  using TL = Tile<R>;
  constexpr int TD = D / R, TH = H / R, kThreads = NW * 64;
  constexpr int S1 = H + MlpLds<R>::kPad, S2 = D + MlpLds<R>::kPad;
  extern __shared__ float lds[];
  float* W1s = lds;                 // D rows of S1: W1s[channel][hidden]
  float* W2s = W1s + D * S1;        // H rows of S2: W2s[hidden][channel]
  float* b1s = W2s + H * S2;        // H
  float* cs = b1s + H;              // D
  float* es = cs + D;               // D
  const int dT = p.d, hT = p.h;
  for (int i = threadIdx.x; i < D * H; i += kThreads) {
    const int k1 = i / H, m1 = i % H, k2 = i / D, m2 = i % D;
    W1s[k1 * S1 + m1] = (k1 < dT && m1 < hT) ? p.W1[k1 * hT + m1] : 0.0f;
    W2s[k2 * S2 + m2] = (k2 < hT && m2 < dT) ? p.W2[k2 * dT + m2] : 0.0f;
  }
  for (int i = threadIdx.x; i < H; i += kThreads) b1s[i] = i < hT ? p.b1[i] : 0.0f;
  for (int i = threadIdx.x; i < D; i += kThreads) {
    cs[i] = i < dT ? p.c[i] : 0.0f;
    es[i] = i < dT ? p.e[i] : 0.0f;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane / R, n = lane % R;
  // A wave with no batch row left is done; in the last partial wave the surplus lanes shadow the last row: they read
  // what its lane reads, draw the same noise, and (re)write the same values to the same addresses -- no lane masks.
  const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * R;
  if (row0 >= p.B) return;
  const int64_t row = row0 + n < p.B ? row0 + n : p.B - 1;
  NoiseKey key = p.key;
  if (p.key_dev != nullptr) {
    const uint64_t ent = *p.key_dev;
    key.k0 = (uint32_t)ent;
    key.k1 = (uint32_t)(ent >> 32);
  }

  // register r of tile t in this lane = channel 16 t + 4 part + r of batch row `row` (one 16-byte quad per tile).
  // Addresses are a wave-uniform base (SGPRs) + ONE 32-bit lane offset per row width + a constant per tile, so that
  // no per-tile 64-bit address lives in vector registers across the sweep (the caller keeps rows * width < 2^30).
  const uint32_t off_d = (uint32_t)(row * dT) + 4 * part, off_h = (uint32_t)(row * hT) + 4 * part;
  const uint64_t quad0 = (key.elem0 + (uint64_t)(row * dT) + 4 * part) >> 2;   // RNG quad of tile 0; tile t: + 4 t
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

  f32x4 lam[TD], acc_rate[TD], acc_shift[TD];
#pragma unroll
  for (int t = 0; t < TD; ++t) {
    lam[t] = load_tile(p.lam, t);
    acc_rate[t] = {0.0f, 0.0f, 0.0f, 0.0f};
    acc_shift[t] = {0.0f, 0.0f, 0.0f, 0.0f};
  }

  // y_k is loaded one step ahead (issued before the last product of the previous step, which does not read it): a
  // load consumed right behind its issue would expose the memory latency once per step -- and per tile in the diffusion
  // terms, which use the same registers again instead of re-reading.
  f32x4 y[TD];
#pragma unroll
  for (int t = 0; t < TD; ++t) y[t] = load_tile(p.ys_all + (int64_t)(p.k_hi - 1 - p.ys_first) * p.B * dT, t);
  int jg = p.grad_last;
  for (int k = p.k_hi - 1; k >= p.k_lo; --k) {
    const float* srow = p.rows + (int64_t)k * 8;
    const float dt = srow[0], sw = srow[4];
    const uint32_t cell = p.cells[k];
    const int64_t slot = k - p.k_lo;
    const bool arrives = jg >= 0 && p.grad_step[jg] == k + 1;     // an output sits on boundary k + 1

#pragma unroll
    for (int t = 0; t < TD; ++t) {
      if (arrives) lam[t] += load_tile(p.grad_ys + (int64_t)jg * p.B * dT, t);
      store_tile(p.stash_lam + slot * p.B * dT, t, lam[t] * dt);
    }
    if (arrives) --jg;
    __builtin_amdgcn_sched_barrier(0);

    // ---- hidden layer: z^T = W1^T y^T; keep act'(z), stash act(z) ---------------------------------------------------
    f32x4 hid[TH];
#pragma unroll
    for (int th = 0; th < TH; ++th) {
      f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < TD; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) z = TL::mfma(W1s[(R * t + 4 * part + r) * S1 + R * th + n], y[t][r], z);
      }
      // the A operands arrive as 2 TD two-address reads (ds_read2_b32: rows r, r+1 of W1s), each feeding two MFMAs: keep
      // four of them in flight ahead of the MFMAs instead of hipcc's read -> wait -> two MFMAs
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
      store_hidden(p.stash_hid + slot * p.B * hT, th, value);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- u^T = W2 lam^T (rows of W2s, four consecutive channels per lane); delta = u * act'(z) * dt ----------------
#pragma unroll
    for (int th = 0; th < TH; ++th) {
      f32x4 u = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < TD; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&W2s[(R * th + n) * S2 + R * t + 4 * part]);
#pragma unroll
        for (int r = 0; r < 4; ++r) u = TL::mfma(a[r], lam[t][r], u);
        if ((t + 1) % 4 == 0) __builtin_amdgcn_sched_barrier(0);
      }
      hid[th] = (u * hid[th]) * dt;
      store_hidden(p.stash_delta + slot * p.B * hT, th, hid[th]);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- diffusion: dW_k again from the counter RNG; lam <- lam + lam c dW ------------------------------------------
#pragma unroll
    for (int t = 0; t < TD; ++t) {
      const int ch = R * t + 4 * part;
      float zn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      // (opaque to the optimiser: otherwise the step-invariant first Philox round of every tile is hoisted out of the
      //  sweep and pinned in ~4 registers per tile)
      uint64_t quad = quad0 + 4 * t;
      asm volatile("" : "+v"(quad));
      if (real_d(t)) normal4<float>(key, quad, cell, 0, kStreamW, zn);
      const f32x4 yt = y[t];
      const f32x4 cq = lds_quad(cs, ch);
      const f32x4 eq = lds_quad(es, ch);
      if (p.method == TSDE_TRAJ_EULER) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // dg/de = q, dg/dc = q y, dg/dy = q c   (affine: q = 1)
          const float q = diffusion_value(p.diff_kind == TSDE_DIFF_SIGMOID, p.diff_amp, cq[r], eq[r], yt[r]).q;
          const float lw = (lam[t][r] * (zn[r] * sw)) * q;
          acc_shift[t][r] += lw;
          acc_rate[t][r] += lw * yt[r];
          lam[t][r] += lw * cq[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float w = zn[r] * sw, cc = cq[r];
          const float v = milstein_v<float>(w, dt, 0.5f, p.method == TSDE_TRAJ_MILSTEIN_ITO);
          const float cy = cc * yt[r];
          const float l = lam[t][r];
          acc_shift[t][r] += l * (w + cc * v);
          acc_rate[t][r] += l * (yt[r] * w + v * ((cy + eq[r]) + cy));
          lam[t][r] += l * (cc * w + (cc * cc) * v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    if (k > p.k_lo) {
#pragma unroll
      for (int t = 0; t < TD; ++t) y[t] = load_tile(p.ys_all + (int64_t)(k - 1 - p.ys_first) * p.B * dT, t);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- lam^T += W1 delta^T (rows of W1s, four consecutive hidden units per lane) ------------------------------------
#pragma unroll
    for (int t = 0; t < TD; ++t) {
      f32x4 back = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int th = 0; th < TH; ++th) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&W1s[(R * t + n) * S1 + R * th + 4 * part]);
#pragma unroll
        for (int r = 0; r < 4; ++r) back = TL::mfma(a[r], hid[th][r], back);
        if ((th + 1) % 4 == 0) __builtin_amdgcn_sched_barrier(0);
      }
      lam[t] += back;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

#pragma unroll
  for (int t = 0; t < TD; ++t) {
    store_tile(p.lam, t, lam[t]);
    store_tile(p.row_rate, t, load_tile(p.row_rate, t) + acc_rate[t]);
    store_tile(p.row_shift, t, load_tile(p.row_shift, t) + acc_shift[t]);
  }
}

template <int D, int H, int ACT, int NW, bool FULL>
static hipError_t launch_back_variant(const MlpBackArgs& p, hipStream_t s) {
  constexpr int R = 16;
  const size_t lds_bytes = (size_t)(D * (H + MlpLds<R>::kPad) + H * (D + MlpLds<R>::kPad) + H + 2 * D) * sizeof(float);
  static bool configured = false;   // per instantiation
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_kernel<D, H, ACT, NW, FULL>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int64_t rows_per_block = NW * R;
  const int64_t blocks = (p.B + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((mlp_backward_kernel<D, H, ACT, NW, FULL>), dim3((unsigned)blocks), dim3(NW * 64), lds_bytes, s, p);
  return hipGetLastError();
}

// 8-wave blocks (two waves per SIMD, 256 registers each) everywhere except the padded 128-channel case, whose bounds
// tests push the sweep past 256 registers: 4-wave blocks there (one wave per SIMD, 512 registers) instead of spilling.
template <int D, int H, int ACT>
static hipError_t launch_back_shape(const MlpBackArgs& p, hipStream_t s) {
  if (p.d == D && p.h == H) return launch_back_variant<D, H, ACT, 8, true>(p, s);
  if constexpr (D == 128) return launch_back_variant<D, H, ACT, 4, false>(p, s);
  else return launch_back_variant<D, H, ACT, 8, false>(p, s);
}

template <int D, int H>
static hipError_t launch_back_act(const MlpBackArgs& p, int act, hipStream_t s) {
  if (act == TSDE_ACT_TANH) return launch_back_shape<D, H, TSDE_ACT_TANH>(p, s);
  if (act == TSDE_ACT_SOFTPLUS) return launch_back_shape<D, H, TSDE_ACT_SOFTPLUS>(p, s);
  return hipErrorInvalidValue;
}

template <int D>
static hipError_t launch_back_h(const MlpBackArgs& p, int act, hipStream_t s) {
  if (p.h <= 32) return launch_back_act<D, 32>(p, act, s);
  if (p.h <= 64) return launch_back_act<D, 64>(p, act, s);
  if (p.h <= 128) return launch_back_act<D, 128>(p, act, s);
  if constexpr (D <= 64) {                 // both weight arrays must fit the LDS of a CU: d * hidden <= 16384
    if (p.h <= 256) return launch_back_act<D, 256>(p, act, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_trajectory_mlp_diag_backward(void* lam, void* stash_lam, void* stash_hid, void* stash_delta,
                                               void* row_rate, void* row_shift, const void* ys_all,
                                               int32_t ys_first, const void* grad_ys, const int32_t* grad_step, int32_t grad_last,
                                               int64_t rows, int64_t d, int64_t h, const void* W1, const void* b1,
                                               const void* W2, const void* c, const void* e, int diff_kind,
                                               double diff_amp, int act, int method, const tsde_traj_t* tr,
                                               int32_t k_lo, int32_t k_hi, NoiseKey key, const uint64_t* key_dev,
                                               hipStream_t s) {
  MlpBackArgs p;
  p.lam = (float*)lam;
  p.stash_lam = (float*)stash_lam;
  p.stash_hid = (float*)stash_hid;
  p.stash_delta = (float*)stash_delta;
  p.row_rate = (float*)row_rate;
  p.row_shift = (float*)row_shift;
  p.ys_all = (const float*)ys_all;
  p.ys_first = ys_first;
  p.grad_ys = (const float*)grad_ys;
  p.grad_step = grad_step;
  p.grad_last = grad_last;
  p.W1 = (const float*)W1;
  p.b1 = (const float*)b1;
  p.W2 = (const float*)W2;
  p.c = (const float*)c;
  p.e = (const float*)e;
  p.method = method;
  p.diff_kind = diff_kind;
  p.diff_amp = (float)diff_amp;
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
  if (d <= 32) return launch_back_h<32>(p, act, s);
  if (d <= 64) return launch_back_h<64>(p, act, s);
  if (d <= 128) return launch_back_h<128>(p, act, s);
  return hipErrorInvalidValue;
}

// ---- C = A^T B (and the column sums of A) over a tall K --------------------------------------------------------------
// A block owns a contiguous K range and walks it in tiles of 32 rows: 16-byte coalesced loads of both operands into
// registers (the NEXT tile's loads are in flight while the current one is multiplied), one LDS copy per tile, double
// buffered (one barrier per tile). Waves are arranged 4 (along M) x 2 (along N); a wave holds MT x NT accumulator tiles
// of 16 x 16 and reads its operands from LDS in MFMA layout:
//   A operand of v_mfma_f32_16x16x4_f32: lane l supplies A^T[i = l % 16][k = l / 16] = tile[k0 + l / 16][16 ti + l % 16]
//   B operand:                           lane l supplies B[k = l / 16][j = l % 16]   = tile[k0 + l / 16][16 tj + l % 16]
// (row stride padded by 16 floats: the two K rows of a half-wave then sit in opposite halves of the 32 banks).
// The loader's threads keep fixed columns, so the column sums of A (the bias gradients) cost one add per loaded value.
constexpr int kGramRows = 32;

template <int MT, int NT, bool VEC>
__global__ void __launch_bounds__(512) gram_kernel(float* __restrict__ partials, float* __restrict__ colsums,
                                                   const float* __restrict__ A, int64_t lda,
                                                   const float* __restrict__ Bm, int64_t ldb, int64_t K, int32_t M,
                                                   int32_t N, int64_t rows_per_block) {
  using TL = Tile<16>;
  constexpr int MP = 64 * MT, NP = 32 * NT;                 // padded widths
  constexpr int SA = MP + 16, SB = NP + 16;                 // LDS row strides (floats)
  constexpr int QA = MP / 4, QB = NP / 4;                   // 16-byte slots per tile row
  constexpr int LA = (kGramRows * QA + 511) / 512, LB = (kGramRows * QB + 511) / 512;   // slots per thread
  __shared__ float As[2][kGramRows * SA];
  __shared__ float Bs[2][kGramRows * SB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = lane >> 4, n = lane & 15;
  const int wm = wave & 3, wn = wave >> 2;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 csum = {0.0f, 0.0f, 0.0f, 0.0f};                    // columns 4 (tid % QA) .. + 3 of A, this thread's rows

  const int64_t k0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t k1 = k0 + rows_per_block < K ? k0 + rows_per_block : K;
  const int ntiles = k1 > k0 ? (int)((k1 - k0 + kGramRows - 1) / kGramRows) : 0;

  // Loader: a slot is 16 bytes of one tile row. `fetch` only ISSUES the load (from a clamped, always valid address);
  // whether the slot lies inside the tile is applied when the registers are copied to LDS, one tile of matrix
  // instructions later -- anything that touches the loaded value earlier (a select, the column sum) would put the
  // wait for the load in front of those instructions.
  f32x4 ra[LA], rb[LB];
  bool in_a[LA], in_b[LB];
  auto fetch = [&](const float* __restrict__ src, int64_t ld, int32_t width, int q_per_row, int slot, int64_t kk,
                   bool& inside) {
    const int r = slot / q_per_row, c = 4 * (slot % q_per_row);
    const int64_t krow = kk + r;
    inside = r < kGramRows && krow < k1 && c < width;
    if constexpr (VEC) {
      const int64_t row_c = krow < K ? krow : K - 1;        // (k1 > k0 >= 0 here: K - 1 is a valid row)
      const int col_c = c < width ? c : width - 4;
      return *reinterpret_cast<const f32x4*>(src + row_c * ld + col_c);
    } else {
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (inside) {
        const float* q = src + krow * ld + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < width) v[e] = q[e];
      }
      return v;
    }
  };
  auto load_tile = [&](int tile) {
    const int64_t kk = k0 + (int64_t)tile * kGramRows;
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[i] = fetch(A, lda, M, QA, tid + 512 * i, kk, in_a[i]);
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[i] = fetch(Bm, ldb, N, QB, tid + 512 * i, kk, in_b[i]);
  };
  auto stage_tile = [&](int buf) {
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int slot = tid + 512 * i;
      const f32x4 v = in_a[i] ? ra[i] : zero;
      if (slot < kGramRows * QA) *reinterpret_cast<f32x4*>(&As[buf][(slot / QA) * SA + 4 * (slot % QA)]) = v;
      csum += v;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int slot = tid + 512 * i;
      const f32x4 v = in_b[i] ? rb[i] : zero;
      if (slot < kGramRows * QB) *reinterpret_cast<f32x4*>(&Bs[buf][(slot / QB) * SB + 4 * (slot % QB)]) = v;
    }
  };

  if (ntiles > 0) {
    load_tile(0);
    stage_tile(0);
  }
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) load_tile(tile + 1);
#pragma unroll
    for (int g = 0; g < kGramRows / 4; ++g) {
      float fa[MT], fb[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) fa[a] = As[buf][(4 * g + part) * SA + 16 * (wm * MT + a) + n];
#pragma unroll
      for (int b = 0; b < NT; ++b) fb[b] = Bs[buf][(4 * g + part) * SB + 16 * (wn * NT + b) + n];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = TL::mfma(fa[a], fb[b], acc[a][b]);
    }
    if (tile + 1 < ntiles) stage_tile(buf ^ 1);
    __syncthreads();
  }

  float* out = partials + (int64_t)blockIdx.x * M * N;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * (wm * MT + a) + 4 * part + r, j = 16 * (wn * NT + b) + n;
        if (i < M && j < N) out[(int64_t)i * N + j] = acc[a][b][r];
      }

  if (colsums != nullptr) {
    // thread tid summed columns 4 (tid % QA) ..+3 over its rows: 512 / QA threads per column quad, added up in a
    // fixed order through LDS (the tile buffers are free now: the loop ended on a barrier)
    float* scratch = &As[0][0];                             // (512 / QA) x MP floats <= 2 * 32 * SA
    *reinterpret_cast<f32x4*>(&scratch[(tid / QA) * MP + 4 * (tid % QA)]) = csum;
    __syncthreads();
    if (tid < M) {
      float total = 0.0f;
      for (int j = 0; j < 512 / QA; ++j) total += scratch[j * MP + tid];
      colsums[(int64_t)blockIdx.x * M + tid] = total;
    }
  }
}

template <int MT, int NT>
static hipError_t launch_gram_vec(float* partials, float* colsums, const float* A, int64_t lda, const float* Bm,
                                  int64_t ldb, int64_t K, int32_t M, int32_t N, int32_t blocks, int64_t rows_per_block,
                                  hipStream_t s) {
  const bool vec = M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) & 15u) == 0;
  if (vec)
    hipLaunchKernelGGL((gram_kernel<MT, NT, true>), dim3(blocks), dim3(512), 0, s, partials, colsums, A, lda, Bm, ldb, K,
                       M, N, rows_per_block);
  else
    hipLaunchKernelGGL((gram_kernel<MT, NT, false>), dim3(blocks), dim3(512), 0, s, partials, colsums, A, lda, Bm, ldb,
                       K, M, N, rows_per_block);
  return hipGetLastError();
}

template <int MT>
static hipError_t launch_gram_n(float* partials, float* colsums, const float* A, int64_t lda, const float* Bm,
                                int64_t ldb, int64_t K, int32_t M, int32_t N, int32_t blocks, int64_t rows_per_block,
                                hipStream_t s) {
  if (N <= 32) return launch_gram_vec<MT, 1>(partials, colsums, A, lda, Bm, ldb, K, M, N, blocks, rows_per_block, s);
  if (N <= 64) return launch_gram_vec<MT, 2>(partials, colsums, A, lda, Bm, ldb, K, M, N, blocks, rows_per_block, s);
  if (N <= 128) return launch_gram_vec<MT, 4>(partials, colsums, A, lda, Bm, ldb, K, M, N, blocks, rows_per_block, s);
  return hipErrorInvalidValue;
}

hipError_t launch_gram_partials(void* partials, void* colsums, const void* A, int64_t lda, const void* Bm, int64_t ldb,
                                int64_t K, int64_t M, int64_t N, int32_t blocks, hipStream_t s) {
  if (K <= 0 || M <= 0 || N <= 0 || blocks <= 0) return hipErrorInvalidValue;
  int64_t rows_per_block = (K + blocks - 1) / blocks;
  rows_per_block = (rows_per_block + kGramRows - 1) / kGramRows * kGramRows;       // whole tiles
  if (M <= 64)
    return launch_gram_n<1>((float*)partials, (float*)colsums, (const float*)A, lda, (const float*)Bm, ldb, K,
                            (int32_t)M, (int32_t)N, blocks, rows_per_block, s);
  if (M <= 128)
    return launch_gram_n<2>((float*)partials, (float*)colsums, (const float*)A, lda, (const float*)Bm, ldb, K,
                            (int32_t)M, (int32_t)N, blocks, rows_per_block, s);
  return hipErrorInvalidValue;
}

}  // namespace tsde
