// dL/dW2, dL/db2 of a general-noise diffusion net's LAST layer in the backward sweep of the reversible Heun pair
// (tsde_neural_rheun.h; reference: the parameter part of the vector-Jacobian product misc.vjp forms in
// methods/reversible_heun.py:119-126). The cotangent of that layer's (rows, d, m) output is never materialised by the sweep --
// it is  p (x) wa + q (x) wb  with two state-sized vectors and the two increments of the evaluation -- so the sweep stashes
// (hid, p, q, wa, wb) per evaluation and row, and this kernel forms, over all N = evaluations x rows stash rows,
//     pre  = hid W2 + b2                         (recomputed: 2 N H d m flop)
//     cot  = (p_i wa_j + q_i wb_j) final'(pre)   (o = i m + j)
//     gW2 += hid^T cot,  gb2 += column sums      (2 N H d m flop)
// on the matrix cores, without the three (N, d m) temporaries and the library products a torch statement needs (measured: 233 of
// the 309 ms of a configs[2]-sized forward + backward before this kernel existed). Deterministic: every row block writes its own
// partial sums, the host adds them in a fixed order.
//
// A block = 4 waves works on 16 stash rows at a time and on 16 consecutive output tiles of 16 outputs (its slice of W2 staged in
// LDS once per block); wave w owns tiles 4 w .. 4 w + 3 of the slice. Per tile 16 MFMAs recompute pre (ROWS x outputs: a lane
// then holds one output and four rows 4 part + r), the closing arithmetic runs on those four, and 16 MFMAs add hid^T cot with the
// rows as the K index in the order the lanes hold them (MFMA r: k = part stands for row 4 part + r) -- the cotangent tile is
// consumed from the registers it was born in. p, q, wa, wb sit transposed in LDS ([channel][row], rows padded to 20: a lane's
// four rows are one conflict-free 16-byte read). The next group's rows are fetched into registers while this group computes.
#include <type_traits>

#include "tsde_common.h"
#include "tsde_launch.h"
#include "tsde_mlp.h"

namespace tsde {

struct LastLayerArgs {
  float* gw;          // (row_blocks, hidden, out) partial sums, input-major like the kernels' w2
  float* gb;          // (row_blocks, out)
  const float *hid, *p, *q, *wa, *wb;
  const float *w2, *b2;      // (hidden, out), (out)
  int64_t N;
  int32_t d, m, hidden, out, final;
  int32_t sh, sd, sm;        // row strides of hid / p, q / wa, wb
};

template <int FINAL>
TSDE_D float final_slope(float x) {
  if constexpr (FINAL == TSDE_FINAL_SIGMOID) {
    const float v = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    return v * (1.0f - v);
  } else if constexpr (FINAL == TSDE_FINAL_TANH) {
    const float e2x = __builtin_amdgcn_exp2f(x * (2.0f * 1.4426950408889634f));
    const float v = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e2x + 1.0f);
    return 1.0f - v * v;
  } else {
    return 1.0f;
  }
}

constexpr int kOT = 4;                  // output tiles per wave
constexpr int kSlice = 4 * kOT * 16;    // outputs per block

constexpr int kST = 20;                 // row stride of the transposed p / q / wa / wb tiles

template <int H>
__global__ void __launch_bounds__(256, 2) rheun_last_layer_kernel(const LastLayerArgs p) {
  constexpr int TH = H / 16, SH = H + 4, SW = kSlice + 4;
  extern __shared__ float lds[];
  float* W2L = lds;                       // [H][SW]: this block's slice of W2
  float* hidL = W2L + H * SW;             // [16][SH]
  float* pT = hidL + 16 * SH;             // [sd][kST], [sd][kST], [sm][kST], [sm][kST]
  float* qT = pT + p.sd * kST;
  float* waT = qT + p.sd * kST;
  float* wbT = waT + p.sm * kST;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, part = lane >> 4, n = lane & 15;
  const int o_base = blockIdx.y * kSlice;
  for (int i = threadIdx.x; i < H * kSlice; i += 256) {
    const int u = i / kSlice, c = i % kSlice, o = o_base + c;
    W2L[u * SW + c] = (u < p.hidden && o < p.out) ? p.w2[(int64_t)u * p.out + o] : 0.0f;
  }
  // per tile: this lane's output o = (i, j), its bias, where its four rows of p / q and of wa / wb sit
  float bias[kOT], live[kOT];
  int pofs[kOT], wofs[kOT];
#pragma unroll
  for (int tl = 0; tl < kOT; ++tl) {
    const int o = o_base + 16 * (kOT * wave + tl) + n;
    const bool have = o < p.out;
    const int i = have ? o / p.m : 0, j = have ? o - i * p.m : 0;
    bias[tl] = have ? p.b2[o] : 0.0f;
    live[tl] = have ? 1.0f : 0.0f;
    pofs[tl] = i * kST + 4 * part;
    wofs[tl] = j * kST + 4 * part;
  }
  f32x4 acc[kOT][TH];
  float gbacc[kOT];
#pragma unroll
  for (int tl = 0; tl < kOT; ++tl) {
    gbacc[tl] = 0.0f;
#pragma unroll
    for (int th = 0; th < TH; ++th) acc[tl][th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // a group's rows on their way to LDS: 16 x H of hid (one 16-byte piece per thread for H = 64), 16 x sd of p and q (sd <= 64:
  // up to four elements per thread), 16 x sm of wa and wb (sm <= 16: one)
  constexpr int kPQ = 4;
  f32x4 r_hid;
  float r_p[kPQ], r_q[kPQ], r_wa, r_wb;
  const int hid_row = threadIdx.x / (H / 4), hid_col = 4 * (threadIdx.x % (H / 4));
  auto fetch = [&](int64_t row0) {
    r_hid = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (hid_row < 16 && row0 + hid_row < p.N && hid_col < p.sh)
      r_hid = *reinterpret_cast<const f32x4*>(p.hid + (row0 + hid_row) * p.sh + hid_col);
#pragma unroll
    for (int e = 0; e < kPQ; ++e) {
      const int i = threadIdx.x + 256 * e, rr = i / p.sd, c = i - rr * p.sd;
      const bool have = i < 16 * p.sd && row0 + rr < p.N;
      r_p[e] = have ? p.p[(row0 + rr) * p.sd + c] : 0.0f;
      r_q[e] = have ? p.q[(row0 + rr) * p.sd + c] : 0.0f;
    }
    {
      const int i = threadIdx.x, rr = i / p.sm, c = i - rr * p.sm;
      const bool have = i < 16 * p.sm && row0 + rr < p.N;
      r_wa = have ? p.wa[(row0 + rr) * p.sm + c] : 0.0f;
      r_wb = have ? p.wb[(row0 + rr) * p.sm + c] : 0.0f;
    }
  };
  auto deposit = [&]() {
    if (hid_row < 16) {
      f32x4 v = r_hid;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (hid_col + r >= p.hidden) v[r] = 0.0f;          // (padded units of the stash hold act(0), not 0)
      }
      *reinterpret_cast<f32x4*>(hidL + hid_row * SH + hid_col) = v;
    }
#pragma unroll
    for (int e = 0; e < kPQ; ++e) {
      const int i = threadIdx.x + 256 * e, rr = i / p.sd, c = i - rr * p.sd;
      if (i < 16 * p.sd) {
        pT[c * kST + rr] = r_p[e];
        qT[c * kST + rr] = r_q[e];
      }
    }
    {
      const int i = threadIdx.x, rr = i / p.sm, c = i - rr * p.sm;
      if (i < 16 * p.sm) {
        waT[c * kST + rr] = r_wa;
        wbT[c * kST + rr] = r_wb;
      }
    }
  };
  auto compute = [&](auto kind) {
    constexpr int FINAL = decltype(kind)::value;
    // hid as the A operand of pre (lane (part, n): row n, unit 4 kk + part) and of hid^T cot (unit 16 th + n of row 4 part + r)
    float hA[H / 4], hT[4][TH];
#pragma unroll
    for (int kk = 0; kk < H / 4; ++kk) hA[kk] = hidL[n * SH + 4 * kk + part];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int th = 0; th < TH; ++th) hT[r][th] = hidL[(4 * part + r) * SH + 16 * th + n];
    }
#pragma unroll
    for (int tl = 0; tl < kOT; ++tl) {
      const int col = 16 * (kOT * wave + tl);
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(pT + pofs[tl]), q4 = *reinterpret_cast<const f32x4*>(qT + pofs[tl]);
      const f32x4 wa4 = *reinterpret_cast<const f32x4*>(waT + wofs[tl]), wb4 = *reinterpret_cast<const f32x4*>(wbT + wofs[tl]);
      f32x4 pre = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kk = 0; kk < H / 4; ++kk) {
        const float b = W2L[(4 * kk + part) * SW + col + n];
        pre = Tile<16>::mfma(hA[kk], b, pre);
      }
      f32x4 cot;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = final_slope<FINAL>(pre[r] + bias[tl]);
        cot[r] = ((p4[r] * wa4[r] + q4[r] * wb4[r]) * s) * live[tl];
        gbacc[tl] += cot[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int th = 0; th < TH; ++th) acc[tl][th] = Tile<16>::mfma(hT[r][th], cot[r], acc[tl][th]);
      }
    }
  };
  const int64_t groups = (p.N + 15) / 16;
  const bool wave_has_outputs = o_base + 16 * kOT * wave < p.out;      // (a small net: the other waves only stage and wait)
  auto sweep = [&](auto kind) {           // (the closing function is uniform over the launch: one branch around the whole sweep)
    if ((int64_t)blockIdx.x < groups) fetch((int64_t)blockIdx.x * 16);
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
      __syncthreads();                    // (the previous group's tiles are no longer read; W2L is complete on the first pass)
      deposit();
      __syncthreads();
      if (grp + gridDim.x < groups) fetch((grp + gridDim.x) * 16);
      if (wave_has_outputs) compute(kind);
    }
  };
  if (p.final == TSDE_FINAL_SIGMOID) sweep(std::integral_constant<int, TSDE_FINAL_SIGMOID>{});
  else if (p.final == TSDE_FINAL_TANH) sweep(std::integral_constant<int, TSDE_FINAL_TANH>{});
  else sweep(std::integral_constant<int, TSDE_FINAL_NONE>{});
  // partial sums of this row block
#pragma unroll
  for (int tl = 0; tl < kOT; ++tl) {
    const int o = o_base + 16 * (kOT * wave + tl) + n;
#pragma unroll
    for (int th = 0; th < TH; ++th) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = 16 * th + 4 * part + r;
        if (u < p.hidden && o < p.out) p.gw[((int64_t)blockIdx.x * p.hidden + u) * p.out + o] = acc[tl][th][r];
      }
    }
    // column sums: a lane holds its output's sum over the rows 4 part + r of every group; the four lane quarters add up
    float s = gbacc[tl];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (part == 0 && o < p.out) p.gb[(int64_t)blockIdx.x * p.out + o] = s;
  }
}

hipError_t launch_rheun_last_layer_grad(void* gw, void* gb, const void* hid, const void* pp, const void* q, const void* wa,
                                        const void* wb, int64_t N, int64_t d, int64_t m, const tsde_deep_mlp_t* net,
                                        int32_t stride_h, int32_t stride_d, int32_t stride_m, int32_t row_blocks, hipStream_t s) {
  LastLayerArgs a;
  a.gw = (float*)gw;
  a.gb = (float*)gb;
  a.hid = (const float*)hid;
  a.p = (const float*)pp;
  a.q = (const float*)q;
  a.wa = (const float*)wa;
  a.wb = (const float*)wb;
  a.w2 = (const float*)net->w2;
  a.b2 = (const float*)net->b2;
  a.N = N;
  a.d = (int32_t)d;
  a.m = (int32_t)m;
  a.hidden = net->hidden;
  a.out = net->out;
  a.final = net->final;
  a.sh = stride_h;
  a.sd = stride_d;
  a.sm = stride_m;
  if (N <= 0) return hipSuccess;
  const int H = net->hidden <= 32 ? 32 : 64;
  if (stride_d > 64 || stride_m > 16 || stride_h > H) return hipErrorInvalidValue;      // (what a thread's fetch registers hold)
  const size_t lds = ((size_t)H * (kSlice + 4) + 16 * (H + 4) + 2 * kST * (size_t)stride_d + 2 * kST * (size_t)stride_m) *
                     sizeof(float);
  const dim3 grid((unsigned)row_blocks, (unsigned)((net->out + kSlice - 1) / kSlice));
  if (H == 32) {
    static bool configured = false;
    if (!configured) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rheun_last_layer_kernel<32>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
      if (e != hipSuccess) return e;
      configured = true;
    }
    TSDE_LAUNCH((rheun_last_layer_kernel<32>), grid, dim3(256), lds, s, a);
  } else {
    static bool configured = false;
    if (!configured) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rheun_last_layer_kernel<64>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
      if (e != hipSuccess) return e;
      configured = true;
    }
    TSDE_LAUNCH((rheun_last_layer_kernel<64>), grid, dim3(256), lds, s, a);
  }
  return hipGetLastError();
}

}  // namespace tsde
