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
// A block = 4 waves works on 16 stash rows at a time (hid, p, q, wa, wb of those rows staged in LDS once) and on 16 consecutive
// output tiles of 16 outputs (its slice of W2 staged in LDS once per block); wave w owns tiles 4 w .. 4 w + 3 of the slice.
// Per tile: 16 MFMAs recompute pre^T (outputs x rows), the cotangent tile goes through a 16 x 16 LDS transpose (rows become the
// K index), 16 MFMAs add hid^T cot to the wave's accumulators.
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

template <int H>
__global__ void __launch_bounds__(256) rheun_last_layer_kernel(const LastLayerArgs p) {
  constexpr int TH = H / 16, SH = H + 4, SW = kSlice + 4, SC = 20;
  extern __shared__ float lds[];
  float* W2L = lds;                       // [H][SW]: this block's slice of W2
  float* hidL = W2L + H * SW;             // [16][SH]
  float* pL = hidL + 16 * SH;             // [16][sd], [16][sd], [16][sm], [16][sm]
  float* qL = pL + 16 * p.sd;
  float* waL = qL + 16 * p.sd;
  float* wbL = waL + 16 * p.sm;
  float* cotL = wbL + 16 * p.sm;          // [4 waves][16][SC]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, part = lane >> 4, n = lane & 15;
  const int o_base = blockIdx.y * kSlice;
  for (int i = threadIdx.x; i < H * kSlice; i += 256) {
    const int u = i / kSlice, c = i % kSlice, o = o_base + c;
    W2L[u * SW + c] = (u < p.hidden && o < p.out) ? p.w2[(int64_t)u * p.out + o] : 0.0f;
  }
  float* cotW = cotL + wave * 16 * SC;
  // per tile and register: the output's bias, its state channel i and Brownian channel j (packed), whether it exists
  f32x4 bias[kOT];
  int ij[kOT][4];
#pragma unroll
  for (int tl = 0; tl < kOT; ++tl) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o_base + 16 * (kOT * wave + tl) + 4 * part + r;
      const bool have = o < p.out;
      const int i = have ? o / p.m : 0;
      ij[tl][r] = have ? (i << 8) | (o - i * p.m) : -1;
      bias[tl][r] = have ? p.b2[o] : 0.0f;
    }
  }
  f32x4 acc[kOT][TH], gbacc[kOT];
#pragma unroll
  for (int tl = 0; tl < kOT; ++tl) {
    gbacc[tl] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int th = 0; th < TH; ++th) acc[tl][th] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  const int64_t groups = (p.N + 15) / 16;
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int64_t row0 = grp * 16;
    __syncthreads();                      // (the previous group's tiles are no longer read; W2L is complete on the first pass)
    for (int i = threadIdx.x; i < 16 * (H / 4); i += 256) {
      const int rr = i / (H / 4), c4 = 4 * (i % (H / 4));
      f32x4 v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (row0 + rr < p.N && c4 < p.sh) v = *reinterpret_cast<const f32x4*>(p.hid + (row0 + rr) * p.sh + c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (c4 + r >= p.hidden) v[r] = 0.0f;          // (padded units of the stash hold act(0), not 0)
      }
      *reinterpret_cast<f32x4*>(hidL + rr * SH + c4) = v;
    }
    for (int i = threadIdx.x; i < 16 * p.sd; i += 256) {
      const int rr = i / p.sd, c = i % p.sd;
      const bool have = row0 + rr < p.N;
      pL[i] = have ? p.p[(row0 + rr) * p.sd + c] : 0.0f;
      qL[i] = have ? p.q[(row0 + rr) * p.sd + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < 16 * p.sm; i += 256) {
      const int rr = i / p.sm, c = i % p.sm;
      const bool have = row0 + rr < p.N;
      waL[i] = have ? p.wa[(row0 + rr) * p.sm + c] : 0.0f;
      wbL[i] = have ? p.wb[(row0 + rr) * p.sm + c] : 0.0f;
    }
    __syncthreads();
    // hid as the B operand of pre^T (lane (part, n): row n, units 16 th + 4 part + r) and as the A operand of hid^T cot
    // (lane (part, n): unit 16 th + n of row 4 kb + part)
    f32x4 hb[TH];
    float ha[4][TH];
#pragma unroll
    for (int th = 0; th < TH; ++th) {
      hb[th] = *reinterpret_cast<const f32x4*>(hidL + n * SH + 16 * th + 4 * part);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) ha[kb][th] = hidL[(4 * kb + part) * SH + 16 * th + n];
    }
#pragma unroll
    for (int tl = 0; tl < kOT; ++tl) {
      const int col = 16 * (kOT * wave + tl);
      if (o_base + col >= p.out) continue;               // (wave-uniform: a tile past the net's outputs)
      f32x4 pre = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = W2L[(16 * th + 4 * part + r) * SW + col + n];
          pre = Tile<16>::mfma(a, hb[th][r], pre);
        }
      }
      f32x4 cot;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int code = ij[tl][r];
        float c = 0.0f;
        if (code >= 0) {
          const int i = code >> 8, j = code & 255;
          const float x = pre[r] + bias[tl][r];
          const float s = p.final == TSDE_FINAL_SIGMOID ? final_slope<TSDE_FINAL_SIGMOID>(x)
                          : p.final == TSDE_FINAL_TANH ? final_slope<TSDE_FINAL_TANH>(x) : 1.0f;
          c = (pL[n * p.sd + i] * waL[n * p.sm + j] + qL[n * p.sd + i] * wbL[n * p.sm + j]) * s;
        }
        cot[r] = c;
        gbacc[tl][r] += c;
      }
      // rows become the K index: through a 16 x 16 transpose in this wave's own LDS tile
      *reinterpret_cast<f32x4*>(cotW + n * SC + 4 * part) = cot;
      __builtin_amdgcn_s_waitcnt(0xc07f);                // (lgkmcnt(0): the wave's own writes have landed)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float b = cotW[(4 * kb + part) * SC + n];
#pragma unroll
        for (int th = 0; th < TH; ++th) acc[tl][th] = Tile<16>::mfma(ha[kb][th], b, acc[tl][th]);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
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
    // column sums: over the 16 rows a lane quarter holds (lanes n = 0 .. 15)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = gbacc[tl][r];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      const int ob = o_base + 16 * (kOT * wave + tl) + 4 * part + r;
      if (n == 0 && ob < p.out) p.gb[(int64_t)blockIdx.x * p.out + ob] = s;
    }
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
  const size_t lds = ((size_t)H * (kSlice + 4) + 16 * (H + 4) + 32 * (size_t)stride_d + 32 * (size_t)stride_m + 4 * 16 * 20) *
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
