// Shared pieces of the perceptron-drift kernels (mlp_trajectory.hip: sampling; mlp_backward.hip: its gradient):
// the two f32 MFMA tile shapes, the activations on the hardware transcendentals, and the LDS footprint.
#pragma once
#include "tsde_common.h"

namespace tsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Activations on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each): the libm forms
// (tanhf, log1pf(expf)) expand to ~100 instructions with divergent special-case branches, which made the
// activation -- not the matrix products -- the longest part of a step. Absolute error <= 3e-7, well inside the
// tolerance at which this kernel is compared with the stepwise path (summation order already differs).
template <int ACT>
TSDE_D float activate(float x) {
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  if constexpr (ACT == TSDE_ACT_TANH) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); exp(2x) = 2^(2x log2 e); saturates cleanly to +-1
    const float e2x = __builtin_amdgcn_exp2f(x * (2.0f * kLog2e));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e2x + 1.0f);
  } else {
    // softplus, torch's threshold-20 form (aten/src/ATen/native/cuda/ActivationSoftplusKernel.cu): log(1 + e^x)
    const float ex = __builtin_amdgcn_exp2f(x * kLog2e);
    const float sp = __builtin_amdgcn_logf(1.0f + ex) * kLn2;
    return x > 20.0f ? x : sp;
  }
}

// The activation together with its derivative, from the same exponential (mlp_backward.hip).
template <int ACT>
TSDE_D void activate_with_slope(float x, float& value, float& slope) {
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  if constexpr (ACT == TSDE_ACT_TANH) {
    const float e2x = __builtin_amdgcn_exp2f(x * (2.0f * kLog2e));
    value = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e2x + 1.0f);
    slope = 1.0f - value * value;
  } else {
    // d/dx log(1 + e^x) = e^x / (1 + e^x) = 1 - 1 / (1 + e^x); torch's backward returns exactly 1 past the threshold
    const float ex = __builtin_amdgcn_exp2f(x * kLog2e);
    const float sp = __builtin_amdgcn_logf(1.0f + ex) * kLn2;
    value = x > 20.0f ? x : sp;
    slope = x > 20.0f ? 1.0f : 1.0f - __builtin_amdgcn_rcpf(1.0f + ex);
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The two f32 MFMA shapes. R = rows of the batch one wave owns = the tile edge.
//   R = 32: v_mfma_f32_32x32x2_f32, 16 accumulator registers per tile, two lane halves  (K = 2 per instruction)
//   R = 16: v_mfma_f32_16x16x4_f32,  4 accumulator registers per tile, four lane quarters (K = 4 per instruction)
// `part` = lane / R selects the K index a lane feeds and the rows of the accumulator it holds; for a fixed register
// the parts hold channels that differ by 4, which is the K grouping both operands are addressed with.
template <int R>
struct Tile;
template <>
struct Tile<32> {
  static constexpr int kRegs = 16, kQuads = 4;
  using acc_t = f32x16;
  TSDE_D static constexpr int row(int r, int part) { return (r & 3) + 8 * (r >> 2) + 4 * part; }
  TSDE_D static constexpr int quad_base(int q, int part) { return 8 * q + 4 * part; }   // channels of regs 4q..4q+3
  TSDE_D static acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};
template <>
struct Tile<16> {
  static constexpr int kRegs = 4, kQuads = 1;
  using acc_t = f32x4;
  TSDE_D static constexpr int row(int r, int part) { return 4 * part + r; }
  TSDE_D static constexpr int quad_base(int q, int part) { return 4 * part; }
  TSDE_D static acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// Four consecutive per-channel constants (biases, diffusion coefficients) from LDS, re-read at every use: the index is
// made opaque so that the loads are NOT hoisted out of the step loop -- hoisted, the constants of all tiles pin more
// than a hundred registers per lane for the whole solve (one ds_read_b128 per quad and step is noise next to the
// hundreds of operand reads of the matrix products).
TSDE_D f32x4 lds_quad(const float* base, int index) {
  asm volatile("" : "+v"(index));
  return *reinterpret_cast<const f32x4*>(base + index);
}

// Diagonal diffusion of one channel and its derivative w.r.t. the shift e (q: then dg/dc = q*y and dg/dy = q*c):
//   affine   g = c*y + e                      q = 1
//   sigmoid  g = amp * sigmoid(c*y + e)       q = amp * s * (1 - s)
// `sigmoid` is uniform over the launch (a scalar branch); the explicitly scheduled sampling kernel passes a constant.
struct DiffusionValue {
  float g, q;
};
TSDE_D DiffusionValue diffusion_value(bool sigmoid, float amp, float c, float e, float y) {
  const float u = c * y + e;
  DiffusionValue v = {u, 1.0f};
  if (sigmoid) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
    v.g = amp * s;
    v.q = v.g * (1.0f - s);
  }
  return v;
}

template <int R>
struct MlpLds {
  static constexpr int kPad = (R == 16) ? 4 : 0;
  static constexpr size_t bytes(int d, int h) { return (size_t)(d * (h + kPad) + h * (d + kPad) + h + 3 * d) * sizeof(float); }
};

}  // namespace tsde
