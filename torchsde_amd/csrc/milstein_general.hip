// Derivative-free Milstein for GENERAL noise (opt-in extension; SURVEY.md section 8 note N1 -- the reference's Milstein
// rejects general noise, torchsde/_core/methods/milstein.py:25).
//
// The scheme is the reference's own derivative-free idea (milstein.py:58-67: g' = g(y0 + dt*f + g*sqrt_dt),
// gdg = (g' - g) * v / (2*sqrt_dt)) applied per Brownian channel, i.e. the explicit order-1.0 scheme of Kloeden & Platen:
//     Y_k      = (y0 + dt*f) + g[:, :, k] * sqrt_dt          (Ito;  Stratonovich: y0 + g[:, :, k] * sqrt_dt)
//     corr_i   = ( sum_{k,l} (g_{i,l}(Y_k) - g_{i,l}(y0)) * I_{k,l} ) / sqrt_dt
//     y1       = ((y0 + f*dt) + g . W) + corr                 with I_{k,l} = (W_k W_l - delta_kl dt)/2 + A_{k,l}
// in place of the m Jacobian-vector products of the derivative form (base_sde.py:164-183). Two kernels:
//   support     builds all m supporting states as ONE (m*B, d) batch, so the user's g is called once on it;
//   correction  streams the (m, B, d, m) result of that call once -- the kernel is bound by exactly that stream --
//               one wave per batch row, the row's g in registers, I_{k,:} through 16-byte broadcast loads, and the
//               m-long sums finished by an xor-shuffle reduction (the layout of general_rows_kernel, steps.hip).
#include "tsde_common.h"
#include "tsde_launch.h"

namespace tsde {

template <typename T>
struct GfGeneralArgs {
  T* out;              // support: yk (m, B, d)      correction: corr (B, d)
  const T *y0, *f;     // support only
  const T* g;          // (B, d, m)
  const T* gk;         // correction: (m, B, d, m)
  const T* I;          // correction: (B, m, m)
  int64_t B, d, m;
  Coef<T> dt_, sqrt_dt_;
  int ito;
};

// One thread per (b, i): reads the m-long row g[b, i, :] with 16-byte loads (consecutive threads -> consecutive rows) and
// writes its m supporting values, one per slab k (consecutive threads -> consecutive addresses inside every slab).
template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock) gf_support_kernel(const GfGeneralArgs<T> a) {
  const T dt = a.dt_.get(), sqrt_dt = a.sqrt_dt_.get();
  const int64_t n = a.B * a.d;
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < n; o += (int64_t)gridDim.x * kBlock) {
    const T base = a.ito ? (a.y0[o] + dt * a.f[o]) : (a.y0[o] + (T)0);
    const T* grow = a.g + o * a.m;
    if constexpr (VEC) {
      for (int64_t k4 = 0; k4 < a.m; k4 += 4) {
        const Pack<T, 4> q = load<T, 4>(grow, k4);
#pragma unroll
        for (int j = 0; j < 4; ++j) a.out[(k4 + j) * n + o] = base + q.v[j] * sqrt_dt;
      }
    } else {
      for (int64_t k = 0; k < a.m; ++k) a.out[k * n + o] = base + grow[k] * sqrt_dt;
    }
  }
}

// One wave per batch row; NC 64-lane 16-byte loads cover the row's d*m entries (d*m/4 = NC*64). Lane L sits on channel
// quad L % G (G = m/4) of output (lane / G) of each chunk.
template <typename T, int NC>
__global__ void __launch_bounds__(kBlock) gf_correction_rows_kernel(const GfGeneralArgs<T> a) {
  const T sqrt_dt = a.sqrt_dt_.get();
  const int G = (int)(a.m >> 2);
  const int logG = __builtin_ctz(G);
  const int lane = threadIdx.x & 63;
  const int lp = lane & (G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  const int outs_per_chunk = 64 >> logG;
  const int64_t row_elems = (int64_t)NC * 64 * 4;          // d * m
  const int64_t slab = a.B * row_elems;                    // elements of one k-slab of gk
  for (int64_t row = wave; row < a.B; row += n_waves) {
    const T* grow = a.g + row * row_elems;
    const T* krow = a.gk + row * row_elems;
    const T* irow = a.I + row * a.m * a.m + (int64_t)lp * 4;
    Pack<T, 4> gq[NC];
    T acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      gq[c] = load<T, 4>(grow, (int64_t)(c * 64 + lane) * 4);
      acc[c] = (T)0;
    }
    for (int64_t k = 0; k < a.m; k += 2) {                  // two slabs in flight per iteration (m % 4 == 0)
      Pack<T, 4> q0[NC], q1[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        q0[c] = load<T, 4, true>(krow + k * slab, (int64_t)(c * 64 + lane) * 4);
        q1[c] = load<T, 4, true>(krow + (k + 1) * slab, (int64_t)(c * 64 + lane) * 4);
      }
      const Pack<T, 4> i0 = load<T, 4>(irow, k * a.m), i1 = load<T, 4>(irow, (k + 1) * a.m);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        acc[c] = acc[c] + ((((q0[c].v[0] - gq[c].v[0]) * i0.v[0] + (q0[c].v[1] - gq[c].v[1]) * i0.v[1]) +
                            (q0[c].v[2] - gq[c].v[2]) * i0.v[2]) + (q0[c].v[3] - gq[c].v[3]) * i0.v[3]);
        acc[c] = acc[c] + ((((q1[c].v[0] - gq[c].v[0]) * i1.v[0] + (q1[c].v[1] - gq[c].v[1]) * i1.v[1]) +
                            (q1[c].v[2] - gq[c].v[2]) * i1.v[2]) + (q1[c].v[3] - gq[c].v[3]) * i1.v[3]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      T part = acc[c];
      for (int off = 1; off < G; off <<= 1) part += __shfl_xor(part, off, 64);
      if (lp == 0) a.out[row * a.d + c * outs_per_chunk + (lane >> logG)] = part / sqrt_dt;
    }
  }
}

// Any d, m: one thread per output (b, i), scalar loads.
template <typename T>
__global__ void __launch_bounds__(kBlock) gf_correction_generic_kernel(const GfGeneralArgs<T> a) {
  const T sqrt_dt = a.sqrt_dt_.get();
  const int64_t n = a.B * a.d;
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < n; o += (int64_t)gridDim.x * kBlock) {
    const int64_t b = o / a.d;
    const T* grow = a.g + o * a.m;
    const T* irow = a.I + b * a.m * a.m;
    T acc = (T)0;
    for (int64_t k = 0; k < a.m; ++k) {
      const T* krow = a.gk + (k * n + o) * a.m;
      T part = (T)0;
      for (int64_t l = 0; l < a.m; ++l) part += (krow[l] - grow[l]) * irow[k * a.m + l];
      acc += part;
    }
    a.out[o] = acc / sqrt_dt;
  }
}

template <typename T>
hipError_t launch_milstein_gf_general_support(void* yk, const void* y0, const void* f, const void* g, int64_t B,
                                              int64_t d, int64_t m, double dt, double sqrt_dt, int ito, hipStream_t s) {
  if (B <= 0 || d <= 0 || m <= 0) return hipSuccess;
  GfGeneralArgs<T> a{(T*)yk, (const T*)y0, (const T*)f, (const T*)g, nullptr, nullptr, B, d, m, coef<T>(dt),
                     coef<T>(sqrt_dt), ito};
  const int grid = grid_for(B * d);
  if (m % 4 == 0 && aligned16(g))
    hipLaunchKernelGGL((gf_support_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s, a);
  else
    hipLaunchKernelGGL((gf_support_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_milstein_gf_general_correction(void* corr, const void* g, const void* gk, const void* I, int64_t B,
                                                 int64_t d, int64_t m, double sqrt_dt, hipStream_t s) {
  if (B <= 0 || d <= 0 || m <= 0) return hipSuccess;
  GfGeneralArgs<T> a{(T*)corr, nullptr, nullptr, (const T*)g, (const T*)gk, (const T*)I, B, d, m, coef<T>(0.0),
                     coef<T>(sqrt_dt), 0};
  const int64_t G = m / 4;
  const bool pow2 = (m % 4 == 0) && G >= 1 && G <= 64 && ((G & (G - 1)) == 0);
  const bool fast = pow2 && aligned16(g) && aligned16(gk) && aligned16(I) && (d * G) % 64 == 0;
  const int64_t nc = fast ? (d * G) / 64 : 0;
  if (fast && (nc == 1 || nc == 2 || nc == 4)) {
    int64_t blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);   // one wave per row; B/4 blocks, uncapped below 2^20
    if (blocks > (1 << 20)) blocks = 1 << 20;
    if (nc == 1) TSDE_LAUNCH((gf_correction_rows_kernel<T, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    else if (nc == 2) TSDE_LAUNCH((gf_correction_rows_kernel<T, 2>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    else TSDE_LAUNCH((gf_correction_rows_kernel<T, 4>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    return hipGetLastError();
  }
  TSDE_LAUNCH(gf_correction_generic_kernel<T>, dim3(grid_for(B * d)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

template hipError_t launch_milstein_gf_general_support<float>(void*, const void*, const void*, const void*, int64_t,
                                                              int64_t, int64_t, double, double, int, hipStream_t);
template hipError_t launch_milstein_gf_general_support<double>(void*, const void*, const void*, const void*, int64_t,
                                                               int64_t, int64_t, double, double, int, hipStream_t);
template hipError_t launch_milstein_gf_general_correction<float>(void*, const void*, const void*, const void*, int64_t,
                                                                 int64_t, int64_t, double, hipStream_t);
template hipError_t launch_milstein_gf_general_correction<double>(void*, const void*, const void*, const void*, int64_t,
                                                                  int64_t, int64_t, double, hipStream_t);

}  // namespace tsde
