// Error norm of adaptive step doubling as ONE fused, deterministic reduction
// (replaces the ~8 elementwise/reduction torch kernels of torchsde/_core/adaptive_stepping.py:42-76
//  `compute_error`, called from base_solver.py:125-128):
//
//   tol_i   = max(eps, rtol * max(|yf_i|, |yh_i|) + atol)
//   err     = max(eps, sqrt( sum_i ((yf_i - yh_i) / tol_i)^2 / n ))
//
// The per-element terms are evaluated in the state dtype with the reference's operation order; the sum is
// accumulated in double with a FIXED tree (per-lane strided partials -> wave shuffle -> block -> one partial per
// block -> a single-block second pass), so the value -- and therefore the accept/reject decisions of a solve -- is
// reproducible run to run (no atomics).
#include "tsde_common.h"
#include "tsde_launch.h"

namespace tsde {

constexpr int kMaxPartials = 1024;

TSDE_D double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over the block, valid in thread 0.
TSDE_D double block_sum(double v, double* lds) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double total = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) total += lds[w];
  }
  return total;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) error_partials_kernel(double* __restrict__ partials,
                                                                const T* __restrict__ yf, const T* __restrict__ yh,
                                                                int64_t n, T rtol, T atol, T eps, int vec) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  auto term = [&](T a, T b) {
    const T fa = a < (T)0 ? -a : a, fb = b < (T)0 ? -b : b;
    T tol = rtol * (fa > fb ? fa : fb) + atol;
    tol = tol < eps ? eps : tol;
    const T ratio = (a - b) / tol;
    return (double)(ratio * ratio);
  };
  if (vec) {
    const int64_t nq = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nq; q += (int64_t)gridDim.x * kBlock) {
      const Pack<T, 4> a = load<T, 4>(yf, q << 2), b = load<T, 4>(yh, q << 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += term(a.v[j], b.v[j]);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
      acc += term(yf[i], yh[i]);
  }
  const double total = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) error_final_kernel(double* __restrict__ out,
                                                             const double* __restrict__ partials, int count,
                                                             double n, double eps) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partials[i];
  const double total = block_sum(acc, lds);
  if (threadIdx.x == 0) {
    const double err = sqrt(total / n);
    out[0] = (err < eps) ? eps : err;   // NaN stays NaN: the host checks it like the reference's assert
  }
}

template <typename T>
hipError_t launch_error_norm(double* out, double* workspace, const void* yf, const void* yh, int64_t n, double rtol,
                             double atol, double eps, hipStream_t s) {
  const bool vec = (n % 4 == 0) && aligned16(yf) && aligned16(yh);
  int64_t blocks = ((vec ? n >> 2 : n) + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxPartials) blocks = kMaxPartials;
  hipLaunchKernelGGL(error_partials_kernel<T>, dim3((unsigned)blocks), dim3(kBlock), 0, s, workspace, (const T*)yf,
                     (const T*)yh, n, (T)rtol, (T)atol, (T)eps, vec ? 1 : 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(error_final_kernel, dim3(1), dim3(kBlock), 0, s, out, (const double*)workspace, (int)blocks,
                     (double)n, (double)(T)eps);
  return hipGetLastError();
}

template hipError_t launch_error_norm<float>(double*, double*, const void*, const void*, int64_t, double, double,
                                             double, hipStream_t);
template hipError_t launch_error_norm<double>(double*, double*, const void*, const void*, int64_t, double, double,
                                              double, hipStream_t);

}  // namespace tsde
