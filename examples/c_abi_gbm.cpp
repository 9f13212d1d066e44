// The drop-in boundary without Python or torch: geometric Brownian motion  dy = mu y dt + sigma y dW  solved through the
// C ABI of libtorchsde_amd.so alone (include/torchsde_amd.h), twice --
//   1. step by step, the way a host framework drives it: f = mu*y and g = sigma*y as two "user" launches
//      (tsde_lincomb2 stands in for the framework's elementwise ops), then tsde_step_diag with the Brownian increment
//      of the step generated in registers (euler.py:29-37 of the reference);
//   2. as ONE launch of tsde_trajectory_affine_diag on the same Brownian path;
// checks that the two agree bit for bit, and the sample mean against E[y_T] = y0 exp(mu T).
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude examples/c_abi_gbm.cpp -Ltorchsde_amd/csrc -ltorchsde_amd \
//         -Wl,-rpath,$PWD/torchsde_amd/csrc -o /tmp/c_abi_gbm && /tmp/c_abi_gbm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "torchsde_amd.h"

#define HIP_OK(call)                                                                     \
  do {                                                                                   \
    const hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                    \
      std::exit(1);                                                                      \
    }                                                                                    \
  } while (0)
#define TSDE_OK(call)                                                                    \
  do {                                                                                   \
    if ((call) != 0) {                                                                   \
      std::fprintf(stderr, "%s: %s\n", #call, tsde_last_error());                        \
      std::exit(1);                                                                      \
    }                                                                                    \
  } while (0)

template <typename T>
static T* to_device(const std::vector<T>& host) {
  T* dev = nullptr;
  HIP_OK(hipMalloc(&dev, host.size() * sizeof(T)));
  HIP_OK(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return dev;
}

int main() {
  if (tsde_abi_version() != TSDE_ABI_VERSION) {
    std::fprintf(stderr, "ABI version mismatch\n");
    return 1;
  }
  const int64_t rows = 16384, d = 64, n = rows * d;
  const int n_steps = 256;
  const double dt = 1.0 / 256.0;            // dyadic: dt, sqrt(dt) are exact in float32
  const float mu = 0.3f, sigma = 0.4f, y_start = 1.0f;
  const uint64_t entropy = 20240601u, elem0 = 0;
  hipStream_t stream = nullptr;
  HIP_OK(hipStreamCreate(&stream));

  // ---- 1. the stepwise solve -----------------------------------------------------------------------------------------
  float *y[2], *f, *g;
  const std::vector<float> y0(n, y_start);
  y[0] = to_device(y0);
  HIP_OK(hipMalloc(&y[1], n * sizeof(float)));
  HIP_OK(hipMalloc(&f, n * sizeof(float)));
  HIP_OK(hipMalloc(&g, n * sizeof(float)));
  for (int k = 0; k < n_steps; ++k) {
    float* cur = y[k & 1];
    TSDE_OK(tsde_lincomb2(f, cur, cur, n, mu, 0.0, TSDE_F32, stream));          // "user code": f = mu * y
    TSDE_OK(tsde_lincomb2(g, cur, cur, n, sigma, 0.0, TSDE_F32, stream));       // "user code": g = sigma * y
    tsde_noise_t noise;
    std::memset(&noise, 0, sizeof noise);
    noise.entropy = entropy;
    noise.elem0 = elem0;
    noise.cell = (uint32_t)k;                                                    // one Brownian cell per step
    noise.h = dt;
    TSDE_OK(tsde_step_diag(y[(k + 1) & 1], cur, f, g, n, dt, 1.0, &noise, TSDE_F32, stream));
  }
  std::vector<float> stepwise(n);
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(stepwise.data(), y[n_steps & 1], n * sizeof(float), hipMemcpyDeviceToHost));

  // ---- 2. the same solve as one launch ------------------------------------------------------------------------------------
  std::vector<float> step_rows((size_t)n_steps * 8, 0.0f);
  std::vector<uint32_t> cells(n_steps);
  for (int k = 0; k < n_steps; ++k) {
    float* row = &step_rows[(size_t)k * 8];
    row[0] = (float)dt;
    row[1] = (float)(0.5 * dt);
    row[2] = (float)(1.0 / dt);
    row[3] = (float)std::sqrt(dt);
    row[4] = (float)std::sqrt(dt);            // sqrt(h): the cell is the step
    row[5] = (float)std::sqrt(dt / 12.0);
    row[6] = (float)dt;
    cells[k] = (uint32_t)k;
  }
  const std::vector<int32_t> out_step = {n_steps};
  const std::vector<float> out_w = {0.0f, 1.0f};
  tsde_traj_t traj;
  traj.step_rows = to_device(step_rows);
  traj.cells = to_device(cells);
  traj.out_step = to_device(out_step);
  traj.out_w = to_device(out_w);
  traj.n_steps = n_steps;
  traj.n_out = 1;
  float* coef[4];
  const float values[4] = {mu, 0.0f, sigma, 0.0f};      // drift rate, drift shift, diffusion rate, diffusion shift
  for (int i = 0; i < 4; ++i) coef[i] = to_device(std::vector<float>(d, values[i]));
  float *y_in = to_device(y0), *ys = nullptr;
  HIP_OK(hipMalloc(&ys, n * sizeof(float)));
  TSDE_OK(tsde_trajectory_affine_diag(ys, y_in, rows, d, coef[0], coef[1], coef[2], coef[3], TSDE_TRAJ_EULER, &traj,
                                      entropy, elem0, nullptr, TSDE_F32, stream));
  std::vector<float> one_launch(n);
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(one_launch.data(), ys, n * sizeof(float), hipMemcpyDeviceToHost));

  // ---- checks ----------------------------------------------------------------------------------------------------------
  const bool identical = std::memcmp(stepwise.data(), one_launch.data(), n * sizeof(float)) == 0;
  double mean = 0.0;
  for (float v : stepwise) mean += v;
  mean /= (double)n;
  const double T = n_steps * dt, expected = y_start * std::exp(mu * T);
  // Monte Carlo error of the mean: sd(y_T) / sqrt(n) with sd(y_T) = E[y_T] sqrt(exp(sigma^2 T) - 1); Euler's weak
  // error at this dt is far below it
  const double tolerance = 5.0 * expected * std::sqrt(std::exp(sigma * sigma * T) - 1.0) / std::sqrt((double)n);
  std::printf("stepwise == one launch: %s\nmean %.6f, E[y_T] %.6f, |diff| %.2e (tolerance %.2e)\n",
              identical ? "bit-identical" : "DIFFERENT", mean, expected, std::fabs(mean - expected), tolerance);
  return identical && std::fabs(mean - expected) < tolerance ? 0 : 1;
}
