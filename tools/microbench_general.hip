// C3 contraction y1 = (y0 + f dt) + sum_j g[b,i,j] dW[b,j] at B=16384, d=32, m=16: the register-only design the
// library ships (one wave per row, increments regenerated per lane, xor-shuffle reduce; csrc/steps.hip
// general_rows_kernel) against the LDS-staged design north_star proposed (the tile's increments generated ONCE per
// block into LDS, barrier, every lane reads its 4 weights from LDS, barrier before the next tile).
// Both variants produce the same numbers (checked). Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/microbench_general.hip -o tools/microbench_general
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../torchsde_amd/csrc/tsde_common.h"
using namespace tsde;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int D = 32, M = 16, G = M / 4;          // G = lanes that share one output (4)
constexpr int ROW_Q = D * G;                      // 16-byte groups of g per batch row (128 = two wave loads)

// ---- (a) registers only: the shipped design -----------------------------------------------------------------
__global__ void __launch_bounds__(256) contraction_registers(float* __restrict__ y1, const float* __restrict__ y0,
                                                             const float* __restrict__ f, const float* __restrict__ g,
                                                             int64_t B, float dt, float sw, NoiseKey key, uint32_t cell) {
  const int lane = threadIdx.x & 63, lp = lane & (G - 1);
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t row = wave; row < B; row += n_waves) {
    const float* grow = g + row * (int64_t)(ROW_Q * 4);
    Pack<float, 4> gq[2];
    float y0v[2], fv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      gq[c] = load<float, 4>(grow, (int64_t)(c * 64 + lane) * 4);
      const int64_t o = row * D + c * 16 + (lane >> 2);
      if (lp == 0) { y0v[c] = y0[o]; fv[c] = f[o]; }
    }
    float n[4];
    normal4<float>(key, (uint64_t)(row * M + lp * 4) >> 2, cell, 0, kStreamW, n);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float part = ((gq[c].v[0] * (n[0] * sw) + gq[c].v[1] * (n[1] * sw)) + gq[c].v[2] * (n[2] * sw)) + gq[c].v[3] * (n[3] * sw);
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      if (lp == 0) y1[row * D + c * 16 + (lane >> 2)] = (y0v[c] + fv[c] * dt) + part;
    }
  }
}

// ---- (b) LDS-staged: a block owns TILE rows per iteration ---------------------------------------------------------
template <int TILE>
__global__ void __launch_bounds__(256) contraction_lds(float* __restrict__ y1, const float* __restrict__ y0,
                                                       const float* __restrict__ f, const float* __restrict__ g, int64_t B,
                                                       float dt, float sw, NoiseKey key, uint32_t cell) {
  __shared__ float w[TILE * M];
  const int64_t n_tiles = (B + TILE - 1) / TILE;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TILE;
    // the tile's TILE*M increments, one Philox call per 4 of them, generated ONCE (not once per consuming lane)
    for (int q = threadIdx.x; q < TILE * G; q += 256) {
      float n[4];
      normal4<float>(key, (uint64_t)(row0 * M + q * 4) >> 2, cell, 0, kStreamW, n);
#pragma unroll
      for (int j = 0; j < 4; ++j) w[q * 4 + j] = n[j] * sw;
    }
    __syncthreads();
    // TILE*ROW_Q 16-byte groups of g, coalesced; group v of the tile -> row v / ROW_Q, channel quad v % G
    for (int v = threadIdx.x; v < TILE * ROW_Q; v += 256) {
      const int r = v / ROW_Q, rem = v - r * ROW_Q, lp = rem & (G - 1);
      const int64_t row = row0 + r;
      float part = 0.f;
      if (row < B) {
        const Pack<float, 4> gq = load<float, 4>(g, (row * ROW_Q + rem) * 4);
        const float* wr = w + r * M + lp * 4;
        part = ((gq.v[0] * wr[0] + gq.v[1] * wr[1]) + gq.v[2] * wr[2]) + gq.v[3] * wr[3];
      }
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      if (row < B && lp == 0) {
        const int64_t o = row * D + (rem >> 2);
        y1[o] = (y0[o] + f[o] * dt) + part;
      }
    }
    __syncthreads();
  }
}

template <typename Launch>
int timeit(const char* name, Launch launch, int64_t B, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, bytes = 4.0 * B * (D * M + 3 * D);
  printf("%-52s %7.2f us  %7.1f GB/s  (%4.1f %% of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
  return 0;
}

int main() {
  const int64_t B = getenv("TSDE_ROWS") ? atoll(getenv("TSDE_ROWS")) : 16384;
  float *y[3], *f, *g;
  for (int i = 0; i < 3; ++i) CK(hipMalloc(&y[i], B * D * 4));
  CK(hipMalloc(&f, B * D * 4)); CK(hipMalloc(&g, B * D * M * 4));
  {
    std::vector<float> h(B * D * M);
    for (auto& v : h) v = 0.05f + 0.3f * (float)rand() / RAND_MAX;
    CK(hipMemcpy(g, h.data(), B * D * M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(y[0], h.data(), B * D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f, h.data() + B * D, B * D * 4, hipMemcpyHostToDevice));
  }
  const NoiseKey key{123u, 456u, 0};
  const float dt = 1.0f / 1024, sw = sqrtf(dt);
  const int grid = 2048;
  printf("C3 contraction, B=%lld d=%d m=%d, random data, %d back-to-back launches per variant (algorithmic bytes: g + y0 + f + y1)\n",
         (long long)B, D, M, 300);
  // same numbers from both designs
  hipLaunchKernelGGL(contraction_registers, dim3(grid), dim3(256), 0, 0, y[1], y[0], f, g, B, dt, sw, key, 7u);
  hipLaunchKernelGGL((contraction_lds<8>), dim3(grid), dim3(256), 0, 0, y[2], y[0], f, g, B, dt, sw, key, 7u);
  CK(hipDeviceSynchronize());
  std::vector<float> a(B * D), b(B * D);
  CK(hipMemcpy(a.data(), y[1], B * D * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), y[2], B * D * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int64_t i = 0; i < B * D; ++i) worst = fmax(worst, fabs((double)a[i] - b[i]));
  printf("max |registers - lds| = %.3g\n", worst);
  for (int rep = 0; rep < 2; ++rep) {
    timeit("registers only (one wave per row; shipped)", [&](int i) {
      hipLaunchKernelGGL(contraction_registers, dim3(grid), dim3(256), 0, 0, y[1], y[0], f, g, B, dt, sw, key, (uint32_t)i); }, B, 300);
    timeit("LDS-staged increments, 8 rows per tile", [&](int i) {
      hipLaunchKernelGGL((contraction_lds<8>), dim3(grid), dim3(256), 0, 0, y[2], y[0], f, g, B, dt, sw, key, (uint32_t)i); }, B, 300);
    timeit("LDS-staged increments, 16 rows per tile", [&](int i) {
      hipLaunchKernelGGL((contraction_lds<16>), dim3(grid), dim3(256), 0, 0, y[2], y[0], f, g, B, dt, sw, key, (uint32_t)i); }, B, 300);
    timeit("LDS-staged increments, 32 rows per tile", [&](int i) {
      hipLaunchKernelGGL((contraction_lds<32>), dim3(1024), dim3(256), 0, 0, y[2], y[0], f, g, B, dt, sw, key, (uint32_t)i); }, B, 300);
    printf("--\n");
  }
  return worst < 1e-6 ? 0 : 2;
}
