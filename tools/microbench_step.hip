// Micro-benchmark of variants of the diagonal Euler step kernel (build: hipcc, run on the GPU box).
// Prints average kernel time and algorithmic GB/s (4 streams x n x 4 B) per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../torchsde_amd/csrc/tsde_common.h"
using namespace tsde;
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool RNG, int UNROLL, bool NT>
__global__ void __launch_bounds__(256) step_variant(float* __restrict__ y1, const float* __restrict__ y0,
                                                    const float* __restrict__ f, const float* __restrict__ g,
                                                    int64_t nq, float dt, NoiseKey key, uint32_t cell, float sw) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t q0 = tid; q0 < nq; q0 += stride * UNROLL) {
    v4f a[UNROLL], b[UNROLL], c[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t q = q0 + u * stride;
      if (q < nq) {
        a[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(y0) + q) : reinterpret_cast<const v4f*>(y0)[q];
        b[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(f) + q) : reinterpret_cast<const v4f*>(f)[q];
        c[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g) + q) : reinterpret_cast<const v4f*>(g)[q];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t q = q0 + u * stride;
      if (q < nq) {
        float w[4] = {1.f, 1.f, 1.f, 1.f};
        if (RNG) {
          normal4<float>(key, (uint64_t)q, cell, 0, kStreamW, w);
          for (int j = 0; j < 4; ++j) w[j] *= sw;
        }
        v4f o;
        o.x = (a[u].x + b[u].x * dt) + c[u].x * w[0];
        o.y = (a[u].y + b[u].y * dt) + c[u].y * w[1];
        o.z = (a[u].z + b[u].z * dt) + c[u].z * w[2];
        o.w = (a[u].w + b[u].w * dt) + c[u].w * w[3];
        if (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(y1) + q);
        else reinterpret_cast<v4f*>(y1)[q] = o;
      }
    }
  }
}

struct Bufs { float *y[2], *f, *g; };

template <bool RNG, int UNROLL, bool NT>
int run(const char* name, int grid, const Bufs& b, int64_t n, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  NoiseKey key{1u, 2u, 0};
  const int64_t nq = n / 4;
  for (int i = 0; i < 10; ++i)
    hipLaunchKernelGGL((step_variant<RNG, UNROLL, NT>), dim3(grid), dim3(256), 0, 0, b.y[(i + 1) & 1], b.y[i & 1], b.f, b.g, nq, 1e-3f, key, (uint32_t)i, 0.03f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((step_variant<RNG, UNROLL, NT>), dim3(grid), dim3(256), 0, 0, b.y[(i + 1) & 1], b.y[i & 1], b.f, b.g, nq, 1e-3f, key, (uint32_t)i, 0.03f);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("%-44s grid=%6d  %7.2f us  %7.1f GB/s\n", name, grid, us, 16.0 * n / us / 1e3);
  return 0;
}

int main() {
  const int64_t n = (getenv("TSDE_ROWS") ? atoll(getenv("TSDE_ROWS")) : 65536LL) * 64;
  Bufs b;
  for (int i = 0; i < 2; ++i) CK(hipMalloc(&b.y[i], n * 4));
  CK(hipMalloc(&b.f, n * 4)); CK(hipMalloc(&b.g, n * 4));
  const bool random_data = (getenv("TSDE_RANDOM") != nullptr);
  if (random_data) {
    std::vector<float> h(n);
    for (int k = 0; k < 4; ++k) {
      for (int64_t i = 0; i < n; ++i) h[i] = (k == 2 ? -0.5f : 0.05f) + 0.3f * (float)rand() / RAND_MAX;
      CK(hipMemcpy(k == 0 ? b.y[0] : k == 1 ? b.y[1] : k == 2 ? b.f : b.g, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    printf("random (live-like) data\n");
  } else {
    CK(hipMemset(b.y[0], 0, n * 4)); CK(hipMemset(b.y[1], 0, n * 4)); CK(hipMemset(b.f, 0, n * 4)); CK(hipMemset(b.g, 0, n * 4));
    printf("zero-filled data\n");
  }
  const int64_t nq = n / 4;
  const int full = (int)((nq + 255) / 256);
  const int it = n > (1LL << 26) ? 40 : 300;
  for (int rep = 0; rep < 2; ++rep) {
    run<false, 1, false>("copy-like (no RNG) u1", 2048, b, n, it);
    run<false, 1, false>("copy-like (no RNG) u1", full, b, n, it);
    run<false, 2, false>("copy-like (no RNG) u2", 2048, b, n, it);
    run<false, 4, false>("copy-like (no RNG) u4", 1024, b, n, it);
    run<true, 1, false>("rng u1", 2048, b, n, it);
    run<true, 1, false>("rng u1", full, b, n, it);
    run<true, 1, false>("rng u1", 1024, b, n, it);
    run<true, 2, false>("rng u2", 2048, b, n, it);
    run<true, 2, false>("rng u2", 1024, b, n, it);
    run<true, 4, false>("rng u4", 1024, b, n, it);
    run<true, 4, false>("rng u4", 512, b, n, it);
    run<true, 1, true>("rng u1 nontemporal", 2048, b, n, it);
    run<true, 2, true>("rng u2 nontemporal", 2048, b, n, it);
    run<false, 2, true>("copy-like u2 nontemporal", 2048, b, n, it);
    printf("--\n");
  }
  return 0;
}
