// What one bridge-tree node visit of the Brownian query costs, piece by piece: per-wave cycles of
//   (a) Philox-4x32-10 alone, (b) + Box-Muller (4 normals), (c) + the W-only split arithmetic, (d) two streams (W and H)
// measured as wall time of a kernel in which every lane runs N dependent visits (no memory traffic but one store).
// The query kernel's floor is visits x (b..d); what it spends above that is bookkeeping (tsde_bridge.h).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/microbench_rng.hip -o tools/microbench_rng
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../torchsde_amd/csrc/tsde_rng.h"
using namespace tsde;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) visits(float* out, int n_visits, NoiseKey key) {
  const uint64_t quad = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  float W[4] = {0.1f, 0.2f, 0.3f, 0.4f}, H[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t acc = 0;
  uint64_t node = 1;
  for (int v = 0; v < n_visits; ++v) {
    if (MODE == 0) {
      const u32x4 r = noise_bits(key, quad, 7u, node, kStreamW);
      acc ^= r.x ^ r.y ^ r.z ^ r.w;
    } else {
      float x1[4], x2[4];
      normal4<float>(key, quad, 7u, node, kStreamW, x1);
      if (MODE == 3) normal4<float>(key, quad, 7u, node, kStreamH, x2);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) W[j] += x1[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float wl = (0.5f * W[j]) * 1.0f + 0.25f * x1[j];
          W[j] = (v & 1) ? wl : W[j] - wl;
          if (MODE == 3) H[j] = (0.25f * H[j] - 0.1f * x1[j]) + 0.07f * x2[j];
        }
      }
    }
    node = 2 * node + (v & 1);
  }
  out[quad] = W[0] + W[1] + W[2] + W[3] + H[0] + H[1] + H[2] + H[3] + (float)acc;
}

template <int MODE>
static void run(const char* name, float* out, int64_t quads, int n_visits, double clock_ghz) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const NoiseKey key{11u, 22u, 0};
  hipLaunchKernelGGL(visits<MODE>, dim3((unsigned)(quads / 256)), dim3(256), 0, 0, out, n_visits, key);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(visits<MODE>, dim3((unsigned)(quads / 256)), dim3(256), 0, 0, out, n_visits, key);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3;
  const double waves = quads / 64.0;
  const double cycles_per_visit = us * 1e-6 * clock_ghz * 1e9 * 1024.0 / (waves * n_visits);   // 1024 SIMDs
  printf("%-44s %8.1f us  %6.3f us per visit of 1M quads  ~%5.0f SIMD cycles per wave-visit\n", name, us,
         us / n_visits * (1048576.0 / quads), cycles_per_visit);
}

int main() {
  const int64_t quads = 1 << 20;   // 65536 x 64 elements / 4
  float* out;
  CK(hipMalloc(&out, quads * 4));
  int clock_khz = 0;
  CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
  const double ghz = clock_khz / 1e6;
  printf("device clock %.2f GHz, 1M quads (C2 size), 66 dependent visits per lane\n", ghz);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("Philox-4x32-10 only", out, quads, 66, ghz);
    run<1>("Philox + Box-Muller (4 normals)", out, quads, 66, ghz);
    run<2>("... + W-only split arithmetic", out, quads, 66, ghz);
    run<3>("two streams (W, H) + split arithmetic", out, quads, 66, ghz);
  }
  return 0;
}
