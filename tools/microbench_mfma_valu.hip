// Do v_mfma_f32_16x16x4_f32 and ordinary vector instructions of the SAME SIMD overlap in time on gfx950?
// One block per CU-slot, W waves per SIMD; each wave runs a loop of `M` independent-accumulator MFMAs and `V` independent
// v_fma_f32 per iteration. Reported: cycles per iteration per wave for (M, 0), (0, V), (M, V) -- if (M, V) costs
// max(MFMA, VALU) the pipes overlap, if it costs the sum they do not. Build:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_mfma_valu.hip -o tools/microbench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int M, int V, bool INTERLEAVE>
__global__ void __launch_bounds__(512) k(float* out, int iters, float seed) {
  f32x4 acc[8];
  float v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = {seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
  for (int it = 0; it < iters; ++it) {
    if (INTERLEAVE) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (M) {
#pragma unroll
          for (int j = 0; j < M / 8; ++j) acc[(g + j) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(g + j) & 7], 0, 0, 0);
        }
        if (V) {
#pragma unroll
          for (int j = 0; j < V / 8; ++j) v[(g * (V / 8) + j) & 15] = __builtin_fmaf(v[(g * (V / 8) + j) & 15], a, b);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < M; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 7], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], a, b);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M, int V, bool IL>
int run(const char* name, int waves_per_simd, float* out, double ghz) {
  const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;   // 4 SIMDs per CU
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<M, V, IL>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k<M, V, IL>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double cyc = ms * 1e-3 * ghz * 1e9 / iters;      // cycles per iteration (per SIMD: all its waves together)
  printf("%-44s waves/SIMD=%d  M=%3d V=%3d  %8.1f SIMD-cycles per iteration  (MFMA alone would be %5d, VALU alone %5d per wave)\n",
         name, waves_per_simd, M, V, cyc, 32 * M * waves_per_simd, 4 * V * waves_per_simd);
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 512 * 4));
  int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz / 1e6;
  printf("clock %.2f GHz (nominal; measured cycles scale with the actual clock)\n", ghz);
  for (int w = 1; w <= 2; ++w) {
    run<32, 0, false>("MFMA only", w, out, ghz);
    run<0, 128, false>("VALU only (v_fma_f32)", w, out, ghz);
    run<32, 128, false>("MFMA block then VALU block", w, out, ghz);
    run<32, 128, true>("interleaved: 4 MFMA, 16 VALU, ...", w, out, ghz);
    run<32, 64, true>("interleaved: 4 MFMA, 8 VALU, ...", w, out, ghz);
    run<32, 256, true>("interleaved: 4 MFMA, 32 VALU, ...", w, out, ghz);
    printf("--\n");
  }
  return 0;
}
