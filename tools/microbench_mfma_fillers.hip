// Which vector instructions hide in the shadow of v_mfma_f32_16x16x4_f32 on gfx950, one wave per SIMD?
// A loop of 32 MFMAs (8 rotating accumulators) with PER fillers of one KIND after every MFMA; every instruction is volatile
// inline assembly, so neither the order nor the instruction selection is the compiler's (tools/microbench_mfma_valu.hip left
// both to hipcc, which packs neighbouring v_fma_f32 into v_pk_fma_f32). Reported: SIMD cycles per MFMA for PER = 0..6 --
// a filler that hides leaves the figure at ~32 until the gap is full; one that does not adds its own issue time each.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_mfma_fillers.hip -o tools/microbench_mfma_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { FMA = 0, EXP, RCP, PKFMA, MULLO, ADDU, MOV, CNDMASK, DSREAD, MIXED, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_pk_fma_f32", "v_mul_lo_u32", "v_add_u32", "v_mov_b32",
                                    "v_cndmask_b32", "ds_read_b32", "exp,add,rcp,fma"};

template <int KIND>
__device__ __forceinline__ void filler(float& x, float& y, f32x2& xx, uint32_t& u, int j, const float* lds_p) {
  if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if constexpr (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
  if constexpr (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(xx));
  if constexpr (KIND == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(0x9E3779B9u));
  if constexpr (KIND == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(0x9E3779B9u));
  if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : );
  if constexpr (KIND == DSREAD) asm volatile("ds_read_b32 %0, %1" : "+v"(x) : "v"((uint32_t)(threadIdx.x * 4)) : "memory");
  if constexpr (KIND == MIXED) {
    switch (j & 3) {
      case 0: asm volatile("v_exp_f32 %0, %0" : "+v"(x)); break;
      case 1: asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x)); break;
      case 2: asm volatile("v_rcp_f32 %0, %0" : "+v"(x)); break;
      default: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y)); break;
    }
  }
}

template <int KIND, int PER, int MFMAS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  __shared__ float lds[512];
  lds[threadIdx.x] = seed;
  lds[threadIdx.x + 256] = seed;
  __syncthreads();
  f32x4 acc[8];
  float x[8], y = seed + 0.5f;
  f32x2 xx = {seed, seed};
  uint32_t u = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = {seed, seed, seed, seed}; x[i] = seed + i; }
  const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if constexpr (MFMAS) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < PER; ++f) filler<KIND>(x[(j * PER + f) & 7], y, xx, u, j * PER + f, lds);
    }
    if constexpr (KIND == DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = xx[0] + xx[1] + (float)u;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int PER, int MFMAS>
double run(float* out, double ghz) {
  const int iters = 2000, blocks = 256, threads = 256;      // one wave per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, PER, MFMAS>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<KIND, PER, MFMAS>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * ghz * 1e9 / iters / 32;      // cycles per slot (one MFMA + PER fillers)
}

template <int KIND>
void row(float* out, double ghz) {
  printf("%-16s with MFMA:", kNames[KIND]);
  printf(" %6.1f", run<KIND, 0, 1>(out, ghz));
  printf(" %6.1f", run<KIND, 1, 1>(out, ghz));
  printf(" %6.1f", run<KIND, 2, 1>(out, ghz));
  printf(" %6.1f", run<KIND, 3, 1>(out, ghz));
  printf(" %6.1f", run<KIND, 4, 1>(out, ghz));
  printf(" %6.1f", run<KIND, 6, 1>(out, ghz));
  printf(" %6.1f", run<KIND, 8, 1>(out, ghz));
  printf("   alone (no MFMA):");
  printf(" %6.1f", run<KIND, 1, 0>(out, ghz));
  printf(" %6.1f", run<KIND, 4, 0>(out, ghz));
  printf(" %6.1f", run<KIND, 8, 0>(out, ghz));
  printf("\n");
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 256 * 4));
  int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz / 1e6;
  printf("clock %.2f GHz nominal; SIMD cycles per slot = one v_mfma_f32_16x16x4_f32 + PER fillers, one wave per SIMD\n", ghz);
  printf("%-16s           PER =      0      1      2      3      4      6      8                     PER =      1      4      8\n", "filler");
  row<FMA>(out, ghz);
  row<EXP>(out, ghz);
  row<RCP>(out, ghz);
  row<PKFMA>(out, ghz);
  row<MULLO>(out, ghz);
  row<ADDU>(out, ghz);
  row<MOV>(out, ghz);
  row<CNDMASK>(out, ghz);
  row<DSREAD>(out, ghz);
  row<MIXED>(out, ghz);
  return 0;
}
