// Issue cost of the VALU instructions the whole-trajectory kernels are made of (Philox-4x32-10 + Box-Muller + the scheme's
// arithmetic), measured on the device the way the roofline of those kernels needs it: SIMD cycles per wave64 instruction
// when the SIMD has nothing else to do but issue that instruction from 8 resident waves (8 independent chains per wave, so
// neither dependent-issue latency nor the other pipes bound it).
//
// Two clocks per row: (a) the shader clock itself (s_memtime ticks between a wave's first and last instruction, median
// over waves, divided by the instructions its SIMD issued in that window = 8 waves x N) -- independent of DVFS -- and
// (b) wall time at the nominal 2.4 GHz. bench.py's `roofline` (bound "valu") multiplies the per-instruction cycles (a) by
// the instruction histogram of the kernel's step loop (tools/valu_model.py).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_valu.hip -o tools/microbench_valu
//   run:   tools/microbench_valu [out.json]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kChains = 8;      // independent dependency chains per wave
constexpr int kUnroll = 8;      // instructions per chain and loop iteration
constexpr int kIters = 2000;
constexpr int kWavesPerSimd = 8;
constexpr int kSimds = 1024;

// One instruction per chain: X(i) expands to the asm of chain i.
#define EIGHT(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__device__ __forceinline__ void body(float (&a)[kChains], float (&b)[kChains], uint32_t (&u)[kChains], uint32_t (&w)[kChains],
                                     uint64_t (&q)[kChains], uint64_t (&r)[kChains], uint32_t s0, uint32_t s1) {
  const uint64_t mask = 0x5555555555555555ull ^ (uint64_t)s0;      // a wave-uniform lane mask in an SGPR pair
#define V_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define V_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define V_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define V_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(r[i]));
#define V_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(r[i]));
#define V_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q[i]) : "v"(r[i]));
#define V_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(u[i]), "s"(s0) : "vcc"); \
                   asm volatile("" : "+v"(u[i]) : "v"(q[i]));
#define V_MAD64D(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\tv_mov_b32 %1, %0" : "=&v"(q[i]), "+v"(u[i]) : "s"(s0) : "vcc");
#define V_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "s"(s0));
#define V_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "s"(s0));
#define V_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(w[i]), "s"(s1));
#define V_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define V_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define V_NOT(i) asm volatile("v_not_b32 %0, %0" : "+v"(u[i]));
#define V_CNDMASK(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(w[i]), "s"(mask));
#define V_BITOP3V(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
#define V_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(w[i]));
#define V_MAD64V(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(u[i]), "v"(w[i]) : "vcc");
#define V_XORS(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "s"(s1));
#define V_CNDVCC(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(w[i]));
#define V_MULS(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
#define V_PKMULS(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(q[i]) : "v"(r[i]));
#define V_XOR2(i) asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %2, %0" : "+v"(u[i]) : "v"(w[i]), "s"(s1));
#define V_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(w[i]) : "vcc");
#define V_CVT(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define V_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define V_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define V_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define V_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define V_SIN(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
#define V_COS(i) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
#define V_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(r[i]));
#define V_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(q[i]) : "v"(r[i]));
#define V_FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a[i]) : "v"(b[i]));
  if constexpr (OP == 0) { EIGHT(V_FMA) }
  if constexpr (OP == 1) { EIGHT(V_MUL) }
  if constexpr (OP == 2) { EIGHT(V_ADD) }
  if constexpr (OP == 3) { EIGHT(V_PKMUL) }
  if constexpr (OP == 4) { EIGHT(V_PKADD) }
  if constexpr (OP == 5) { EIGHT(V_PKFMA) }
  if constexpr (OP == 6) { EIGHT(V_MAD64) }
  if constexpr (OP == 7) { EIGHT(V_MULLO) }
  if constexpr (OP == 8) { EIGHT(V_MULHI) }
  if constexpr (OP == 9) { EIGHT(V_BITOP3) }
  if constexpr (OP == 10) { EIGHT(V_XOR) }
  if constexpr (OP == 11) { EIGHT(V_ADDU) }
  if constexpr (OP == 12) { EIGHT(V_NOT) }
  if constexpr (OP == 13) { EIGHT(V_CNDMASK) }
  if constexpr (OP == 14) { EIGHT(V_CMP) }
  if constexpr (OP == 15) { EIGHT(V_CVT) }
  if constexpr (OP == 16) { EIGHT(V_LOG) }
  if constexpr (OP == 17) { EIGHT(V_EXP) }
  if constexpr (OP == 18) { EIGHT(V_RCP) }
  if constexpr (OP == 19) { EIGHT(V_SQRT) }
  if constexpr (OP == 20) { EIGHT(V_SIN) }
  if constexpr (OP == 21) { EIGHT(V_COS) }
  if constexpr (OP == 22) { EIGHT(V_LSHLADD64) }
  if constexpr (OP == 23) { EIGHT(V_MOV64) }
  if constexpr (OP == 24) { EIGHT(V_FMAMK) }
  if constexpr (OP == 25) { EIGHT(V_BITOP3V) }
  if constexpr (OP == 26) { EIGHT(V_MOV) }
  if constexpr (OP == 27) { EIGHT(V_MAD64V) }
  if constexpr (OP == 28) { EIGHT(V_XORS) }
  if constexpr (OP == 29) { EIGHT(V_CNDVCC) }
  if constexpr (OP == 30) { EIGHT(V_MULS) }
  if constexpr (OP == 31) { EIGHT(V_PKMULS) }
  if constexpr (OP == 32) { EIGHT(V_XOR2) }
}

static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
                               "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_bitop3_b32", "v_xor_b32", "v_add_u32",
                               "v_not_b32", "v_cndmask_b32", "v_cmp_lt_u32", "v_cvt_f32_u32", "v_log_f32", "v_exp_f32",
                               "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_lshl_add_u64", "v_mov_b64",
                               "v_fmamk_f32", "v_bitop3_b32(vgpr)", "v_mov_b32", "v_mad_u64_u32(vgpr)", "v_xor_b32(sgpr)",
                               "v_cndmask_b32(vcc)", "v_mul_f32(sgpr)", "v_pk_mul_f32(op_sel)", "2x v_xor_b32(v,s)"};
constexpr int kOps = 33;

template <int OP>
__global__ void __launch_bounds__(256) issue(uint64_t* ticks, float* sink, uint32_t s0, uint32_t s1) {
  float a[kChains], b[kChains];
  uint32_t u[kChains], w[kChains];
  uint64_t q[kChains], r[kChains];
#pragma unroll
  for (int i = 0; i < kChains; ++i) {
    a[i] = 0.3f + 0.01f * (threadIdx.x + i);
    b[i] = 0.999f;
    u[i] = threadIdx.x * 2654435761u + i;
    w[i] = threadIdx.x + 77u * i;
    q[i] = ((uint64_t)__float_as_uint(0.5f) << 32) | __float_as_uint(0.25f);
    r[i] = ((uint64_t)__float_as_uint(0.999f) << 32) | __float_as_uint(1.001f);
  }
  const uint64_t t0 = clock64();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) body<OP>(a, b, u, w, q, r, s0, s1);
  }
  const uint64_t t1 = clock64();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < kChains; ++i) acc += a[i] + (float)u[i] + (float)(uint32_t)q[i] + (float)(uint32_t)(q[i] >> 32);
  const int gid = blockIdx.x * 256 + threadIdx.x;
  sink[gid] = acc;
  if ((threadIdx.x & 63) == 0) ticks[gid >> 6] = t1 - t0;
}

struct Row {
  std::string name;
  double cyc_shader, cyc_wall;
};

template <int OP>
static Row run(uint64_t* ticks_d, float* sink_d, int waves) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = waves / 4;
  hipLaunchKernelGGL(issue<OP>, dim3(blocks), dim3(256), 0, 0, ticks_d, sink_d, 0xD2511F53u, 0x9E3779B9u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  std::vector<uint64_t> ticks(waves);
  double best_ticks = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(issue<OP>, dim3(blocks), dim3(256), 0, 0, ticks_d, sink_d, 0xD2511F53u, 0x9E3779B9u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
    CK(hipMemcpy(ticks.data(), ticks_d, waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::nth_element(ticks.begin(), ticks.begin() + waves / 2, ticks.end());
    best_ticks = std::min(best_ticks, (double)ticks[waves / 2]);
  }
  const double per_wave = (double)kIters * kUnroll * kChains;   // instructions one wave issues
  Row row;
  row.name = kNames[OP];
  // s_memtime on gfx950 counts at 100 MHz on some stacks and at the shader clock on others: report what it gives and let
  // the caller see both columns (the JSON carries the tick rate measured against wall time)
  row.cyc_shader = best_ticks / (per_wave * kWavesPerSimd);
  row.cyc_wall = (double)best * 1e-3 * 2.4e9 / (per_wave * kWavesPerSimd);
  return row;
}

template <int OP>
static void run_all(std::vector<Row>& rows, uint64_t* ticks_d, float* sink_d, int waves) {
  rows.push_back(run<OP>(ticks_d, sink_d, waves));
  if constexpr (OP + 1 < kOps) run_all<OP + 1>(rows, ticks_d, sink_d, waves);
}

int main(int argc, char** argv) {
  const int waves = kSimds * kWavesPerSimd;
  uint64_t* ticks_d;
  float* sink_d;
  CK(hipMalloc(&ticks_d, waves * sizeof(uint64_t)));
  CK(hipMalloc(&sink_d, (size_t)waves * 64 * sizeof(float)));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs, clockRate %.2f GHz; %d waves (8 per SIMD), %d chains x %d x %d instructions per wave\n", prop.name,
         prop.multiProcessorCount, prop.clockRate / 1e6, waves, kChains, kUnroll, kIters);
  std::vector<Row> rows;
  run_all<0>(rows, ticks_d, sink_d, waves);
  // the tick unit: if s_memtime ran at the shader clock, v_fma_f32's two columns agree; otherwise scale by their ratio
  const double tick_scale = rows[0].cyc_wall / rows[0].cyc_shader;
  const bool ticks_are_cycles = tick_scale > 0.7 && tick_scale < 1.4;
  printf("%-16s %14s %14s\n", "instruction", "cyc (s_memtime)", "cyc (wall@2.4)");
  for (const Row& r : rows) printf("%-16s %14.2f %14.2f\n", r.name.c_str(), r.cyc_shader, r.cyc_wall);
  printf("s_memtime ticks %s shader cycles (wall/tick ratio on v_fma_f32: %.3f)\n", ticks_are_cycles ? "ARE" : "are NOT",
         tick_scale);
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    fprintf(f, "{\"device\": \"%s\", \"waves_per_simd\": %d, \"ticks_are_shader_cycles\": %s, \"wall_over_tick_v_fma\": %.4f,\n",
            prop.name, kWavesPerSimd, ticks_are_cycles ? "true" : "false", tick_scale);
    fprintf(f, " \"unit\": \"SIMD cycles per wave64 instruction (issue-bound, 8 waves per SIMD)\",\n \"cycles\": {");
    for (size_t i = 0; i < rows.size(); ++i)
      fprintf(f, "%s\"%s\": {\"shader\": %.3f, \"wall_2p4ghz\": %.3f}", i ? ", " : "", rows[i].name.c_str(),
              rows[i].cyc_shader, rows[i].cyc_wall);
    fprintf(f, "}}\n");
    fclose(f);
  }
  return 0;
}
