// Launch-shape variants of the diagonal Euler step kernel in the SHARD regime (2-16 MiB per stream: what every GPU of
// the 8-GPU configuration runs), timed as 200 launches replayed from ONE hipGraph (no host launch cost in the figure)
// on random, live-like data with the increment generated in registers like the shipped kernel does.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/microbench_shard.hip -o tools/microbench_shard
//   run:   tools/microbench_shard            (also meaningful under `rocprofv3 --kernel-trace --stats`: each variant is
//                                            its own template instantiation, so the trace has a row per variant)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../torchsde_amd/csrc/tsde_common.h"
using namespace tsde;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
  float* y1;
  const float *y0, *f, *g;
  int64_t nq;   // 16-byte groups
  float dt, sw;
  NoiseKey key;
  uint32_t cell;
};

__device__ __forceinline__ v4f update(const Args& a, int64_t q, v4f y, v4f f, v4f g) {
  float w[4];
  normal4<float>(a.key, (uint64_t)q, a.cell, 0, kStreamW, w);
  v4f o;
  o.x = (y.x + f.x * a.dt) + 1.0f * (g.x * (w[0] * a.sw));
  o.y = (y.y + f.y * a.dt) + 1.0f * (g.y * (w[1] * a.sw));
  o.z = (y.z + f.z * a.dt) + 1.0f * (g.z * (w[2] * a.sw));
  o.w = (y.w + f.w * a.dt) + 1.0f * (g.w * (w[3] * a.sw));
  return o;
}

// BLOCK threads, QPT groups per thread (consecutive tiles of BLOCK groups), all loads of a thread issued before its first
// use; the grid covers the problem exactly (no grid-stride loop) unless CAPPED, which is the shipped kernel's shape.
template <int BLOCK, int QPT, bool CAPPED>
__global__ void __launch_bounds__(BLOCK) step_shape(const Args a) {
  constexpr int64_t kChunk = (int64_t)BLOCK * QPT;
  for (int64_t base = (int64_t)blockIdx.x * kChunk; base < a.nq; base += (int64_t)gridDim.x * kChunk) {
    v4f y[QPT], f[QPT], g[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int64_t q = base + (int64_t)u * BLOCK + threadIdx.x;
      if (q < a.nq) {
        y[u] = reinterpret_cast<const v4f*>(a.y0)[q];
        f[u] = reinterpret_cast<const v4f*>(a.f)[q];
        g[u] = reinterpret_cast<const v4f*>(a.g)[q];
      }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int64_t q = base + (int64_t)u * BLOCK + threadIdx.x;
      if (q < a.nq) reinterpret_cast<v4f*>(a.y1)[q] = update(a, q, y[u], f[u], g[u]);
    }
    if (!CAPPED) break;
  }
}

// 8 bytes per lane: twice the waves for the same bytes (two lanes share one Philox quad; the pair recomputes it)
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) step_half(const Args a) {
  const int64_t h = (int64_t)blockIdx.x * BLOCK + threadIdx.x;   // index of an 8-byte half group
  if (h >= 2 * a.nq) return;
  const v2f y = reinterpret_cast<const v2f*>(a.y0)[h], f = reinterpret_cast<const v2f*>(a.f)[h],
            g = reinterpret_cast<const v2f*>(a.g)[h];
  float w[4];
  normal4<float>(a.key, (uint64_t)(h >> 1), a.cell, 0, kStreamW, w);
  const int o = (int)(h & 1) * 2;
  v2f r;
  r.x = (y.x + f.x * a.dt) + 1.0f * (g.x * (w[o] * a.sw));
  r.y = (y.y + f.y * a.dt) + 1.0f * (g.y * (w[o + 1] * a.sw));
  reinterpret_cast<v2f*>(a.y1)[h] = r;
}

struct Bufs { float *y[2], *f, *g; };

template <typename Launch>
static void time_variant(const char* name, int64_t n, Launch launch, const Bufs& b) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int iters = 200;
  for (int i = 0; i < 5; ++i) launch(s, i);
  CK(hipStreamSynchronize(s));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < iters; ++i) launch(s, i);
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / iters;
  printf("  %-46s %7.2f us/launch  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, us, 16.0 * n / us / 1e3, 16.0 * n / us / 1e3 / 80.0);
  CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(s));
}

template <int BLOCK, int QPT, bool CAPPED>
static void run_shape(const char* name, int64_t n, const Bufs& b) {
  const int64_t nq = n / 4;
  int64_t grid = (nq + (int64_t)BLOCK * QPT - 1) / ((int64_t)BLOCK * QPT);
  if (CAPPED && grid > 2048) grid = 2048;
  time_variant(name, n, [&](hipStream_t s, int i) {
    Args a{b.y[(i + 1) & 1], b.y[i & 1], b.f, b.g, nq, 0.0009765625f, 0.03125f, NoiseKey{1u, 2u, 0}, (uint32_t)i};
    hipLaunchKernelGGL((step_shape<BLOCK, QPT, CAPPED>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
  }, b);
}

template <int BLOCK>
static void run_half(const char* name, int64_t n, const Bufs& b) {
  const int64_t nq = n / 4;
  const int64_t grid = (2 * nq + BLOCK - 1) / BLOCK;
  time_variant(name, n, [&](hipStream_t s, int i) {
    Args a{b.y[(i + 1) & 1], b.y[i & 1], b.f, b.g, nq, 0.0009765625f, 0.03125f, NoiseKey{1u, 2u, 0}, (uint32_t)i};
    hipLaunchKernelGGL((step_half<BLOCK>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
  }, b);
}

int main() {
  const int64_t shapes[][2] = {{16384, 32}, {32768, 32}, {32768, 64}, {65536, 64}, {32768, 128}};
  for (auto& sh : shapes) {
    const int64_t n = sh[0] * sh[1];
    Bufs b;
    std::vector<float> h(n);
    for (int k = 0; k < 4; ++k) {
      float** dst = k == 0 ? &b.y[0] : k == 1 ? &b.y[1] : k == 2 ? &b.f : &b.g;
      CK(hipMalloc(dst, n * 4));
      for (int64_t i = 0; i < n; ++i) h[i] = (k == 2 ? -0.5f : 0.05f) + 0.3f * (float)rand() / (float)RAND_MAX;
      CK(hipMemcpy(*dst, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    printf("B=%lld d=%lld  (%.1f MiB per stream, %lld 16-byte groups)\n", (long long)sh[0], (long long)sh[1],
           n * 4 / 1048576.0, (long long)(n / 4));
    run_shape<256, 1, true>("block 256, 1 group/thread, cap 2048 (shipped)", n, b);
    run_shape<256, 1, false>("block 256, 1 group/thread, uncapped", n, b);
    run_shape<256, 2, false>("block 256, 2 groups/thread", n, b);
    run_shape<256, 4, false>("block 256, 4 groups/thread", n, b);
    run_shape<512, 1, false>("block 512, 1 group/thread", n, b);
    run_shape<512, 2, false>("block 512, 2 groups/thread", n, b);
    run_shape<1024, 1, false>("block 1024, 1 group/thread", n, b);
    run_shape<128, 1, false>("block 128, 1 group/thread", n, b);
    run_shape<64, 1, false>("block 64, 1 group/thread", n, b);
    run_shape<64, 2, false>("block 64, 2 groups/thread", n, b);
    run_half<256>("block 256, 8 bytes/lane", n, b);
    run_half<512>("block 512, 8 bytes/lane", n, b);
    for (int k = 0; k < 2; ++k) CK(hipFree(b.y[k]));
    CK(hipFree(b.f)); CK(hipFree(b.g));
  }
  return 0;
}
