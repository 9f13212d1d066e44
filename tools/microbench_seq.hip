// In-situ micro-benchmark: the per-step kernel SEQUENCE of the C2 workload (producer kernels f = mu*y, g = sigma*y
// standing in for the user's torch ops, then the fused Euler step), with random data, timing each kernel with events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../torchsde_amd/csrc/tsde_common.h"
using namespace tsde;
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// producer: out = coef[col] * y  (d = 64 -> 16 quads per row); BLK threads, UNR quads per thread, torch-like tiling
template <int BLK, int UNR>
__global__ void __launch_bounds__(BLK) bcast_mul(float* __restrict__ out, const float* __restrict__ y, const float* __restrict__ coef, int64_t nq) {
  const int64_t base = (int64_t)blockIdx.x * BLK * UNR;
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int64_t q = base + u * BLK + threadIdx.x;
    if (q < nq) {
      const v4f a = reinterpret_cast<const v4f*>(y)[q];
      const v4f c = reinterpret_cast<const v4f*>(coef)[q & 15];
      reinterpret_cast<v4f*>(out)[q] = a * c;
    }
  }
}

template <int QPT, bool NTL = false, bool NTS = false>   // quads per thread, contiguous per block; nontemporal loads / stores
__global__ void __launch_bounds__(256) step_blocked(float* __restrict__ y1, const float* __restrict__ y0, const float* __restrict__ f,
                                                    const float* __restrict__ g, int64_t nq, float dt, NoiseKey key, uint32_t cell, float sw) {
  for (int64_t blk = blockIdx.x; blk * 256 * QPT < nq; blk += gridDim.x) {
    const int64_t base = blk * 256 * QPT;
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int64_t q = base + u * 256 + threadIdx.x;
      if (q < nq) {
        const v4f a = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(y0) + q) : reinterpret_cast<const v4f*>(y0)[q];
        const v4f b = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(f) + q) : reinterpret_cast<const v4f*>(f)[q];
        const v4f c = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g) + q) : reinterpret_cast<const v4f*>(g)[q];
        float w[4];
        normal4<float>(key, (uint64_t)q, cell, 0, kStreamW, w);
        v4f o;
        o.x = (a.x + b.x * dt) + c.x * (w[0] * sw); o.y = (a.y + b.y * dt) + c.y * (w[1] * sw);
        o.z = (a.z + b.z * dt) + c.z * (w[2] * sw); o.w = (a.w + b.w * dt) + c.w * (w[3] * sw);
        if (NTS) __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(y1) + q);
        else reinterpret_cast<v4f*>(y1)[q] = o;
      }
    }
  }
}

int main() {
  const int64_t n = (getenv("TSDE_ROWS") ? atoll(getenv("TSDE_ROWS")) : 65536LL) * 64, nq = n / 4;
  float *y[2], *f, *g, *mu, *sg;
  for (int i = 0; i < 2; ++i) CK(hipMalloc(&y[i], n * 4));
  CK(hipMalloc(&f, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&mu, 256)); CK(hipMalloc(&sg, 256));
  std::vector<float> h(n);
  for (int64_t i = 0; i < n; ++i) h[i] = 0.05f + 0.1f * (float)rand() / RAND_MAX;
  CK(hipMemcpy(y[0], h.data(), n * 4, hipMemcpyHostToDevice));
  float hm[64], hs[64];
  for (int i = 0; i < 64; ++i) { hm[i] = -0.3f - 0.5f * (float)rand() / RAND_MAX; hs[i] = 0.2f + 0.5f * (float)rand() / RAND_MAX; }
  CK(hipMemcpy(mu, hm, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sg, hs, 256, hipMemcpyHostToDevice));
  NoiseKey key{1u, 2u, 0};
  const int iters = n > (1LL << 26) ? 30 : 300;
  std::vector<hipEvent_t> ev(4 * iters);
  for (auto& evt : ev) CK(hipEventCreate(&evt));
  auto seq = [&](int variant, int it, bool timed) {
    float* yi = y[it & 1]; float* yo = y[(it + 1) & 1];
    if (timed) hipEventRecord(ev[4 * it + 0], 0);
    if (true) {   // torch-like producer tiling: 128 threads x 4
      hipLaunchKernelGGL((bcast_mul<128, 4>), dim3((nq + 511) / 512), dim3(128), 0, 0, f, yi, mu, nq);
      if (timed) hipEventRecord(ev[4 * it + 1], 0);
      hipLaunchKernelGGL((bcast_mul<128, 4>), dim3((nq + 511) / 512), dim3(128), 0, 0, g, yi, sg, nq);
    } else {                               // producer tiled like the consumer: 256 threads x 1 (4 KiB chunks)
      hipLaunchKernelGGL((bcast_mul<256, 1>), dim3((nq + 255) / 256), dim3(256), 0, 0, f, yi, mu, nq);
      if (timed) hipEventRecord(ev[4 * it + 1], 0);
      hipLaunchKernelGGL((bcast_mul<256, 1>), dim3((nq + 255) / 256), dim3(256), 0, 0, g, yi, sg, nq);
    }
    if (timed) hipEventRecord(ev[4 * it + 2], 0);
    switch (variant) {
      case 0: hipLaunchKernelGGL((step_blocked<1>), dim3(2048), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 1: hipLaunchKernelGGL((step_blocked<2>), dim3(2048), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 2: hipLaunchKernelGGL((step_blocked<2>), dim3(1024), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 3: hipLaunchKernelGGL((step_blocked<4>), dim3(1024), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 4: hipLaunchKernelGGL((step_blocked<4>), dim3(512), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 5: hipLaunchKernelGGL((step_blocked<8>), dim3(512), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 6: hipLaunchKernelGGL((step_blocked<2, true, false>), dim3(2048), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 7: hipLaunchKernelGGL((step_blocked<2, true, true>), dim3(2048), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 8: hipLaunchKernelGGL((step_blocked<1, true, false>), dim3(2048), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 9: hipLaunchKernelGGL((step_blocked<1, true, true>), dim3(2048), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 10: hipLaunchKernelGGL((step_blocked<1, true, true>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
      case 11: hipLaunchKernelGGL((step_blocked<2, true, true>), dim3((unsigned)((nq + 511) / 512)), dim3(256), 0, 0, yo, yi, f, g, nq, 9.765625e-4f, key, (uint32_t)it, 0.03125f); break;
    }
    if (timed) hipEventRecord(ev[4 * it + 3], 0);
  };
  const char* names[] = {"QPT1 g2048", "QPT2 g2048", "QPT2 g1024", "QPT4 g1024", "QPT4 g512", "QPT8 g512", "QPT2 NT-loads", "QPT2 NT-loads+store", "QPT1 NT-loads", "QPT1 NT-loads+store", "QPT1 NT l+s full grid", "QPT2 NT l+s full grid"};
  for (int rep = 0; rep < 2; ++rep)
    for (int variant = 0; variant < 12; ++variant) {
      for (int it = 0; it < 20; ++it) seq(variant, it, false);
      CK(hipDeviceSynchronize());
      for (int it = 0; it < iters; ++it) seq(variant, it, true);
      CK(hipDeviceSynchronize());
      double tf = 0, tg = 0, ts = 0;
      for (int it = 0; it < iters; ++it) {
        float a, b, c;
        hipEventElapsedTime(&a, ev[4 * it], ev[4 * it + 1]); hipEventElapsedTime(&b, ev[4 * it + 1], ev[4 * it + 2]); hipEventElapsedTime(&c, ev[4 * it + 2], ev[4 * it + 3]);
        tf += a; tg += b; ts += c;
      }
      printf("%-52s f %6.2f us  g %6.2f us  step %6.2f us (%6.1f GB/s)  total %6.2f us\n", names[variant], tf * 1e3 / iters, tg * 1e3 / iters,
             ts * 1e3 / iters, 16.0 * n / (ts * 1e3 / iters) / 1e3, (tf + tg + ts) * 1e3 / iters);
    }
  return 0;
}
