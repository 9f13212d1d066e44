// Shared launch plumbing for the gfx950 kernels: 16-byte packs, grid sizing, noise sources.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tsde_bridge.h"
#include "tsde_rng.h"

namespace tsde {

// Per-dispatch timing for bench.py's roofline (tsde_prof_begin / _end in capi.hip): when a C-ABI entry point has armed
// a pair of events, the next kernel launch is issued with hipExtLaunchKernelGGL, which binds the events to THAT dispatch
// -- their elapsed time is the dispatch's own start-to-end interval, what rocprofv3's kernel trace reports, without
// the marker packets (and their ~2.5 us) that hipEventRecord calls around a launch put on the stream.
struct LaunchTiming {
  hipEvent_t start, stop;
  bool armed;
};
inline LaunchTiming& launch_timing() {
  static thread_local LaunchTiming t{nullptr, nullptr, false};
  return t;
}
#define TSDE_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                   \
  do {                                                                                                         \
    ::tsde::LaunchTiming& lt_ = ::tsde::launch_timing();                                                       \
    if (lt_.armed) {                                                                                           \
      lt_.armed = false;                                                                                       \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, lt_.start, lt_.stop, 0, __VA_ARGS__);          \
    } else {                                                                                                   \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                     \
    }                                                                                                          \
  } while (0)

constexpr int kBlock = 256;         // 4 waves of 64 lanes
constexpr int kMaxGrid = 256 * 8;   // 256 CUs x 8 resident blocks; the rest is grid-stride

// W contiguous elements of T. W=4 is the vector path: 16 B (fp32) or 2x16 B (fp64) per lane.
template <typename T, int W>
struct Pack {
  T v[W];
};

typedef float tsde_f4 __attribute__((ext_vector_type(4)));
typedef double tsde_d2 __attribute__((ext_vector_type(2)));

// NT = nontemporal (streaming) access: used when the kernel's streams exceed the 256 MiB Infinity Cache, where it
// is worth +30 % (tools/microbench_seq.hip at 1M rows: 216 -> 166 us); inside the cache it costs 5-8 %.
template <typename T, int W, bool NT = false>
TSDE_D Pack<T, W> load(const T* __restrict__ p, int64_t i) {
  Pack<T, W> r;
  if constexpr (W == 4 && sizeof(T) == 4) {
    const tsde_f4* q4 = reinterpret_cast<const tsde_f4*>(p + i);
    const tsde_f4 q = NT ? __builtin_nontemporal_load(q4) : *q4;
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else if constexpr (W == 4 && sizeof(T) == 8) {
    const tsde_d2* q2 = reinterpret_cast<const tsde_d2*>(p + i);
    const tsde_d2 q0 = NT ? __builtin_nontemporal_load(q2) : q2[0];
    const tsde_d2 q1 = NT ? __builtin_nontemporal_load(q2 + 1) : q2[1];
    r.v[0] = q0.x; r.v[1] = q0.y; r.v[2] = q1.x; r.v[3] = q1.y;
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j) r.v[j] = p[i + j];
  }
  return r;
}

template <typename T, int W, bool NT = false>
TSDE_D void store(T* __restrict__ p, int64_t i, const Pack<T, W>& r) {
  if constexpr (W == 4 && sizeof(T) == 4) {
    const tsde_f4 q = {r.v[0], r.v[1], r.v[2], r.v[3]};
    if (NT) __builtin_nontemporal_store(q, reinterpret_cast<tsde_f4*>(p + i));
    else *reinterpret_cast<tsde_f4*>(p + i) = q;
  } else if constexpr (W == 4 && sizeof(T) == 8) {
    const tsde_d2 q0 = {r.v[0], r.v[1]}, q1 = {r.v[2], r.v[3]};
    if (NT) {
      __builtin_nontemporal_store(q0, reinterpret_cast<tsde_d2*>(p + i));
      __builtin_nontemporal_store(q1, reinterpret_cast<tsde_d2*>(p + i) + 1);
    } else {
      reinterpret_cast<tsde_d2*>(p + i)[0] = q0;
      reinterpret_cast<tsde_d2*>(p + i)[1] = q1;
    }
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j) p[i + j] = r.v[j];
  }
}

// A scalar coefficient of a step kernel: a launch-time constant, or a word in DEVICE memory that an earlier kernel on
// the stream wrote (adaptive stepping without a host round trip per attempt: the step size, dt/2, sqrt(dt), 1/dt and
// the interpolation weights of an attempt are produced by the controller kernel, csrc/adaptive.hip). On the C ABI such
// a coefficient travels in the SAME `double` argument as a constant would, as a quiet NaN whose low 48 bits are the
// device address (include/torchsde_amd.h: TSDE_DEV_SCALAR); `coef<T>()` decodes it on the host side of the launch.
template <typename T>
struct Coef {
  T v;
  const T* p;
  TSDE_D T get() const { return p ? *p : v; }
};

constexpr uint64_t kDevScalarTag = 0x7FFCull;   // sign 0, exponent all ones, quiet bit and bit 50 set

inline bool is_dev_scalar(double x) {
  uint64_t bits;
  memcpy(&bits, &x, sizeof(bits));
  return (bits >> 48) == kDevScalarTag;
}

template <typename T>
inline Coef<T> coef(double x) {
  uint64_t bits;
  memcpy(&bits, &x, sizeof(bits));
  if ((bits >> 48) == kDevScalarTag)
    return Coef<T>{(T)0, reinterpret_cast<const T*>((uintptr_t)(bits & 0xFFFFFFFFFFFFull))};
  return Coef<T>{(T)x, nullptr};
}

// The Brownian increment of ONE grid cell, as the step kernels consume it.
//   dW  == nullptr : generate in registers from (key, cell, h)              [fused, 0 bytes of HBM]
//   dW  != nullptr : read a materialised increment (foreign bm / replay).   [+4 B per element]
// `bcast_d` > 0 reads the external tensor with one value per row of d state channels (scalar noise).
template <typename T>
struct CellNoise {
  const T* dW;
  const T* dU;
  NoiseKey key;
  uint32_t cell;
  double h;
  T sw, sh, th;      // sqrt(h), sqrt(h/12), h rounded to T -- on the HOST (set_width), so that no kernel runs a
                     // double-precision sqrt / divide per row (general_rows_kernel: 8.4 -> 6 us at the C3 shape)
  int64_t bcast_d;
  const uint64_t* key_dev;   // optional run-time entropy (HIP-graph replay with a new seed)
};

// Width of the cell and the scale factors derived from it (host side; IEEE sqrt and divide are correctly rounded on
// both sides, so these are the bits the kernels used to compute for themselves).
template <typename T>
inline void set_width(CellNoise<T>& c, double h) {
  c.h = h;
  c.sw = (T)sqrt(h);
  c.sh = (T)sqrt(h / 12.0);
  c.th = (T)h;
}

// The key a kernel actually draws with: the baked one, or the device word if present (wave-uniform load).
template <typename T>
TSDE_D NoiseKey live_key(const CellNoise<T>& nz) {
  NoiseKey k = nz.key;
  if (nz.key_dev != nullptr) {
    const uint64_t e = *nz.key_dev;
    k.k0 = (uint32_t)e;
    k.k1 = (uint32_t)(e >> 32);
  }
  return k;
}

// W and (optionally) U = h (W/2 + H) for W consecutive elements starting at local index i.
template <typename T, int W, bool NEED_U>
TSDE_D void cell_noise(const CellNoise<T>& nz, int64_t i, Pack<T, W>& w, Pack<T, W>& u) {
  if (nz.dW != nullptr) {
    if (nz.bcast_d > 0) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const int64_t r = (i + j) / nz.bcast_d;
        w.v[j] = nz.dW[r];
        if (NEED_U) u.v[j] = nz.dU[r];
      }
    } else {
      w = load<T, W>(nz.dW, i);
      if (NEED_U) u = load<T, W>(nz.dU, i);
    }
    return;
  }
  const T sw = nz.sw, sh = nz.sh, th = nz.th;
  const NoiseKey key = live_key(nz);
  const uint64_t e = key.elem0 + (uint64_t)i;
  if constexpr (W == 4) {
    T n[4];
    normal4<T>(key, e >> 2, nz.cell, 0, kStreamW, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) w.v[j] = n[j] * sw;
    if (NEED_U) {
      normal4<T>(key, e >> 2, nz.cell, 0, kStreamH, n);
#pragma unroll
      for (int j = 0; j < 4; ++j) u.v[j] = th * ((T)0.5 * w.v[j] + n[j] * sh);
    }
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      w.v[j] = normal1<T>(key, e + j, nz.cell, 0, kStreamW) * sw;
      if (NEED_U) u.v[j] = th * ((T)0.5 * w.v[j] + normal1<T>(key, e + j, nz.cell, 0, kStreamH) * sh);
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int grid_for(int64_t work_items) {
  int64_t g = (work_items + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

// Elementwise driver. `Op` provides  template<int W> __device__ void run(int64_t i) const  acting on
// elements [i, i+W). The vector path (W=4) is taken when `vec` is set by the launcher (all pointers
// 16-byte aligned, n % 4 == 0 and the global noise offset % 4 == 0).
// Vector path tiling: a block owns QPT * 256 CONSECUTIVE 16-byte groups. Measured in situ on MI355X
// (tools/microbench_seq.hip, C2 shapes, between producer kernels): QPT=2 -> 12.0 us per step kernel,
// 1 -> 13.4 us, 4 or 8 -> 13.3-13.7 us. Small problems keep QPT=1 so that every CU still gets >= 8 blocks.
template <typename Op, int QPT, bool NT>
__global__ void __launch_bounds__(kBlock) elementwise_kernel(const Op op, const int64_t n, const int vec) {
  if (vec == 2) {
    // XCD-matched tiling (launch_elementwise; `TSDE_XCD_MATCH=0` switches it off). Workgroups go to the 8 XCDs round-robin, and so do the
    // workgroups of the torch kernels that produced f and g just before this launch (and that will read y1 right
    // after it): ATen's broadcasting elementwise kernel gives workgroup c the 512 consecutive elements c*512 ..
    // (elementwise_kernel_manual_unroll<128, 4>), i.e. 128 16-byte groups, so chunk c was written through the L2 of XCD
    // c % 8. This tiling hands block b the 128-group chunks c with c % 8 == b % 8 -- its own XCD's lines.
    const int64_t nq = n >> 2;
    const int64_t n_chunks = (nq + 127) >> 7;                         // 128-group chunks
    const int64_t per_xcd = (n_chunks + 7) >> 3;                      // chunks whose index is = x (mod 8), upper bound
    const int x = blockIdx.x & 7;
    const int64_t lane_block = blockIdx.x >> 3, blocks_per_xcd = gridDim.x >> 3;
    constexpr int kSub = (kBlock * QPT) / 128;                        // chunks per block iteration
    for (int64_t j0 = lane_block * kSub; j0 < per_xcd; j0 += blocks_per_xcd * kSub) {
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const int idx = u * kBlock + threadIdx.x;
        const int64_t c = ((j0 + (idx >> 7)) << 3) + x;               // chunk index
        const int64_t q = (c << 7) + (idx & 127);
        if (q < nq) op.template run<4, NT>(q << 2);
      }
    }
  } else if (vec) {
    const int64_t nq = n >> 2;
    constexpr int64_t kChunk = (int64_t)kBlock * QPT;
    for (int64_t base = (int64_t)blockIdx.x * kChunk; base < nq; base += (int64_t)gridDim.x * kChunk) {
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const int64_t q = base + (int64_t)u * kBlock + threadIdx.x;
        if (q < nq) op.template run<4, NT>(q << 2);
      }
    }
  } else {
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = tid; i < n; i += stride) op.template run<1, false>(i);
  }
}

// Per-stream size above which the streaming (nontemporal, uncapped-grid) variant is used: 96 MiB per stream, i.e.
// a 4-stream kernel whose live data no longer fits the 256 MiB Infinity Cache. TSDE_FORCE_NT=1 forces it on and
// TSDE_FORCE_NT=0 off (tests, and the on/off measurement of tools/bench_kernels.py).
inline bool use_streaming_variant(int64_t n, size_t elem_size) {
  static const int forced = [] {
    const char* e = getenv("TSDE_FORCE_NT");
    return (e && e[0] == '1') ? 1 : (e && e[0] == '0') ? 0 : -1;
  }();
  if (forced >= 0) return forced == 1;
  return (uint64_t)n * elem_size >= (96ull << 20);
}

// The XCD-matched tiling of elementwise_kernel is the default; TSDE_XCD_MATCH=0 restores consecutive tiles (the on / off
// measurement: profiles/r4_xcd_matched_tiling.txt -- stepwise GBM solves 8-20 % faster, nothing slower than 1 %).
inline bool xcd_matched() {
  static const bool on = [] {
    const char* e = getenv("TSDE_XCD_MATCH");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <typename Op>
inline hipError_t launch_elementwise(const Op& op, int64_t n, bool vec, hipStream_t stream, size_t elem_size = 4) {
  if (n <= 0) return hipSuccess;
  const int64_t nq = n >> 2;
  if (vec && use_streaming_variant(n, elem_size)) {
    int64_t blocks = (nq + kBlock - 1) / kBlock;                 // one 16-B group per thread, no grid cap
    if (blocks > (1 << 30)) blocks = 1 << 30;
    TSDE_LAUNCH((elementwise_kernel<Op, 1, true>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, op, n, 1);
  } else if (vec && nq >= (int64_t)2 * kBlock * kMaxGrid) {
    TSDE_LAUNCH((elementwise_kernel<Op, 2, false>), dim3(grid_for((nq + 1) / 2)), dim3(kBlock), 0, stream, op,
                       n, xcd_matched() ? 2 : 1);
  } else {
    const int grid = grid_for(vec ? nq : n);
    TSDE_LAUNCH((elementwise_kernel<Op, 1, false>), dim3(grid), dim3(kBlock), 0, stream, op,
                       n, vec ? ((xcd_matched() && grid >= 8 && grid % 8 == 0) ? 2 : 1) : 0);
  }
  return hipGetLastError();
}

}  // namespace tsde
