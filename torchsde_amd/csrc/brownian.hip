// Standalone Brownian-increment kernels behind BrownianInterval.__call__
// (replaces torchsde/_brownian/brownian_interval.py:589-687 and the tree/LRU/seed machinery under it).
#include "tsde_common.h"
#include "tsde_launch.h"

namespace tsde {

// ---- raw normals (test / diagnostics) -----------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock) normals_kernel(T* __restrict__ out, int64_t n, NoiseKey key, uint32_t cell,
                                                         uint64_t node, uint32_t stream_id) {
  const uint64_t q0 = key.elem0 >> 2;
  const uint64_t q1 = (key.elem0 + (uint64_t)n + 3) >> 2;
  const int64_t nq = (int64_t)(q1 - q0);
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < nq; t += (int64_t)gridDim.x * kBlock) {
    const uint64_t quad = q0 + (uint64_t)t;
    T v[4];
    normal4<T>(key, quad, cell, node, stream_id, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = (int64_t)(quad * 4 + j) - (int64_t)key.elem0;
      if (i >= 0 && i < n) out[i] = v[j];
    }
  }
}

// ---- general interval query ----------------------------------------------------------------------
struct QueryBounds {
  double a, b;
  int64_t ca, cb;
};

// BrownianInterval.locate on the device: a, b clamped to the grid; ca = (number of edges <= a) - 1 and
// cb = (number of edges < b) - 1, both clamped to [0, n_cells - 1]. Wave-uniform (every thread does the same walk).
TSDE_D QueryBounds locate_bounds(const double* __restrict__ edges, int64_t n_cells, double a, double b) {
  const double lo = edges[0], hi = edges[n_cells];
  a = a < lo ? lo : (a > hi ? hi : a);
  b = b < lo ? lo : (b > hi ? hi : b);
  int64_t l = 0, r = n_cells + 1;          // first index with edges[i] > a
  while (l < r) {
    const int64_t mid = (l + r) >> 1;
    if (edges[mid] <= a) l = mid + 1; else r = mid;
  }
  int64_t ca = l - 1;
  l = 0, r = n_cells + 1;                  // first index with edges[i] >= b
  while (l < r) {
    const int64_t mid = (l + r) >> 1;
    if (edges[mid] < b) l = mid + 1; else r = mid;
  }
  int64_t cb = l - 1;
  ca = ca < 0 ? 0 : (ca > n_cells - 1 ? n_cells - 1 : ca);
  cb = cb < 0 ? 0 : (cb > n_cells - 1 ? n_cells - 1 : cb);
  return QueryBounds{a, b, ca, cb};
}

template <typename T, bool HAVE_H>
__global__ void __launch_bounds__(kBlock) query_kernel(T* __restrict__ W, T* __restrict__ U, T* __restrict__ H,
                                                       int64_t n, NoiseKey key, QueryArgs qa, int vec) {
  if (qa.key_dev != nullptr) {
    const uint64_t e = *qa.key_dev;
    key.k0 = (uint32_t)e;
    key.k1 = (uint32_t)(e >> 32);
  }
  const uint64_t q0 = key.elem0 >> 2;
  const uint64_t q1 = (key.elem0 + (uint64_t)n + 3) >> 2;
  const int64_t nq = (int64_t)(q1 - q0);
  // The interval: launch-time constants, or two doubles in device memory written by an earlier kernel on the stream
  // (adaptive stepping: the attempt's bounds come from the controller kernel, csrc/adaptive.hip). In that case the
  // cells are located here, with the host's rule (BrownianInterval.locate): ca = last edge <= a, cb = last edge < b.
  QueryBounds qb{qa.a, qa.b, qa.ca, qa.cb};
  if (qa.ab_dev != nullptr) {
    qb = locate_bounds(qa.edges, qa.n_cells, qa.ab_dev[0], qa.ab_dev[1]);
    if (!(qb.a < qb.b)) {       // empty interval (an attempt after the last output time): the increment is zero
      for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        W[i] = (T)0;
        if (HAVE_H && U) U[i] = (T)0;
        if (HAVE_H && H) H[i] = (T)0;
      }
      return;
    }
  }
  const double hq = qb.b - qb.a;
  // Bridge-split coefficients of every tree level on the way to a (in cell ca) and to b (in cell cb): computed
  // once per block by threads 0..max_depth, shared through LDS (tsde_bridge.h: DescentTable).
  __shared__ T rows_a[kMaxLevels * DescentTable<T, HAVE_H>::N];
  __shared__ T rows_b[kMaxLevels * DescentTable<T, HAVE_H>::N];
  const DescentTable<T, HAVE_H> ta{rows_a}, tb{rows_b};
  // (an end point that sits on a cell edge needs no split, so its table is neither built nor read)
  if (qb.a != qa.edges[qb.ca]) ta.template build<false>(qa.edges[qb.ca], qa.edges[qb.ca + 1], qb.a, qa.cfg);
  if (qb.b != qa.edges[qb.cb + 1]) tb.template build<true>(qa.edges[qb.cb], qa.edges[qb.cb + 1], qb.b, qa.cfg);
  __syncthreads();
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < nq; t += (int64_t)gridDim.x * kBlock) {
    const uint64_t quad = q0 + (uint64_t)t;
    PieceAcc<T, HAVE_H> acc;
    acc.clear();
    WH4<T> root;
    {
      const double s = qa.edges[qb.ca], e = qa.edges[qb.ca + 1];
      cell_root<T, HAVE_H>(key, quad, (uint32_t)qb.ca, e - s, root);
      if (qa.rootW != nullptr) {
        // Pinned top-level interval: the user supplied W (and maybe H) of the single cell
        // (brownian_interval.py:553-561, arguments `W=` / `H=`).
        const T* rw = (const T*)qa.rootW;
        const T* rh = (const T*)qa.rootH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t i = (int64_t)(quad * 4 + j) - (int64_t)key.elem0;
          if (i >= 0 && i < n) {
            root.W[j] = rw[i];
            if (HAVE_H && rh) root.H[j] = rh[i];
          }
        }
      }
    }
    if (qb.ca == qb.cb) {
      const double s = qa.edges[qb.ca], e = qa.edges[qb.ca + 1];
      cell_range<T, HAVE_H>(key, quad, (uint32_t)qb.ca, s, e, qb.a, qb.b, root, qa.cfg, ta, tb, acc);
    } else {
      {
        const double s = qa.edges[qb.ca], e = qa.edges[qb.ca + 1];
        cell_range<T, HAVE_H>(key, quad, (uint32_t)qb.ca, s, e, qb.a, e, root, qa.cfg, ta, tb, acc);
      }
      for (int64_t c = qb.ca + 1; c < qb.cb; ++c) {
        const double h = qa.edges[c + 1] - qa.edges[c];
        WH4<T> P;
        cell_root<T, HAVE_H>(key, quad, (uint32_t)c, h, P);
        acc.push_right(P, h);
      }
      {
        const double s = qa.edges[qb.cb], e = qa.edges[qb.cb + 1];
        WH4<T> P;
        cell_root<T, HAVE_H>(key, quad, (uint32_t)qb.cb, e - s, P);
        cell_range<T, HAVE_H>(key, quad, (uint32_t)qb.cb, s, e, s, qb.b, P, qa.cfg, ta, tb, acc);
      }
    }
    Pack<T, 4> w, u, hh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w.v[j] = acc.v.W[j];
      hh.v[j] = acc.v.H[j];
      u.v[j] = (T)hq * ((T)0.5 * acc.v.W[j] + acc.v.H[j]);  // _H_to_U, brownian_interval.py:102-103
    }
    const int64_t i0 = (int64_t)(quad * 4) - (int64_t)key.elem0;
    if (vec) {
      store<T, 4>(W, i0, w);
      if (HAVE_H && U) store<T, 4>(U, i0, u);
      if (HAVE_H && H) store<T, 4>(H, i0, hh);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i >= 0 && i < n) {
          W[i] = w.v[j];
          if (HAVE_H && U) U[i] = u.v[j];
          if (HAVE_H && H) H[i] = hh.v[j];
        }
      }
    }
  }
}

template <typename T>
hipError_t launch_normals(void* out, int64_t n, NoiseKey key, uint32_t cell, uint64_t node, uint32_t stream_id,
                          hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(normals_kernel<T>, dim3(grid_for((n + 3) / 4 + 1)), dim3(kBlock), 0, s, (T*)out, n, key, cell,
                     node, stream_id);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_query(void* W, void* U, void* H, int64_t n, NoiseKey key, const QueryArgs& qa, bool have_h,
                        hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const bool vec = (key.elem0 % 4 == 0) && (n % 4 == 0) && aligned16(W) && (!U || aligned16(U)) &&
                   (!H || aligned16(H));
  const dim3 grid(grid_for((n + 3) / 4 + 1));
  if (have_h) {
    TSDE_LAUNCH((query_kernel<T, true>), grid, dim3(kBlock), 0, s, (T*)W, (T*)U, (T*)H, n, key, qa, vec ? 1 : 0);
  } else {
    TSDE_LAUNCH((query_kernel<T, false>), grid, dim3(kBlock), 0, s, (T*)W, (T*)nullptr, (T*)nullptr, n, key, qa,
                vec ? 1 : 0);
  }
  return hipGetLastError();
}

template hipError_t launch_normals<float>(void*, int64_t, NoiseKey, uint32_t, uint64_t, uint32_t, hipStream_t);
template hipError_t launch_normals<double>(void*, int64_t, NoiseKey, uint32_t, uint64_t, uint32_t, hipStream_t);
template hipError_t launch_query<float>(void*, void*, void*, int64_t, NoiseKey, const QueryArgs&, bool, hipStream_t);
template hipError_t launch_query<double>(void*, void*, void*, int64_t, NoiseKey, const QueryArgs&, bool, hipStream_t);

}  // namespace tsde
