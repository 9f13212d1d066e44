// Reversible Heun (Stratonovich, arXiv:2105.13493) and its exact-gradient adjoint: fused elementwise kernels.
// Reference: torchsde/_core/methods/reversible_heun.py:48-73 (forward step), :98-144 (adjoint step).
// Same conventions as steps.hip: reference operation order, one rounding per op, increment generated in
// registers from the step's grid cell (or read from memory for foreign Brownian objects).
#include "tsde_common.h"
#include "tsde_launch.h"

namespace tsde {

template <typename T>
static CellNoise<T> rh_noise(const tsde_noise_t* nz) {
  CellNoise<T> c;
  c.dW = (const T*)nz->dW;
  c.dU = (const T*)nz->dU;
  c.key.k0 = (uint32_t)nz->entropy;
  c.key.k1 = (uint32_t)(nz->entropy >> 32);
  c.key.elem0 = nz->elem0;
  c.cell = nz->cell;
  set_width<T>(c, nz->h);
  c.bcast_d = nz->bcast_d;
  c.key_dev = nz->entropy_dev;
  return c;
}

static bool rh_noise_vec(const tsde_noise_t* nz) {
  if (nz->dW == nullptr) return (nz->elem0 % 4) == 0;
  if (nz->bcast_d > 0) return (nz->bcast_d % 4) == 0;
  return aligned16(nz->dW);
}

// z1 = ((2*y0 - z0) + s*(f0*dt)) + s*(g0*dW)      reversible_heun.py:69 (s=+1), :109 (s=-1)
template <typename T>
struct RheunZOp {
  T* z1;
  const T *y0, *z0, *f0, *g0;
  T dt, sgn;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const Pack<T, W> y = load<T, W, NT>(y0, i), z = load<T, W, NT>(z0, i), f = load<T, W, NT>(f0, i), g = load<T, W, NT>(g0, i);
    Pack<T, W> w, u, o;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = ((((T)2 * y.v[j]) - z.v[j]) + sgn * (f.v[j] * dt)) + sgn * (g.v[j] * w.v[j]);
    store<T, W, NT>(z1, i, o);
  }
};

// y1 = (y0 + s*((f0+f1)*half_dt)) + s*((g0+g1)*(0.5*dW))     reversible_heun.py:71 (s=+1), :130-131 (s=-1)
template <typename T>
struct RheunYOp {
  T* y1;
  const T *y0, *f0, *f1, *g0, *g1;
  T half_dt, sgn;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const Pack<T, W> y = load<T, W, NT>(y0, i), a = load<T, W, NT>(f0, i), b = load<T, W, NT>(f1, i), c = load<T, W, NT>(g0, i),
                     d = load<T, W, NT>(g1, i);
    Pack<T, W> w, u, o;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const T drift = (a.v[j] + b.v[j]) * half_dt;
      const T diff = (c.v[j] + d.v[j]) * ((T)0.5 * w.v[j]);
      o.v[j] = (y.v[j] + sgn * drift) + sgn * diff;
    }
    store<T, W, NT>(y1, i, o);
  }
};

// out = a*x + b*y
template <typename T>
struct Lincomb2Op {
  T* out;
  const T *x, *y;
  T a, b;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const Pack<T, W> p = load<T, W, NT>(x, i), q = load<T, W, NT>(y, i);
    Pack<T, W> o;
#pragma unroll
    for (int j = 0; j < W; ++j) o.v[j] = a * p.v[j] + b * q.v[j];
    store<T, W, NT>(out, i, o);
  }
};

// adjoint stage A (diagonal noise): af0' = af0 + ay*half_dt ; ag0' = ag0 + ay*(0.5*dW)     :106-117
template <typename T>
struct RheunAdjAOp {
  T *af0_out, *ag0_out;
  const T *ay, *af0, *ag0;
  T half_dt;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const Pack<T, W> a = load<T, W, NT>(ay, i), f = load<T, W, NT>(af0, i), g = load<T, W, NT>(ag0, i);
    Pack<T, W> w, u, of, og;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      of.v[j] = f.v[j] + a.v[j] * half_dt;
      og.v[j] = g.v[j] + a.v[j] * ((T)0.5 * w.v[j]);
    }
    store<T, W, NT>(af0_out, i, of);
    store<T, W, NT>(ag0_out, i, og);
  }
};

// adjoint stage B (diagonal noise), after the VJP through f_and_g(z0):                       :127,134-137
//   az0' = az0 + vjp_z ; ay1 = ay + 2*az0' ; az1 = -az0' ; af1 = ay*half_dt + az0'*dt ; ag1 = ay*(0.5*dW) + az0'*dW
template <typename T>
struct RheunAdjBOp {
  T *ay1, *az1, *af1, *ag1;
  const T *ay, *az0, *vjp_z;
  T dt, half_dt;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const Pack<T, W> a = load<T, W, NT>(ay, i), z = load<T, W, NT>(az0, i), v = load<T, W, NT>(vjp_z, i);
    Pack<T, W> w, u, oy, oz, of, og;
    cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const T zz = z.v[j] + v.v[j];
      oy.v[j] = a.v[j] + (T)2 * zz;
      oz.v[j] = -zz;
      of.v[j] = a.v[j] * half_dt + zz * dt;
      og.v[j] = a.v[j] * ((T)0.5 * w.v[j]) + zz * w.v[j];
    }
    store<T, W, NT>(ay1, i, oy);
    store<T, W, NT>(az1, i, oz);
    store<T, W, NT>(af1, i, of);
    store<T, W, NT>(ag1, i, og);
  }
};

// ---- Heun / Euler-Heun final stage (Stratonovich predictor-corrector) ------------------------------------------
//   mode 0, Heun        (heun.py:35-48):        y1 = y0 + (((dt*(f + fp)) + g*dW) + gp*dW) * 0.5
//   mode 1, Euler-Heun  (euler_heun.py:29-42):  y1 = (y0 + dt*f) + ((g*dW + gp*dW) * 0.5)
// PROD = true: g / gp already hold the diffusion-vector products (user-supplied g_prod, or a contraction).
template <typename T, bool PROD>
struct HeunFinalOp {
  T* y1;
  const T *y0, *f, *fp, *g, *gp;
  Coef<T> dt_;
  int mode;
  CellNoise<T> nz;
  template <int W, bool NT = false>
  TSDE_D void run(int64_t i) const {
    const T dt = dt_.get();
    const Pack<T, W> y = load<T, W, NT>(y0, i), a = load<T, W, NT>(f, i), c = load<T, W, NT>(g, i), d = load<T, W, NT>(gp, i);
    Pack<T, W> b, w, u, o;
    if (mode == 0) b = load<T, W, NT>(fp, i);
    if (!PROD) cell_noise<T, W, false>(nz, i, w, u);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const T p0 = PROD ? c.v[j] : c.v[j] * w.v[j];
      const T p1 = PROD ? d.v[j] : d.v[j] * w.v[j];
      if (mode == 0) {
        o.v[j] = y.v[j] + (((dt * (a.v[j] + b.v[j])) + p0) + p1) * (T)0.5;
      } else {
        o.v[j] = (y.v[j] + dt * a.v[j]) + ((p0 + p1) * (T)0.5);
      }
    }
    store<T, W, NT>(y1, i, o);
  }
};

template <typename T>
hipError_t launch_heun_final(void* y1, const void* y0, const void* f, const void* fp, const void* g, const void* gp,
                             int64_t n, double dt, int mode, int prod, const tsde_noise_t* nz, hipStream_t s) {
  bool vec = (n % 4 == 0) && aligned16(y1) && aligned16(y0) && aligned16(f) && (!fp || aligned16(fp)) && aligned16(g) &&
             aligned16(gp);
  if (prod) {
    HeunFinalOp<T, true> op{(T*)y1, (const T*)y0, (const T*)f, (const T*)fp, (const T*)g, (const T*)gp, coef<T>(dt), mode,
                            CellNoise<T>{}};
    return launch_elementwise(op, n, vec, s, sizeof(T));
  }
  HeunFinalOp<T, false> op{(T*)y1, (const T*)y0, (const T*)f, (const T*)fp, (const T*)g, (const T*)gp, coef<T>(dt), mode,
                           rh_noise<T>(nz)};
  vec = vec && rh_noise_vec(nz);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

// ---- Davie / Foster approximation of the Levy area of one interval (brownian_interval.py:78-99) ------------------
//   A[b,i,j] = H_i W_j - W_i H_j + std_ij (N_ij - N_ji),  std = sqrt(h^2/12) (Davie) or
//   sqrt(0.1h (0.1h + H_i^2 + H_j^2)) (Foster);  N: independent normals of stream A keyed on (cell, node).
template <typename T>
__global__ void __launch_bounds__(kBlock) levy_area_kernel(T* __restrict__ A, const T* __restrict__ W,
                                                           const T* __restrict__ H, int64_t B, int64_t m, double h,
                                                           int foster, NoiseKey key, const uint64_t* key_dev,
                                                           uint32_t cell, uint64_t node) {
  if (key_dev != nullptr) {
    const uint64_t e = *key_dev;
    key.k0 = (uint32_t)e;
    key.k1 = (uint32_t)(e >> 32);
  }
  const int64_t total = B * m * m;
  const T tenth_h = (T)(0.1 * h);
  const T davie_std = (T)sqrt(h * h / 12.0);
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t b = t / (m * m);
    const int64_t r = t - b * m * m;
    const int64_t i = r / m, j = r - i * m;
    const T Wi = W[b * m + i], Wj = W[b * m + j], Hi = H[b * m + i], Hj = H[b * m + j];
    T a = Hi * Wj - Wi * Hj;
    if (i != j) {
      const uint64_t eij = key.elem0 * (uint64_t)m + (uint64_t)t;
      const uint64_t eji = key.elem0 * (uint64_t)m + (uint64_t)(b * m * m + j * m + i);
      const T nij = normal1<T>(key, eij, cell, node, kStreamA);
      const T nji = normal1<T>(key, eji, cell, node, kStreamA);
      // Hi^2 + Hj^2 first: commutative, so A stays exactly antisymmetric
      const T sd = foster ? (T)sqrt((double)(tenth_h * (tenth_h + (Hi * Hi + Hj * Hj)))) : davie_std;
      a += sd * (nij - nji);
    }
    A[t] = a;
  }
}

// The same values, one wave per batch row: the row's m*m normals are drawn ONCE -- m*m/4 Philox calls, four normals each,
// the very quads `normal1` above indexes into -- and staged in LDS, from where entry (i, j) picks N_ij and N_ji. The
// kernel above makes two Philox calls per entry and keeps one of the eight normals they produce (33 us at the C3 shape
// for a 16 MiB result; this one is write-bound). Needs the row's first entry on a quad boundary (m even) and the row
// in a quarter of 48 KB of LDS (m <= 54 in float32, m <= 38 in float64). `Idt` != nullptr fuses tsde_iterated_integrals: I = 0.5*(W_i W_j - [i==j] dt) + A is
// written instead of A (same operation order as the two kernels in sequence), saving A's round trip through HBM.
constexpr int kLevyRowsPerBlock = kBlock / 64;

template <typename T>
__global__ void __launch_bounds__(kBlock) levy_area_rows_kernel(T* __restrict__ out, const T* __restrict__ W,
                                                                const T* __restrict__ H, int64_t B, int m, double h,
                                                                int foster, NoiseKey key, const uint64_t* key_dev,
                                                                uint32_t cell, uint64_t node, int fuse, T dt, int ito) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  T* lds = reinterpret_cast<T*>(lds_raw);
  if (key_dev != nullptr) {
    const uint64_t e = *key_dev;
    key.k0 = (uint32_t)e;
    key.k1 = (uint32_t)(e >> 32);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mm = m * m;
  T* nrow = lds + (int64_t)wave * (mm + 2 * m);   // [mm normals | m of W | m of H]
  T* wrow = nrow + mm;
  T* hrow = wrow + m;
  const T tenth_h = (T)(0.1 * h);
  const T davie_std = (T)sqrt(h * h / 12.0);
  for (int64_t row0 = (int64_t)blockIdx.x * kLevyRowsPerBlock; row0 < B; row0 += (int64_t)gridDim.x * kLevyRowsPerBlock) {
    const int64_t b = row0 + wave;
    const bool live = b < B;
    if (live) {
      const uint64_t base = key.elem0 * (uint64_t)m + (uint64_t)b * (uint64_t)mm;   // % 4 == 0 (checked by the launcher)
      for (int q = lane; q < mm / 4; q += 64) {
        T n[4];
        normal4<T>(key, (base >> 2) + (uint64_t)q, cell, node, kStreamA, n);
#pragma unroll
        for (int c = 0; c < 4; ++c) nrow[4 * q + c] = n[c];
      }
      for (int i = lane; i < m; i += 64) {
        wrow[i] = W[b * m + i];
        hrow[i] = H[b * m + i];
      }
    }
    __syncthreads();
    if (live) {
      for (int t = lane; t < mm; t += 64) {
        const int i = t / m, j = t - i * m;
        const T Wi = wrow[i], Wj = wrow[j], Hi = hrow[i], Hj = hrow[j];
        T a = Hi * Wj - Wi * Hj;
        if (i != j) {
          const T sd = foster ? (T)sqrt((double)(tenth_h * (tenth_h + (Hi * Hi + Hj * Hj)))) : davie_std;
          a += sd * (nrow[t] - nrow[j * m + i]);
        }
        if (fuse) {
          T v = Wi * Wj;
          if (ito && i == j) v = v - dt;
          a = (T)0.5 * v + a;
        }
        out[b * (int64_t)mm + t] = a;
      }
    }
    __syncthreads();
  }
}

// ---- iterated integrals for general-noise Milstein (extension, SURVEY.md section 8 note N1) ----------------------
//   I[b,k,l] = 0.5*(W_k W_l - [k==l] dt) + A[b,k,l]   (Ito; Stratonovich drops the dt term; A may be absent)
template <typename T>
__global__ void __launch_bounds__(kBlock) iterated_integrals_kernel(T* __restrict__ I, const T* __restrict__ W,
                                                                    const T* __restrict__ A, int64_t B, int64_t m,
                                                                    T dt, int ito) {
  const int64_t total = B * m * m;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t b = t / (m * m);
    const int64_t r = t - b * m * m;
    const int64_t k = r / m, l = r - k * m;
    T v = W[b * m + k] * W[b * m + l];
    if (ito && k == l) v = v - dt;
    v = (T)0.5 * v;
    if (A != nullptr) v = v + A[t];
    I[t] = v;
  }
}

template <typename T>
hipError_t launch_iterated_integrals(void* I, const void* W, const void* A, int64_t B, int64_t m, double dt, int ito,
                                     hipStream_t s) {
  const int64_t total = B * m * m;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(iterated_integrals_kernel<T>, dim3(grid_for(total)), dim3(kBlock), 0, s, (T*)I, (const T*)W,
                     (const T*)A, B, m, (T)dt, ito);
  return hipGetLastError();
}

// `fuse`: write I = 0.5*(W W^T - [diag] dt) + A instead of A (tsde_levy_iterated_integrals). Returns
// hipErrorNotSupported when the fused form is asked for a shape only the per-entry kernel serves.
template <typename T>
hipError_t launch_levy_area(void* A, const void* W, const void* H, int64_t B, int64_t m, double h, int foster,
                            NoiseKey key, const uint64_t* key_dev, uint32_t cell, uint64_t node, hipStream_t s,
                            int fuse, double dt, int ito) {
  const int64_t total = B * m * m;
  if (total <= 0) return hipSuccess;
  static const bool rows_off = [] { const char* e = getenv("TSDE_LEVY_ROWS"); return e && e[0] == '0'; }();
  const size_t lds = (size_t)kLevyRowsPerBlock * (size_t)(m * m + 2 * m) * sizeof(T);
  const bool rows_ok = !rows_off && (m % 2 == 0) && lds <= (48u << 10) && ((key.elem0 * (uint64_t)m) % 4 == 0);
  if (rows_ok) {
    int64_t blocks = (B + kLevyRowsPerBlock - 1) / kLevyRowsPerBlock;
    if (blocks > kMaxGrid * 4) blocks = kMaxGrid * 4;
    hipLaunchKernelGGL(levy_area_rows_kernel<T>, dim3((unsigned)blocks), dim3(kBlock), lds, s, (T*)A, (const T*)W,
                       (const T*)H, B, (int)m, h, foster, key, key_dev, cell, node, fuse, (T)dt, ito);
    return hipGetLastError();
  }
  if (fuse) return hipErrorNotSupported;
  hipLaunchKernelGGL(levy_area_kernel<T>, dim3(grid_for(total)), dim3(kBlock), 0, s, (T*)A, (const T*)W, (const T*)H, B,
                     m, h, foster, key, key_dev, cell, node);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_rheun_z(void* z1, const void* y0, const void* z0, const void* f0, const void* g0, int64_t n, double dt,
                          double sgn, const tsde_noise_t* nz, hipStream_t s) {
  RheunZOp<T> op{(T*)z1, (const T*)y0, (const T*)z0, (const T*)f0, (const T*)g0, (T)dt, (T)sgn, rh_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(z1) && aligned16(y0) && aligned16(z0) && aligned16(f0) && aligned16(g0) &&
                   rh_noise_vec(nz);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_rheun_y(void* y1, const void* y0, const void* f0, const void* f1, const void* g0, const void* g1,
                          int64_t n, double half_dt, double sgn, const tsde_noise_t* nz, hipStream_t s) {
  RheunYOp<T> op{(T*)y1,      (const T*)y0, (const T*)f0, (const T*)f1,   (const T*)g0,
                 (const T*)g1, (T)half_dt,   (T)sgn,       rh_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(y1) && aligned16(y0) && aligned16(f0) && aligned16(f1) && aligned16(g0) &&
                   aligned16(g1) && rh_noise_vec(nz);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_lincomb2(void* out, const void* x, const void* y, int64_t n, double a, double b, hipStream_t s) {
  Lincomb2Op<T> op{(T*)out, (const T*)x, (const T*)y, (T)a, (T)b};
  const bool vec = (n % 4 == 0) && aligned16(out) && aligned16(x) && aligned16(y);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_rheun_adj_a(void* af0_out, void* ag0_out, const void* ay, const void* af0, const void* ag0, int64_t n,
                              double half_dt, const tsde_noise_t* nz, hipStream_t s) {
  RheunAdjAOp<T> op{(T*)af0_out, (T*)ag0_out, (const T*)ay, (const T*)af0, (const T*)ag0, (T)half_dt, rh_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(af0_out) && aligned16(ag0_out) && aligned16(ay) && aligned16(af0) &&
                   aligned16(ag0) && rh_noise_vec(nz);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

template <typename T>
hipError_t launch_rheun_adj_b(void* ay1, void* az1, void* af1, void* ag1, const void* ay, const void* az0,
                              const void* vjp_z, int64_t n, double dt, double half_dt, const tsde_noise_t* nz,
                              hipStream_t s) {
  RheunAdjBOp<T> op{(T*)ay1, (T*)az1, (T*)af1, (T*)ag1, (const T*)ay, (const T*)az0, (const T*)vjp_z,
                    (T)dt,   (T)half_dt, rh_noise<T>(nz)};
  const bool vec = (n % 4 == 0) && aligned16(ay1) && aligned16(az1) && aligned16(af1) && aligned16(ag1) &&
                   aligned16(ay) && aligned16(az0) && aligned16(vjp_z) && rh_noise_vec(nz);
  return launch_elementwise(op, n, vec, s, sizeof(T));
}

#define TSDE_RH_INSTANTIATE(T)                                                                                       \
  template hipError_t launch_heun_final<T>(void*, const void*, const void*, const void*, const void*, const void*,   \
                                           int64_t, double, int, int, const tsde_noise_t*, hipStream_t);             \
  template hipError_t launch_iterated_integrals<T>(void*, const void*, const void*, int64_t, int64_t, double, int,   \
                                                   hipStream_t);                                                     \
  template hipError_t launch_levy_area<T>(void*, const void*, const void*, int64_t, int64_t, double, int, NoiseKey,  \
                                          const uint64_t*, uint32_t, uint64_t, hipStream_t, int, double, int);       \
  template hipError_t launch_rheun_z<T>(void*, const void*, const void*, const void*, const void*, int64_t, double,  \
                                        double, const tsde_noise_t*, hipStream_t);                                   \
  template hipError_t launch_rheun_y<T>(void*, const void*, const void*, const void*, const void*, const void*,      \
                                        int64_t, double, double, const tsde_noise_t*, hipStream_t);                  \
  template hipError_t launch_lincomb2<T>(void*, const void*, const void*, int64_t, double, double, hipStream_t);     \
  template hipError_t launch_rheun_adj_a<T>(void*, void*, const void*, const void*, const void*, int64_t, double,    \
                                            const tsde_noise_t*, hipStream_t);                                       \
  template hipError_t launch_rheun_adj_b<T>(void*, void*, void*, void*, const void*, const void*, const void*,       \
                                            int64_t, double, double, const tsde_noise_t*, hipStream_t);
TSDE_RH_INSTANTIATE(float)
TSDE_RH_INSTANTIATE(double)

}  // namespace tsde
