// Brownian-bridge algebra on (W, H) pairs and the virtual cell tree walk.
//
// The formulas restate the reference's `_Interval._increment_and_space_time_levy_area`
// (torchsde/_brownian/brownian_interval.py:188-241: split of a parent (W,H) into children),
// the sub-interval merge in `BrownianInterval.__call__` (:643-672) and `_H_to_U` (:102-103),
// with the same operation order and the same "coefficients in double, tensors in T" convention
// (the reference computes coefficients as Python floats and multiplies them into tensors).
//
// What is new is WHERE the normals come from: instead of a stored binary tree of seeds plus an
// LRU cache, the Brownian path over a top-level *cell* [s,e] is a virtual dyadic tree whose node
// normals are Philox outputs keyed on (entropy, element, cell, heap index) -- tsde_rng.h -- so
// any sub-interval can be recomputed from nothing, in registers, in any order.
#pragma once
#include "tsde_rng.h"

namespace tsde {

// Four lanes (one Philox quad) of (W,H).
template <typename T>
struct WH4 {
  T W[4];
  T H[4];
};

// Root draw of a cell of width h:  W ~ sqrt(h) N,  H ~ sqrt(h/12) N   (brownian_interval.py:553-558).
template <typename T, bool HAVE_H>
TSDE_D void cell_root(const NoiseKey& key, uint64_t quad, uint32_t cell, double h, WH4<T>& o) {
  T n[4];
  normal4<T>(key, quad, cell, 0, kStreamW, n);
  const T sw = (T)sqrt(h);
#pragma unroll
  for (int j = 0; j < 4; ++j) o.W[j] = n[j] * sw;
  if (HAVE_H) {
    normal4<T>(key, quad, cell, 0, kStreamH, n);
    const T sh = (T)sqrt(h / 12.0);
#pragma unroll
    for (int j = 0; j < 4; ++j) o.H[j] = n[j] * sh;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o.H[j] = (T)0;
  }
}

// Coefficients of one bridge split of [lo,hi] at x. They depend on the three times only, so a wave computes them
// once per tree level (lane k <-> level k, `DescentTable` below) instead of once per lane per level: the double-
// precision divide and square roots were ~45 % of a misaligned query's instructions.
template <typename T, bool HAVE_H>
struct SplitCoef {
  // without H: tl, th, sd.  with H: fcl, scl, third, fl2, ta, cr, fcr, scr, fr2, tb, cl.
  T c[HAVE_H ? 11 : 3];
};

template <typename T, bool HAVE_H>
TSDE_D SplitCoef<T, HAVE_H> split_coef(double lo, double x, double hi) {
  SplitCoef<T, HAVE_H> k;
  const double hrec = 1.0 / (hi - lo);
  const double l = x - lo;
  const double r = hi - x;
  if constexpr (HAVE_H) {
    const double l2 = l * l, r2 = r * r;
    const double l3 = l * l2, r3 = r * r2;
    const double v = 0.5 * sqrt(l * r / (l3 + r3));
    const double a = v * l2 * hrec;
    const double b = v * r2 * hrec;
    const double c = v * 0.57735026918962584;  // 1/sqrt(3)
    const double fl = l * hrec, fr = r * hrec;
    k.c[0] = (T)fl;
    k.c[1] = (T)(6.0 * fl * r * hrec);
    k.c[2] = (T)(2.0 * (a * l + b * r) * hrec);
    k.c[3] = (T)(fl * fl);
    k.c[4] = (T)a;
    k.c[5] = (T)(c * r);
    k.c[6] = (T)fr;
    k.c[7] = (T)(6.0 * fr * l * hrec);
    k.c[8] = (T)(fr * fr);
    k.c[9] = (T)b;
    k.c[10] = (T)(c * l);
  } else {
    k.c[0] = (T)l;
    k.c[1] = (T)hrec;
    k.c[2] = (T)sqrt(l * r * hrec);
  }
  return k;
}

// L, R from the parent P and the node normals X1 (W stream), X2 (H stream).
template <typename T, bool HAVE_H>
TSDE_D void split_apply(const SplitCoef<T, HAVE_H>& k, const T (&X1)[4], const T (&X2)[4], const WH4<T>& P, WH4<T>& L,
                        WH4<T>& R) {
  if constexpr (HAVE_H) {
    const T fcl = k.c[0], scl = k.c[1], third = k.c[2], fl2 = k.c[3], ta = k.c[4], cr = k.c[5];
    const T fcr = k.c[6], scr = k.c[7], fr2 = k.c[8], tb = k.c[9], cl = k.c[10];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T W = P.W[j], H = P.H[j];
      L.W[j] = (fcl * W + scl * H) + third * X1[j];
      L.H[j] = (fl2 * H - ta * X1[j]) + cr * X2[j];
      R.W[j] = (fcr * W - scr * H) - third * X1[j];
      R.H[j] = (fr2 * H - tb * X1[j]) - cl * X2[j];
    }
  } else {
    const T tl = k.c[0], th = k.c[1], sd = k.c[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T W = P.W[j];
      const T wl = (tl * W) * th + sd * X1[j];
      L.W[j] = wl;
      R.W[j] = W - wl;
      L.H[j] = (T)0;
      R.H[j] = (T)0;
    }
  }
}

// Split parent P over [lo,hi] at x into L=[lo,x], R=[x,hi] using the parent's node normals.
template <typename T, bool HAVE_H>
TSDE_D void bridge_split(const NoiseKey& key, uint64_t quad, uint32_t cell, uint64_t node, const SplitCoef<T, HAVE_H>& k,
                         const WH4<T>& P, WH4<T>& L, WH4<T>& R) {
  T X1[4], X2[4] = {(T)0, (T)0, (T)0, (T)0};
  normal4<T>(key, quad, cell, node, kStreamW, X1);
  if constexpr (HAVE_H) normal4<T>(key, quad, cell, node, kStreamH, X2);
  split_apply<T, HAVE_H>(k, X1, X2, P, L, R);
}

// Concatenate A over an interval of length ha with B over the adjacent interval of length hb
// (brownian_interval.py:647-672, left = accumulated, right = new piece).
template <typename T, bool HAVE_H>
TSDE_D void interval_merge(WH4<T>& A, double ha, const WH4<T>& B, double hb) {
  if (HAVE_H) {
    const T tha = (T)ha, thb = (T)hb, tsum = (T)(ha + hb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T term1 = thb * (B.H[j] + (T)0.5 * A.W[j]);
      const T term2 = tha * (A.H[j] - (T)0.5 * B.W[j]);
      A.H[j] = (term1 + term2) / tsum;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) A.W[j] = A.W[j] + B.W[j];
}

// Accumulator for time-ordered pieces; `len == 0` means empty.
template <typename T, bool HAVE_H>
struct PieceAcc {
  WH4<T> v;
  double len;
  TSDE_D void clear() {
    len = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v.W[j] = v.H[j] = (T)0;
  }
  // Append a piece on the right.
  TSDE_D void push_right(const WH4<T>& p, double h) {
    if (len == 0.0) {
      v = p;
    } else {
      interval_merge<T, HAVE_H>(v, len, p, h);
    }
    len += h;
  }
  // Prepend a piece on the left.
  TSDE_D void push_left(const WH4<T>& p, double h) {
    if (len == 0.0) {
      v = p;
    } else {
      WH4<T> q = p;
      interval_merge<T, HAVE_H>(q, h, v, len);
      v = q;
    }
    len += h;
  }
};

struct WalkCfg {
  int max_depth;  // dyadic levels before the leaf rule applies
  int snap;       // leaf rule: 0 = split the leaf exactly at the query point, 1 = snap to the nearer edge
};

// Where the in-cell walk bisects at `depth`: the dyadic midpoint, or (leaf rule, exact mode) the query point itself.
TSDE_D double split_point(int depth, double lo, double hi, double p, const WalkCfg& cfg) {
  return (depth >= cfg.max_depth) ? p : 0.5 * (lo + hi);
}

// Split coefficients along the descent from the cell [s,e] toward the point p, one tree level per table row.
// The descent is a pure function of (s, e, p, cfg) -- the same for every element -- so instead of every lane
// redoing the double-precision divide/sqrt at every level, 64 threads of the block each replay the cheap
// bookkeeping down to their own level and compute that level's coefficients ONCE into LDS; after a barrier the
// walk fetches row `depth` (a wave-uniform, conflict-free broadcast read). Needs max_depth + 1 <= kMaxLevels
// (the C ABI caps max_depth at 40).
//   TOWARD_B = false: the left end a of a query  (go left iff a < x,  as walk_suffix / cell_range do)
//   TOWARD_B = true : the right end b of a query (go right iff b > x, as walk_prefix / cell_range do)
constexpr int kMaxLevels = 64;

template <typename T, bool HAVE_H>
struct DescentTable {
  static constexpr int N = HAVE_H ? 11 : 3;
  T* rows;   // LDS, kMaxLevels x N

  // Called by ALL threads of the block (the caller places the barrier after the last build).
  template <bool TOWARD_B>
  TSDE_D void build(double s, double e, double p, const WalkCfg& cfg) const {
    const int level = (int)threadIdx.x;
    if (level >= kMaxLevels || level > cfg.max_depth) return;
    double lo = s, hi = e;
    for (int d = 0; d < level; ++d) {
      if (TOWARD_B ? (p == hi) : (p == lo)) return;   // the walk stops here: deeper rows are never read
      const double x = split_point(d, lo, hi, p, cfg);
      const bool left = TOWARD_B ? !(p > x) : (p < x);
      if (left) hi = x; else lo = x;
    }
    if (TOWARD_B ? (p == hi) : (p == lo)) return;
    const SplitCoef<T, HAVE_H> k = split_coef<T, HAVE_H>(lo, split_point(level, lo, hi, p, cfg), hi);
#pragma unroll
    for (int i = 0; i < N; ++i) rows[level * N + i] = k.c[i];
  }

  // Coefficients of level `depth` (wave-uniform index).
  TSDE_D SplitCoef<T, HAVE_H> at(int depth) const {
    SplitCoef<T, HAVE_H> k;
#pragma unroll
    for (int i = 0; i < N; ++i) k.c[i] = rows[depth * N + i];
    return k;
  }
};

// (W,H) of [a, hi] inside node `node`=[lo,hi] whose value is P. Pieces are prepended to acc.
template <typename T, bool HAVE_H>
TSDE_D void walk_suffix(const NoiseKey& key, uint64_t quad, uint32_t cell, uint64_t node, int depth, double lo,
                        double hi, double a, WH4<T> P, const WalkCfg& cfg, const DescentTable<T, HAVE_H>& ta,
                        PieceAcc<T, HAVE_H>& acc) {
  for (;;) {
    if (a == lo) {
      acc.push_left(P, hi - lo);
      return;
    }
    if (depth >= cfg.max_depth && cfg.snap) {
      if ((a - lo) < (hi - a)) acc.push_left(P, hi - lo);
      return;
    }
    const double x = split_point(depth, lo, hi, a, cfg);
    WH4<T> L, R;
    bridge_split<T, HAVE_H>(key, quad, cell, node, ta.at(depth), P, L, R);
    ++depth;
    if (a < x) {
      acc.push_left(R, hi - x);
      P = L;
      hi = x;
      node = 2 * node;
    } else {
      P = R;
      lo = x;
      node = 2 * node + 1;
    }
  }
}

// (W,H) of [lo, b] inside node=[lo,hi]. Pieces are appended to acc.
template <typename T, bool HAVE_H>
TSDE_D void walk_prefix(const NoiseKey& key, uint64_t quad, uint32_t cell, uint64_t node, int depth, double lo,
                        double hi, double b, WH4<T> P, const WalkCfg& cfg, const DescentTable<T, HAVE_H>& tb,
                        PieceAcc<T, HAVE_H>& acc) {
  for (;;) {
    if (b == hi) {
      acc.push_right(P, hi - lo);
      return;
    }
    if (depth >= cfg.max_depth && cfg.snap) {
      if ((hi - b) <= (b - lo)) acc.push_right(P, hi - lo);
      return;
    }
    const double x = split_point(depth, lo, hi, b, cfg);
    WH4<T> L, R;
    bridge_split<T, HAVE_H>(key, quad, cell, node, tb.at(depth), P, L, R);
    ++depth;
    if (b > x) {
      acc.push_right(L, x - lo);
      P = R;
      lo = x;
      node = 2 * node + 1;
    } else {
      P = L;
      hi = x;
      node = 2 * node;
    }
  }
}

// (W,H) of [a,b] inside the cell [s,e]  (s <= a < b <= e) whose root value is P. Appends to acc (time-ordered).
// `ta` / `tb`: descent tables of this cell toward a / toward b (a table whose end point is a cell edge is never read).
template <typename T, bool HAVE_H>
TSDE_D void cell_range(const NoiseKey& key, uint64_t quad, uint32_t cell, double s, double e, double a, double b,
                       WH4<T> P, const WalkCfg& cfg, const DescentTable<T, HAVE_H>& ta,
                       const DescentTable<T, HAVE_H>& tb, PieceAcc<T, HAVE_H>& acc) {
  uint64_t node = 1;
  int depth = 0;
  double lo = s, hi = e;
  for (;;) {
    if (a == lo && b == hi) {
      acc.push_right(P, hi - lo);
      return;
    }
    if (depth >= cfg.max_depth && cfg.snap) {
      const bool a_lo = (a - lo) < (hi - a);
      const bool b_hi = (hi - b) <= (b - lo);
      if (a_lo && b_hi) acc.push_right(P, hi - lo);
      return;
    }
    // Above the fork both descents visit this node; below max_depth they also split it at the same point.
    // At the leaf rule the split point is a if a is interior, else b.
    const bool use_a = a > lo;
    const double x = split_point(depth, lo, hi, use_a ? a : b, cfg);
    WH4<T> L, R;
    bridge_split<T, HAVE_H>(key, quad, cell, node, use_a ? ta.at(depth) : tb.at(depth), P, L, R);
    ++depth;
    if (b <= x) {
      P = L;
      hi = x;
      node = 2 * node;
    } else if (a >= x) {
      P = R;
      lo = x;
      node = 2 * node + 1;
    } else {
      PieceAcc<T, HAVE_H> left;
      left.clear();
      walk_suffix<T, HAVE_H>(key, quad, cell, 2 * node, depth, lo, x, a, L, cfg, ta, left);
      if (left.len != 0.0) acc.push_right(left.v, left.len);
      walk_prefix<T, HAVE_H>(key, quad, cell, 2 * node + 1, depth, x, hi, b, R, cfg, tb, acc);
      return;
    }
  }
}

}  // namespace tsde
