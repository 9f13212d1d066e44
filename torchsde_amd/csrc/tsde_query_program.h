// The Brownian query as a straight-line PROGRAM.
//
// Which tree nodes a query [a, b] visits, where each is split, which pieces are kept and with what weights they are
// joined depends on the query times and the cell grid only -- it is the same for every element of the batch. The walk
// of tsde_bridge.h (cell_range / walk_suffix / walk_prefix, kept as the checker of this file) nevertheless redoes that
// bookkeeping in every lane at every level: double-precision midpoints, comparisons, piece lengths, their conversions
// for the merge formula -- 22-27 % on top of the Philox + Box-Muller work that the generator's definition makes
// unavoidable (profiles/r3a_microbench_rng_node_visit.txt). Here one thread of the block walks the tree ONCE, without
// touching any random number, and writes down what the lanes have to do as a list of operations in LDS (the split
// coefficients of an operation, the expensive part, are then computed by one thread per operation); every lane runs that
// list: per operation one broadcast LDS read, the node's normals, the split, a merge. Same splits, same merges, same
// order, same coefficient arithmetic as the walk: bit-identical results (tests/test_gpu_query_program.py compares the
// two kernels; tests/golden/query_kernel_r1.pt pins both).
#pragma once
#include "tsde_bridge.h"

namespace tsde {

constexpr int kMaxOps = 192;   // 3 roots + 2 * (kMaxLevels/2 + 10) splits + pushes: max_depth <= 40 needs < 140

enum : uint32_t {
  kOpRoot = 1u,        // P = root draw of cell `cell` (c[0] = sqrt(h), c[1] = sqrt(h/12))
  kOpSplit = 2u,       // (L, R) = split of P at this node; keep one child in P, drop / push / save the other
  kOpPushP = 3u,       // push P whole
  kOpLeftToAcc = 4u,   // acc.push_right(left)
  kOpRestore = 5u,     // P = the child saved at the fork
  kOpCodeMask = 0xFu,
  kOpKeepRight = 1u << 8,    // split: P = R (else P = L)
  kOpOtherPush = 1u << 9,    // split: the other child is pushed
  kOpOtherSave = 1u << 10,   // split: the other child is saved for kOpRestore
  kOpToLeft = 1u << 11,      // push target: the `left` accumulator, prepending (push_left); else `acc`, appending
  kOpEmpty = 1u << 12,       // push: the target accumulator is empty (the piece is copied, not merged)
  kOpPinned = 1u << 13,      // root: overwrite the draw with the user's pinned (W, H)
};

// One operation as the lanes read it: 16-byte aligned, so that a row is a few ds_read_b128 (wave-uniform address:
// broadcast reads, no bank conflicts).
template <typename T, bool HAVE_H>
struct alignas(16) OpRow {
  static constexpr int NC = HAVE_H ? 11 : 3;
  uint32_t code, cell, node_lo, node_hi;
  T c[NC];     // split coefficients (root: c[0] = sqrt(h), c[1] = sqrt(h/12))
  T m[3];      // (T)ha, (T)hb, (T)(ha + hb) of interval_merge(A, ha, B, hb)
};

template <typename T, bool HAVE_H>
struct QueryProgram {
  static constexpr int NC = HAVE_H ? 11 : 3;
  OpRow<T, HAVE_H>* rows;   // [kMaxOps]
  double* times;            // [kMaxOps][3]: (lo, x, hi) of a split, (h, -, -) of a root -- input of the coefficient pass
  int* n_ops;               // [1]
};

// The builder's scratch (LDS, doubles): what the skeleton pass leaves for the coefficient pass.
struct ProgramScratch {
  uint32_t* code;
  uint32_t* cell;
  uint64_t* node;
  double* merge64;
};

// Host-of-the-block side: the skeleton. Mirrors cell_range / walk_suffix / walk_prefix / PieceAcc statement by
// statement, with the random-number work replaced by "emit an operation".
struct ProgramBuilder {
  uint32_t* code;
  uint32_t* cell;
  uint64_t* node;
  double* times;
  double* merge64;   // [kMaxOps][3] as doubles here; converted to T in the coefficient pass
  int n;
  double acc_len, left_len;

  TSDE_D int emit(uint32_t c, uint32_t cl, uint64_t nd, double t0, double t1, double t2) {
    const int i = n < kMaxOps ? n : kMaxOps - 1;      // (cannot overflow for max_depth <= 40; stay in bounds regardless)
    code[i] = c;
    cell[i] = cl;
    node[i] = nd;
    times[3 * i] = t0;
    times[3 * i + 1] = t1;
    times[3 * i + 2] = t2;
    merge64[3 * i] = merge64[3 * i + 1] = merge64[3 * i + 2] = 0.0;
    n = i + 1;
    return i;
  }
  // PieceAcc::push_right(p, h) on `acc` / push_left(p, h) on `left`: flag bits and merge scalars of operation i
  TSDE_D void push(int i, bool to_left, double h) {
    double& len = to_left ? left_len : acc_len;
    uint32_t f = to_left ? kOpToLeft : 0u;
    if (len == 0.0) {
      f |= kOpEmpty;
    } else if (to_left) {      // interval_merge(q = p, h, v, len)
      merge64[3 * i] = h;
      merge64[3 * i + 1] = len;
      merge64[3 * i + 2] = h + len;
    } else {                   // interval_merge(v, len, p, h)
      merge64[3 * i] = len;
      merge64[3 * i + 1] = h;
      merge64[3 * i + 2] = len + h;
    }
    code[i] |= f;
    len += h;
  }
  TSDE_D void push_p(uint32_t cl, bool to_left, double h) {
    const int i = emit(kOpPushP, cl, 0, 0.0, 0.0, 0.0);
    push(i, to_left, h);
  }

  // walk_suffix: (W,H) of [a, hi] inside node=[lo,hi]; pieces prepended to `left` (to_left) or `acc`.
  TSDE_D void suffix(uint32_t cl, uint64_t nd, int depth, double lo, double hi, double a, const WalkCfg& cfg,
                     bool to_left) {
    for (;;) {
      if (a == lo) {
        push_p(cl, to_left, hi - lo);
        return;
      }
      if (depth >= cfg.max_depth && cfg.snap) {
        if ((a - lo) < (hi - a)) push_p(cl, to_left, hi - lo);
        return;
      }
      const double x = split_point(depth, lo, hi, a, cfg);
      ++depth;
      if (a < x) {      // keep L, push R = [x, hi]
        const int i = emit(kOpSplit | kOpOtherPush, cl, nd, lo, x, hi);
        push(i, to_left, hi - x);
        hi = x;
        nd = 2 * nd;
      } else {
        emit(kOpSplit | kOpKeepRight, cl, nd, lo, x, hi);
        lo = x;
        nd = 2 * nd + 1;
      }
    }
  }
  // walk_prefix: (W,H) of [lo, b] inside node=[lo,hi]; pieces appended to `acc`.
  TSDE_D void prefix(uint32_t cl, uint64_t nd, int depth, double lo, double hi, double b, const WalkCfg& cfg) {
    for (;;) {
      if (b == hi) {
        push_p(cl, false, hi - lo);
        return;
      }
      if (depth >= cfg.max_depth && cfg.snap) {
        if ((hi - b) <= (b - lo)) push_p(cl, false, hi - lo);
        return;
      }
      const double x = split_point(depth, lo, hi, b, cfg);
      ++depth;
      if (b > x) {      // keep R, push L = [lo, x]
        const int i = emit(kOpSplit | kOpKeepRight | kOpOtherPush, cl, nd, lo, x, hi);
        push(i, false, x - lo);
        lo = x;
        nd = 2 * nd + 1;
      } else {
        emit(kOpSplit, cl, nd, lo, x, hi);
        hi = x;
        nd = 2 * nd;
      }
    }
  }
  // cell_range: (W,H) of [a,b] inside the cell [s,e] whose root value is in P; appends to `acc`.
  TSDE_D void range(uint32_t cl, double s, double e, double a, double b, const WalkCfg& cfg) {
    uint64_t nd = 1;
    int depth = 0;
    double lo = s, hi = e;
    for (;;) {
      if (a == lo && b == hi) {
        push_p(cl, false, hi - lo);
        return;
      }
      if (depth >= cfg.max_depth && cfg.snap) {
        const bool a_lo = (a - lo) < (hi - a);
        const bool b_hi = (hi - b) <= (b - lo);
        if (a_lo && b_hi) push_p(cl, false, hi - lo);
        return;
      }
      const bool use_a = a > lo;
      const double x = split_point(depth, lo, hi, use_a ? a : b, cfg);
      ++depth;
      if (b <= x) {
        emit(kOpSplit, cl, nd, lo, x, hi);
        hi = x;
        nd = 2 * nd;
      } else if (a >= x) {
        emit(kOpSplit | kOpKeepRight, cl, nd, lo, x, hi);
        lo = x;
        nd = 2 * nd + 1;
      } else {
        // the fork: P = L for the suffix walk into a fresh `left`, R is saved for the prefix walk
        emit(kOpSplit | kOpOtherSave, cl, nd, lo, x, hi);
        left_len = 0.0;
        suffix(cl, 2 * nd, depth, lo, x, a, cfg, true);
        if (left_len != 0.0) {
          const int i = emit(kOpLeftToAcc, cl, 0, 0.0, 0.0, 0.0);
          const double h = left_len;
          push(i, false, h);          // acc.push_right(left.v, left.len)
          code[i] &= ~kOpToLeft;
        }
        emit(kOpRestore, cl, 0, 0.0, 0.0, 0.0);
        prefix(cl, 2 * nd + 1, depth, x, hi, b, cfg);
        return;
      }
    }
  }
};

// Runs in ALL threads of the block. `mid_lo .. mid_hi`: the whole cells strictly between the end cells are NOT part of
// the program (there may be thousands): the kernel loops over them between the two halves; `split_at` receives the
// index of the first operation of the last cell's part (n_ops if the query lives in one cell).
template <typename T, bool HAVE_H>
TSDE_D void build_query_program(const QueryProgram<T, HAVE_H>& pg, const ProgramScratch& sc,
                                const double* __restrict__ edges, int64_t ca, int64_t cb, double a, double b,
                                const WalkCfg& cfg, bool pinned, int* split_at, double* acc_len_mid) {
  if (threadIdx.x == 0) {
    ProgramBuilder pb{sc.code, sc.cell, sc.node, pg.times, sc.merge64, 0, 0.0, 0.0};
    const double s = edges[ca], e = edges[ca + 1];
    pb.emit(kOpRoot | (pinned ? kOpPinned : 0u), (uint32_t)ca, 0, e - s, 0.0, 0.0);
    if (ca == cb) {
      pb.range((uint32_t)ca, s, e, a, b, cfg);
      *split_at = pb.n;
      *acc_len_mid = pb.acc_len;
    } else {
      pb.range((uint32_t)ca, s, e, a, e, cfg);
      *split_at = pb.n;
      *acc_len_mid = pb.acc_len;
      // the middle cells are merged by the kernel's own loop: account for their lengths here
      for (int64_t c = ca + 1; c < cb; ++c) pb.acc_len += edges[c + 1] - edges[c];
      const double s2 = edges[cb], e2 = edges[cb + 1];
      pb.emit(kOpRoot, (uint32_t)cb, 0, e2 - s2, 0.0, 0.0);
      pb.range((uint32_t)cb, s2, e2, s2, b, cfg);
    }
    *pg.n_ops = pb.n;
  }
  __syncthreads();
  const int n = *pg.n_ops;
  constexpr int NC = QueryProgram<T, HAVE_H>::NC;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    OpRow<T, HAVE_H> row;
    row.code = sc.code[i];
    row.cell = sc.cell[i];
    row.node_lo = (uint32_t)sc.node[i];
    row.node_hi = (uint32_t)(sc.node[i] >> 32);
#pragma unroll
    for (int j = 0; j < NC; ++j) row.c[j] = (T)0;
    const uint32_t c = row.code & kOpCodeMask;
    if (c == kOpSplit) {
      const SplitCoef<T, HAVE_H> k = split_coef<T, HAVE_H>(pg.times[3 * i], pg.times[3 * i + 1], pg.times[3 * i + 2]);
#pragma unroll
      for (int j = 0; j < NC; ++j) row.c[j] = k.c[j];
    } else if (c == kOpRoot) {
      const double h = pg.times[3 * i];
      row.c[0] = (T)sqrt(h);
      row.c[1] = (T)sqrt(h / 12.0);
    }
    row.m[0] = (T)sc.merge64[3 * i];
    row.m[1] = (T)sc.merge64[3 * i + 1];
    row.m[2] = (T)sc.merge64[3 * i + 2];
    pg.rows[i] = row;
  }
  __syncthreads();
}

// interval_merge with the three converted lengths read from the program: A = A (+) B.
template <typename T, bool HAVE_H>
TSDE_D void merge_with(WH4<T>& A, const WH4<T>& B, const T* __restrict__ m) {
  if (HAVE_H) {
    const T tha = m[0], thb = m[1], tsum = m[2];
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

// One push of `piece`: into `left` (prepending) or `acc` (appending), as PieceAcc does.
template <typename T, bool HAVE_H>
TSDE_D void program_push(uint32_t flags, const T* __restrict__ m, const WH4<T>& piece, WH4<T>& acc, WH4<T>& left) {
  if (flags & kOpToLeft) {
    if (flags & kOpEmpty) {
      left = piece;
    } else {                 // q = p; interval_merge(q, h, v, len); v = q
      WH4<T> q = piece;
      merge_with<T, HAVE_H>(q, left, m);
      left = q;
    }
  } else {
    if (flags & kOpEmpty) {
      acc = piece;
    } else {
      merge_with<T, HAVE_H>(acc, piece, m);
    }
  }
}

// Operations [first, last) of the program for one Philox quad. The next row is fetched while the current one runs.
template <typename T, bool HAVE_H>
TSDE_D void run_query_program(const QueryProgram<T, HAVE_H>& pg, int first, int last, const NoiseKey& key, uint64_t quad,
                              const T* __restrict__ pinW, const T* __restrict__ pinH, int64_t i0, int64_t n,
                              WH4<T>& P, WH4<T>& saved, WH4<T>& acc, WH4<T>& left) {
  constexpr int NC = QueryProgram<T, HAVE_H>::NC;
  if (first >= last) return;
  OpRow<T, HAVE_H> next = pg.rows[first];
  for (int i = first; i < last; ++i) {
    const OpRow<T, HAVE_H> row = next;
    if (i + 1 < last) next = pg.rows[i + 1];
    // wave-uniform by construction: into scalar registers, so that the control flow below is scalar branches
    const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)row.code);
    const uint32_t c = flags & kOpCodeMask;
    if (c == kOpSplit) {
      const uint32_t cl = (uint32_t)__builtin_amdgcn_readfirstlane((int)row.cell);
      const uint64_t nd = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)row.node_hi) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)row.node_lo);
      SplitCoef<T, HAVE_H> k;
#pragma unroll
      for (int j = 0; j < NC; ++j) k.c[j] = row.c[j];
      WH4<T> L, R;
      bridge_split<T, HAVE_H>(key, quad, cl, nd, k, P, L, R);
      if (flags & kOpKeepRight) {
        if (flags & kOpOtherPush) program_push<T, HAVE_H>(flags, row.m, L, acc, left);
        P = R;
      } else {
        if (flags & kOpOtherPush) program_push<T, HAVE_H>(flags, row.m, R, acc, left);
        if (flags & kOpOtherSave) saved = R;
        P = L;
      }
    } else if (c == kOpPushP) {
      program_push<T, HAVE_H>(flags, row.m, P, acc, left);
    } else if (c == kOpLeftToAcc) {
      const WH4<T> piece = left;
      program_push<T, HAVE_H>(flags, row.m, piece, acc, left);
    } else if (c == kOpRestore) {
      P = saved;
    } else if (c == kOpRoot) {
      const uint32_t cl = (uint32_t)__builtin_amdgcn_readfirstlane((int)row.cell);
      const T sw = row.c[0], sh = row.c[1];
      T nrm[4];
      normal4<T>(key, quad, cl, 0, kStreamW, nrm);
#pragma unroll
      for (int j = 0; j < 4; ++j) P.W[j] = nrm[j] * sw;
      if (HAVE_H) {
        normal4<T>(key, quad, cl, 0, kStreamH, nrm);
#pragma unroll
        for (int j = 0; j < 4; ++j) P.H[j] = nrm[j] * sh;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) P.H[j] = (T)0;
      }
      if ((flags & kOpPinned) && pinW != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t e = i0 + j;
          if (e >= 0 && e < n) {
            P.W[j] = pinW[e];
            if (HAVE_H && pinH) P.H[j] = pinH[e];
          }
        }
      }
    }
  }
}

}  // namespace tsde
