/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY. Nothing under torchsde_amd/ may import, link or call this.
 *
 * Plain-C CPU twin of the counter-RNG Brownian generator (torchsde_amd/csrc/tsde_rng.h,
 * tsde_bridge.h, brownian.hip), written independently (scalar, recursive) so that the HIP kernels
 * can be checked against it:
 *
 *   - Philox-4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3",
 *     SC'11 (Random123 v1.14 `philox4x32_R(10, ...)`). Pinned by Random123's known-answer vectors
 *     (tests/test_oracle_rng.py). Bit-exact contract with the GPU.
 *   - Box-Muller in double precision from the canonical definition
 *         u1 = (a + 0.5) / 2^32,  theta = 2 pi b / 2^32,  n0 = r cos(theta), n1 = r sin(theta).
 *     The GPU evaluates the same definition with fp32 hardware transcendentals: tolerance contract.
 *   - Brownian bridge split / merge / H->U: restated from the reference,
 *     torchsde/_brownian/brownian_interval.py:188-241 (split), :643-672 (merge), :102-103 (H->U),
 *     :553-558 (top-level draw). Pinned against the reference itself by tests/golden/bridge_*.npz
 *     (tests/test_oracle_bridge.py feeds the reference's own normals through orc_bridge_split).
 *
 * Tensor arithmetic is done in REAL (float or double) with coefficients computed in double and cast
 * at use, mirroring `python_float * tensor` in the reference.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* ------------------------------------------------------------------ Philox ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    if (round > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    uint64_t prod0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    uint64_t prod1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    uint32_t n0 = (uint32_t)(prod1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)prod1;
    uint32_t n2 = (uint32_t)(prod0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)prod0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* counter layout, see tsde_rng.h */
void orc_noise_counter(uint64_t quad, uint32_t cell, uint64_t node, uint32_t stream, uint32_t out[4]) {
  out[0] = (uint32_t)(quad & 0xFFFFFFFFu);
  out[1] = cell;
  out[2] = (uint32_t)(node & 0xFFFFFFFFu);
  out[3] = (stream << 30) | ((uint32_t)((quad >> 32) & 0xFFFFFu) << 10) | (uint32_t)((node >> 32) & 0x3FFu);
}

/* standard normal of one global element, double precision */
double orc_normal(uint64_t entropy, uint64_t elem, uint32_t cell, uint64_t node, uint32_t stream) {
  uint32_t ctr[4], key[2], r[4];
  key[0] = (uint32_t)(entropy & 0xFFFFFFFFu);
  key[1] = (uint32_t)(entropy >> 32);
  orc_noise_counter(elem >> 2, cell, node, stream, ctr);
  orc_philox4x32_10(ctr, key, r);
  unsigned lane = (unsigned)(elem & 3u);
  uint32_t a = (lane & 2u) ? r[2] : r[0];
  uint32_t b = (lane & 2u) ? r[3] : r[1];
  double u1 = ((double)a + 0.5) / 4294967296.0;
  double theta = 6.283185307179586476925286766559 * ((double)b / 4294967296.0);
  double rad = sqrt(-2.0 * log(u1));
  return (lane & 1u) ? rad * sin(theta) : rad * cos(theta);
}

void orc_normals(double* out, int64_t n, uint64_t entropy, uint64_t elem0, uint32_t cell, uint64_t node,
                 uint32_t stream) {
  for (int64_t i = 0; i < n; ++i) out[i] = orc_normal(entropy, elem0 + (uint64_t)i, cell, node, stream);
}

/* ------------------------------------------------------------ bridge algebra ---- */
#define DEFINE_BRIDGE(REAL, SUF)                                                                            \
  /* children (Wl,Hl),(Wr,Hr) of parent (W,H) over [lo,hi] split at x, with normals X1, X2 */             \
  void orc_bridge_split_##SUF(REAL W, REAL H, double lo, double x, double hi, REAL X1, REAL X2, int have_h, \
                              REAL* Wl, REAL* Hl, REAL* Wr, REAL* Hr) {                                    \
    double h_reciprocal = 1.0 / (hi - lo);                                                                 \
    double left_diff = x - lo, right_diff = hi - x;                                                        \
    if (have_h) {                                                                                          \
      double l2 = left_diff * left_diff, r2 = right_diff * right_diff;                                     \
      double l3 = left_diff * l2, r3 = right_diff * r2;                                                    \
      double v = 0.5 * sqrt(left_diff * right_diff / (l3 + r3));                                           \
      double a = v * l2 * h_reciprocal, b = v * r2 * h_reciprocal, c = v * (1.0 / sqrt(3.0));              \
      double third_coeff = 2.0 * (a * left_diff + b * right_diff) * h_reciprocal;                          \
      {                                                                                                    \
        double first_coeff = left_diff * h_reciprocal;                                                     \
        double second_coeff = 6.0 * first_coeff * right_diff * h_reciprocal;                               \
        REAL t1 = (REAL)first_coeff * W, t2 = (REAL)second_coeff * H, t3 = (REAL)third_coeff * X1;         \
        *Wl = (t1 + t2) + t3;                                                                              \
        REAL u1 = (REAL)(first_coeff * first_coeff) * H, u2 = (REAL)a * X1, u3 = (REAL)(c * right_diff) * X2; \
        *Hl = (u1 - u2) + u3;                                                                              \
      }                                                                                                    \
      {                                                                                                    \
        double first_coeff = right_diff * h_reciprocal;                                                    \
        double second_coeff = 6.0 * first_coeff * left_diff * h_reciprocal;                                \
        REAL t1 = (REAL)first_coeff * W, t2 = (REAL)second_coeff * H, t3 = (REAL)third_coeff * X1;         \
        *Wr = (t1 - t2) - t3;                                                                              \
        REAL u1 = (REAL)(first_coeff * first_coeff) * H, u2 = (REAL)b * X1, u3 = (REAL)(c * left_diff) * X2; \
        *Hr = (u1 - u2) - u3;                                                                              \
      }                                                                                                    \
    } else {                                                                                               \
      REAL mean = ((REAL)left_diff * W) * (REAL)h_reciprocal;                                              \
      double var = left_diff * right_diff * h_reciprocal;                                                  \
      REAL left_W = mean + (REAL)sqrt(var) * X1;                                                           \
      *Wl = left_W;                                                                                        \
      *Wr = W - left_W;                                                                                    \
      *Hl = (REAL)0;                                                                                       \
      *Hr = (REAL)0;                                                                                       \
    }                                                                                                      \
  }                                                                                                        \
  /* (W,H) over [s,u] merged with (Wi,Hi) over [u,t] -> over [s,t]; lengths ha = u-s, hb = t-u */          \
  void orc_interval_merge_##SUF(REAL* W, REAL* H, double ha, REAL Wi, REAL Hi, double hb, int have_h) {    \
    if (have_h) {                                                                                          \
      REAL term1 = (REAL)hb * (Hi + (REAL)0.5 * (*W));                                                     \
      REAL term2 = (REAL)ha * (*H - (REAL)0.5 * Wi);                                                       \
      *H = (term1 + term2) / (REAL)(ha + hb);                                                              \
    }                                                                                                      \
    *W = *W + Wi;                                                                                          \
  }                                                                                                        \
                                                                                                           \
  typedef struct {                                                                                         \
    uint64_t entropy, elem;                                                                                \
    uint32_t cell;                                                                                         \
    int have_h, max_depth, snap;                                                                           \
    REAL W, H;                                                                                             \
    double len;                                                                                            \
  } walk_##SUF;                                                                                            \
                                                                                                           \
  static void push_##SUF(walk_##SUF* w, REAL pW, REAL pH, double h) {                                      \
    if (w->len == 0.0) {                                                                                   \
      w->W = pW;                                                                                           \
      w->H = pH;                                                                                           \
    } else {                                                                                               \
      orc_interval_merge_##SUF(&w->W, &w->H, w->len, pW, pH, h, w->have_h);                                \
    }                                                                                                      \
    w->len += h;                                                                                           \
  }                                                                                                        \
                                                                                                           \
  /* append the pieces of [a,b] inside node=[lo,hi] (value PW,PH) in time order */                         \
  static void range_##SUF(walk_##SUF* w, uint64_t node, int depth, double lo, double hi, REAL PW, REAL PH,  \
                          double a, double b) {                                                            \
    if (a == lo && b == hi) {                                                                              \
      push_##SUF(w, PW, PH, hi - lo);                                                                      \
      return;                                                                                              \
    }                                                                                                      \
    double x;                                                                                              \
    if (depth >= w->max_depth) {                                                                           \
      if (w->snap) {                                                                                       \
        int a_at_lo = (a == lo) || ((a - lo) < (hi - a));                                                  \
        int b_at_hi = (b == hi) || ((hi - b) <= (b - lo));                                                 \
        if (a_at_lo && b_at_hi) push_##SUF(w, PW, PH, hi - lo);                                            \
        return;                                                                                            \
      }                                                                                                    \
      x = (a > lo) ? a : b;                                                                                \
    } else {                                                                                               \
      x = 0.5 * (lo + hi);                                                                                 \
    }                                                                                                      \
    REAL X1 = (REAL)orc_normal(w->entropy, w->elem, w->cell, node, 0);                                     \
    REAL X2 = w->have_h ? (REAL)orc_normal(w->entropy, w->elem, w->cell, node, 1) : (REAL)0;               \
    REAL Wl, Hl, Wr, Hr;                                                                                   \
    orc_bridge_split_##SUF(PW, PH, lo, x, hi, X1, X2, w->have_h, &Wl, &Hl, &Wr, &Hr);                      \
    if (b <= x) {                                                                                          \
      range_##SUF(w, 2 * node, depth + 1, lo, x, Wl, Hl, a, b);                                            \
    } else if (a >= x) {                                                                                   \
      range_##SUF(w, 2 * node + 1, depth + 1, x, hi, Wr, Hr, a, b);                                        \
    } else {                                                                                               \
      range_##SUF(w, 2 * node, depth + 1, lo, x, Wl, Hl, a, x);                                            \
      range_##SUF(w, 2 * node + 1, depth + 1, x, hi, Wr, Hr, x, b);                                        \
    }                                                                                                      \
  }                                                                                                        \
                                                                                                           \
  static void root_##SUF(uint64_t entropy, uint64_t elem, uint32_t cell, double h, int have_h, REAL* W,    \
                         REAL* H) {                                                                        \
    *W = (REAL)orc_normal(entropy, elem, cell, 0, 0) * (REAL)sqrt(h);                                      \
    *H = have_h ? (REAL)orc_normal(entropy, elem, cell, 0, 1) * (REAL)sqrt(h / 12.0) : (REAL)0;            \
  }                                                                                                        \
                                                                                                           \
  /* Increment over [a,b]: W, U = h (W/2 + H), H. edges: host array of n_cells+1 doubles. */               \
  void orc_query_##SUF(REAL* W, REAL* U, REAL* H, int64_t n, uint64_t entropy, uint64_t elem0,             \
                       const double* edges, int64_t ca, int64_t cb, double a, double b, const REAL* rootW, \
                       const REAL* rootH, int have_h, int max_depth, int snap) {                           \
    for (int64_t i = 0; i < n; ++i) {                                                                      \
      walk_##SUF w;                                                                                        \
      w.entropy = entropy;                                                                                 \
      w.elem = elem0 + (uint64_t)i;                                                                        \
      w.have_h = have_h;                                                                                   \
      w.max_depth = max_depth;                                                                             \
      w.snap = snap;                                                                                       \
      w.W = w.H = (REAL)0;                                                                                 \
      w.len = 0.0;                                                                                         \
      for (int64_t c = ca; c <= cb; ++c) {                                                                 \
        double s = edges[c], e = edges[c + 1];                                                             \
        REAL PW, PH;                                                                                       \
        root_##SUF(entropy, w.elem, (uint32_t)c, e - s, have_h, &PW, &PH);                                 \
        if (c == ca && rootW) {                                                                            \
          PW = rootW[i];                                                                                   \
          if (have_h && rootH) PH = rootH[i];                                                              \
        }                                                                                                  \
        double qa = (c == ca) ? a : s, qb = (c == cb) ? b : e;                                             \
        w.cell = (uint32_t)c;                                                                              \
        range_##SUF(&w, 1, 0, s, e, PW, PH, qa, qb);                                                       \
      }                                                                                                    \
      W[i] = w.W;                                                                                          \
      if (H) H[i] = w.H;                                                                                   \
      if (U) U[i] = (REAL)(b - a) * ((REAL)0.5 * w.W + w.H);                                               \
    }                                                                                                      \
  }

DEFINE_BRIDGE(float, f32)
DEFINE_BRIDGE(double, f64)
