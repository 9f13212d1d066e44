// extern "C" surface of libtorchsde_amd.so (declared in include/torchsde_amd.h).
#include <stdio.h>
#include <string.h>

#include <vector>

#include "tsde_common.h"
#include "tsde_launch.h"

namespace {

thread_local char g_err[256] = "";

int fail(hipError_t e, const char* where) {
  if (e != hipSuccess) snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

int bad_arg(const char* where, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, what);
  return (int)hipErrorInvalidValue;
}

tsde::NoiseKey make_key(uint64_t entropy, uint64_t elem0) {
  tsde::NoiseKey k;
  k.k0 = (uint32_t)entropy;
  k.k1 = (uint32_t)(entropy >> 32);
  k.elem0 = elem0;
  return k;
}

// ---- optional per-launch HIP-event timing of one kernel family -------------------------------------
struct Prof {
  int kid = 0;
  int cap = 0;
  int used = 0;
  std::vector<hipEvent_t> ev;  // 2 per launch
} g_prof;

// `dispatch_timed`: the entry point's launcher goes through TSDE_LAUNCH (tsde_common.h), so the events are bound to the
// kernel dispatch itself; otherwise they are recorded on the stream around the call (a bracket: + marker latency).
struct ProfScope {
  bool on;
  hipStream_t s;
  int slot;
  ProfScope(int kid, hipStream_t stream, bool dispatch_timed = false) : on(false), s(stream), slot(0) {
    if (g_prof.kid == kid && g_prof.used < g_prof.cap) {
      slot = g_prof.used++;
      if (dispatch_timed) {
        tsde::LaunchTiming& lt = tsde::launch_timing();
        lt.start = g_prof.ev[2 * slot];
        lt.stop = g_prof.ev[2 * slot + 1];
        lt.armed = true;
      } else {
        on = true;
        (void)hipEventRecord(g_prof.ev[2 * slot], s);
      }
    }
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
    tsde::LaunchTiming& lt = tsde::launch_timing();
    if (lt.armed) {      // the call launched nothing (an empty problem): stamp the pair so that it reads as zero time
      lt.armed = false;
      (void)hipEventRecord(lt.start, s);
      (void)hipEventRecord(lt.stop, s);
    }
  }
};

#define TSDE_DISPATCH(dtype, where, expr_f32, expr_f64)         \
  do {                                                          \
    if ((dtype) == TSDE_F32) return fail((expr_f32), where);    \
    if ((dtype) == TSDE_F64) return fail((expr_f64), where);    \
    return bad_arg(where, "dtype must be TSDE_F32 or TSDE_F64"); \
  } while (0)

}  // namespace

extern "C" {

int tsde_abi_version(void) { return TSDE_ABI_VERSION; }

const char* tsde_last_error(void) { return g_err; }

void tsde_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const tsde::u32x4 r = tsde::philox4x32_10({ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  out[3] = r.w;
}

void tsde_noise_counter(uint64_t quad, uint32_t cell, uint64_t node, uint32_t stream, uint32_t out[4]) {
  const tsde::u32x4 c = tsde::noise_counter(quad, cell, node, stream);
  out[0] = c.x;
  out[1] = c.y;
  out[2] = c.z;
  out[3] = c.w;
}

int tsde_brownian_normals(void* out, int64_t n, uint64_t entropy, uint64_t elem0, uint32_t cell, uint64_t node,
                          uint32_t stream_id, int dtype, void* stream) {
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  TSDE_DISPATCH(dtype, "tsde_brownian_normals", tsde::launch_normals<float>(out, n, key, cell, node, stream_id, s),
                tsde::launch_normals<double>(out, n, key, cell, node, stream_id, s));
}

int tsde_brownian_query(void* W, void* U, void* H, int64_t n, uint64_t entropy, uint64_t elem0, const double* edges,
                        int64_t ca, int64_t cb, double a, double b, const void* rootW, const void* rootH, int have_h,
                        int max_depth, int snap, const uint64_t* entropy_dev, int dtype, void* stream) {
  if (!W || !edges) return bad_arg("tsde_brownian_query", "W and edges are required");
  if (ca < 0 || cb < ca || !(a < b)) return bad_arg("tsde_brownian_query", "need 0 <= ca <= cb and a < b");
  if ((U || H) && !have_h) return bad_arg("tsde_brownian_query", "U/H requested without have_h");
  if (max_depth < 0 || max_depth > 40) return bad_arg("tsde_brownian_query", "max_depth must be in [0, 40]");
  if (rootW && (ca != 0 || cb != 0)) return bad_arg("tsde_brownian_query", "a pinned root needs a single cell");
  if (rootH && !rootW) return bad_arg("tsde_brownian_query", "rootH without rootW");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  tsde::QueryArgs qa;
  qa.edges = edges;
  qa.ca = ca;
  qa.cb = cb;
  qa.a = a;
  qa.b = b;
  qa.rootW = rootW;
  qa.rootH = rootH;
  qa.key_dev = entropy_dev;
  qa.ab_dev = nullptr;
  qa.n_cells = 0;
  qa.cfg.max_depth = max_depth;
  qa.cfg.snap = snap;
  ProfScope p(TSDE_KID_BROWNIAN_QUERY, s, true);
  TSDE_DISPATCH(dtype, "tsde_brownian_query", tsde::launch_query<float>(W, U, H, n, key, qa, have_h != 0, s),
                tsde::launch_query<double>(W, U, H, n, key, qa, have_h != 0, s));
}

int tsde_brownian_query_dev(void* W, void* U, void* H, int64_t n, uint64_t entropy, uint64_t elem0, const double* edges,
                            int64_t n_cells, const double* ab_dev, int have_h, int max_depth,
                            const uint64_t* entropy_dev, int dtype, void* stream) {
  if (!W || !edges || !ab_dev) return bad_arg("tsde_brownian_query_dev", "W, edges and ab_dev are required");
  if (n_cells < 1) return bad_arg("tsde_brownian_query_dev", "need at least one cell");
  if ((U || H) && !have_h) return bad_arg("tsde_brownian_query_dev", "U/H requested without have_h");
  if (max_depth < 0 || max_depth > 40) return bad_arg("tsde_brownian_query_dev", "max_depth must be in [0, 40]");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  tsde::QueryArgs qa;
  qa.edges = edges;
  qa.ca = qa.cb = 0;
  qa.a = qa.b = 0.0;
  qa.rootW = qa.rootH = nullptr;
  qa.key_dev = entropy_dev;
  qa.ab_dev = ab_dev;
  qa.n_cells = n_cells;
  qa.cfg.max_depth = max_depth;
  qa.cfg.snap = 0;
  ProfScope p(TSDE_KID_BROWNIAN_QUERY, s, true);
  TSDE_DISPATCH(dtype, "tsde_brownian_query_dev", tsde::launch_query<float>(W, U, H, n, key, qa, have_h != 0, s),
                tsde::launch_query<double>(W, U, H, n, key, qa, have_h != 0, s));
}

int tsde_cell_increment(void* W_out, void* U_out, int64_t n, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!W_out || !noise) return bad_arg("tsde_cell_increment", "null argument");
  if (noise->dW) return bad_arg("tsde_cell_increment", "noise must describe a generated cell (dW == NULL)");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_cell_increment", tsde::launch_cell_increment<float>(W_out, U_out, n, noise, s),
                tsde::launch_cell_increment<double>(W_out, U_out, n, noise, s));
}

int tsde_step_diag(void* y1, const void* y0, const void* f, const void* g, int64_t n, double cf, double cg,
                   const tsde_noise_t* noise, int dtype, void* stream) {
  if (!y1 || !y0 || !f || !g || !noise) return bad_arg("tsde_step_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_STEP_DIAG, s, true);
  TSDE_DISPATCH(dtype, "tsde_step_diag", tsde::launch_step_diag<float>(y1, y0, f, g, n, cf, cg, noise, s),
                tsde::launch_step_diag<double>(y1, y0, f, g, n, cf, cg, noise, s));
}

int tsde_step_prod(void* y1, const void* y0, const void* f, const void* gp, int64_t n, double cf, double cg, int dtype,
                   void* stream) {
  if (!y1 || !y0 || !f || !gp) return bad_arg("tsde_step_prod", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_step_prod", tsde::launch_step_prod<float>(y1, y0, f, gp, n, cf, cg, s),
                tsde::launch_step_prod<double>(y1, y0, f, gp, n, cf, cg, s));
}

int tsde_step_general(void* y1, const void* y0, const void* f, const void* g, int64_t B, int64_t d, int64_t m,
                      double cf, double cg, const tsde_noise_t* noise, int dtype, void* stream) {
  return tsde_step_general_w(y1, y0, f, g, B, d, m, 1.0, cf, cg, 0, 0.0, 0.0, 0.0, noise, dtype, stream);
}

int tsde_step_general_w(void* y1, const void* y0, const void* f, const void* g, int64_t B, int64_t d, int64_t m,
                        double ca, double cf, double cg, int weight_mode, double cw, double cu, double rdt,
                        const tsde_noise_t* noise, int dtype, void* stream) {
  if (!y1 || !y0 || !f || !g || !noise) return bad_arg("tsde_step_general", "null argument");
  if (weight_mode < 0 || weight_mode > 2) return bad_arg("tsde_step_general", "weight_mode must be 0, 1 or 2");
  if (weight_mode != 0 && noise->dW && !noise->dU) return bad_arg("tsde_step_general", "weights need dU");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_STEP_GENERAL, s, true);
  TSDE_DISPATCH(dtype, "tsde_step_general",
                tsde::launch_step_general<float>(y1, y0, f, g, B, d, m, ca, cf, cg, weight_mode, cw, cu, rdt, noise, s),
                tsde::launch_step_general<double>(y1, y0, f, g, B, d, m, ca, cf, cg, weight_mode, cw, cu, rdt, noise,
                                                  s));
}

int tsde_step_shared(void* y1, const void* y0, const void* f, const void* S, int64_t B, int64_t d, int64_t m, double ca,
                     double cf, double cg, int weight_mode, double cw, double cu, double rdt, const tsde_noise_t* noise,
                     int dtype, void* stream) {
  if (!y1 || !y0 || !f || !S || !noise) return bad_arg("tsde_step_shared", "null argument");
  if (weight_mode < 0 || weight_mode > 2) return bad_arg("tsde_step_shared", "weight_mode must be 0, 1 or 2");
  if (weight_mode != 0 && noise->dW && !noise->dU) return bad_arg("tsde_step_shared", "weights need dU");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_STEP_SHARED, s, true);
  TSDE_DISPATCH(dtype, "tsde_step_shared",
                tsde::launch_step_shared<float>(y1, y0, f, S, B, d, m, ca, cf, cg, weight_mode, cw, cu, rdt, noise, s),
                tsde::launch_step_shared<double>(y1, y0, f, S, B, d, m, ca, cf, cg, weight_mode, cw, cu, rdt, noise, s));
}

int tsde_milstein_v(void* v_out, void* W_out, int64_t n, double dt, int ito, double scale, const tsde_noise_t* noise,
                    int dtype, void* stream) {
  if (!v_out || !noise) return bad_arg("tsde_milstein_v", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_milstein_v",
                tsde::launch_milstein_v<float>(v_out, W_out, nullptr, n, dt, ito, scale, noise, s),
                tsde::launch_milstein_v<double>(v_out, W_out, nullptr, n, dt, ito, scale, noise, s));
}

int tsde_milstein_weight(void* out, const void* g, int64_t n, double dt, int ito, double scale,
                         const tsde_noise_t* noise, int dtype, void* stream) {
  if (!out || !g || !noise) return bad_arg("tsde_milstein_weight", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_milstein_weight",
                tsde::launch_milstein_v<float>(out, nullptr, g, n, dt, ito, scale, noise, s),
                tsde::launch_milstein_v<double>(out, nullptr, g, n, dt, ito, scale, noise, s));
}

int tsde_milstein_diag(void* y1, const void* y0, const void* f, const void* g, const void* gdg, int64_t n, double dt,
                       const tsde_noise_t* noise, int dtype, void* stream) {
  if (!y1 || !y0 || !f || !g || !gdg || !noise) return bad_arg("tsde_milstein_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_MILSTEIN_DIAG, s, true);
  TSDE_DISPATCH(dtype, "tsde_milstein_diag", tsde::launch_milstein_diag<float>(y1, y0, f, g, gdg, n, dt, noise, s),
                tsde::launch_milstein_diag<double>(y1, y0, f, g, gdg, n, dt, noise, s));
}

int tsde_milstein_gf_prime(void* yp, const void* y0, const void* f, const void* g, int64_t n, double dt,
                           double sqrt_dt, int ito, int dtype, void* stream) {
  if (!yp || !y0 || !f || !g) return bad_arg("tsde_milstein_gf_prime", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_milstein_gf_prime",
                tsde::launch_milstein_gf_prime<float>(yp, y0, f, g, n, dt, sqrt_dt, ito, s),
                tsde::launch_milstein_gf_prime<double>(yp, y0, f, g, n, dt, sqrt_dt, ito, s));
}

int tsde_milstein_gf_diag(void* y1, const void* y0, const void* f, const void* g, const void* gprime, int64_t n,
                          double dt, double sqrt_dt, int ito, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!y1 || !y0 || !f || !g || !gprime || !noise) return bad_arg("tsde_milstein_gf_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_MILSTEIN_DIAG, s, true);
  TSDE_DISPATCH(dtype, "tsde_milstein_gf_diag",
                tsde::launch_milstein_gf_diag<float>(y1, y0, f, g, gprime, n, dt, sqrt_dt, ito, noise, s),
                tsde::launch_milstein_gf_diag<double>(y1, y0, f, g, gprime, n, dt, sqrt_dt, ito, noise, s));
}

int tsde_srk_diag_stage(int stage, void* const out[3], const void* const in[5], int64_t n, double dt, double rdt,
                        double sqrt_dt, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!out || !in || !noise) return bad_arg("tsde_srk_diag_stage", "null argument");
  if (stage < 1 || stage > 4) return bad_arg("tsde_srk_diag_stage", "stage must be 1..4");
  static const int n_in[5] = {0, 3, 5, 4, 2}, n_out[5] = {0, 3, 3, 2, 1};
  for (int j = 0; j < n_in[stage]; ++j)
    if (!in[j]) return bad_arg("tsde_srk_diag_stage", "missing input pointer (stage 1: 3, 2: 5, 3: 4, 4: 2 inputs)");
  for (int j = 0; j < n_out[stage]; ++j)
    if (!out[j]) return bad_arg("tsde_srk_diag_stage", "missing output pointer (stage 1: 3, 2: 3, 3: 2, 4: 1 outputs)");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_SRK_STAGE, s, true);
  TSDE_DISPATCH(dtype, "tsde_srk_diag_stage",
                tsde::launch_srk_stage<float>(stage, out, in, n, dt, rdt, sqrt_dt, noise, s),
                tsde::launch_srk_stage<double>(stage, out, in, n, dt, rdt, sqrt_dt, noise, s));
}

int tsde_milstein_gf_general_support(void* yk, const void* y0, const void* f, const void* g, int64_t B, int64_t d,
                                     int64_t m, double dt, double sqrt_dt, int ito, int dtype, void* stream) {
  if (!yk || !y0 || !f || !g) return bad_arg("tsde_milstein_gf_general_support", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_milstein_gf_general_support",
                tsde::launch_milstein_gf_general_support<float>(yk, y0, f, g, B, d, m, dt, sqrt_dt, ito, s),
                tsde::launch_milstein_gf_general_support<double>(yk, y0, f, g, B, d, m, dt, sqrt_dt, ito, s));
}

int tsde_milstein_gf_general_correction(void* corr, const void* g, const void* gk, const void* I, int64_t B, int64_t d,
                                        int64_t m, double sqrt_dt, int dtype, void* stream) {
  if (!corr || !g || !gk || !I) return bad_arg("tsde_milstein_gf_general_correction", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_MILSTEIN_GF_GENERAL, s, true);
  TSDE_DISPATCH(dtype, "tsde_milstein_gf_general_correction",
                tsde::launch_milstein_gf_general_correction<float>(corr, g, gk, I, B, d, m, sqrt_dt, s),
                tsde::launch_milstein_gf_general_correction<double>(corr, g, gk, I, B, d, m, sqrt_dt, s));
}

int tsde_heun_final(void* y1, const void* y0, const void* f, const void* fp, const void* g, const void* gp, int64_t n,
                    double dt, int mode, int prod, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!y1 || !y0 || !f || !g || !gp) return bad_arg("tsde_heun_final", "null argument");
  if (mode != 0 && mode != 1) return bad_arg("tsde_heun_final", "mode must be 0 (heun) or 1 (euler_heun)");
  if (mode == 0 && !fp) return bad_arg("tsde_heun_final", "heun needs the second drift evaluation");
  if (!prod && !noise) return bad_arg("tsde_heun_final", "noise required unless prod");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_heun_final", tsde::launch_heun_final<float>(y1, y0, f, fp, g, gp, n, dt, mode, prod, noise, s),
                tsde::launch_heun_final<double>(y1, y0, f, fp, g, gp, n, dt, mode, prod, noise, s));
}

int tsde_iterated_integrals(void* I, const void* W, const void* A, int64_t B, int64_t m, double dt, int ito, int dtype,
                            void* stream) {
  if (!I || !W) return bad_arg("tsde_iterated_integrals", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_iterated_integrals", tsde::launch_iterated_integrals<float>(I, W, A, B, m, dt, ito, s),
                tsde::launch_iterated_integrals<double>(I, W, A, B, m, dt, ito, s));
}

int tsde_levy_area(void* A, const void* W, const void* H, int64_t B, int64_t m, double h, int foster, uint64_t entropy,
                   uint64_t elem0, uint32_t cell, uint64_t node, const uint64_t* entropy_dev, int dtype, void* stream) {
  if (!A || !W || !H) return bad_arg("tsde_levy_area", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  TSDE_DISPATCH(dtype, "tsde_levy_area",
                tsde::launch_levy_area<float>(A, W, H, B, m, h, foster, key, entropy_dev, cell, node, s),
                tsde::launch_levy_area<double>(A, W, H, B, m, h, foster, key, entropy_dev, cell, node, s));
}

int tsde_levy_iterated_integrals(void* I, const void* W, const void* H, int64_t B, int64_t m, double h, int foster,
                                 uint64_t entropy, uint64_t elem0, uint32_t cell, uint64_t node,
                                 const uint64_t* entropy_dev, double dt, int ito, int dtype, void* stream) {
  if (!I || !W || !H) return bad_arg("tsde_levy_iterated_integrals", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  TSDE_DISPATCH(dtype, "tsde_levy_iterated_integrals",
                tsde::launch_levy_area<float>(I, W, H, B, m, h, foster, key, entropy_dev, cell, node, s, 1, dt, ito),
                tsde::launch_levy_area<double>(I, W, H, B, m, h, foster, key, entropy_dev, cell, node, s, 1, dt, ito));
}

int tsde_rheun_z_diag(void* z1, const void* y0, const void* z0, const void* f0, const void* g0, int64_t n, double dt,
                      double sign, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!z1 || !y0 || !z0 || !f0 || !g0 || !noise) return bad_arg("tsde_rheun_z_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_RHEUN, s);
  TSDE_DISPATCH(dtype, "tsde_rheun_z_diag", tsde::launch_rheun_z<float>(z1, y0, z0, f0, g0, n, dt, sign, noise, s),
                tsde::launch_rheun_z<double>(z1, y0, z0, f0, g0, n, dt, sign, noise, s));
}

int tsde_rheun_y_diag(void* y1, const void* y0, const void* f0, const void* f1, const void* g0, const void* g1,
                      int64_t n, double half_dt, double sign, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!y1 || !y0 || !f0 || !f1 || !g0 || !g1 || !noise) return bad_arg("tsde_rheun_y_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_RHEUN, s);
  TSDE_DISPATCH(dtype, "tsde_rheun_y_diag",
                tsde::launch_rheun_y<float>(y1, y0, f0, f1, g0, g1, n, half_dt, sign, noise, s),
                tsde::launch_rheun_y<double>(y1, y0, f0, f1, g0, g1, n, half_dt, sign, noise, s));
}

int tsde_lincomb2(void* out, const void* x, const void* y, int64_t n, double a, double b, int dtype, void* stream) {
  if (!out || !x || !y) return bad_arg("tsde_lincomb2", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_lincomb2", tsde::launch_lincomb2<float>(out, x, y, n, a, b, s),
                tsde::launch_lincomb2<double>(out, x, y, n, a, b, s));
}

int tsde_rheun_adj_a_diag(void* af0_out, void* ag0_out, const void* ay, const void* af0, const void* ag0, int64_t n,
                          double half_dt, const tsde_noise_t* noise, int dtype, void* stream) {
  if (!af0_out || !ag0_out || !ay || !af0 || !ag0 || !noise) return bad_arg("tsde_rheun_adj_a_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_rheun_adj_a_diag",
                tsde::launch_rheun_adj_a<float>(af0_out, ag0_out, ay, af0, ag0, n, half_dt, noise, s),
                tsde::launch_rheun_adj_a<double>(af0_out, ag0_out, ay, af0, ag0, n, half_dt, noise, s));
}

int tsde_rheun_adj_b_diag(void* ay1, void* az1, void* af1, void* ag1, const void* ay, const void* az0,
                          const void* vjp_z, int64_t n, double dt, double half_dt, const tsde_noise_t* noise, int dtype,
                          void* stream) {
  if (!ay1 || !az1 || !af1 || !ag1 || !ay || !az0 || !vjp_z || !noise)
    return bad_arg("tsde_rheun_adj_b_diag", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_rheun_adj_b_diag",
                tsde::launch_rheun_adj_b<float>(ay1, az1, af1, ag1, ay, az0, vjp_z, n, dt, half_dt, noise, s),
                tsde::launch_rheun_adj_b<double>(ay1, az1, af1, ag1, ay, az0, vjp_z, n, dt, half_dt, noise, s));
}

int tsde_aug_update(const tsde_seg_t* segs, int nseg, double cF, double cG, int dtype, void* stream) {
  if (!segs || nseg < 0) return bad_arg("tsde_aug_update", "bad segment list");
  if (dtype != TSDE_F32 && dtype != TSDE_F64) return bad_arg("tsde_aug_update", "dtype");
  const hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < nseg; ++i) {
    if (segs[i].n > 0 && (!segs[i].out || !segs[i].s)) return bad_arg("tsde_aug_update", "segment without state");
  }
  ProfScope p(TSDE_KID_AUG_UPDATE, s, true);
  if (dtype == TSDE_F32) return fail(tsde::launch_aug_segments<float>(segs, nseg, cF, cG, s), "tsde_aug_update");
  return fail(tsde::launch_aug_segments<double>(segs, nseg, cF, cG, s), "tsde_aug_update");
}

int tsde_linear_interp(void* out, const void* ya, const void* yb, int64_t n, double w0, double w1, int dtype,
                       void* stream) {
  if (!out || !ya || !yb) return bad_arg("tsde_linear_interp", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_linear_interp", tsde::launch_interp<float>(out, ya, yb, n, w0, w1, s),
                tsde::launch_interp<double>(out, ya, yb, n, w0, w1, s));
}

int tsde_error_norm(double* out, double* workspace, const void* y_full, const void* y_half, int64_t n, double rtol,
                    double atol, double eps, int dtype, void* stream) {
  if (!out || !workspace || !y_full || !y_half) return bad_arg("tsde_error_norm", "null argument");
  if (n <= 0) return bad_arg("tsde_error_norm", "n must be positive");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_error_norm",
                tsde::launch_error_norm<float>(out, workspace, y_full, y_half, n, rtol, atol, eps, s),
                tsde::launch_error_norm<double>(out, workspace, y_full, y_half, n, rtol, atol, eps, s));
}

int tsde_adaptive_begin(double* ctl, void* scal, double out_t, const double* stage_fracs, int n_fracs, int dtype,
                        void* stream) {
  if (!ctl || !scal || (n_fracs > 0 && !stage_fracs)) return bad_arg("tsde_adaptive_begin", "null argument");
  if (n_fracs < 1 || n_fracs > TSDE_ADAPTIVE_MAX_STAGES - 1) return bad_arg("tsde_adaptive_begin", "bad number of stages");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_adaptive_begin",
                tsde::launch_adaptive_begin<float>(ctl, scal, out_t, nullptr, 0, stage_fracs, n_fracs, s),
                tsde::launch_adaptive_begin<double>(ctl, scal, out_t, nullptr, 0, stage_fracs, n_fracs, s));
}

int tsde_adaptive_begin_outputs(double* ctl, void* scal, const double* out_times, int32_t n_out, const double* stage_fracs,
                                int n_fracs, int dtype, void* stream) {
  const char* where = "tsde_adaptive_begin_outputs";
  if (!ctl || !scal || !out_times || !stage_fracs) return bad_arg(where, "null argument");
  if (n_out < 1) return bad_arg(where, "need at least one output time");
  if (n_fracs < 1 || n_fracs > TSDE_ADAPTIVE_MAX_STAGES - 1) return bad_arg(where, "bad number of stages");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, where,
                tsde::launch_adaptive_begin<float>(ctl, scal, 0.0, out_times, n_out, stage_fracs, n_fracs, s),
                tsde::launch_adaptive_begin<double>(ctl, scal, 0.0, out_times, n_out, stage_fracs, n_fracs, s));
}

int tsde_adaptive_control_outputs(double* ctl, void* scal, const double* error, const double* out_times,
                                  double* accept_log, int32_t log_capacity, const double* stage_fracs, int n_fracs, int dtype,
                                  void* stream) {
  const char* where = "tsde_adaptive_control_outputs";
  if (!ctl || !scal || !error || !out_times || !stage_fracs) return bad_arg(where, "null argument");
  if (n_fracs < 1 || n_fracs > TSDE_ADAPTIVE_MAX_STAGES - 1) return bad_arg(where, "bad number of stages");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, where,
                tsde::launch_adaptive_control<float>(ctl, scal, error, out_times, accept_log, log_capacity, stage_fracs,
                                                     n_fracs, s),
                tsde::launch_adaptive_control<double>(ctl, scal, error, out_times, accept_log, log_capacity, stage_fracs,
                                                      n_fracs, s));
}

int tsde_adaptive_emit(const void* ys_slot, const void* prev_y, const void* curr_y, int64_t n, const double* ctl,
                       const double* out_times, int dtype, void* stream) {
  const char* where = "tsde_adaptive_emit";
  if (!ys_slot || !prev_y || !curr_y || !ctl || !out_times) return bad_arg(where, "null argument");
  if (n <= 0) return bad_arg(where, "n must be positive");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, where, tsde::launch_adaptive_emit<float>(ys_slot, prev_y, curr_y, n, ctl, out_times, s),
                tsde::launch_adaptive_emit<double>(ys_slot, prev_y, curr_y, n, ctl, out_times, s));
}

int tsde_adaptive_control(double* ctl, void* scal, const double* error, const double* stage_fracs, int n_fracs,
                          int dtype, void* stream) {
  if (!ctl || !scal || !error || !stage_fracs) return bad_arg("tsde_adaptive_control", "null argument");
  if (n_fracs < 1 || n_fracs > TSDE_ADAPTIVE_MAX_STAGES - 1) return bad_arg("tsde_adaptive_control", "bad number of stages");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_adaptive_control",
                tsde::launch_adaptive_control<float>(ctl, scal, error, nullptr, nullptr, 0, stage_fracs, n_fracs, s),
                tsde::launch_adaptive_control<double>(ctl, scal, error, nullptr, nullptr, 0, stage_fracs, n_fracs, s));
}

int tsde_adaptive_commit(void* prev_y, void* curr_y, const void* y_next, int64_t n, const void* scal, int dtype,
                         void* stream) {
  if (!prev_y || !curr_y || !y_next || !scal) return bad_arg("tsde_adaptive_commit", "null argument");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_adaptive_commit", tsde::launch_adaptive_commit<float>(prev_y, curr_y, y_next, n, scal, s),
                tsde::launch_adaptive_commit<double>(prev_y, curr_y, y_next, n, scal, s));
}

int tsde_merge_halves(void* W, void* U, const void* Wa, const void* Ha, const void* Wb, const void* Hb, int64_t n,
                      const double* ctl, double ha, double hb, int dtype, void* stream) {
  if (!W || !Wa || !Wb) return bad_arg("tsde_merge_halves", "null argument");
  if (!ctl && !(ha + hb > 0.0)) return bad_arg("tsde_merge_halves", "need ctl or positive widths");
  if (U && (!Ha || !Hb)) return bad_arg("tsde_merge_halves", "U needs the space-time Levy areas of both halves");
  const hipStream_t s = (hipStream_t)stream;
  TSDE_DISPATCH(dtype, "tsde_merge_halves", tsde::launch_merge_halves<float>(W, U, Wa, Ha, Wb, Hb, n, ctl, ha, hb, s),
                tsde::launch_merge_halves<double>(W, U, Wa, Ha, Wb, Hb, n, ctl, ha, hb, s));
}

static int trajectory_affine_diag(const char* where, void* ys, void* sens, const void* y0, int64_t rows, int64_t d,
                                  const void* drift_rate, const void* drift_shift, const void* diff_rate,
                                  const void* diff_shift, int64_t coef_step_stride, int method, const tsde_traj_t* traj,
                                  uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                                  void* stream) {
  if (!ys || !y0 || !drift_rate || !drift_shift || !diff_rate || !diff_shift || !traj) return bad_arg(where, "null argument");
  if (coef_step_stride != 0 && coef_step_stride < d) return bad_arg(where, "coef_step_stride must be 0 or >= d");
  if (coef_step_stride != 0 && sens) return bad_arg(where, "per-step coefficients: values only");
  if (rows < 0 || d <= 0) return bad_arg(where, "need rows >= 0 and d > 0");
  if (method < TSDE_TRAJ_EULER || method > TSDE_TRAJ_EULER_HEUN) return bad_arg(where, "unknown method");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  TSDE_DISPATCH(dtype, where,
                tsde::launch_trajectory_affine_diag<float>(ys, sens, y0, rows, d, drift_rate, drift_shift, diff_rate,
                                                           diff_shift, coef_step_stride, method, traj, key, entropy_dev, s),
                tsde::launch_trajectory_affine_diag<double>(ys, sens, y0, rows, d, drift_rate, drift_shift, diff_rate,
                                                            diff_shift, coef_step_stride, method, traj, key, entropy_dev,
                                                            s));
}

int tsde_trajectory_affine_diag(void* ys, const void* y0, int64_t rows, int64_t d, const void* drift_rate,
                                const void* drift_shift, const void* diff_rate, const void* diff_shift, int method,
                                const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                                const uint64_t* entropy_dev, int dtype, void* stream) {
  return trajectory_affine_diag("tsde_trajectory_affine_diag", ys, nullptr, y0, rows, d, drift_rate, drift_shift,
                                diff_rate, diff_shift, 0, method, traj, entropy, elem0, entropy_dev, dtype, stream);
}

int tsde_trajectory_affine_diag_timed(void* ys, const void* y0, int64_t rows, int64_t d, const void* drift_rate,
                                      const void* drift_shift, const void* diff_rate, const void* diff_shift,
                                      int64_t coef_step_stride, int method, const tsde_traj_t* traj, uint64_t entropy,
                                      uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream) {
  return trajectory_affine_diag("tsde_trajectory_affine_diag_timed", ys, nullptr, y0, rows, d, drift_rate, drift_shift,
                                diff_rate, diff_shift, coef_step_stride, method, traj, entropy, elem0, entropy_dev, dtype,
                                stream);
}

int tsde_trajectory_affine_diag_sens(void* ys, void* sens, const void* y0, int64_t rows, int64_t d,
                                     const void* drift_rate, const void* drift_shift, const void* diff_rate,
                                     const void* diff_shift, int method, const tsde_traj_t* traj, uint64_t entropy,
                                     uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream) {
  if (!sens) return bad_arg("tsde_trajectory_affine_diag_sens", "null argument");
  return trajectory_affine_diag("tsde_trajectory_affine_diag_sens", ys, sens, y0, rows, d, drift_rate, drift_shift,
                                diff_rate, diff_shift, 0, method, traj, entropy, elem0, entropy_dev, dtype, stream);
}

int tsde_trajectory_mlp_diag(void* ys, const void* y0, int64_t rows, int64_t d, int64_t hidden, const void* w1,
                             const void* b1, const void* w2, const void* b2, const void* diff_rate,
                             const void* diff_shift, int diff_kind, double diff_amp, int activation, int method,
                             const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev,
                             int dtype, void* stream) {
  const char* where = "tsde_trajectory_mlp_diag";
  if (diff_kind != TSDE_DIFF_AFFINE && diff_kind != TSDE_DIFF_SIGMOID) return bad_arg(where, "unknown diffusion kind");
  if (!ys || !y0 || !w1 || !b1 || !w2 || !b2 || !diff_rate || !diff_shift || !traj) return bad_arg(where, "null argument");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (rows < 0) return bad_arg(where, "need rows >= 0");
  if (d < 4 || d > 128 || d % 4 != 0 || hidden < 1 || hidden > 256 || (hidden > 128 && d > 64))
    return bad_arg(where, "need d a multiple of 4 in [4, 128] and hidden in [1, 128] (up to 256 for d <= 64)");
  if ((reinterpret_cast<uintptr_t>(y0) | reinterpret_cast<uintptr_t>(ys)) & 15u)
    return bad_arg(where, "ys and y0 must be 16-byte aligned");
  if (rows * d >= (int64_t(1) << 30)) return bad_arg(where, "need rows * d < 2^30 (32-bit lane offsets)");
  if (activation != TSDE_ACT_TANH && activation != TSDE_ACT_SOFTPLUS) return bad_arg(where, "unknown activation");
  if (method != TSDE_TRAJ_EULER && method != TSDE_TRAJ_MILSTEIN_ITO && method != TSDE_TRAJ_MILSTEIN_STRAT &&
      method != TSDE_TRAJ_MIDPOINT && method != TSDE_TRAJ_SRK)
    return bad_arg(where, "method must be Euler, Milstein, midpoint or SRK");
  if (elem0 % 4 != 0) return bad_arg(where, "elem0 must be a multiple of 4");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  return fail(tsde::launch_trajectory_mlp_diag(ys, y0, rows, d, hidden, w1, b1, w2, b2, diff_rate, diff_shift,
                                               diff_kind, diff_amp, activation, method, traj, make_key(entropy, elem0), entropy_dev, s),
              where);
}

static int prog_diag(const char* where, void* ys, void* sens, const int8_t* param_slot, const void* y0, int64_t rows,
                     int64_t d, const uint32_t* code, int32_t f_len, int32_t g_len, int32_t dg_len, const void* consts,
                     int32_t n_const, int scalar_noise, int method, const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                     const uint64_t* entropy_dev, int dtype, void* stream) {
  if (!ys || !y0 || !code || !traj) return bad_arg(where, "null argument");
  if (f_len < 1 || g_len < 1 || dg_len < 0 || f_len + g_len + dg_len > 96)
    return bad_arg(where, "program lengths out of range (at most 96 words together)");
  if (n_const < 0 || n_const > 64 || (n_const > 0 && !consts)) return bad_arg(where, "constant table missing or above 64 rows");
  if (rows < 0 || d <= 0) return bad_arg(where, "need rows >= 0 and d > 0");
  if (method < TSDE_TRAJ_EULER || method > TSDE_TRAJ_EULER_HEUN) return bad_arg(where, "unknown method");
  if ((method == TSDE_TRAJ_MILSTEIN_ITO || method == TSDE_TRAJ_MILSTEIN_STRAT) && dg_len < 1)
    return bad_arg(where, "Milstein needs the program of the diffusion's derivative");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  if (sens && param_slot)
    for (int k = 0; k < n_const; ++k)
      if (param_slot[k] == 0 || param_slot[k] >= TSDE_TRAJ_SENS || param_slot[k] < -1)
        return bad_arg(where, "param_slot entries must be -1 or in 1 .. TSDE_TRAJ_SENS - 1");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  TSDE_DISPATCH(dtype, where,
                tsde::launch_trajectory_prog_diag<float>(ys, sens, param_slot, y0, rows, d, code, f_len, g_len, dg_len, consts,
                                                         n_const, scalar_noise != 0, method, traj, key, entropy_dev, s),
                tsde::launch_trajectory_prog_diag<double>(ys, sens, param_slot, y0, rows, d, code, f_len, g_len, dg_len, consts,
                                                          n_const, scalar_noise != 0, method, traj, key, entropy_dev, s));
}

int tsde_trajectory_prog_diag(void* ys, const void* y0, int64_t rows, int64_t d, const uint32_t* code, int32_t f_len,
                              int32_t g_len, int32_t dg_len, const void* consts, int32_t n_const, int scalar_noise, int method,
                              const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev,
                              int dtype, void* stream) {
  return prog_diag("tsde_trajectory_prog_diag", ys, nullptr, nullptr, y0, rows, d, code, f_len, g_len, dg_len, consts, n_const,
                   scalar_noise, method, traj, entropy, elem0, entropy_dev, dtype, stream);
}

int tsde_trajectory_prog_diag_sens(void* ys, void* sens, const void* y0, int64_t rows, int64_t d, const uint32_t* code,
                                   int32_t f_len, int32_t g_len, int32_t dg_len, const void* consts, int32_t n_const,
                                   const int8_t* param_slot, int scalar_noise, int method, const tsde_traj_t* traj,
                                   uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream) {
  if (!sens || !param_slot) return bad_arg("tsde_trajectory_prog_diag_sens", "sens and param_slot are required");
  return prog_diag("tsde_trajectory_prog_diag_sens", ys, sens, param_slot, y0, rows, d, code, f_len, g_len, dg_len, consts,
                   n_const, scalar_noise, method, traj, entropy, elem0, entropy_dev, dtype, stream);
}

int tsde_trajectory_mlp_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const tsde_mlp_t* drift,
                                 const void* g_table, int g_time_dependent, int method, const tsde_traj_t* traj,
                                 uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream) {
  const char* where = "tsde_trajectory_mlp_additive";
  if (!ys || !y0 || !drift || !g_table || !traj) return bad_arg(where, "null argument");
  if (!drift->w1 || !drift->b1 || !drift->w2 || !drift->b2) return bad_arg(where, "a perceptron without weights or biases");
  if (drift->hidden < 1 || drift->hidden > 128) return bad_arg(where, "hidden sizes must be in [1, 128]");
  if (drift->activation != TSDE_ACT_TANH && drift->activation != TSDE_ACT_SOFTPLUS) return bad_arg(where, "unknown activation");
  if (drift->precision != TSDE_PRECISION_F32) return bad_arg(where, "the drift runs in exact f32");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (rows < 0) return bad_arg(where, "need rows >= 0");
  if (d < 1 || d > 64) return bad_arg(where, "need d in [1, 64]");
  if (m < 1 || m > 16) return bad_arg(where, "need 1 <= m <= 16 Brownian channels");
  if (drift->out != d || drift->final != TSDE_FINAL_NONE || drift->scale != 1.0)
    return bad_arg(where, "the drift maps to d channels, with no output function and scale 1");
  // (rows are read and written as 16-byte groups only when d is a multiple of 4 -- then every row of every output is aligned
  //  with the bases; any other width goes element by element, mlp_general.hip `row_quads`)
  if (((reinterpret_cast<uintptr_t>(y0) | reinterpret_cast<uintptr_t>(ys)) & (d % 4 == 0 ? 15u : 3u)) != 0)
    return bad_arg(where, "ys and y0 must be 16-byte aligned (4-byte when d is not a multiple of 4)");
  if (rows * d >= (int64_t(1) << 30)) return bad_arg(where, "need rows * d < 2^30 (32-bit lane offsets)");
  if (method != TSDE_TRAJ_EULER && method != TSDE_TRAJ_MIDPOINT && method != TSDE_TRAJ_SRK)
    return bad_arg(where, "method must be Euler, midpoint or SRK (SRA1)");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  const hipError_t e = tsde::launch_trajectory_mlp_additive(ys, y0, rows, d, m, drift, g_table, g_time_dependent != 0, method,
                                                            traj, make_key(entropy, elem0), entropy_dev, s);
  if (e == hipErrorInvalidValue) return bad_arg(where, "no kernel for this shape");
  return fail(e, where);
}

int tsde_trajectory_prog_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const uint32_t* code,
                                  int32_t f_len, const void* consts, int32_t n_const, const void* g_table,
                                  int g_time_dependent, int method, const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                                  const uint64_t* entropy_dev, int dtype, void* stream) {
  const char* where = "tsde_trajectory_prog_additive";
  if (!ys || !y0 || !code || !traj || !g_table) return bad_arg(where, "null argument");
  if (f_len < 1 || f_len > 96) return bad_arg(where, "the drift program has 1 to 96 words");
  if (n_const < 0 || n_const > 64 || (n_const > 0 && !consts)) return bad_arg(where, "constant table missing or above 64 rows");
  if (rows < 0 || d <= 0) return bad_arg(where, "need rows >= 0 and d > 0");
  if (m < 1 || m > 16) return bad_arg(where, "need 1 <= m <= 16 Brownian channels");
  if (method != TSDE_TRAJ_EULER && method != TSDE_TRAJ_MIDPOINT && method != TSDE_TRAJ_SRK)
    return bad_arg(where, "method must be Euler, midpoint or SRK (SRA1)");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  TSDE_DISPATCH(dtype, where,
                tsde::launch_trajectory_prog_additive<float>(ys, y0, rows, d, m, code, f_len, consts, n_const, g_table,
                                                             g_time_dependent != 0, method, traj, key, entropy_dev, s),
                tsde::launch_trajectory_prog_additive<double>(ys, y0, rows, d, m, code, f_len, consts, n_const, g_table,
                                                              g_time_dependent != 0, method, traj, key, entropy_dev, s));
}

int64_t tsde_trajectory_mlp_general_lds(int64_t d, int64_t m, int64_t drift_hidden, int64_t diffusion_hidden,
                                        int64_t diffusion_out, int noise) {
  if (d < 1 || drift_hidden < 1 || diffusion_hidden < 1 || diffusion_out < 1) return 0;
  return (int64_t)tsde::neural_footprint(d, m, drift_hidden, diffusion_hidden, diffusion_out, noise);
}

int tsde_trajectory_mlp_general(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                                const tsde_mlp_t* drift, const tsde_mlp_t* diffusion, int method, const tsde_traj_t* traj,
                                uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype, void* stream) {
  const char* where = "tsde_trajectory_mlp_general";
  if (!ys || !y0 || !drift || !diffusion || !traj) return bad_arg(where, "null argument");
  for (const tsde_mlp_t* net : {drift, diffusion}) {
    if (!net->w1 || !net->b1 || !net->w2 || !net->b2) return bad_arg(where, "a perceptron without weights or biases");
    if (net->hidden < 1 || net->hidden > (noise == TSDE_NOISE_GENERAL ? 64 : 128))
      return bad_arg(where, "hidden sizes must be in [1, 64] (general noise) or [1, 128]");
    if (net->activation != TSDE_ACT_TANH && net->activation != TSDE_ACT_SOFTPLUS) return bad_arg(where, "unknown activation");
    if (net->final != TSDE_FINAL_NONE && net->final != TSDE_FINAL_SIGMOID) return bad_arg(where, "unknown output function");
    if (net->precision != TSDE_PRECISION_F32 && net->precision != TSDE_PRECISION_BF16X3) return bad_arg(where, "unknown precision");
  }
  if (drift->precision != TSDE_PRECISION_F32 || (diffusion->precision != TSDE_PRECISION_F32 && noise != TSDE_NOISE_GENERAL))
    return bad_arg(where, "split-bf16 products are for the diffusion net under general noise only");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (rows < 0) return bad_arg(where, "need rows >= 0");
  if (d < 1 || d > 64) return bad_arg(where, "need d in [1, 64]");
  if (drift->out != d || drift->final != TSDE_FINAL_NONE || drift->scale != 1.0)
    return bad_arg(where, "the drift maps to d channels, with no output function and scale 1");
  if (noise == TSDE_NOISE_GENERAL) {
    if (m < 1 || m > 32) return bad_arg(where, "general noise: m must be in [1, 32]");
    if (diffusion->out != d * m) return bad_arg(where, "general noise: the diffusion net maps to d * m outputs");
  } else if (noise == TSDE_NOISE_DIAGONAL || noise == TSDE_NOISE_SCALAR) {
    if (diffusion->out != d) return bad_arg(where, "diagonal / scalar noise: the diffusion net maps to d outputs");
    if (m != (noise == TSDE_NOISE_DIAGONAL ? d : 1)) return bad_arg(where, "m must be d (diagonal) or 1 (scalar)");
  } else {
    return bad_arg(where, "unknown noise kind");
  }
  if (tsde::neural_footprint(d, m, drift->hidden, diffusion->hidden, diffusion->out, noise) == 0)
    return bad_arg(where, "no kernel for this shape");
  // (rows are read and written as 16-byte groups only when d is a multiple of 4 -- then every row of every output is aligned
  //  with the bases; any other width goes element by element, mlp_general.hip `row_quads`)
  if (((reinterpret_cast<uintptr_t>(y0) | reinterpret_cast<uintptr_t>(ys)) & (d % 4 == 0 ? 15u : 3u)) != 0)
    return bad_arg(where, "ys and y0 must be 16-byte aligned (4-byte when d is not a multiple of 4)");
  if (rows * d >= (int64_t(1) << 30)) return bad_arg(where, "need rows * d < 2^30 (32-bit lane offsets)");
  if (method != TSDE_TRAJ_EULER && method != TSDE_TRAJ_MIDPOINT && method != TSDE_TRAJ_SRK)
    return bad_arg(where, "method must be Euler, midpoint or SRK");
  if (method == TSDE_TRAJ_SRK && noise == TSDE_NOISE_GENERAL)
    return bad_arg(where, "SRK (SRID2) takes diagonal or scalar noise, like the reference's (srk.py:34-35)");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  const hipError_t e = tsde::launch_trajectory_mlp_general(ys, y0, rows, d, m, noise, drift, diffusion, method, traj,
                                                           make_key(entropy, elem0), entropy_dev, s);
  if (e == hipErrorInvalidValue) return bad_arg(where, "the weights of this shape do not fit the LDS of a CU");
  return fail(e, where);
}

int tsde_trajectory_expr_diag(void* ys, const void* y0, int64_t rows, int64_t d, const void* const coef[8], int f_kind,
                              int g_kind, int method, const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                              const uint64_t* entropy_dev, int dtype, void* stream) {
  return tsde_trajectory_expr_diag_timed(ys, y0, rows, d, coef, 0, f_kind, g_kind, method, traj, entropy, elem0,
                                         entropy_dev, dtype, stream);
}

int tsde_trajectory_expr_diag_timed(void* ys, const void* y0, int64_t rows, int64_t d, const void* const coef[8],
                                    int64_t coef_step_stride, int f_kind, int g_kind, int method,
                                    const tsde_traj_t* traj, uint64_t entropy, uint64_t elem0,
                                    const uint64_t* entropy_dev, int dtype, void* stream) {
  const char* where = coef_step_stride ? "tsde_trajectory_expr_diag_timed" : "tsde_trajectory_expr_diag";
  if (!ys || !y0 || !coef || !traj) return bad_arg(where, "null argument");
  if (coef_step_stride != 0 && coef_step_stride < d) return bad_arg(where, "coef_step_stride must be 0 or >= d");
  for (int c = 0; c < 8; ++c)
    if (!coef[c]) return bad_arg(where, "null coefficient array");
  if (rows < 0 || d <= 0) return bad_arg(where, "need rows >= 0 and d > 0");
  if (method < TSDE_TRAJ_EULER || method > TSDE_TRAJ_EULER_HEUN) return bad_arg(where, "unknown method");
  if (f_kind < TSDE_FN_IDENTITY || f_kind > TSDE_FN_POLY3 || g_kind < TSDE_FN_IDENTITY || g_kind > TSDE_FN_POLY3)
    return bad_arg(where, "unknown function code");
  if (traj->n_steps < 0 || traj->n_out < 0) return bad_arg(where, "negative schedule length");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return bad_arg(where, "schedule without output map");
  const hipStream_t s = (hipStream_t)stream;
  const tsde::NoiseKey key = make_key(entropy, elem0);
  ProfScope p(TSDE_KID_TRAJECTORY, s);
  TSDE_DISPATCH(dtype, where,
                tsde::launch_trajectory_expr_diag<float>(ys, y0, rows, d, coef, coef_step_stride, f_kind, g_kind, method,
                                                         traj, key, entropy_dev, s),
                tsde::launch_trajectory_expr_diag<double>(ys, y0, rows, d, coef, coef_step_stride, f_kind, g_kind, method,
                                                          traj, key, entropy_dev, s));
}

int tsde_adjoint_mlp_diag(void* y, void* a, void* stash_a, void* stash_hid, void* stash_delta, void* stash_y,
                          void* row_rate, void* row_shift, int64_t rows, int64_t d, int64_t hidden, const void* w1,
                          const void* b1, const void* w2, const void* b2, const void* diff_rate, const void* diff_shift,
                          int diff_kind, double diff_amp, int activation, int ito, const tsde_traj_t* traj, int32_t k_lo,
                          int32_t k_hi, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                          void* stream) {
  const char* where = "tsde_adjoint_mlp_diag";
  if (!y || !a || !stash_a || !stash_hid || !stash_delta || !stash_y || !row_rate || !row_shift || !w1 || !b1 || !w2 ||
      !b2 || !diff_rate || !diff_shift || !traj)
    return bad_arg(where, "null argument");
  if (diff_kind != TSDE_DIFF_AFFINE && diff_kind != TSDE_DIFF_SIGMOID) return bad_arg(where, "unknown diffusion kind");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (rows < 0) return bad_arg(where, "need rows >= 0");
  if (d < 4 || d > 128 || d % 4 != 0 || hidden < 4 || hidden > 256 || hidden % 4 != 0 || (hidden > 128 && d > 64))
    return bad_arg(where, "need d and hidden multiples of 4, d in [4, 128], hidden in [4, 128] (up to 256 for d <= 64)");
  if (rows * (d > hidden ? d : hidden) >= (int64_t(1) << 30))
    return bad_arg(where, "need rows * max(d, hidden) < 2^30 (32-bit lane offsets)");
  const void* aligned[] = {y, a, stash_a, stash_hid, stash_delta, stash_y, row_rate, row_shift};
  for (const void* q : aligned)
    if (reinterpret_cast<uintptr_t>(q) & 15u) return bad_arg(where, "state-shaped buffers must be 16-byte aligned");
  if (activation != TSDE_ACT_TANH && activation != TSDE_ACT_SOFTPLUS) return bad_arg(where, "unknown activation");
  if (elem0 % 4 != 0) return bad_arg(where, "elem0 must be a multiple of 4");
  if (k_lo < 0 || k_hi < k_lo || k_hi > traj->n_steps) return bad_arg(where, "need 0 <= k_lo <= k_hi <= n_steps");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_MLP_ADJOINT, s);
  return fail(tsde::launch_adjoint_mlp_diag(y, a, stash_a, stash_hid, stash_delta, stash_y, row_rate, row_shift, rows, d,
                                            hidden, w1, b1, w2, b2, diff_rate, diff_shift, diff_kind, diff_amp,
                                            activation, ito & 3, traj, k_lo, k_hi, make_key(entropy, elem0),
                                            entropy_dev, s),
              where);
}

int tsde_trajectory_mlp_diag_backward(void* lam, void* stash_lam, void* stash_hid, void* stash_delta, void* row_rate,
                                      void* row_shift, const void* ys_all, int32_t ys_first, const void* grad_ys,
                                      const int32_t* grad_step, int32_t grad_last, int64_t rows, int64_t d,
                                      int64_t hidden, const void* w1, const void* b1, const void* w2,
                                      const void* diff_rate, const void* diff_shift, int diff_kind, double diff_amp,
                                      int activation, int method, const tsde_traj_t* traj, int32_t k_lo, int32_t k_hi,
                                      uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                                      void* stream) {
  const char* where = "tsde_trajectory_mlp_diag_backward";
  if (!lam || !stash_lam || !stash_hid || !stash_delta || !row_rate || !row_shift || !ys_all || !w1 || !b1 || !w2 ||
      !diff_rate || !diff_shift || !traj)
    return bad_arg(where, "null argument");
  if (method != TSDE_TRAJ_EULER && method != TSDE_TRAJ_MILSTEIN_ITO && method != TSDE_TRAJ_MILSTEIN_STRAT)
    return bad_arg(where, "method must be Euler or Milstein");
  if (diff_kind != TSDE_DIFF_AFFINE && !(diff_kind == TSDE_DIFF_SIGMOID && method == TSDE_TRAJ_EULER))
    return bad_arg(where, "diffusion kind must be affine, or sigmoid with Euler");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (rows < 0) return bad_arg(where, "need rows >= 0");
  if (d < 4 || d > 128 || d % 4 != 0 || hidden < 4 || hidden > 256 || hidden % 4 != 0 || (hidden > 128 && d > 64))
    return bad_arg(where, "need d and hidden multiples of 4, d in [4, 128], hidden in [4, 128] (up to 256 for d <= 64)");
  if (rows * (d > hidden ? d : hidden) >= (int64_t(1) << 30))
    return bad_arg(where, "need rows * max(d, hidden) < 2^30 (32-bit lane offsets)");
  const void* aligned[] = {lam, stash_lam, stash_hid, stash_delta, row_rate, row_shift, ys_all, grad_ys};
  for (const void* q : aligned)
    if (reinterpret_cast<uintptr_t>(q) & 15u) return bad_arg(where, "state-shaped buffers must be 16-byte aligned");
  if (activation != TSDE_ACT_TANH && activation != TSDE_ACT_SOFTPLUS) return bad_arg(where, "unknown activation");
  if (elem0 % 4 != 0) return bad_arg(where, "elem0 must be a multiple of 4");
  if (k_lo < 0 || k_hi < k_lo || k_hi > traj->n_steps) return bad_arg(where, "need 0 <= k_lo <= k_hi <= n_steps");
  if (ys_first < 0 || ys_first > k_lo) return bad_arg(where, "need 0 <= ys_first <= k_lo");
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return bad_arg(where, "schedule without step rows");
  if (grad_last >= 0 && (!grad_ys || !grad_step)) return bad_arg(where, "grad_last >= 0 without cotangents");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_MLP_BACKWARD, s);
  return fail(tsde::launch_trajectory_mlp_diag_backward(lam, stash_lam, stash_hid, stash_delta, row_rate, row_shift,
                                                        ys_all, ys_first, grad_ys, grad_step, grad_last, rows, d, hidden, w1, b1,
                                                        w2, diff_rate, diff_shift, diff_kind, diff_amp, activation, method,
                                                        traj, k_lo, k_hi,
                                                        make_key(entropy, elem0), entropy_dev, s),
              where);
}

int tsde_gram_partials(void* partials, void* colsum_partials, const void* a, int64_t lda, const void* b, int64_t ldb,
                       int64_t k, int64_t m, int64_t n, int32_t blocks, int dtype, void* stream) {
  const char* where = "tsde_gram_partials";
  if (!partials || !a || !b) return bad_arg(where, "null argument");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (k < 1 || m < 1 || m > 128 || n < 1 || n > 128 || blocks < 1) return bad_arg(where, "need k, blocks >= 1 and m, n in [1, 128]");
  if (lda < m || ldb < n) return bad_arg(where, "need lda >= m and ldb >= n");
  return fail(tsde::launch_gram_partials(partials, colsum_partials, a, lda, b, ldb, k, m, n, blocks, (hipStream_t)stream),
              where);
}

static const char* deep_mlp_problem(const tsde_deep_mlp_t* n, int64_t d, int64_t out, const char* which) {
  static thread_local char msg[160];
  const char* what = nullptr;
  if (!n) what = "null";
  else if (!n->w1 || !n->b1 || !n->w2 || !n->b2) what = "a perceptron without weights or biases";
  else if (n->hidden < 1 || n->hidden > 64) what = "hidden sizes must be in [1, 64]";
  else if (n->n_mid < 0 || n->n_mid > 2) what = "0 to 2 hidden-to-hidden layers";
  else if ((n->n_mid > 0 && (!n->wm[0] || !n->bm[0])) || (n->n_mid > 1 && (!n->wm[1] || !n->bm[1])))
    what = "a hidden-to-hidden layer without weights or biases";
  else if (n->activation != TSDE_ACT_TANH && n->activation != TSDE_ACT_SOFTPLUS && n->activation != TSDE_ACT_SILU)
    what = "unknown activation";
  else if (n->final != TSDE_FINAL_NONE && n->final != TSDE_FINAL_SIGMOID && n->final != TSDE_FINAL_TANH)
    what = "unknown output function";
  else if (n->out != out) what = "wrong number of outputs";
  if (!what) return nullptr;
  snprintf(msg, sizeof(msg), "%s net: %s", which, what);
  (void)d;
  return msg;
}

static const char* rheun_mlp_problem(int64_t rows, int64_t d, int64_t m, int noise, const tsde_deep_mlp_t* drift,
                                     const tsde_deep_mlp_t* diffusion, const tsde_traj_t* traj, const void* times, int dtype) {
  if (!drift || !diffusion || !traj || !times) return "null argument";
  if (dtype != TSDE_F32) return "dtype must be TSDE_F32";
  if (rows < 0) return "need rows >= 0";
  if (d < 1 || d > 64) return "need d in [1, 64]";
  int64_t out = d;
  if (noise == TSDE_NOISE_GENERAL) {
    if (m < 1 || m > 16) return "general noise: m must be in [1, 16]";
    out = d * m;
  } else if (noise == TSDE_NOISE_DIAGONAL || noise == TSDE_NOISE_SCALAR) {
    if (m != (noise == TSDE_NOISE_DIAGONAL ? d : 1)) return "m must be d (diagonal) or 1 (scalar)";
  } else {
    return "unknown noise kind";
  }
  if (const char* bad = deep_mlp_problem(drift, d, d, "drift")) return bad;
  if (const char* bad = deep_mlp_problem(diffusion, d, out, "diffusion")) return bad;
  if (tsde::rheun_footprint(d, m, drift->hidden, diffusion->hidden, diffusion->out, noise, drift->n_mid, diffusion->n_mid) == 0)
    return "no kernel for this shape";
  if (rows * d >= (int64_t(1) << 30)) return "need rows * d < 2^30 (32-bit lane offsets)";
  if (traj->n_steps < 0 || traj->n_out < 0) return "negative schedule length";
  if (traj->n_steps > 0 && (!traj->step_rows || !traj->cells)) return "schedule without step rows";
  if (traj->n_out > 0 && (!traj->out_step || !traj->out_w)) return "schedule without output map";
  return nullptr;
}

int tsde_rheun_mlp_forward(void* ys, void* z_out, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                           const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, const tsde_traj_t* traj,
                           const void* times, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                           void* stream) {
  return tsde_deep_mlp_forward(ys, z_out, y0, rows, d, m, noise, drift, diffusion, TSDE_TRAJ_REVERSIBLE_HEUN, traj, times, entropy,
                               elem0, entropy_dev, dtype, stream);
}

int tsde_deep_mlp_forward(void* ys, void* z_out, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                          const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, int method, const tsde_traj_t* traj,
                          const void* times, uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, int dtype,
                          void* stream) {
  const char* where = method == TSDE_TRAJ_REVERSIBLE_HEUN ? "tsde_rheun_mlp_forward" : "tsde_deep_mlp_forward";
  if (method != TSDE_TRAJ_REVERSIBLE_HEUN && method != TSDE_TRAJ_EULER && method != TSDE_TRAJ_MIDPOINT &&
      method != TSDE_TRAJ_HEUN && method != TSDE_TRAJ_EULER_HEUN)
    return bad_arg(where, "method must be Euler, midpoint, Heun, Euler-Heun or reversible Heun");
  if (!ys || !y0) return bad_arg(where, "null argument");
  if (const char* bad = rheun_mlp_problem(rows, d, m, noise, drift, diffusion, traj, times, dtype)) return bad_arg(where, bad);
  const uintptr_t mask = d % 4 == 0 ? 15u : 3u;
  if (((reinterpret_cast<uintptr_t>(y0) | reinterpret_cast<uintptr_t>(ys) | reinterpret_cast<uintptr_t>(z_out)) & mask) != 0)
    return bad_arg(where, "ys, z_out and y0 must be 16-byte aligned (4-byte when d is not a multiple of 4)");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_RHEUN_MLP, s, true);
  const hipError_t e = tsde::launch_rheun_mlp_forward(ys, z_out, y0, rows, d, m, noise, drift, diffusion, method, traj, times,
                                                      make_key(entropy, elem0), entropy_dev, s);
  if (e == hipErrorInvalidValue) return bad_arg(where, "the weights of this shape do not fit the LDS of a CU");
  return fail(e, where);
}

int64_t tsde_rheun_mlp_lds(int64_t d, int64_t m, int64_t drift_hidden, int64_t diffusion_hidden, int64_t diffusion_out,
                           int noise, int drift_mid, int diffusion_mid) {
  return (int64_t)tsde::rheun_footprint(d, m, drift_hidden, diffusion_hidden, diffusion_out, noise, drift_mid, diffusion_mid);
}

int tsde_rheun_mlp_backward(const tsde_rheun_state_t* state, const tsde_rheun_stash_t* stash, const void* ys_all,
                            const void* grad_ys, int64_t rows, int64_t d, int64_t m, int noise,
                            const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, const tsde_traj_t* traj,
                            const void* times, int32_t j_hi, int32_t j_lo, uint64_t entropy, uint64_t elem0,
                            const uint64_t* entropy_dev, int dtype, void* stream) {
  const char* where = "tsde_rheun_mlp_backward";
  if (!state || !stash || !ys_all || !grad_ys) return bad_arg(where, "null argument");
  if (!state->y || !state->z || !state->a_y || !state->a_z || !state->a_f || !state->p) return bad_arg(where, "null state buffer");
  if (const char* bad = rheun_mlp_problem(rows, d, m, noise, drift, diffusion, traj, times, dtype)) return bad_arg(where, bad);
  if (j_lo < 0 || j_hi < j_lo || j_hi > traj->n_steps) return bad_arg(where, "need 0 <= j_lo <= j_hi <= n_steps");
  for (int i = 0; i < traj->n_out; ++i) {
    (void)i;      // (the output map lives on the device: that every output sits on a step boundary is the caller's contract)
  }
  if ((stash->stride_d | stash->stride_m | stash->stride_hf | stash->stride_hg) & 3) return bad_arg(where, "stash strides must be multiples of 4");
  const void* aligned[] = {state->y, state->z, state->a_y, state->a_z, state->a_f, state->p, ys_all, grad_ys, stash->z, stash->cf,
                           stash->p, stash->q, stash->wa, stash->wb};
  const uintptr_t mask = d % 4 == 0 ? 15u : 3u;
  for (const void* q : aligned)
    if (reinterpret_cast<uintptr_t>(q) & mask) return bad_arg(where, "state-shaped buffers must be 16-byte aligned");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope p(TSDE_KID_RHEUN_MLP, s, true);
  const hipError_t e = tsde::launch_rheun_mlp_backward(state, stash, ys_all, grad_ys, rows, d, m, noise, drift, diffusion, traj,
                                                       times, j_hi, j_lo, make_key(entropy, elem0), entropy_dev, s);
  if (e == hipErrorInvalidValue) return bad_arg(where, "the weights of this shape do not fit the LDS of a CU");
  return fail(e, where);
}

int tsde_rheun_last_layer_grad(void* gw, void* gb, const void* hid, const void* p, const void* q, const void* wa, const void* wb,
                               int64_t n_rows, int64_t d, int64_t m, const tsde_deep_mlp_t* diffusion, int32_t stride_h,
                               int32_t stride_d, int32_t stride_m, int32_t row_blocks, int dtype, void* stream) {
  const char* where = "tsde_rheun_last_layer_grad";
  if (!gw || !gb || !hid || !p || !q || !wa || !wb || !diffusion) return bad_arg(where, "null argument");
  if (dtype != TSDE_F32) return bad_arg(where, "dtype must be TSDE_F32");
  if (!diffusion->w2 || !diffusion->b2) return bad_arg(where, "a net without its last layer");
  if (n_rows < 0 || d < 1 || m < 1 || m > 255 || row_blocks < 1) return bad_arg(where, "need n_rows >= 0, d >= 1, 1 <= m <= 255, row_blocks >= 1");
  if (diffusion->hidden < 1 || diffusion->hidden > 64) return bad_arg(where, "hidden sizes must be in [1, 64]");
  if (diffusion->out != d * m) return bad_arg(where, "the net maps to d * m outputs");
  if (diffusion->final != TSDE_FINAL_NONE && diffusion->final != TSDE_FINAL_SIGMOID && diffusion->final != TSDE_FINAL_TANH)
    return bad_arg(where, "unknown output function");
  if (((stride_h | stride_d | stride_m) & 3) || stride_h < 4 || stride_d < d || stride_m < m)
    return bad_arg(where, "strides must be multiples of 4 that cover the widths");
  if ((reinterpret_cast<uintptr_t>(hid) & 15u) != 0) return bad_arg(where, "hid must be 16-byte aligned");
  const hipStream_t s = (hipStream_t)stream;
  ProfScope prof(TSDE_KID_RHEUN_MLP, s, true);
  return fail(tsde::launch_rheun_last_layer_grad(gw, gb, hid, p, q, wa, wb, n_rows, d, m, diffusion, stride_h, stride_d, stride_m,
                                                 row_blocks, s), where);
}

int tsde_prof_begin(int kid, int capacity) {
  if (capacity <= 0) return bad_arg("tsde_prof_begin", "capacity must be positive");
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.ev.resize(2 * (size_t)capacity);
  for (auto& e : g_prof.ev) {
    const hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return fail(r, "tsde_prof_begin");
  }
  g_prof.kid = kid;
  g_prof.cap = capacity;
  g_prof.used = 0;
  return 0;
}

namespace {
__global__ void delay_kernel(long long ticks) {
  const long long start = wall_clock64();   // constant 100 MHz counter
  while (wall_clock64() - start < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

int tsde_graph_memset_nodes_to_kernels(void* hip_graph, int* n_memset, int* n_replaced) {
  if (!hip_graph || !n_memset || !n_replaced) return bad_arg("tsde_graph_memset_nodes_to_kernels", "null argument");
  return fail(tsde::memset_nodes_to_kernels((hipGraph_t)hip_graph, n_memset, n_replaced),
              "tsde_graph_memset_nodes_to_kernels");
}

int tsde_delay_us(double microseconds, void* stream) {
  if (!(microseconds >= 0.0) || microseconds > 2.0e6) return bad_arg("tsde_delay_us", "delay must be in [0, 2 s]");
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long)(microseconds * 100.0));
  return fail(hipGetLastError(), "tsde_delay_us");
}

namespace {
__global__ void timed_spin_kernel(long long ticks, long long* measured) {
  const long long start = wall_clock64();
  long long now = start;
  while (now - start < ticks) now = wall_clock64();
  *measured = now - start;
}
}  // namespace

int tsde_prof_bracket_overhead(int n, double spin_us, double* overhead_ms, void* stream) {
  // What an event bracket adds to the kernel inside it: bracket a single-thread kernel that spins for `spin_us`
  // and reports its own duration from the constant-rate wall clock; overhead = mean(bracket - self-measured).
  if (n <= 0 || !overhead_ms || !(spin_us > 0.0)) return bad_arg("tsde_prof_bracket_overhead", "bad arguments");
  const hipStream_t s = (hipStream_t)stream;
  int dev = 0, khz = 0;
  hipError_t r = hipGetDevice(&dev);
  if (r == hipSuccess) r = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (r != hipSuccess || khz <= 0) return fail(r == hipSuccess ? hipErrorUnknown : r, "tsde_prof_bracket_overhead");
  const double ticks_per_us = khz * 1e-3;
  long long* measured = nullptr;
  r = hipMalloc(&measured, sizeof(long long) * (size_t)n);
  if (r != hipSuccess) return fail(r, "tsde_prof_bracket_overhead");
  std::vector<hipEvent_t> ev(2 * (size_t)n);
  for (auto& e : ev) {
    r = hipEventCreate(&e);
    if (r != hipSuccess) return fail(r, "tsde_prof_bracket_overhead");
  }
  for (int i = 0; i < n; ++i) {
    (void)hipEventRecord(ev[2 * i], s);
    hipLaunchKernelGGL(timed_spin_kernel, dim3(1), dim3(1), 0, s, (long long)(spin_us * ticks_per_us), measured + i);
    (void)hipEventRecord(ev[2 * i + 1], s);
  }
  std::vector<long long> host(n);
  r = hipStreamSynchronize(s);
  if (r == hipSuccess) r = hipMemcpy(host.data(), measured, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost);
  double sum = 0.0;
  for (int i = 0; i < n && r == hipSuccess; ++i) {
    float ms = 0.f;
    r = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
    sum += (double)ms - (double)host[i] / ticks_per_us * 1e-3;
  }
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  (void)hipFree(measured);
  *overhead_ms = sum / n;
  return fail(r, "tsde_prof_bracket_overhead");
}

int tsde_prof_bracket_open(int kid, void* stream) {
  if (g_prof.kid != kid || g_prof.used >= g_prof.cap) return -1;
  const int slot = g_prof.used++;
  (void)hipEventRecord(g_prof.ev[2 * slot], (hipStream_t)stream);
  return slot;
}

int tsde_prof_bracket_close(int slot, void* stream) {
  if (slot < 0 || slot >= g_prof.used) return bad_arg("tsde_prof_bracket_close", "no such bracket");
  return fail(hipEventRecord(g_prof.ev[2 * slot + 1], (hipStream_t)stream), "tsde_prof_bracket_close");
}

int tsde_prof_read(double* ms, int capacity, int* used) {
  hipError_t r = hipSuccess;
  if (used) *used = g_prof.used < capacity ? g_prof.used : capacity;
  for (int i = 0; i < g_prof.used && i < capacity && r == hipSuccess; ++i) {
    r = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    float t = 0.f;
    if (r == hipSuccess) r = hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    ms[i] = t;
  }
  return fail(r, "tsde_prof_read");
}

int tsde_prof_end(double* total_ms, int64_t* launches) {
  double sum = 0.0;
  hipError_t r = hipSuccess;
  for (int i = 0; i < g_prof.used && r == hipSuccess; ++i) {
    r = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    float ms = 0.f;
    if (r == hipSuccess) r = hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    sum += ms;
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = g_prof.used;
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.kid = 0;
  g_prof.cap = 0;
  g_prof.used = 0;
  return fail(r, "tsde_prof_end");
}

}  // extern "C"
