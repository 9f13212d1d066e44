// Typed launchers behind the C ABI (capi.hip dispatches on dtype).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/torchsde_amd.h"
#include "tsde_bridge.h"
#include "tsde_rng.h"

namespace tsde {

struct QueryArgs {
  const double* edges;  // device pointer, n_cells + 1 doubles
  int64_t ca, cb;
  double a, b;
  const void* rootW;    // optional pinned (W,H) of the single top-level cell
  const void* rootH;
  const uint64_t* key_dev;  // optional run-time entropy
  const double* ab_dev;     // optional: (a, b) in device memory; then ca, cb, a, b above are ignored
  int64_t n_cells;          // number of cells behind `edges` (needed to locate a, b on the device)
  WalkCfg cfg;
};

template <typename T>
hipError_t launch_normals(void* out, int64_t n, NoiseKey key, uint32_t cell, uint64_t node, uint32_t stream_id,
                          hipStream_t s);
template <typename T>
hipError_t launch_query(void* W, void* U, void* H, int64_t n, NoiseKey key, const QueryArgs& qa, bool have_h,
                        hipStream_t s);
template <typename T>
hipError_t launch_cell_increment(void* W_out, void* U_out, int64_t n, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_step_diag(void* y1, const void* y0, const void* f, const void* g, int64_t n, double cf, double cg,
                            const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_step_prod(void* y1, const void* y0, const void* f, const void* gp, int64_t n, double cf, double cg,
                            hipStream_t s);
template <typename T>
hipError_t launch_step_general(void* y1, const void* y0, const void* f, const void* g, int64_t B, int64_t d, int64_t m,
                               double ca, double cf, double cg, int weight_mode, double cw, double cu, double rdt,
                               const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_step_shared(void* y1, const void* y0, const void* f, const void* S, int64_t B, int64_t d, int64_t m,
                              double ca, double cf, double cg, int weight_mode, double cw, double cu, double rdt,
                              const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_milstein_v(void* v_out, void* W_out, const void* g, int64_t n, double dt, int ito, double scale,
                             const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_milstein_diag(void* y1, const void* y0, const void* f, const void* g, const void* gdg, int64_t n,
                                double dt, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_milstein_gf_prime(void* yp, const void* y0, const void* f, const void* g, int64_t n, double dt,
                                    double sqrt_dt, int ito, hipStream_t s);
template <typename T>
hipError_t launch_milstein_gf_diag(void* y1, const void* y0, const void* f, const void* g, const void* gp, int64_t n,
                                   double dt, double sqrt_dt, int ito, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_srk_stage(int stage, void* const out[3], const void* const in[5], int64_t n, double dt, double rdt,
                            double sqrt_dt, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_milstein_gf_general_support(void* yk, const void* y0, const void* f, const void* g, int64_t B,
                                              int64_t d, int64_t m, double dt, double sqrt_dt, int ito, hipStream_t s);
template <typename T>
hipError_t launch_milstein_gf_general_correction(void* corr, const void* g, const void* gk, const void* I, int64_t B,
                                                 int64_t d, int64_t m, double sqrt_dt, hipStream_t s);
template <typename T>
hipError_t launch_aug_segments(const tsde_seg_t* segs, int nseg, double cF, double cG, hipStream_t s);
template <typename T>
hipError_t launch_interp(void* out, const void* ya, const void* yb, int64_t n, double w0, double w1, hipStream_t s);
}  // namespace tsde

namespace tsde {
// rheun.hip
template <typename T>
hipError_t launch_heun_final(void* y1, const void* y0, const void* f, const void* fp, const void* g, const void* gp,
                             int64_t n, double dt, int mode, int prod, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_iterated_integrals(void* I, const void* W, const void* A, int64_t B, int64_t m, double dt, int ito,
                                     hipStream_t s);
template <typename T>
hipError_t launch_levy_area(void* A, const void* W, const void* H, int64_t B, int64_t m, double h, int foster,
                            NoiseKey key, const uint64_t* key_dev, uint32_t cell, uint64_t node, hipStream_t s,
                            int fuse = 0, double dt = 0.0, int ito = 0);
template <typename T>
hipError_t launch_rheun_z(void* z1, const void* y0, const void* z0, const void* f0, const void* g0, int64_t n, double dt,
                          double sgn, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_rheun_y(void* y1, const void* y0, const void* f0, const void* f1, const void* g0, const void* g1,
                          int64_t n, double half_dt, double sgn, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_lincomb2(void* out, const void* x, const void* y, int64_t n, double a, double b, hipStream_t s);
template <typename T>
hipError_t launch_rheun_adj_a(void* af0_out, void* ag0_out, const void* ay, const void* af0, const void* ag0, int64_t n,
                              double half_dt, const tsde_noise_t* nz, hipStream_t s);
template <typename T>
hipError_t launch_rheun_adj_b(void* ay1, void* az1, void* af1, void* ag1, const void* ay, const void* az0,
                              const void* vjp_z, int64_t n, double dt, double half_dt, const tsde_noise_t* nz,
                              hipStream_t s);
// reduce.hip
template <typename T>
hipError_t launch_error_norm(double* out, double* workspace, const void* yf, const void* yh, int64_t n, double rtol,
                             double atol, double eps, hipStream_t s);
// adaptive.hip
template <typename T>
hipError_t launch_adaptive_begin(double* ctl, void* scal, double out_t, const double* out_times, int n_out,
                                 const double* fracs, int n_fracs, hipStream_t s);
template <typename T>
hipError_t launch_adaptive_control(double* ctl, void* scal, const double* error, const double* out_times,
                                   double* accept_log, int log_capacity, const double* fracs, int n_fracs, hipStream_t s);
template <typename T>
hipError_t launch_adaptive_emit(const void* ys_slot, const void* prev_y, const void* curr_y, int64_t n, const double* ctl,
                                const double* out_times, hipStream_t s);
template <typename T>
hipError_t launch_adaptive_commit(void* prev_y, void* curr_y, const void* y_next, int64_t n, const void* scal,
                                  hipStream_t s);
template <typename T>
hipError_t launch_merge_halves(void* W, void* U, const void* Wa, const void* Ha, const void* Wb, const void* Hb,
                               int64_t n, const double* ctl, double ha, double hb, hipStream_t s);
// trajectory.hip
template <typename T>
hipError_t launch_trajectory_affine_diag(void* ys, void* sens, const void* y0, int64_t rows, int64_t d, const void* a,
                                         const void* b, const void* c, const void* e, int64_t cstride, int method,
                                         const tsde_traj_t* tr, NoiseKey key, const uint64_t* key_dev, hipStream_t s);
template <typename T>
hipError_t launch_trajectory_expr_diag(void* ys, const void* y0, int64_t rows, int64_t d, const void* const coef[8],
                                       int64_t cstride, int f_kind, int g_kind, int method, const tsde_traj_t* tr,
                                       NoiseKey key, const uint64_t* key_dev, hipStream_t s);
template <typename T>
hipError_t launch_trajectory_prog_diag(void* ys, void* sens, const int8_t* param_slot, const void* y0, int64_t rows, int64_t d,
                                       const uint32_t* code, int f_len, int g_len, int dg_len, const void* consts, int n_const,
                                       int scalar_noise, int method, const tsde_traj_t* tr, NoiseKey key,
                                       const uint64_t* key_dev, hipStream_t s);
template <typename T>
hipError_t launch_trajectory_prog_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const uint32_t* code,
                                           int f_len, const void* consts, int n_const, const void* gtab, int time_dependent,
                                           int method, const tsde_traj_t* tr, NoiseKey key, const uint64_t* key_dev,
                                           hipStream_t s);
// mlp_trajectory.hip
hipError_t launch_trajectory_mlp_diag(void* ys, const void* y0, int64_t rows, int64_t d, int64_t h, const void* W1,
                                      const void* b1, const void* W2, const void* b2, const void* c, const void* e,
                                      int diff_kind, double diff_amp, int act, int method, const tsde_traj_t* tr,
                                      NoiseKey key, const uint64_t* key_dev, hipStream_t s);
// mlp_general.hip
hipError_t launch_trajectory_mlp_general(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                                         const tsde_mlp_t* drift, const tsde_mlp_t* diffusion, int method,
                                         const tsde_traj_t* tr, NoiseKey key, const uint64_t* key_dev, hipStream_t s);
hipError_t launch_trajectory_mlp_additive(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const tsde_mlp_t* drift,
                                          const void* gtab, int time_dependent, int method, const tsde_traj_t* tr,
                                          NoiseKey key, const uint64_t* key_dev, hipStream_t s);
size_t neural_footprint(int64_t d, int64_t m, int64_t hf, int64_t hg, int64_t out, int noise);
// mlp_backward.hip
hipError_t launch_trajectory_mlp_diag_backward(void* lam, void* stash_lam, void* stash_hid, void* stash_delta,
                                               void* row_rate, void* row_shift, const void* ys_all,
                                               int32_t ys_first, const void* grad_ys, const int32_t* grad_step, int32_t grad_last,
                                               int64_t rows, int64_t d, int64_t h, const void* W1, const void* b1,
                                               const void* W2, const void* c, const void* e, int diff_kind,
                                               double diff_amp, int act, int method, const tsde_traj_t* tr,
                                               int32_t k_lo, int32_t k_hi, NoiseKey key, const uint64_t* key_dev,
                                               hipStream_t s);
// mlp_adjoint.hip
hipError_t launch_adjoint_mlp_diag(void* y, void* a, void* stash_a, void* stash_hid, void* stash_delta, void* stash_y,
                                   void* row_rate, void* row_shift, int64_t rows, int64_t d, int64_t h, const void* W1,
                                   const void* b1, const void* W2, const void* b2, const void* c, const void* e,
                                   int diff_kind, double diff_amp, int act, int ito, const tsde_traj_t* tr, int32_t k_lo,
                                   int32_t k_hi, NoiseKey key, const uint64_t* key_dev, hipStream_t s);
hipError_t launch_gram_partials(void* partials, void* colsums, const void* A, int64_t lda, const void* Bm, int64_t ldb,
                                int64_t K, int64_t M, int64_t N, int32_t blocks, hipStream_t s);
// graph_nodes.hip: memset nodes of a captured, not yet instantiated graph -> fill-kernel nodes with the same edges
hipError_t memset_nodes_to_kernels(hipGraph_t graph, int* n_memset, int* n_replaced);
// neural_rheun.hip
hipError_t launch_rheun_mlp_forward(void* ys, void* z_out, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                                    const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, int method,
                                    const tsde_traj_t* tr, const void* times, NoiseKey key, const uint64_t* key_dev,
                                    hipStream_t s);
hipError_t launch_rheun_mlp_backward(const tsde_rheun_state_t* state, const tsde_rheun_stash_t* stash, const void* ys_all,
                                     const void* grad_ys, int64_t rows, int64_t d, int64_t m, int noise,
                                     const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, const tsde_traj_t* tr,
                                     const void* times, int j_hi, int j_lo, NoiseKey key, const uint64_t* key_dev,
                                     hipStream_t s);
hipError_t launch_rheun_last_layer_grad(void* gw, void* gb, const void* hid, const void* p, const void* q, const void* wa,
                                        const void* wb, int64_t N, int64_t d, int64_t m, const tsde_deep_mlp_t* net,
                                        int32_t stride_h, int32_t stride_d, int32_t stride_m, int32_t row_blocks, hipStream_t s);
size_t rheun_footprint(int64_t d, int64_t m, int64_t hf, int64_t hg, int64_t out, int noise, int nmf, int nmg);
}  // namespace tsde
