// Launchers of the reversible-Heun neural-SDE kernels (tsde_neural_rheun.h): tsde_rheun_mlp_forward / _backward.
#include "tsde_neural_rheun.h"

namespace tsde {

static size_t rheun_lds_limit() {
  static const size_t limit = [] {
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&bytes, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && bytes > 0)
      return (size_t)bytes;
    return (size_t)(64 * 1024);
  }();
  return limit;
}

// (general noise: a multiple of 32 -- the kernel takes the diffusion's output tiles two at a time)
// (up to 32 state channels the padded width D * mode: the kernels' row stride is then a compile-time constant)
static int rheun_outp(int D, int d, int out, int mode) {
  return mode >= 4 ? ((D <= 32 ? D : d) * mode + 31) / 32 * 32 : (out + 15) / 16 * 16;
}

template <int D, int H, int MODE, bool BACKWARD>
static hipError_t launch_rheun_mode(const RheunArgs& p, hipStream_t s) {
  const int outp = rheun_outp(D, p.d, p.g.out, MODE);
  const size_t lds_bytes = rheun_lds_floats(D, H, outp, p.f.n_mid, p.g.n_mid) * sizeof(float);
  if (lds_bytes > rheun_lds_limit()) return hipErrorInvalidValue;
  static bool configured = false;   // per instantiation
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&neural_rheun_kernel<D, H, MODE, BACKWARD>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int64_t groups = (p.B + 15) / 16;
  int64_t blocks = (groups + 3) / 4;
  const int64_t per_cu = (int64_t)((160 * 1024) / lds_bytes) < 1 ? 1 : (int64_t)((160 * 1024) / lds_bytes);
  const int64_t resident = 256 * (per_cu > 8 ? 8 : per_cu);
  if (blocks > resident) blocks = resident;
  TSDE_LAUNCH((neural_rheun_kernel<D, H, MODE, BACKWARD>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, p, outp);
  return hipGetLastError();
}

template <int D, int H, bool BACKWARD>
static hipError_t launch_rheun_dh(const RheunArgs& p, int noise, hipStream_t s) {
  if (noise == TSDE_NOISE_DIAGONAL) return launch_rheun_mode<D, H, 0, BACKWARD>(p, s);
  if (noise == TSDE_NOISE_SCALAR) return launch_rheun_mode<D, H, 1, BACKWARD>(p, s);
  if (noise == TSDE_NOISE_GENERAL) {
    if (p.m < 1 || p.m > 16) return hipErrorInvalidValue;
    if (p.m <= 4) return launch_rheun_mode<D, H, 4, BACKWARD>(p, s);
    return launch_rheun_mode<D, H, 16, BACKWARD>(p, s);
  }
  return hipErrorInvalidValue;
}

template <bool BACKWARD>
static hipError_t launch_rheun(const RheunArgs& p, int noise, hipStream_t s) {
  const int h = p.f.hidden > p.g.hidden ? p.f.hidden : p.g.hidden;
  if (p.B <= 0) return hipSuccess;
  if (p.d <= 16) {
    if (h <= 32) return launch_rheun_dh<16, 32, BACKWARD>(p, noise, s);
    if (h <= 64) return launch_rheun_dh<16, 64, BACKWARD>(p, noise, s);
  } else if (p.d <= 32) {
    if (h <= 32) return launch_rheun_dh<32, 32, BACKWARD>(p, noise, s);
    if (h <= 64) return launch_rheun_dh<32, 64, BACKWARD>(p, noise, s);
  } else if (p.d <= 64) {
    if (h <= 32) return launch_rheun_dh<64, 32, BACKWARD>(p, noise, s);
    if (h <= 64) return launch_rheun_dh<64, 64, BACKWARD>(p, noise, s);
  }
  return hipErrorInvalidValue;
}

// LDS bytes of a shape, or 0 when no instantiation covers it (tsde_rheun_mlp_lds: the host asks before it routes a module here)
size_t rheun_footprint(int64_t d, int64_t m, int64_t hf, int64_t hg, int64_t out, int noise, int nmf, int nmg) {
  const int D = d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : 0;
  const int64_t h = hf > hg ? hf : hg;
  const int H = h <= 32 ? 32 : h <= 64 ? 64 : 0;
  if (D == 0 || H == 0 || d < 1 || nmf < 0 || nmg < 0 || nmf > kMaxMid || nmg > kMaxMid) return 0;
  int mode = noise == TSDE_NOISE_DIAGONAL ? 0 : noise == TSDE_NOISE_SCALAR ? 1 : -1;
  if (noise == TSDE_NOISE_GENERAL) {
    if (m < 1 || m > 16) return 0;
    mode = m <= 4 ? 4 : 16;
  }
  if (mode < 0) return 0;
  return rheun_lds_floats(D, H, rheun_outp(D, (int)d, (int)out, mode), nmf, nmg) * sizeof(float);
}

static DeepNet deep_view(const tsde_deep_mlp_t* n) {
  DeepNet v;
  v.w1 = (const float*)n->w1;
  v.w1t = (const float*)n->w1t;
  v.b1 = (const float*)n->b1;
  for (int l = 0; l < kMaxMid; ++l) {
    v.wm[l] = (const float*)n->wm[l];
    v.bm[l] = (const float*)n->bm[l];
  }
  v.w2 = (const float*)n->w2;
  v.b2 = (const float*)n->b2;
  v.hidden = n->hidden;
  v.out = n->out;
  v.act = n->activation;
  v.final = n->final;
  v.n_mid = n->n_mid;
  v.scale = (float)n->scale;
  v.act_scale = (float)n->act_scale;
  return v;
}

static void fill_common(RheunArgs& p, int64_t rows, int64_t d, int64_t m, const tsde_deep_mlp_t* drift,
                        const tsde_deep_mlp_t* diffusion, const tsde_traj_t* tr, const void* times, NoiseKey key,
                        const uint64_t* key_dev) {
  memset(&p, 0, sizeof(p));
  p.f = deep_view(drift);
  p.g = deep_view(diffusion);
  p.rows = (const float*)tr->step_rows;
  p.times = (const float*)times;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const float*)tr->out_w;
  p.B = rows;
  p.d = (int32_t)d;
  p.m = (int32_t)m;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key = key;
  p.key_dev = key_dev;
}

hipError_t launch_rheun_mlp_forward(void* ys, void* z_out, const void* y0, int64_t rows, int64_t d, int64_t m, int noise,
                                    const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, int method,
                                    const tsde_traj_t* tr, const void* times, NoiseKey key, const uint64_t* key_dev,
                                    hipStream_t s) {
  RheunArgs p;
  fill_common(p, rows, d, m, drift, diffusion, tr, times, key, key_dev);
  p.method = method;
  p.ys = (float*)ys;
  p.z_out = (float*)z_out;
  p.y0 = (const float*)y0;
  return launch_rheun<false>(p, noise, s);
}

hipError_t launch_rheun_mlp_backward(const tsde_rheun_state_t* state, const tsde_rheun_stash_t* stash, const void* ys_all,
                                     const void* grad_ys, int64_t rows, int64_t d, int64_t m, int noise,
                                     const tsde_deep_mlp_t* drift, const tsde_deep_mlp_t* diffusion, const tsde_traj_t* tr,
                                     const void* times, int j_hi, int j_lo, NoiseKey key, const uint64_t* key_dev,
                                     hipStream_t s) {
  RheunArgs p;
  fill_common(p, rows, d, m, drift, diffusion, tr, times, key, key_dev);
  p.s_y = (float*)state->y;
  p.s_z = (float*)state->z;
  p.s_ay = (float*)state->a_y;
  p.s_az = (float*)state->a_z;
  p.s_af = (float*)state->a_f;
  p.s_p = (float*)state->p;
  p.ys_all = (const float*)ys_all;
  p.gys = (const float*)grad_ys;
  p.j_hi = j_hi;
  p.j_lo = j_lo;
  RheunStash& st = p.st;
  st.z = (float*)stash->z;
  st.cf = (float*)stash->cf;
  st.p = (float*)stash->p;
  st.q = (float*)stash->q;
  st.wa = (float*)stash->wa;
  st.wb = (float*)stash->wb;
  for (int l = 0; l <= kMaxMid; ++l) {
    st.hf[l] = (float*)stash->hf[l];
    st.df[l] = (float*)stash->df[l];
    st.hg[l] = (float*)stash->hg[l];
    st.dg[l] = (float*)stash->dg[l];
  }
  st.sd = stash->stride_d;
  st.sm = stash->stride_m;
  st.shf = stash->stride_hf;
  st.shg = stash->stride_hg;
  return launch_rheun<true>(p, noise, s);
}

}  // namespace tsde
