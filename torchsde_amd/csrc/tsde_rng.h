// Counter-based Brownian noise source for the MI355X-native torchsde hot path.
//
// Replaces the reference's per-tree-node `torch.Generator(seed) + torch.randn(size)`
// (torchsde/_brownian/brownian_interval.py:30-32, seeds from numpy SeedSequence :336-339)
// with a stateless Philox-4x32-10 field: every normal is a pure function of
//   (entropy, global element index, cell index, in-cell tree node, stream)
// so any increment can be (re)generated in registers, in any order, on any shard.
//
// This header is shared by the device kernels and by the host-side helpers the C-ABI
// exports for known-answer tests (include/torchsde_amd.h: tsde_philox4x32_10).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TSDE_HD __host__ __device__ __forceinline__
#define TSDE_D __device__ __forceinline__
#else
#define TSDE_HD inline
#endif

namespace tsde {

struct u32x4 {
  uint32_t x, y, z, w;
};

// a ^ b ^ c in one instruction on the device (gfx950 v_bitop3_b32, truth table 0x96); hipcc does not fuse the
// two xors of a Philox round on its own, and they are a third of the round's VALU work.
#if defined(__HIP_DEVICE_COMPILE__)
#define TSDE_XOR3(a, b, c) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96))
#else
#define TSDE_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif

// Philox-4x32-10 (Salmon et al., SC'11). One call = 4 x 32 random bits.
TSDE_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u;
  constexpr uint32_t kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)kM0 * c.x;
    const uint64_t p1 = (uint64_t)kM1 * c.z;
    const u32x4 n = {TSDE_XOR3((uint32_t)(p1 >> 32), c.y, k0), (uint32_t)p1,
                     TSDE_XOR3((uint32_t)(p0 >> 32), c.w, k1), (uint32_t)p0};
    c = n;
    k0 += kW0;
    k1 += kW1;
  }
  return c;
}

// Noise streams of one tree node.
enum : uint32_t { kStreamW = 0, kStreamH = 1, kStreamA = 2 };

// Identity of the noise field a kernel draws from.
struct NoiseKey {
  uint32_t k0, k1;   // entropy (low / high 32 bits)
  uint64_t elem0;    // global index of this shard's element 0 in the (B_global * m) noise tensor
};

// Counter layout:
//   c0 = quad[31:0]            quad = global element index >> 2 (4 normals per Philox call)
//   c1 = cell                  top-level cell of the Brownian grid
//   c2 = node[31:0]            heap index of the in-cell bridge-tree node (0 = the cell's own draw)
//   c3 = stream<<30 | quad[51:32]<<10 | node[41:32]
TSDE_HD u32x4 noise_counter(uint64_t quad, uint32_t cell, uint64_t node, uint32_t stream) {
  u32x4 c;
  c.x = (uint32_t)quad;
  c.y = cell;
  c.z = (uint32_t)node;
  c.w = (stream << 30) | (((uint32_t)(quad >> 32) & 0xFFFFFu) << 10) | ((uint32_t)(node >> 32) & 0x3FFu);
  return c;
}

TSDE_HD u32x4 noise_bits(const NoiseKey& key, uint64_t quad, uint32_t cell, uint64_t node, uint32_t stream) {
  return philox4x32_10(noise_counter(quad, cell, node, stream), key.k0, key.k1);
}

#if defined(__HIPCC__)
// ---- Box-Muller on the device ------------------------------------------------------------------
// Canonical definition (what the CPU oracle evaluates in double precision):
//   u1 = (a + 0.5) / 2^32 in (0,1),  theta = 2*pi * b / 2^32,
//   n0 = sqrt(-2 ln u1) cos(theta),  n1 = sqrt(-2 ln u1) sin(theta).
// fp32: v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (the latter two take revolutions, so theta
// needs no range reduction). Near u1 -> 1 the log is replaced by the series of -ln(1-w), w = 1-u1
// computed exactly from ~a, so small radii keep full relative accuracy.
TSDE_D void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = fmaf((float)a, 0x1p-32f, 0x1p-33f);
  const float w = fmaf((float)(~a), 0x1p-32f, 0x1p-33f);
  const float s_log = -1.3862943611198906f * __builtin_amdgcn_logf(u1);  // -2 ln2 * log2(u1)
  float p = fmaf(w, 1.0f / 6.0f, 0.2f);
  p = fmaf(w, p, 0.25f);
  p = fmaf(w, p, 1.0f / 3.0f);
  p = fmaf(w, p, 0.5f);
  p = fmaf(w, p, 1.0f);
  const float s_ser = 2.0f * w * p;
  const float s = (a >= 0xF0000000u) ? s_ser : s_log;
  const float r = __builtin_amdgcn_sqrtf(s);
  const float t = (float)b * 0x1p-32f;
  n0 = r * __builtin_amdgcn_cosf(t);
  n1 = r * __builtin_amdgcn_sinf(t);
}

TSDE_D void box_muller(uint32_t a, uint32_t b, double& n0, double& n1) {
  const double u1 = ((double)a + 0.5) * 0x1p-32;
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi((double)b * 0x1p-31, &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// Four standard normals of one (quad, cell, node, stream).
template <typename T>
TSDE_D void normal4(const NoiseKey& key, uint64_t quad, uint32_t cell, uint64_t node, uint32_t stream, T (&n)[4]) {
  const u32x4 r = noise_bits(key, quad, cell, node, stream);
  box_muller(r.x, r.y, n[0], n[1]);
  box_muller(r.z, r.w, n[2], n[3]);
}

// One standard normal for a single global element (generic / unaligned paths).
template <typename T>
TSDE_D T normal1(const NoiseKey& key, uint64_t elem, uint32_t cell, uint64_t node, uint32_t stream) {
  const u32x4 r = noise_bits(key, elem >> 2, cell, node, stream);
  const uint32_t lane = (uint32_t)elem & 3u;
  const uint32_t a = (lane & 2u) ? r.z : r.x;
  const uint32_t b = (lane & 2u) ? r.w : r.y;
  T n0, n1;
  box_muller(a, b, n0, n1);
  return (lane & 1u) ? n1 : n0;
}
#endif  // __HIPCC__

}  // namespace tsde
