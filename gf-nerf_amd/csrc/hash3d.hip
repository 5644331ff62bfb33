// hash3d.hip -- Hash3DAnchored lookup / scatter on gfx950.
//
// Replaces gfnerf/bindings/field/Hash3DAnchored_cuda.cu:11-155 (K1, K2) and the
// host wrappers :160-239.  Differences that are design, not numerics:
//   * the f32 master table is gathered directly and rounded to f16 in registers
//     (the reference converts the WHOLE pool to f16 on every call, :185);
//   * gradients are accumulated in f32 (reference: f16 atomics, :151), each
//     contribution quantised exactly as the reference does;
//   * anchors may be int32 (packed sampler output) or int64 (reference ABI).
// The level-overlap addressing quirk (:37 + :74: level base l*T elements, slot
// address pos*2+k) is kept bit-for-bit so reference checkpoints stay loadable.
#include <math.h>

#include "gfn_common.h"

namespace {

struct LevelScales {
  float s[GFN_N_LEVELS];
};

struct Corners {
  uint32_t pos[8];
  float w[8];
};

__device__ __forceinline__ uint32_t sat_u32(float f) {
  // static_cast<unsigned>(floorf(p)) (:44-46); saturating like the PTX cvt
  return f <= 0.f ? 0u : (f >= 4294967040.f ? 0xffffffffu : (uint32_t)f);
}

__device__ __forceinline__ void hash_corners(float px, float py, float pz, float scale, const int* prim,
                                             const float* bias, uint32_t mask, Corners& c) {
  px = px * scale; py = py * scale; pz = pz * scale;  // :29
  px = px + bias[0]; py = py + bias[1]; pz = pz + bias[2];  // :41
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const uint32_t ux = sat_u32(fx), uy = sat_u32(fy), uz = sat_u32(fz);
  const uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  const uint32_t x0 = ux * pa, x1 = (ux + 1u) * pa;
  const uint32_t y0 = uy * pb, y1 = (uy + 1u) * pb;
  const uint32_t z0 = uz * pc, z1 = (uz + 1u) * pc;
  // % local_size with local_size == T == 2^k (:48-55, Hash3DAnchored.cpp:66-67)
  c.pos[0] = (x0 ^ y0 ^ z0) & mask;
  c.pos[1] = (x0 ^ y0 ^ z1) & mask;
  c.pos[2] = (x0 ^ y1 ^ z0) & mask;
  c.pos[3] = (x0 ^ y1 ^ z1) & mask;
  c.pos[4] = (x1 ^ y0 ^ z0) & mask;
  c.pos[5] = (x1 ^ y0 ^ z1) & mask;
  c.pos[6] = (x1 ^ y1 ^ z0) & mask;
  c.pos[7] = (x1 ^ y1 ^ z1) & mask;
  const float a = px - fx, b = py - fy, cc = pz - fz;  // :58-60
  const float na = 1.f - a, nb = 1.f - b, nc = 1.f - cc;
  c.w[0] = (na * nb) * nc;  // :62-69
  c.w[1] = (na * nb) * cc;
  c.w[2] = (na * b) * nc;
  c.w[3] = (na * b) * cc;
  c.w[4] = (a * nb) * nc;
  c.w[5] = (a * nb) * cc;
  c.w[6] = (a * b) * nc;
  c.w[7] = (a * b) * cc;
}

template <typename AnchorT>
__global__ __launch_bounds__(256) void hash_fwd_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, int64_t n_volumes, int log2_T, LevelScales sc,
    const float* __restrict__ feat_pool, const int* __restrict__ prim_pool,
    const float* __restrict__ bias_pool, const float* __restrict__ pts,
    const AnchorT* __restrict__ anchors, int anchor_stride, int accumulate, float* __restrict__ out) {
  __shared__ float s_scale[GFN_N_LEVELS];
  if (threadIdx.x < GFN_N_LEVELS) s_scale[threadIdx.x] = sc.s[threadIdx.x];
  __syncthreads();
  const int64_t n = gfn_live_n(n_max, n_dev);
  const uint32_t T = 1u << log2_T;
  // thread = (point, level); 16 consecutive lanes = 16 levels of one point, so a
  // wave writes 4 points x 32 f32 = 512 contiguous bytes.
  const int level = threadIdx.x & 15;
  for (int64_t pt = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); pt < n; pt += (int64_t)gridDim.x * 16) {
    const int64_t vol = (int64_t)anchors[pt * anchor_stride];
    const int64_t tr = (int64_t)level * n_volumes + vol;  // :32,:40
    Corners c;
    hash_corners(pts[3 * pt], pts[3 * pt + 1], pts[3 * pt + 2], s_scale[level], prim_pool + 3 * tr,
                 bias_pool + 3 * tr, T - 1u, c);
    const float2* base = reinterpret_cast<const float2*>(feat_pool + (int64_t)level * T);
    float2 f[8];
#pragma unroll
    for (int d = 0; d < 8; d++) f[d] = base[c.pos[d]];  // 8 independent gathers in flight
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const float t0 = c.w[d] * gfn_round_f16(f[d].x);  // :185 pool.to(f16), :74-77 float(half)
      const float t1 = c.w[d] * gfn_round_f16(f[d].y);
      a0 = (d == 0) ? t0 : a0 + t0;
      a1 = (d == 0) ? t1 : a1 + t1;
    }
    float2 res = make_float2(gfn_round_f16(a0), gfn_round_f16(a1));  // :73,:195
    float2* dst = reinterpret_cast<float2*>(out) + pt * 16 + level;
    if (accumulate) {  // hash_feats += residual_hash_feats (nerfacto_field.py:486), f32 add
      const float2 prev = *dst;
      res.x = prev.x + res.x; res.y = prev.y + res.y;
    }
    *dst = res;
  }
}

template <typename AnchorT>
__global__ __launch_bounds__(256) void hash_bwd_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, int64_t n_volumes, int log2_T, LevelScales sc,
    const int* __restrict__ prim_pool, const float* __restrict__ bias_pool,
    const float* __restrict__ pts, const AnchorT* __restrict__ anchors, int anchor_stride,
    const float* __restrict__ grad_out, float* __restrict__ grad_pool) {
  __shared__ float s_scale[GFN_N_LEVELS];
  if (threadIdx.x < GFN_N_LEVELS) s_scale[threadIdx.x] = sc.s[threadIdx.x];
  __syncthreads();
  const int64_t n = gfn_live_n(n_max, n_dev);
  const uint32_t T = 1u << log2_T;
  const int level = threadIdx.x & 15;
  for (int64_t pt = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); pt < n; pt += (int64_t)gridDim.x * 16) {
    const float2 g = reinterpret_cast<const float2*>(grad_out)[pt * 16 + level];
    const float g0 = gfn_round_f16(g.x * 128.f);  // :219
    const float g1 = gfn_round_f16(g.y * 128.f);
    if (g0 == 0.f && g1 == 0.f) continue;  // :149
    const int64_t vol = (int64_t)anchors[pt * anchor_stride];
    const int64_t tr = (int64_t)level * n_volumes + vol;
    Corners c;
    hash_corners(pts[3 * pt], pts[3 * pt + 1], pts[3 * pt + 2], s_scale[level], prim_pool + 3 * tr,
                 bias_pool + 3 * tr, T - 1u, c);
    float* base = grad_pool + (int64_t)level * T;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      // :150 (half)(w0*ws[d]); the reference's final /128 (:238) is exact (power of two)
      const float c0 = gfn_round_f16(g0 * c.w[d]) * (1.f / 128.f);
      const float c1 = gfn_round_f16(g1 * c.w[d]) * (1.f / 128.f);
      float* dst = base + (size_t)c.pos[d] * 2;
      atomicAdd(dst, c0);
      atomicAdd(dst + 1, c1);
    }
  }
}

int check_common(const char* fn, int64_t n_max, int64_t n_volumes, int log2_T) {
  GFN_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31), "%s: n_max=%lld out of range", fn, (long long)n_max);
  GFN_CHECK_ARG(n_volumes > 0 && n_volumes < (1ll << 26), "%s: n_volumes=%lld", fn, (long long)n_volumes);
  GFN_CHECK_ARG(log2_T >= 4 && log2_T <= 26, "%s: log2_T=%d must be in [4,26]", fn, log2_T);
  return GFN_OK;
}

unsigned point_grid(int64_t n_max) {
  int64_t blocks = (n_max + 15) / 16;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond ~8 WG of 256 threads per CU
  return (unsigned)(blocks > 0 ? blocks : 1);
}

}  // namespace

extern "C" {

void gfn_hash_level_scales_host(float out_host[GFN_N_LEVELS]) {
  // Hash3DAnchored_cuda.cu:28; exact exp2 (double) of the f32 argument, as the oracle.
  for (int l = 0; l < GFN_N_LEVELS; l++) {
    const float arg = (10.f - 3.f) * (float)l / (float)(GFN_N_LEVELS - 1) + 3.f;
    out_host[l] = (float)exp2((double)arg);
  }
}

int gfn_hash_fwd(const float* feat_pool, const int32_t* prim_pool, const float* bias_pool,
                 const float* pts, const void* anchors, int anchors_i64, int anchor_stride,
                 int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                 int accumulate, float* out, void* stream) {
  if (n_max == 0) return GFN_OK;  // empty batch: pointers may be null
  GFN_CHECK_ARG(feat_pool && prim_pool && bias_pool && pts && anchors && out, "gfn_hash_fwd: null pointer");
  GFN_CHECK_ARG(anchor_stride >= 1, "gfn_hash_fwd: anchor_stride=%d", anchor_stride);
  if (int rc = check_common("gfn_hash_fwd", n_max, n_volumes, log2_T)) return rc;
  if (n_max == 0) return GFN_OK;
  LevelScales sc;
  gfn_hash_level_scales_host(sc.s);
  const unsigned grid = point_grid(n_max);
  if (anchors_i64)
    hash_fwd_kernel<int64_t><<<grid, 256, 0, gfn_stream(stream)>>>(n_max, n_dev, n_volumes, log2_T, sc, feat_pool, prim_pool,
                                                                  bias_pool, pts, (const int64_t*)anchors, anchor_stride, accumulate, out);
  else
    hash_fwd_kernel<int32_t><<<grid, 256, 0, gfn_stream(stream)>>>(n_max, n_dev, n_volumes, log2_T, sc, feat_pool, prim_pool,
                                                                  bias_pool, pts, (const int32_t*)anchors, anchor_stride, accumulate, out);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_hash_bwd(const int32_t* prim_pool, const float* bias_pool, const float* pts,
                 const void* anchors, int anchors_i64, int anchor_stride, const float* grad_out,
                 int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                 float* grad_pool, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(prim_pool && bias_pool && pts && anchors && grad_out && grad_pool, "gfn_hash_bwd: null pointer");
  GFN_CHECK_ARG(anchor_stride >= 1, "gfn_hash_bwd: anchor_stride=%d", anchor_stride);
  if (int rc = check_common("gfn_hash_bwd", n_max, n_volumes, log2_T)) return rc;
  if (n_max == 0) return GFN_OK;
  LevelScales sc;
  gfn_hash_level_scales_host(sc.s);
  const unsigned grid = point_grid(n_max);
  if (anchors_i64)
    hash_bwd_kernel<int64_t><<<grid, 256, 0, gfn_stream(stream)>>>(n_max, n_dev, n_volumes, log2_T, sc, prim_pool, bias_pool,
                                                                  pts, (const int64_t*)anchors, anchor_stride, grad_out, grad_pool);
  else
    hash_bwd_kernel<int32_t><<<grid, 256, 0, gfn_stream(stream)>>>(n_max, n_dev, n_volumes, log2_T, sc, prim_pool, bias_pool,
                                                                  pts, (const int32_t*)anchors, anchor_stride, grad_out, grad_pool);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
