// hash_device.h -- device-side pieces of the Hash3DAnchored lookup (Hash3DAnchored_cuda.cu:11-79) shared by
// hash3d.hip and fused_fwd.hip: corner hashing, trilinear weights, one level's gather + blend.
#pragma once
#include "gfn_common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

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

// :27-69.  Returns the integer cell (ux,uy,uz) through `cell` for run detection.
// MUL3: (u + 1) * p formed as u * p + p -- the same value in the uint32 wraparound arithmetic of the reference (:48-55) with
// three quarter-rate multiplies instead of six; false keeps the six multiplies, whose shorter live ranges one register-tight
// kernel needs (fused_fwd_kernel<1, false, false> spills 4 VGPRs otherwise; tests/test_abi_and_host.py checks for none).
template <bool MUL3 = true>
__device__ __forceinline__ void hash_corners(float px, float py, float pz, float scale, const int* __restrict__ prim,
                                             const float* __restrict__ bias, uint32_t mask, Corners& c,
                                             uint32_t cell[3]) {
  px = px * scale; py = py * scale; pz = pz * scale;  // :29
  if (bias) { px = px + bias[0]; py = py + bias[1]; pz = pz + bias[2]; }  // :41 (null = all-zero bias pool)
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const uint32_t ux = sat_u32(fx), uy = sat_u32(fy), uz = sat_u32(fz);
  cell[0] = ux; cell[1] = uy; cell[2] = uz;
  const uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  const uint32_t x0 = ux * pa, x1 = MUL3 ? x0 + pa : (ux + 1u) * pa;
  const uint32_t y0 = uy * pb, y1 = MUL3 ? y0 + pb : (uy + 1u) * pb;
  const uint32_t z0 = uz * pc, z1 = MUL3 ? z0 + pc : (uz + 1u) * pc;
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

// The same cell through another table's primes (the focal stage's residual table, nerfacto_field.py:485-486): only the 8 slots
// change -- cell and trilinear weights depend on the point alone (bias pools are shared by the tables here).
__device__ __forceinline__ void rehash_corners(const uint32_t cell[3], const int* __restrict__ prim, uint32_t mask, Corners& c) {
  const uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  const uint32_t x0 = cell[0] * pa, x1 = x0 + pa;
  const uint32_t y0 = cell[1] * pb, y1 = y0 + pb;
  const uint32_t z0 = cell[2] * pc, z1 = z0 + pc;
  c.pos[0] = (x0 ^ y0 ^ z0) & mask;
  c.pos[1] = (x0 ^ y0 ^ z1) & mask;
  c.pos[2] = (x0 ^ y1 ^ z0) & mask;
  c.pos[3] = (x0 ^ y1 ^ z1) & mask;
  c.pos[4] = (x1 ^ y0 ^ z0) & mask;
  c.pos[5] = (x1 ^ y0 ^ z1) & mask;
  c.pos[6] = (x1 ^ y1 ^ z0) & mask;
  c.pos[7] = (x1 ^ y1 ^ z1) & mask;
}

// one level of one table: 8 gathers, trilinear blend in f32, result rounded to f16 (:71-78,:195)
template <bool F16>
__device__ __forceinline__ float2 gather_level(const void* __restrict__ table, int64_t level_base, const Corners& c) {
  float2 f[8];
  if (F16) {
    const half2_t* base = reinterpret_cast<const half2_t*>(reinterpret_cast<const _Float16*>(table) + level_base);
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const half2_t v = base[c.pos[d]];
      f[d] = make_float2((float)v[0], (float)v[1]);
    }
  } else {
    const float2* base = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(table) + level_base);
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const float2 v = base[c.pos[d]];
      f[d] = make_float2(gfn_round_f16(v.x), gfn_round_f16(v.y));  // :185 pool.to(f16)
    }
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    const float t0 = c.w[d] * f[d].x, t1 = c.w[d] * f[d].y;  // :74-77, left-to-right sum
    a0 = (d == 0) ? t0 : a0 + t0;
    a1 = (d == 0) ? t1 : a1 + t1;
  }
  return make_float2(gfn_round_f16(a0), gfn_round_f16(a1));
}

}  // namespace
