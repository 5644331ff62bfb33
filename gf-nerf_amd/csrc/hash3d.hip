// hash3d.hip -- Hash3DAnchored lookup / scatter on gfx950.
//
// Replaces gfnerf/bindings/field/Hash3DAnchored_cuda.cu:11-155 (K1, K2) and the
// host wrappers :160-239.  Differences that are design, not numerics:
//   * lane = SAMPLE (64 consecutive samples of a ray per wave), levels walked in a
//     loop: neighbouring samples share cells on the coarse levels, so their gathers
//     collapse onto the same cache lines (the reference's thread = (point, level)
//     grid, :17-18, gives every lane of a gather a different level region);
//   * the table is gathered either from an f16 shadow kept current by the optimiser
//     kernel or straight from the f32 master with the rounding done in registers
//     (the reference converts the WHOLE pool to f16 on every call, :185);
//   * the focal stage's two lookups (global + residual block table,
//     nerfacto_field.py:477-486) share one pass over the samples;
//   * gradients: per-level run-length reduction across the wave (consecutive samples
//     in one cell) before the atomics; accumulation either in f32 or, like the
//     reference (:151), in packed f16 with the x128 loss scale (:209) kept in the
//     buffer.
// The level-overlap addressing quirk (:37 + :74: level base l*T elements, slot
// address pos*2+k) is kept bit-for-bit so reference checkpoints stay loadable.
#include <math.h>

#include "gfn_common.h"
#include "hash_device.h"

namespace {

template <typename AnchorT, bool F16, bool DUAL>
__global__ __launch_bounds__(256) void hash_fwd_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, int64_t n_volumes, int log2_T, LevelScales sc,
    const void* __restrict__ table_a, const void* __restrict__ table_b, const int* __restrict__ prim_pool,
    const int* __restrict__ prim_pool_b, const float* __restrict__ bias_pool, const float* __restrict__ pts,
    const AnchorT* __restrict__ anchors, int anchor_stride, int accumulate, float* __restrict__ out) {
  const int64_t n = gfn_live_n(n_max, n_dev);
  const uint32_t T = 1u << log2_T;
  // wave-uniform trip count: every lane of a wave runs the same number of iterations
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
    const int64_t pt = base + threadIdx.x;
    const bool valid = pt < n;
    const int64_t q = valid ? pt : n - 1;  // clamp: invalid lanes redo the last sample, store nothing
    const float px = pts[3 * q], py = pts[3 * q + 1], pz = pts[3 * q + 2];
    const int64_t vol = (int64_t)anchors[q * anchor_stride];
    float o[32];
#pragma unroll
    for (int level = 0; level < GFN_N_LEVELS; level++) {
      const int64_t tr = (int64_t)level * n_volumes + vol;  // :32,:40
      Corners c;
      uint32_t cell[3];
      hash_corners(px, py, pz, sc.s[level], prim_pool + 3 * tr, bias_pool ? bias_pool + 3 * tr : nullptr, T - 1u, c, cell);
      float2 r = gather_level<F16>(table_a, (int64_t)level * T, c);
      if (DUAL) {  // hash_feats += residual_hash_feats (nerfacto_field.py:486): f32 add of two f16-valued terms
        // every Hash3DAnchored draws its own primes (Hash3DAnchored.cpp:43-50): same cell and weights, other slots
        if (prim_pool_b) hash_corners(px, py, pz, sc.s[level], prim_pool_b + 3 * tr, bias_pool ? bias_pool + 3 * tr : nullptr, T - 1u, c, cell);
        const float2 r2 = gather_level<F16>(table_b, (int64_t)level * T, c);
        r.x = r.x + r2.x; r.y = r.y + r2.y;
      }
      o[2 * level] = r.x; o[2 * level + 1] = r.y;
    }
    if (valid) {
      float4* dst = reinterpret_cast<float4*>(out + pt * 32);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        float4 v = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        if (accumulate) {
          const float4 p = dst[k];
          v.x = p.x + v.x; v.y = p.y + v.y; v.z = p.z + v.z; v.w = p.w + v.w;
        }
        dst[k] = v;
      }
    }
  }
}

// Sum v over the lanes [lane, run_end] (runs are maximal groups of consecutive lanes with the
// same cell); the run head (lowest lane) ends up with the run total.
__device__ __forceinline__ float run_sum(float v, int lane, int run_end) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float u = __shfl_down(v, off, 64);
    if (lane + off <= run_end) v += u;
  }
  return v;
}

template <typename AnchorT, bool GRAD_F16>
__global__ __launch_bounds__(256) void hash_bwd_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, int64_t n_volumes, int log2_T, LevelScales sc,
    const int* __restrict__ prim_pool, const float* __restrict__ bias_pool, const float* __restrict__ pts,
    const AnchorT* __restrict__ anchors, int anchor_stride, const float* __restrict__ grad_out,
    void* __restrict__ grad_pool) {
  __shared__ float s_scale[GFN_N_LEVELS];
  if (threadIdx.x < GFN_N_LEVELS) s_scale[threadIdx.x] = sc.s[threadIdx.x];
  __syncthreads();
  const int64_t n = gfn_live_n(n_max, n_dev);
  const uint32_t T = 1u << log2_T;
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
    const int64_t pt = base + threadIdx.x;
    const bool valid = pt < n;
    const int64_t q = valid ? pt : n - 1;
    const float px = pts[3 * q], py = pts[3 * q + 1], pz = pts[3 * q + 2];
    const int64_t vol = (int64_t)anchors[q * anchor_stride];
    const float2* gp = reinterpret_cast<const float2*>(grad_out + q * 32);
#pragma unroll 1
    for (int level = 0; level < GFN_N_LEVELS; level++) {
      const int64_t tr = (int64_t)level * n_volumes + vol;
      Corners c;
      uint32_t cell[3];
      hash_corners(px, py, pz, s_scale[level], prim_pool + 3 * tr, bias_pool ? bias_pool + 3 * tr : nullptr, T - 1u, c, cell);
      const float2 g = gp[level];
      float g0 = gfn_round_f16(g.x * 128.f);  // :219 grad_in = (grad * 128).half()
      float g1 = gfn_round_f16(g.y * 128.f);
      const bool live = valid && (g0 != 0.f || g1 != 0.f);  // :149
      if (!live) { g0 = 0.f; g1 = 0.f; }
      // run detection: same volume and same integer cell as the previous lane => same 8 slots
      const uint32_t pvol = __shfl_up((uint32_t)vol, 1, 64), pcx = __shfl_up(cell[0], 1, 64);
      const uint32_t pcy = __shfl_up(cell[1], 1, 64), pcz = __shfl_up(cell[2], 1, 64);
      const bool head = (lane == 0) || pvol != (uint32_t)vol || pcx != cell[0] || pcy != cell[1] || pcz != cell[2];
      const unsigned long long heads = __ballot(head);
      float c0[8], c1[8];
#pragma unroll
      for (int d = 0; d < 8; d++) {  // :150 (half)(w0*ws[d]) -- each contribution quantised like the reference
        c0[d] = gfn_round_f16(g0 * c.w[d]);
        c1[d] = gfn_round_f16(g1 * c.w[d]);
      }
      bool issue = live;
      if (__popcll(heads) <= 40) {  // wave-uniform: runs are long enough to pay for the shuffles
        const unsigned long long later = (lane == 63) ? 0ull : (heads >> (lane + 1));
        const int run_end = later ? lane + __builtin_ctzll(later) : 63;
#pragma unroll
        for (int d = 0; d < 8; d++) {
          c0[d] = run_sum(c0[d], lane, run_end);
          c1[d] = run_sum(c1[d], lane, run_end);
        }
        issue = head;  // zero sums are skipped below
      }
      if (issue) {
        if (GRAD_F16) {
          half2_t* baseh = reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(grad_pool) + (int64_t)level * T);
#pragma unroll
          for (int d = 0; d < 8; d++) {
            if (c0[d] != 0.f || c1[d] != 0.f) {
              const half2_t v = {(_Float16)c0[d], (_Float16)c1[d]};  // x128-scaled, like the reference's pool (:222,:151)
              __builtin_amdgcn_global_atomic_fadd_v2f16(
                  (__attribute__((address_space(1))) half2_t*)(baseh + c.pos[d]), v);
            }
          }
        } else {
          float* basef = reinterpret_cast<float*>(grad_pool) + (int64_t)level * T;
#pragma unroll
          for (int d = 0; d < 8; d++) {
            if (c0[d] != 0.f || c1[d] != 0.f) {
              float* dst = basef + (size_t)c.pos[d] * 2;
              atomicAdd(dst, c0[d] * (1.f / 128.f));  // the reference's final /128 (:238), exact (power of two)
              atomicAdd(dst + 1, c1[d] * (1.f / 128.f));
            }
          }
        }
      }
    }
  }
}

int check_common(const char* fn, int64_t n_max, int64_t n_volumes, int log2_T, int anchor_stride) {
  GFN_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31), "%s: n_max=%lld out of range", fn, (long long)n_max);
  GFN_CHECK_ARG(n_volumes > 0 && n_volumes < (1ll << 26), "%s: n_volumes=%lld", fn, (long long)n_volumes);
  GFN_CHECK_ARG(log2_T >= 4 && log2_T <= 26, "%s: log2_T=%d must be in [4,26]", fn, log2_T);
  GFN_CHECK_ARG(anchor_stride >= 1, "%s: anchor_stride=%d", fn, anchor_stride);
  return GFN_OK;
}

unsigned point_grid(int64_t n_max) {
  int64_t blocks = (n_max + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 workgroups per CU
  return (unsigned)(blocks > 0 ? blocks : 1);
}

template <typename AnchorT>
void launch_fwd(bool f16, bool dual, unsigned grid, hipStream_t st, int64_t n_max, const int32_t* n_dev, int64_t V,
                int log2_T, const LevelScales& sc, const void* ta, const void* tb, const int* prim, const int* prim_b,
                const float* bias, const float* pts, const AnchorT* anchors, int stride, int accumulate, float* out) {
#define GFN_FWD(F, D)                                                                                              \
  hash_fwd_kernel<AnchorT, F, D><<<grid, 256, 0, st>>>(n_max, n_dev, V, log2_T, sc, ta, tb, prim, prim_b, bias, pts, anchors, \
                                                       stride, accumulate, out)
  if (f16) { if (dual) GFN_FWD(true, true); else GFN_FWD(true, false); }
  else     { if (dual) GFN_FWD(false, true); else GFN_FWD(false, false); }
#undef GFN_FWD
}

int hash_fwd_impl(const char* fn, const void* ta, const void* tb, bool f16, const int32_t* prim_pool,
                  const int32_t* prim_pool_b, const float* bias_pool, const float* pts, const void* anchors, int anchors_i64, int anchor_stride,
                  int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T, int accumulate, float* out,
                  void* stream) {
  if (n_max == 0) return GFN_OK;  // empty batch: pointers may be null
  GFN_CHECK_ARG(ta && prim_pool && pts && anchors && out, "%s: null pointer", fn);
  if (int rc = check_common(fn, n_max, n_volumes, log2_T, anchor_stride)) return rc;
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0, "%s: out must be 16-byte aligned", fn);
  LevelScales sc;
  gfn_hash_level_scales_host(sc.s);
  const unsigned grid = point_grid(n_max);
  hipStream_t st = gfn_stream(stream);
  if (anchors_i64)
    launch_fwd<int64_t>(f16, tb != nullptr, grid, st, n_max, n_dev, n_volumes, log2_T, sc, ta, tb, prim_pool, prim_pool_b, bias_pool, pts,
                        (const int64_t*)anchors, anchor_stride, accumulate, out);
  else
    launch_fwd<int32_t>(f16, tb != nullptr, grid, st, n_max, n_dev, n_volumes, log2_T, sc, ta, tb, prim_pool, prim_pool_b, bias_pool, pts,
                        (const int32_t*)anchors, anchor_stride, accumulate, out);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (_Float16)src[i];
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
  return hash_fwd_impl("gfn_hash_fwd", feat_pool, nullptr, false, prim_pool, nullptr, bias_pool, pts, anchors, anchors_i64,
                       anchor_stride, n_max, n_dev, n_volumes, log2_T, accumulate, out, stream);
}

int gfn_hash_fwd_f16(const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool,
                     const int32_t* prim_pool2, const float* bias_pool, const float* pts, const void* anchors, int anchors_i64,
                     int anchor_stride, int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                     float* out, void* stream) {
  return hash_fwd_impl("gfn_hash_fwd_f16", table_f16, table2_f16, true, prim_pool, table2_f16 ? prim_pool2 : nullptr, bias_pool, pts, anchors, anchors_i64,
                       anchor_stride, n_max, n_dev, n_volumes, log2_T, 0, out, stream);
}

int gfn_table_to_f16(const float* feat_pool, uint16_t* table_f16, int64_t n_elems, void* stream) {
  if (n_elems == 0) return GFN_OK;
  GFN_CHECK_ARG(feat_pool && table_f16 && n_elems > 0, "gfn_table_to_f16: bad arguments");
  int64_t blocks = (n_elems + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  f32_to_f16_kernel<<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(feat_pool, reinterpret_cast<_Float16*>(table_f16), n_elems);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_hash_bwd(const int32_t* prim_pool, const float* bias_pool, const float* pts,
                 const void* anchors, int anchors_i64, int anchor_stride, const float* grad_out,
                 int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                 void* grad_pool, int grad_f16, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(prim_pool && pts && anchors && grad_out && grad_pool, "gfn_hash_bwd: null pointer");
  if (int rc = check_common("gfn_hash_bwd", n_max, n_volumes, log2_T, anchor_stride)) return rc;
  LevelScales sc;
  gfn_hash_level_scales_host(sc.s);
  const unsigned grid = point_grid(n_max);
  hipStream_t st = gfn_stream(stream);
#define GFN_BWD(A, H)                                                                                               \
  hash_bwd_kernel<A, H><<<grid, 256, 0, st>>>(n_max, n_dev, n_volumes, log2_T, sc, prim_pool, bias_pool, pts, (const A*)anchors, \
                                              anchor_stride, grad_out, grad_pool)
  if (anchors_i64) { if (grad_f16) GFN_BWD(int64_t, true); else GFN_BWD(int64_t, false); }
  else             { if (grad_f16) GFN_BWD(int32_t, true); else GFN_BWD(int32_t, false); }
#undef GFN_BWD
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
