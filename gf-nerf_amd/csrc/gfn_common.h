// gfn_common.h -- shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gfnerf_amd.h"

static_assert(sizeof(gfn_tree_node) == GFN_TREE_NODE_BYTES, "TreeNode ABI (PersSampler.h:39-48)");
static_assert(sizeof(gfn_trans_info) == GFN_TRANS_INFO_BYTES, "TransInfo ABI (PersSampler.h:31-37)");
static_assert(offsetof(gfn_tree_node, trans_idx) == 96, "TreeNode.trans_idx");
static_assert(offsetof(gfn_trans_info, weight) == 384, "TransInfo.weight");
static_assert(offsetof(gfn_trans_info, dis_summary) == 544, "TransInfo.dis_summary");

// error plumbing (defined in runtime.hip)
void gfn_set_error(const char* fmt, ...);
#define GFN_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      gfn_set_error(__VA_ARGS__);     \
      return GFN_ERR_ARG;             \
    }                                 \
  } while (0)
#define GFN_CHECK_HIP(expr)                                                          \
  do {                                                                               \
    hipError_t e__ = (expr);                                                         \
    if (e__ != hipSuccess) {                                                         \
      gfn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return GFN_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)
#define GFN_CHECK_LAUNCH() GFN_CHECK_HIP(hipGetLastError())

// cap on the workgroups (= CUs) of the persistent kernels from the call's launch policy (gfn_launch_opts.decoder_cus), 256 = every CU
inline int gfn_decoder_cu_cap(const gfn_launch_opts* opts) { return (opts && opts->decoder_cus > 0 && opts->decoder_cus < 256) ? opts->decoder_cus : 256; }
// amplitude of the one-plane (PL_RNE) producers' per-sample dither (x3_device.h), 0 = off (gfn_launch_opts.plain_rounding)
inline float gfn_plane_dither_amp(const gfn_launch_opts* opts) { return (opts && opts->plain_rounding) ? 0.f : 0x1p-22f; }

static inline hipStream_t gfn_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Zero `bytes` bytes at p on `stream` with a KERNEL (runtime.hip), never a memset node: a hipMemsetAsync captured into a hipGraph executes in
// full on the first launch of the instantiated graph only -- later replays of the node leave a quarter or a half of a 71 MB buffer, or all of a
// 4-byte one, untouched (ROCm 7.0 runtime of this image; tools/README.md, tests/test_gpu_parity.py::test_fill_zero_replays).  Every zeroing the
// library does may end up inside a caller's capture (a training step, a forward, a render tile), so none of them is a memset.
hipError_t gfn_zero_async(void* p, size_t bytes, hipStream_t stream);

// One-time set-up per (call site, device), e.g. hipFuncSetAttribute for a kernel's dynamic LDS size: true the first time it is
// reached with the current device selected (`seen`: a function-local static, one bit per device ordinal).
static inline bool gfn_first_call_on_device(unsigned long long& seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return true;
  const unsigned long long bit = 1ull << (dev & 63);
  if (seen & bit) return false;
  seen |= bit;
  return true;
}

// live element count: device value if given, else the host capacity.
__device__ __forceinline__ int64_t gfn_live_n(int64_t n_max, const int32_t* n_dev) {
  if (n_dev) {
    int64_t n = (int64_t)(*n_dev);
    return n < n_max ? n : n_max;
  }
  return n_max;
}

// Power-of-two gradient scale of the global-stage decoder backward: gmax * scale lies in [8, 16), which leaves the
// f16 halves three orders of magnitude of headroom for growth through the transposed layers and ~1e-12 of gmax
// as absolute resolution (hi + lo' * 2^-11, f16 subnormals included).
__host__ __device__ __forceinline__ float gfn_grad_scale_of(float gmax) {
  union { float f; uint32_t u; } v;
  v.f = gmax;
  int e = (int)((v.u >> 23) & 0xffu);
  e = e < 27 ? 27 : (e > 227 ? 227 : e);
  v.u = (uint32_t)(257 - e) << 23;  // 2^(130 - e)
  return v.f;
}

// f32 -> f16 -> f32 round trip, RNE (v_cvt_f16_f32 / v_cvt_f32_f16).  The empty asm pins x as
// an f32 VGPR first: without it hipcc folds `(half)(a*b)` into v_fma_mixlo_f16, which rounds the
// exact product ONCE, whereas the reference (mul.f32 then cvt.rn.f16.f32) rounds twice.
__device__ __forceinline__ float gfn_round_f16(float x) {
  asm volatile("" : "+v"(x));
  return (float)(_Float16)x;
}

// Eigen 3.4 fixed-size reduction order (redux_novec_unroller halves the range);
// see oracle/gfn_oracle.c.  Compiled with -ffp-contract=off.
__device__ __forceinline__ float gfn_sum3(float a, float b, float c) { return a + (b + c); }
__device__ __forceinline__ float gfn_sum4(float a, float b, float c, float d) { return (a + b) + (c + d); }
