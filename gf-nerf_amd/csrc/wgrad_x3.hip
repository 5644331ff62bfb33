// wgrad_x3.hip -- weight gradients of the decoder (global stage): dW[m][k] += sum_s dz[s][m] a[s][k].
//
// The contraction runs over SAMPLES, which the fused forward / backward kernels hold on the lane axis
// of their accumulator tiles (MFMA lanes index M or N, never K), so both operands come back from HBM:
// the producers write them as split-f16 "planes" (hi and lo' = (x-hi)*2^11, see field_x3.hip) -- tile-major
// [n / 32][features / 8][32 samples][8 features] (x3_device.h plane_off; `tiled`), or plain rows [n][ld] for callers that
// build planes themselves --, this kernel stages 64-sample slabs in LDS as [feature tile][sample][32] (64-byte rows)
// and reads them TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group a 4-sample x 16-feature block
// comes back feature-major, which is exactly the 32x32x16 operand fragment
// (lane (h, r): feature r, samples 8h .. 8h+7 of the k-step), for A (= dz^T) and B (= a) alike.
// Three MFMAs per (out tile, in tile, k-step) as in field_x3.hip.  Split-K over workgroups, f32 atomics
// at the end (coalesced rows; 31 k weights x <= 256 workgroups).
#include "gfn_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using s4 = __attribute__((ext_vector_type(4))) short;
using s8 = __attribute__((ext_vector_type(8))) short;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int S = 64;  // samples per LDS slab

struct Plane {
  const uint16_t* hi;
  const uint16_t* lo;
  int ld;    // rows: row stride in elements (multiple of 8); tiled: the plane's width W
  int feat;  // stored features per row (multiple of 8, <= 32 * tiles)
};

__device__ __forceinline__ int64_t plane_off(int64_t q, int W, int octet) { return (((q >> 5) * (W >> 3) + octet) * 32 + (q & 31)) * 8; }

__device__ __forceinline__ h8 tr_frag(const _Float16* tile /*[S][32]*/, int ks, int lane) {
  // lane 4q+p of 16-lane group g supplies &T[row0+q][col0+4p]; lane i of the group receives column col0+i of rows row0..row0+3
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const _Float16* a = tile + (16 * ks + 8 * (g >> 1) + q) * 32 + 16 * (g & 1) + 4 * p;
  const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)a);
  const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a + 4 * 32));
  const s8 r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_bit_cast(h8, r);
}

// One slab = 64 samples x (FD + FA) features x {hi, lo'}.  Each thread owns a fixed set of 16-byte pieces; the next
// slab's pieces are fetched into registers while the matrix pipe works on the current one.
template <int FEAT, bool TILED>
struct Stager {
  static constexpr int PIECES = FEAT / 8;                 // 16-byte pieces per sample row
  static constexpr int TOTAL = S * PIECES;                // pieces per slab
  static constexpr int PER_THREAD = (TOTAL + 255) / 256;
  u32x4 hi[PER_THREAD], lo[PER_THREAD];

  // piece e of a slab -> (sample s, first feature f).  Rows: consecutive threads walk one sample's row; tiled: consecutive
  // threads walk the 32 samples of one octet (512 contiguous bytes), the order the producers wrote them in.
  static __device__ __forceinline__ void piece(int e, int& s, int& f) {
    if (TILED) { s = (e & 31) + 32 * ((e >> 5) / PIECES); f = ((e >> 5) % PIECES) * 8; }
    else { s = e / PIECES; f = (e % PIECES) * 8; }
  }
  __device__ __forceinline__ void fetch(const Plane& p, int64_t s0, int64_t n) {
#pragma unroll
    for (int it = 0; it < PER_THREAD; it++) {
      const int e = it * 256 + (int)threadIdx.x;
      int s, f;
      piece(e, s, f);
      hi[it] = u32x4{0u, 0u, 0u, 0u}; lo[it] = u32x4{0u, 0u, 0u, 0u};
      if (e < TOTAL && s0 + s < n) {
        const int64_t o = TILED ? plane_off(s0 + s, p.ld, f >> 3) : (s0 + s) * p.ld + f;
        hi[it] = *reinterpret_cast<const u32x4*>(p.hi + o);
        lo[it] = *reinterpret_cast<const u32x4*>(p.lo + o);
      }
    }
  }
  __device__ __forceinline__ void commit(_Float16* dst_hi, _Float16* dst_lo) const {  // -> [tile][S][32]
#pragma unroll
    for (int it = 0; it < PER_THREAD; it++) {
      const int e = it * 256 + (int)threadIdx.x;
      int s, f;
      piece(e, s, f);
      if (e < TOTAL) {
        *reinterpret_cast<u32x4*>(dst_hi + ((f >> 5) * S + s) * 32 + (f & 31)) = hi[it];
        *reinterpret_cast<u32x4*>(dst_lo + ((f >> 5) * S + s) * 32 + (f & 31)) = lo[it];
      }
    }
  }
};

// WAVE_IS_NI: wave w owns in-tile w and loops over the MO out tiles (NI == 4); else wave w owns out tile w (MO == 4)
// and loops over the NI in tiles.
template <int FD, int FA, bool TILED>
__global__ __launch_bounds__(256) void wgrad_x3_kernel(int64_t n_max, const int32_t* __restrict__ n_dev, Plane dz, Plane a,
                                                       const float* __restrict__ gmax_dev, float* __restrict__ dW, int ldw,
                                                       int m_valid, int k_valid, const int* __restrict__ col_map,
                                                       float* __restrict__ db) {
  constexpr int MO = (FD + 31) / 32, NI = (FA + 31) / 32;
  constexpr bool WAVE_IS_NI = (NI == 4);
  static_assert(WAVE_IS_NI || MO == 4, "one of the tile counts must be 4 (one per wave)");
  constexpr int PPW = WAVE_IS_NI ? MO : NI;  // pairs per wave
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* Ahi = lds;                     // [MO][S][32]
  _Float16* Alo = Ahi + MO * S * 32;
  _Float16* Bhi = Alo + MO * S * 32;       // [NI][S][32]
  _Float16* Blo = Bhi + NI * S * 32;
  // features beyond the stored ones stay zero for the whole kernel
  for (int e = threadIdx.x; e < 2 * (MO + NI) * S * 32 / 8; e += blockDim.x) reinterpret_cast<u32x4*>(lds)[e] = u32x4{0u, 0u, 0u, 0u};
  const int64_t n = gfn_live_n(n_max, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 accm[PPW], accc[PPW];
#pragma unroll
  for (int t = 0; t < PPW; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) { accm[t][r] = 0.f; accc[t][r] = 0.f; }
  float bsum = 0.f;
  const int64_t n_slabs = (n + S - 1) / S;
  Stager<FD, TILED> sd;
  Stager<FA, TILED> sa;
  if ((int64_t)blockIdx.x < n_slabs) { sd.fetch(dz, (int64_t)blockIdx.x * S, n); sa.fetch(a, (int64_t)blockIdx.x * S, n); }
  __syncthreads();
  for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
    sd.commit(Ahi, Alo);
    sa.commit(Bhi, Blo);
    __syncthreads();
    const int64_t next = slab + gridDim.x;
    if (next < n_slabs) { sd.fetch(dz, next * S, n); sa.fetch(a, next * S, n); }  // in flight under the MFMAs below
    if (db) {  // bias gradient: column sums of dz (threads 0..127: hi plane, 128..255: lo' plane)
      const int m = threadIdx.x & 127, part = threadIdx.x >> 7;
      if (m < 32 * MO) {
        const _Float16* col = (part ? Alo : Ahi) + (m >> 5) * S * 32 + (m & 31);
        float acc = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; s++) acc += (float)col[s * 32];
        bsum += part ? acc * 0x1p-11f : acc;
      }
    }
#pragma unroll
    for (int ks = 0; ks < S / 16; ks++) {
      if (WAVE_IS_NI) {
        const h8 bh = tr_frag(Bhi + wave * S * 32, ks, lane), bl = tr_frag(Blo + wave * S * 32, ks, lane);
#pragma unroll
        for (int t = 0; t < PPW; t++) {
          const h8 ah = tr_frag(Ahi + t * S * 32, ks, lane), al = tr_frag(Alo + t * S * 32, ks, lane);
          accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accm[t], 0, 0, 0);
          accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc[t], 0, 0, 0);
          accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc[t], 0, 0, 0);
        }
      } else {
        const h8 ah = tr_frag(Ahi + wave * S * 32, ks, lane), al = tr_frag(Alo + wave * S * 32, ks, lane);
#pragma unroll
        for (int t = 0; t < PPW; t++) {
          const h8 bh = tr_frag(Bhi + t * S * 32, ks, lane), bl = tr_frag(Blo + t * S * 32, ks, lane);
          accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accm[t], 0, 0, 0);
          accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc[t], 0, 0, 0);
          accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  const float inv = gmax_dev ? 1.0f / gfn_grad_scale_of(gmax_dev[0]) : 1.0f;  // the producer's power-of-two gradient scale
  if (db) {
    const int m = threadIdx.x & 127;
    if (m < m_valid && m < 32 * MO && bsum != 0.f) atomicAdd(db + m, bsum * inv);
  }
  const int h = lane >> 5, j = lane & 31;
#pragma unroll
  for (int t = 0; t < PPW; t++) {
    const int mo = WAVE_IS_NI ? t : wave, ni = WAVE_IS_NI ? wave : t;
    if (mo >= MO || ni >= NI) continue;
    int k = 32 * ni + j;
    const bool k_ok = k < k_valid;
    if (col_map && k_ok) k = col_map[k];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = __builtin_fmaf(accc[t][r], 0x1p-11f, accm[t][r]) * inv;
      if (m < m_valid && k_ok && k >= 0 && v != 0.f) atomicAdd(dW + (int64_t)m * ldw + k, v);
    }
  }
}

template <int FD, int FA, bool TILED>
int launch(int64_t n_max, const int32_t* n_dev, const Plane& dz, const Plane& a, const float* gmax_dev, float* dW, int ldw,
           int m_valid, int k_valid, const int* col_map, float* db, hipStream_t st) {
  constexpr int MO = (FD + 31) / 32, NI = (FA + 31) / 32;
  const size_t lds = (size_t)2 * (MO + NI) * S * 32 * sizeof(_Float16);
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen))
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<FD, FA, TILED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int64_t blocks = (n_max + S - 1) / S;
  if (blocks > 512) blocks = 512;
  wgrad_x3_kernel<FD, FA, TILED><<<(unsigned)blocks, 256, lds, st>>>(n_max, n_dev, dz, a, gmax_dev, dW, ldw, m_valid, k_valid, col_map, db);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // namespace

extern "C" {

int gfn_wgrad_x3(const uint16_t* dz_hi, const uint16_t* dz_lo, int ld_dz, int feat_dz, const uint16_t* a_hi, const uint16_t* a_lo,
                 int ld_a, int feat_a, int64_t n_max, const int32_t* n_dev, const float* gmax_dev, int m_valid, int k_valid,
                 const int32_t* col_map, float* dW, int ldw, float* db, int tiled, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(dz_hi && dz_lo && a_hi && a_lo && dW, "gfn_wgrad_x3: null pointer");
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31), "gfn_wgrad_x3: n_max=%lld", (long long)n_max);
  GFN_CHECK_ARG(ld_dz % 8 == 0 && ld_a % 8 == 0 && feat_dz % 8 == 0 && feat_a % 8 == 0 && feat_dz >= 8 && feat_a >= 8 &&
                    feat_dz <= ld_dz && feat_a <= ld_a && feat_dz <= 128 && feat_a <= 128,
                "gfn_wgrad_x3: plane strides / widths must be multiples of 8, widths in [8,128]");
  GFN_CHECK_ARG(m_valid >= 1 && m_valid <= feat_dz && k_valid >= 1 && k_valid <= feat_a && ldw >= 1, "gfn_wgrad_x3: bad m/k/ldw");
  GFN_CHECK_ARG(((reinterpret_cast<uintptr_t>(dz_hi) | reinterpret_cast<uintptr_t>(dz_lo) | reinterpret_cast<uintptr_t>(a_hi) |
                  reinterpret_cast<uintptr_t>(a_lo)) & 15) == 0, "gfn_wgrad_x3: planes must be 16-byte aligned");
  GFN_CHECK_ARG(!tiled || (ld_dz == feat_dz && ld_a == feat_a), "gfn_wgrad_x3: a tile-major plane is addressed whole (ld == width)");
  const Plane dz{dz_hi, dz_lo, ld_dz, feat_dz}, a{a_hi, a_lo, ld_a, feat_a};
  hipStream_t st = gfn_stream(stream);
#define GFN_WG(D, A)                                                                                                       \
  if (feat_dz == D && feat_a == A)                                                                                         \
    return tiled ? launch<D, A, true>(n_max, n_dev, dz, a, gmax_dev, dW, ldw, m_valid, k_valid, col_map, db, st)          \
                 : launch<D, A, false>(n_max, n_dev, dz, a, gmax_dev, dW, ldw, m_valid, k_valid, col_map, db, st)
  GFN_WG(128, 128);  // head layer 2
  GFN_WG(8, 128);    // head layer 3 (3 outputs)
  GFN_WG(128, 64);   // head layer 1 (63 inputs)
  GFN_WG(16, 128);   // base layer 2
  GFN_WG(128, 32);   // base layer 1
#undef GFN_WG
  gfn_set_error("gfn_wgrad_x3: unsupported plane widths %d x %d (supported: 128x128, 8x128, 128x64, 16x128, 128x32)", feat_dz, feat_a);
  return GFN_ERR_ARG;
}

}  // extern "C"
