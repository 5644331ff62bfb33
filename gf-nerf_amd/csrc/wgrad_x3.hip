// wgrad_x3.hip -- weight gradients of the decoder (global stage): dW[m][k] += sum_s dz[s][m] a[s][k].
//
// The contraction runs over SAMPLES, which the fused forward / backward kernels hold on the lane axis
// of their accumulator tiles (MFMA lanes index M or N, never K), so both operands come back from HBM:
// the producers write them as split-f16 "planes" (hi and lo' = (x-hi)*2^11, see field_x3.hip), tile-major
// [n / 32][features / 8][32 samples][8 features] (x3_device.h plane_off).  Slabs of samples are staged in LDS as
// [feature tile][sample][32] (64-byte rows) and read TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group a
// 4-sample x 16-feature block comes back feature-major, which is exactly the 32x32x16 operand fragment
// (lane (h, r): feature r, samples 8h .. 8h+7 of the k-step), for A (= dz^T) and B (= a) alike.
// Three MFMAs per (out tile, in tile, k-step) as in field_x3.hip.  Split-K over workgroups.
//   wgrad_x3_layers_kernel  the training path: all five layers in one launch, 32-sample slabs by LDS-DMA
//                           (global_load_lds_dwordx4, two buffers), bias sums from the MFMA fragments, per-workgroup partial
//                           vectors + wgrad_reduce_kernel instead of final atomics.  1.85 GB of planes in 0.38 ms (4.9 TB/s);
//                           the five register-staged launches with atomics it replaces took 0.47 ms
//                           An operand may come as ONE plane (lo == NULL): its values rounded to f16 once, to nearest (the producers'
//                           PL_RNE mode, x3_device.h) -- the per-term rounding is zero-mean and averages out over the 5e5 samples of the
//                           sum; the term that needs the missing half is not computed (both operands so: one MFMA per fragment pair,
//                           64-sample slabs, 0.93 GB of planes instead of 1.85)
//   wgrad_x3_kernel         planes as plain rows [n][ld] (callers that build planes themselves; tests): 64-sample slabs
//                           staged through registers, f32 atomics at the end
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
template <int FEAT>
struct Stager {
  static constexpr int PIECES = FEAT / 8;                 // 16-byte pieces per sample row
  static constexpr int TOTAL = S * PIECES;                // pieces per slab
  static constexpr int PER_THREAD = (TOTAL + 255) / 256;
  u32x4 hi[PER_THREAD], lo[PER_THREAD];

  // piece e of a slab -> (sample s, first feature f): consecutive threads walk one sample's row
  static __device__ __forceinline__ void piece(int e, int& s, int& f) { s = e / PIECES; f = (e % PIECES) * 8; }
  __device__ __forceinline__ void fetch(const Plane& p, int64_t s0, int64_t n) {
#pragma unroll
    for (int it = 0; it < PER_THREAD; it++) {
      const int e = it * 256 + (int)threadIdx.x;
      int s, f;
      piece(e, s, f);
      hi[it] = u32x4{0u, 0u, 0u, 0u}; lo[it] = u32x4{0u, 0u, 0u, 0u};
      if (e < TOTAL && s0 + s < n) {
        const int64_t o = (s0 + s) * p.ld + f;
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
template <int FD, int FA>
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
  Stager<FD> sd;
  Stager<FA> sa;
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

template <int FD, int FA>
int launch(int64_t n_max, const int32_t* n_dev, const Plane& dz, const Plane& a, const float* gmax_dev, float* dW, int ldw,
           int m_valid, int k_valid, const int* col_map, float* db, hipStream_t st) {
  constexpr int MO = (FD + 31) / 32, NI = (FA + 31) / 32;
  const size_t lds = (size_t)2 * (MO + NI) * S * 32 * sizeof(_Float16);
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen))
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<FD, FA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int64_t blocks = (n_max + S - 1) / S;
  if (blocks > 512) blocks = 512;
  wgrad_x3_kernel<FD, FA><<<(unsigned)blocks, 256, lds, st>>>(n_max, n_dev, dz, a, gmax_dev, dW, ldw, m_valid, k_valid, col_map, db);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}


// ---- tile-major planes: slabs go HBM -> LDS directly (global_load_lds_dwordx4), no register staging -------------------------
// A read-only stream through VGPRs (the kernel above) tops out near 4 TB/s on this chip (the plain-load return path; the
// same loop without its MFMAs and atomics is no faster); LDS-DMA reads reach 6.  A slab is 32 samples; the LDS image keeps
// the [feature tile][sample][32] rows tr_frag reads, so each lane fetches the 16-byte piece (sample s, octet 4t + c) that
// belongs at its lane-linear LDS position: per wave-instruction 4 runs of 256 contiguous bytes (16 samples of one octet).
// Two buffers: the loads of slab i + 1 are issued right after the barrier that retires slab i's and land under its MFMAs.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int FEAT, bool LO, int SG /* samples per slab */>
__device__ __forceinline__ void glds_image(const Plane& p, _Float16* img_hi, _Float16* img_lo, int64_t s0, int64_t n) {
  constexpr int TILES = (FEAT + 31) / 32, PIECES = TILES * SG * 4;  // 16-byte pieces of one image
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int e0 = 0; e0 < PIECES; e0 += 256) {
    const int e = e0 + (int)threadIdx.x;
    if (e - lane < PIECES) {  // wave-uniform
      const int t = e / (4 * SG), s = (e >> 2) % SG, oct = 4 * t + (e & 3);
      if (oct < FEAT / 8 && s0 + s < n) {  // the other pieces keep the zeros the kernel starts with (tail slab: zeroed below)
        const int64_t o = plane_off(s0 + s, p.ld, oct);
        __builtin_amdgcn_global_load_lds((gptr_t)(p.hi + o), (lptr_t)(img_hi + (e - lane) * 8), 16, 0, 0);
        if (LO) __builtin_amdgcn_global_load_lds((gptr_t)(p.lo + o), (lptr_t)(img_lo + (e - lane) * 8), 16, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ float frag_sum(h8 v, float acc) {  // acc + the fragment's 8 values, in f32
  using h2 = __attribute__((ext_vector_type(2))) _Float16;
  const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
  acc = __builtin_amdgcn_fdot2(h2{v[0], v[1]}, one, acc, false);
  acc = __builtin_amdgcn_fdot2(h2{v[2], v[3]}, one, acc, false);
  acc = __builtin_amdgcn_fdot2(h2{v[4], v[5]}, one, acc, false);
  return __builtin_amdgcn_fdot2(h2{v[6], v[7]}, one, acc, false);
}

struct LayerArgs {
  Plane dz, a;
  float* dW; int ldw; int m_valid, k_valid;
  const int* col_map;
  float* db;
  int shape;  // index into the supported (FD, FA) pairs, see GFN_WGRAD_SHAPES
};

// One layer's weight gradient from this workgroup's share of the slabs.  parts == nullptr: added to dW / db with f32 atomics;
// else this workgroup's partial sums are STORED at parts + (dW - base) (every valid entry, zeros included) for wgrad_reduce_kernel:
// 512 workgroups x 31 k atomics into the same 124 KB took 85 us of the 0.48 ms the five layers needed.
// DLO / ALO: the dz / a operand has its lo' plane (else: one plane, rounded to nearest).  SG: samples per slab.
template <int FD, int FA, bool DLO, bool ALO, int SG>
__device__ __forceinline__ void wgrad_phase(const LayerArgs& L, int64_t n, const float* __restrict__ gmax_dev, float* parts_w,
                                            float* parts_b, _Float16* lds) {
  constexpr int MO = (FD + 31) / 32, NI = (FA + 31) / 32;
  constexpr bool WAVE_IS_NI = (NI == 4);
  static_assert(WAVE_IS_NI || MO == 4, "one of the tile counts must be 4 (one per wave)");
  constexpr int PPW = WAVE_IS_NI ? MO : NI;
  constexpr int IMG_A = MO * SG * 32, IMG_B = NI * SG * 32;                // halfs per image
  constexpr int OFF_ALO = IMG_A, OFF_BHI = IMG_A * (DLO ? 2 : 1), OFF_BLO = OFF_BHI + IMG_B;
  constexpr int BUF = OFF_BHI + IMG_B * (ALO ? 2 : 1);                     // lds: [2][Ahi | Alo? | Bhi | Blo?]
  constexpr int ROWS = (MO * (DLO ? 2 : 1) + NI * (ALO ? 2 : 1)) * SG;     // 64-byte rows of one buffer
  __syncthreads();  // (several layers in one kernel) nobody still reads the previous layer's images
  for (int e = threadIdx.x; e < 2 * BUF / 8; e += blockDim.x) reinterpret_cast<u32x4*>(lds)[e] = u32x4{0u, 0u, 0u, 0u};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 accm[PPW], accc[PPW];
#pragma unroll
  for (int t = 0; t < PPW; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) { accm[t][r] = 0.f; accc[t][r] = 0.f; }
  float bs_hi = 0.f, bs_lo = 0.f;  // bias gradient = column sums of dz, taken from the A fragments on their way to the matrix pipe:
                                   // lane (h, r) holds feature r of samples 8h .. 8h+7 of the k-step (column reads from LDS cost more
                                   // than the whole MFMA phase: 2-byte reads, 4-way bank conflicts)
  const int64_t n_slabs = (n + SG - 1) / SG;
  __syncthreads();  // the zeros are in place before the first DMA lands on them
  if ((int64_t)blockIdx.x < n_slabs) {
    glds_image<FD, DLO, SG>(L.dz, lds, lds + OFF_ALO, (int64_t)blockIdx.x * SG, n);
    glds_image<FA, ALO, SG>(L.a, lds + OFF_BHI, lds + OFF_BLO, (int64_t)blockIdx.x * SG, n);
  }
  int cur = 0;
  for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x, cur ^= 1) {
    _Float16* Ahi = lds + cur * BUF;
    _Float16* Alo = Ahi + OFF_ALO;
    _Float16* Bhi = Ahi + OFF_BHI;
    _Float16* Blo = Ahi + OFF_BLO;
    __syncthreads();  // vmcnt(0) + barrier: this slab has landed, and every wave is done reading the other buffer
    if ((slab + 1) * SG > n) {  // the last, partial slab: rows beyond n still hold an earlier slab
      const int first = (int)(n - slab * SG);
      for (int e = threadIdx.x; e < ROWS * 4; e += blockDim.x)
        if (((e >> 2) % SG) >= first) reinterpret_cast<u32x4*>(Ahi)[e] = u32x4{0u, 0u, 0u, 0u};
      __syncthreads();
    }
    const int64_t next = slab + gridDim.x;
    if (next < n_slabs) {
      _Float16* N = lds + (cur ^ 1) * BUF;
      glds_image<FD, DLO, SG>(L.dz, N, N + OFF_ALO, next * SG, n);
      glds_image<FA, ALO, SG>(L.a, N + OFF_BHI, N + OFF_BLO, next * SG, n);
    }
#pragma unroll
    for (int ks = 0; ks < SG / 16; ks++) {
      if (WAVE_IS_NI) {
        const h8 bh = tr_frag(Bhi + wave * SG * 32, ks, lane);
        h8 bl = bh;
        if (ALO) bl = tr_frag(Blo + wave * SG * 32, ks, lane);
#pragma unroll
        for (int t = 0; t < PPW; t++) {
          const h8 ah = tr_frag(Ahi + t * SG * 32, ks, lane);
          h8 al = ah;
          if (DLO) al = tr_frag(Alo + t * SG * 32, ks, lane);
          if (t == wave) { bs_hi = frag_sum(ah, bs_hi); if (DLO) bs_lo = frag_sum(al, bs_lo); }  // MO <= 4: wave t sums out tile t
          accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accm[t], 0, 0, 0);
          if (ALO) accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc[t], 0, 0, 0);
          if (DLO) accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc[t], 0, 0, 0);
        }
      } else {
        const h8 ah = tr_frag(Ahi + wave * SG * 32, ks, lane);
        h8 al = ah;
        if (DLO) al = tr_frag(Alo + wave * SG * 32, ks, lane);
        bs_hi = frag_sum(ah, bs_hi);
        if (DLO) bs_lo = frag_sum(al, bs_lo);
#pragma unroll
        for (int t = 0; t < PPW; t++) {
          const h8 bh = tr_frag(Bhi + t * SG * 32, ks, lane);
          h8 bl = bh;
          if (ALO) bl = tr_frag(Blo + t * SG * 32, ks, lane);
          accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accm[t], 0, 0, 0);
          if (ALO) accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc[t], 0, 0, 0);
          if (DLO) accc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc[t], 0, 0, 0);
        }
      }
    }
  }
  const float inv = gmax_dev ? 1.0f / gfn_grad_scale_of(gmax_dev[0]) : 1.0f;  // the producer's power-of-two gradient scale
  if (L.db && wave < MO) {
    const float half_sum = __builtin_fmaf(bs_lo, 0x1p-11f, bs_hi);
    const float bsum = (half_sum + __shfl_xor(half_sum, 32, 64)) * inv;  // samples 8h .. 8h+7: h = 0 and 1
    const int m = 32 * wave + (lane & 31);
    if (lane < 32 && m < L.m_valid) {
      if (parts_b) parts_b[m] = bsum;
      else if (bsum != 0.f) atomicAdd(L.db + m, bsum);
    }
  }
  const int h = lane >> 5, j = lane & 31;
#pragma unroll
  for (int t = 0; t < PPW; t++) {
    const int mo = WAVE_IS_NI ? t : wave, ni = WAVE_IS_NI ? wave : t;
    if (mo >= MO || ni >= NI) continue;
    int k = 32 * ni + j;
    const bool k_ok = k < L.k_valid;
    if (L.col_map && k_ok) k = L.col_map[k];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = __builtin_fmaf(accc[t][r], 0x1p-11f, accm[t][r]) * inv;
      if (m < L.m_valid && k_ok && k >= 0) {
        if (parts_w) parts_w[(int64_t)m * L.ldw + k] = v;
        else if (v != 0.f) atomicAdd(L.dW + (int64_t)m * L.ldw + k, v);
      }
    }
  }
}

#define GFN_WGRAD_SHAPES(X) X(0, 128, 128) X(1, 8, 128) X(2, 128, 64) X(3, 16, 128) X(4, 128, 32)

constexpr int kMaxLayers = 5;
struct LayersArgs {
  LayerArgs L[kMaxLayers];
  int n_layers;
  float* parts; int64_t part_stride;  // workgroup b's partial gradient vector starts at parts + b * part_stride
  const float* base;                  // dW / db point into the flat gradient vector starting here
};

__host__ __device__ constexpr int slab_of(bool dlo, bool alo) { return (dlo || alo) ? 32 : 64; }  // one plane per operand: half the bytes per sample, twice the samples in flight

// All layers of the decoder in one launch: five kernels of 60-150 us each paid their ramp-up, first-slab latency and tail five times.
// FORMS: the operand form per layer SHAPE, two bits each -- bit 2i: the dz plane of shape i has its lo' half, bit 2i + 1: the a plane
// has.  A compile-time table (one kernel per combination in use: GFN_WGRAD_FORMS) rather than a run-time choice: with all twenty
// (shape, form) bodies in one kernel the register allocator spills.
template <uint32_t FORMS>
__global__ __launch_bounds__(256, 2) void wgrad_x3_layers_kernel(int64_t n_max, const int32_t* __restrict__ n_dev,
                                                                 const float* __restrict__ gmax_dev, LayersArgs A) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int64_t n = gfn_live_n(n_max, n_dev);
  float* mine = A.parts ? A.parts + (int64_t)blockIdx.x * A.part_stride : nullptr;
  for (int l = 0; l < A.n_layers; l++) {
    const LayerArgs& L = A.L[l];
    float* pw = mine ? mine + (L.dW - A.base) : nullptr;
    float* pb = (mine && L.db) ? mine + (L.db - A.base) : nullptr;
    switch (L.shape) {
#define GFN_CASE(ID, D, F)                                                                                                   \
      case ID: {                                                                                                             \
        constexpr bool dlo = (FORMS >> (2 * ID)) & 1u, alo = (FORMS >> (2 * ID + 1)) & 1u;                                  \
        wgrad_phase<D, F, dlo, alo, slab_of(dlo, alo)>(L, n, gmax_dev, pw, pb, lds);                                         \
      } break;
      GFN_WGRAD_SHAPES(GFN_CASE)
#undef GFN_CASE
    }
  }
}

// out[i] += sum over the workgroups' partial vectors, in ONE fixed order whatever the launch's timing (the same bits every run, like the
// cuBLAS GEMMs of gfnerf/mlp.py:35-57): a workgroup owns 64 consecutive outputs; wave w adds the parts w, w + 4, w + 8, ... of its quarter
// into four running sums, the four waves' results meet in LDS and are added as ((w0 + w1) + (w2 + w3)); one plain store per output.
constexpr int kReduceCols = 64;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ parts, int n_parts, int64_t stride, int64_t n,
                                                           float* __restrict__ out) {
  __shared__ float part_sum[4][kReduceCols];
  const int col = threadIdx.x & (kReduceCols - 1), w = threadIdx.x / kReduceCols;
  const int64_t i = (int64_t)blockIdx.x * kReduceCols + col;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int b = w;
    for (; b + 12 < n_parts; b += 16)
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] += parts[(int64_t)(b + 4 * u) * stride + i];
    for (; b < n_parts; b += 4) acc[0] += parts[(int64_t)b * stride + i];
  }
  part_sum[w][col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (w == 0 && i < n) {
    const float v = (part_sum[0][col] + part_sum[1][col]) + (part_sum[2][col] + part_sum[3][col]);
    if (v != 0.f) out[i] += v;
  }
}

constexpr int kWgradBlocks = 512;  // two workgroups per CU; also the number of partial vectors

inline size_t lds_bytes_of(int FD, int FA, bool dlo, bool alo) {
  return (size_t)2 * (((FD + 31) / 32) * (dlo ? 2 : 1) + ((FA + 31) / 32) * (alo ? 2 : 1)) * slab_of(dlo, alo) * 32 * sizeof(_Float16);
}

inline int shape_id(int fd, int fa) {
#define GFN_ID(ID, D, F) if (fd == D && fa == F) return ID;
  GFN_WGRAD_SHAPES(GFN_ID)
#undef GFN_ID
  return -1;
}

// The combinations of operand forms in use: every layer x3 (0x3ff); every operand one plane (0); every dz one plane, every a split
// (0x2aa); the engine's one-plane form, where dza1 -- shape 4 -- keeps its lo' half (0x100), and its "mixed" form (0x3aa).
#define GFN_WGRAD_FORMS(X) X(0x3ffu) X(0x000u) X(0x2aau) X(0x100u) X(0x3aau)

int launch_layers(const LayersArgs& A, int64_t n_max, const int32_t* n_dev, const float* gmax_dev, hipStream_t st) {
  size_t lds = 0;
  uint32_t forms = 0, seen = 0;
  for (int l = 0; l < A.n_layers; l++) {
    const bool dlo = A.L[l].dz.lo != nullptr, alo = A.L[l].a.lo != nullptr;
    const size_t b = lds_bytes_of(A.L[l].dz.feat, A.L[l].a.feat, dlo, alo);
    if (b > lds) lds = b;
    const uint32_t f = (dlo ? 1u : 0u) | (alo ? 2u : 0u), sh = 2 * A.L[l].shape;
    GFN_CHECK_ARG(!((seen >> sh) & 1u) || ((forms >> sh) & 3u) == f, "gfn_wgrad_x3_layers: two layers of one shape differ in which lo' planes they have");
    seen |= 1u << sh;
    forms |= f << sh;
  }
  // shapes no layer has: any form will do -- take them from the first instantiated table that agrees on the shapes present
  uint32_t present = 0;
  for (int i = 0; i < 5; i++)
    if ((seen >> (2 * i)) & 1u) present |= 3u << (2 * i);
  // the grid does not depend on the slab: workgroup b always owns partial vector b, and the caller's reduce reads the first
  // min(ceil(n_max / 32), 512) of them (a workgroup without a slab stores zeros)
  int64_t blocks = (n_max + 31) / 32;
  if (blocks > kWgradBlocks) blocks = kWgradBlocks;
  static unsigned long long attr_seen = 0;
  const bool first = gfn_first_call_on_device(attr_seen);
#define GFN_SET(F) if (first) GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_layers_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_of(128, 128, true, true)));
  GFN_WGRAD_FORMS(GFN_SET)
#undef GFN_SET
#define GFN_TRY(F)                                                                                   \
  if (((F ^ forms) & present) == 0) {                                                                \
    wgrad_x3_layers_kernel<F><<<(unsigned)blocks, 256, lds, st>>>(n_max, n_dev, gmax_dev, A);        \
    GFN_CHECK_LAUNCH();                                                                              \
    return GFN_OK;                                                                                   \
  }
  GFN_WGRAD_FORMS(GFN_TRY)
#undef GFN_TRY
  gfn_set_error("gfn_wgrad_x3_layers: no kernel for this combination of one-plane / two-plane operands (forms 0x%x; see GFN_WGRAD_FORMS)", forms);
  return GFN_ERR_ARG;
}

int one_layer(int shape, int64_t n_max, const int32_t* n_dev, const Plane& dz, const Plane& a, const float* gmax_dev, float* dW, int ldw,
              int m_valid, int k_valid, const int* col_map, float* db, hipStream_t st) {
  LayersArgs A{};
  A.L[0] = LayerArgs{dz, a, dW, ldw, m_valid, k_valid, col_map, db, shape};
  A.n_layers = 1;
  return launch_layers(A, n_max, n_dev, gmax_dev, st);
}

}  // namespace

extern "C" {

int gfn_wgrad_x3(const uint16_t* dz_hi, const uint16_t* dz_lo, int ld_dz, int feat_dz, const uint16_t* a_hi, const uint16_t* a_lo,
                 int ld_a, int feat_a, int64_t n_max, const int32_t* n_dev, const float* gmax_dev, int m_valid, int k_valid,
                 const int32_t* col_map, float* dW, int ldw, float* db, int tiled, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(dz_hi && a_hi && dW && (tiled || (dz_lo && a_lo)), "gfn_wgrad_x3: null pointer (one-plane operands, lo == NULL, need the tile-major form)");
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
    return tiled ? one_layer(shape_id(D, A), n_max, n_dev, dz, a, gmax_dev, dW, ldw, m_valid, k_valid, col_map, db, st)   \
                 : launch<D, A>(n_max, n_dev, dz, a, gmax_dev, dW, ldw, m_valid, k_valid, col_map, db, st)
  GFN_WG(128, 128);  // head layer 2
  GFN_WG(8, 128);    // head layer 3 (3 outputs)
  GFN_WG(128, 64);   // head layer 1 (63 inputs)
  GFN_WG(16, 128);   // base layer 2
  GFN_WG(128, 32);   // base layer 1
#undef GFN_WG
  gfn_set_error("gfn_wgrad_x3: unsupported plane widths %d x %d (supported: 128x128, 8x128, 128x64, 16x128, 128x32)", feat_dz, feat_a);
  return GFN_ERR_ARG;
}

int gfn_wgrad_x3_parts(void) { return kWgradBlocks; }

int gfn_wgrad_x3_layers(const gfn_wgrad_layer* layers, int n_layers, int64_t n_max, const int32_t* n_dev, const float* gmax_dev,
                        const float* grad_base, int64_t grad_len, float* parts, int64_t part_stride, void* stream) {
  if (n_max == 0 || n_layers == 0) return GFN_OK;
  GFN_CHECK_ARG(layers && n_layers > 0 && n_layers <= kMaxLayers, "gfn_wgrad_x3_layers: 1..%d layers", kMaxLayers);
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31), "gfn_wgrad_x3_layers: n_max=%lld", (long long)n_max);
  GFN_CHECK_ARG(!parts || (grad_base && grad_len > 0 && part_stride >= grad_len), "gfn_wgrad_x3_layers: partial vectors need the flat gradient's base, length and a stride >= it");
  LayersArgs A{};
  A.n_layers = n_layers; A.parts = parts; A.part_stride = part_stride; A.base = grad_base;
  for (int l = 0; l < n_layers; l++) {
    const gfn_wgrad_layer& g = layers[l];
    GFN_CHECK_ARG(g.dz_hi && g.a_hi && g.dW, "gfn_wgrad_x3_layers: null pointer in layer %d", l);
    GFN_CHECK_ARG(((reinterpret_cast<uintptr_t>(g.dz_hi) | reinterpret_cast<uintptr_t>(g.dz_lo) | reinterpret_cast<uintptr_t>(g.a_hi) |
                    reinterpret_cast<uintptr_t>(g.a_lo)) & 15) == 0, "gfn_wgrad_x3_layers: planes must be 16-byte aligned");
    const int shape = shape_id(g.feat_dz, g.feat_a);
    GFN_CHECK_ARG(shape >= 0, "gfn_wgrad_x3_layers: unsupported plane widths %d x %d in layer %d", g.feat_dz, g.feat_a, l);
    GFN_CHECK_ARG(g.m_valid >= 1 && g.m_valid <= g.feat_dz && g.k_valid >= 1 && g.k_valid <= g.feat_a && g.ldw >= 1, "gfn_wgrad_x3_layers: bad m/k/ldw in layer %d", l);
    if (parts) {  // the partial vectors mirror the flat gradient: every address the layer writes must lie inside it
      const int64_t w0 = g.dW - grad_base, w1 = w0 + (int64_t)(g.m_valid - 1) * g.ldw + g.ldw;
      GFN_CHECK_ARG(w0 >= 0 && w1 <= grad_len, "gfn_wgrad_x3_layers: dW of layer %d lies outside the flat gradient", l);
      GFN_CHECK_ARG(!g.db || (g.db - grad_base >= 0 && g.db - grad_base + g.m_valid <= grad_len), "gfn_wgrad_x3_layers: db of layer %d lies outside the flat gradient", l);
    }
    A.L[l] = LayerArgs{Plane{g.dz_hi, g.dz_lo, g.feat_dz, g.feat_dz}, Plane{g.a_hi, g.a_lo, g.feat_a, g.feat_a}, g.dW, g.ldw, g.m_valid,
                       g.k_valid, g.col_map, g.db, shape};
  }
  return launch_layers(A, n_max, n_dev, gmax_dev, gfn_stream(stream));
}

int gfn_wgrad_reduce(const float* parts, int n_parts, int64_t part_stride, int64_t n, float* out, void* stream) {
  if (n <= 0 || n_parts <= 0) return GFN_OK;
  GFN_CHECK_ARG(parts && out && part_stride >= n, "gfn_wgrad_reduce: bad arguments");
  wgrad_reduce_kernel<<<(unsigned)((n + kReduceCols - 1) / kReduceCols), 256, 0, gfn_stream(stream)>>>(parts, n_parts, part_stride, n, out);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
