// field_x3.hip -- the fused decoder of field.hip on the f16 matrix pipe at f32 accuracy.
//
// v_mfma_f32_32x32x2_f32 runs at the vector rate (64 FLOP/clk/SIMD); v_mfma_f32_32x32x16_f16 is 16x
// that.  Every f32 operand x is split into two f16 numbers, x ~= hi + lo' * 2^-11 with
// hi = rtz_f16(x), lo' = rtz_f16((x - hi) * 2^11)  (the 2^11 keeps lo' out of the f16 subnormals),
// and a product sum becomes three MFMAs into two f32 accumulators,
//     main += A_hi B_hi            corr += A_hi B_lo' + A_lo' B_hi        result = main + corr * 2^-11,
// dropping only the lo*lo term (2^-22 relative): ~1e-6 of sum|a b|, 5.3x fewer matrix-pipe
// cycles than the exact f32 form.  The reference runs these layers in f32 (nn.Linear,
// gfnerf/mlp.py:35-57); parity bar 1e-3 abs (BASELINE.json), tests hold 1e-5.
//
// Structure is field.hip's: a wave owns 32 samples (MFMA column = lane&31), the weights are the A
// operand, and every layer's 32x32 accumulator tile IS the next layer's B operand -- registers
// 8s..8s+7 of a tile, split and packed, are the 8-element fragment of k-step s, whose k order is
// row = 16s + 8(i>>2) + 4h + (i&3) (h = lane>>5); the weight images are laid out in that order, one
// contiguous 1 KiB chunk (64 lanes x 16 B, a conflict-free ds_read_b128) per (k-step, out tile).
// The images are built once per weight update by field_pack_x3_kernel and copied into LDS by
// every workgroup (146 KiB forward, 112 KiB backward; 512 threads = two waves per SIMD so one
// wave's split/ReLU VALU work overlaps the other's MFMAs).
#include "gfn_common.h"
#include "hash_device.h"
#include "x3_device.h"

namespace {


struct FieldWeights {
  const float* bW0; const float* bb0; const float* bW1; const float* bb1; const float* hW0; const float* hb0;
  const float* hW1; const float* hb1; const float* hW2; const float* hb2;
};

__device__ float fwd_weight(const FieldWeights& w, int c, int h, int r, int i) {
  if (c < C_B1) { const int u = c >> 2, ot = c & 3; return w.bW0[(32 * ot + r) * 32 + 16 * u + 8 * h + i]; }
  if (c < C_H0) { const int u = c - C_B1; return r < 16 ? w.bW1[r * 128 + acc_row(u >> 1, u & 1, h, i)] : 0.f; }
  if (c < C_H1) {
    const int cc = c - C_H0, u = cc >> 2, o = 32 * (cc & 3) + r;
    if (u == 0) { const int row = acc_row(0, 0, h, i); return row == 0 ? 0.f : w.hW0[o * 63 + 15 + row]; }  // row 0 = density channel
    if (u == 1) return w.hW0[o * 63 + 8 * h + i];
    return w.hW0[o * 63 + 31 + 16 * (u - 2) + 8 * h + i];
  }
  if (c < C_H2) { const int cc = c - C_H1, u = cc >> 2, ot = cc & 3; return w.hW1[(32 * ot + r) * 128 + acc_row(u >> 1, u & 1, h, i)]; }
  const int u = c - C_H2;
  return r < 3 ? w.hW2[r * 128 + acc_row(u >> 1, u & 1, h, i)] : 0.f;
}

__device__ float bwd_weight(const FieldWeights& w, int c, int h, int r, int i) {
  if (c < D_G2) { const int it = c; return (h == 0 && i < 3) ? w.hW2[i * 128 + 32 * it + r] : 0.f; }
  if (c < D_G1) { const int cc = c - D_G2, u = cc >> 2, it = cc & 3; return w.hW1[acc_row(u >> 1, u & 1, h, i) * 128 + 32 * it + r]; }
  if (c < D_GB1) { const int u = c - D_G1; return (r >= 1 && r <= 15) ? w.hW0[acc_row(u >> 1, u & 1, h, i) * 63 + 15 + r] : 0.f; }
  if (c < D_GB0) { const int it = c - D_GB1; return w.bW1[acc_row(0, 0, h, i) * 128 + 32 * it + r]; }
  if (c < D_GE) { const int u = c - D_GB0; return w.bW0[acc_row(u >> 1, u & 1, h, i) * 32 + r]; }
  const int u = c - D_GE;
  return w.hW0[acc_row(u >> 1, u & 1, h, i) * 63 + 31 + r];
}

// one thread per (chunk, lane): 8 weights -> 16 B of the hi image and 16 B of the lo image
__global__ void field_pack_x3_kernel(FieldWeights w, u32x4* __restrict__ fwd, u32x4* __restrict__ bwd) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 6, lane = gid & 63, h = lane >> 5, r = lane & 31;
  if (c < FWD_CHUNKS) {
    if (!fwd) return;
    u32x4 hi, lo;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      uint32_t a, b;
      split2(fwd_weight(w, c, h, r, 2 * p), fwd_weight(w, c, h, r, 2 * p + 1), a, b);
      hi[p] = a; lo[p] = b;
    }
    fwd[gid] = hi;
    fwd[FWD_CHUNKS * 64 + gid] = lo;
    if (c == 0) {  // biases behind the two images: bb0[128] bb1[32] hb0[128] hb1[128] hb2[32]
      float* bp = reinterpret_cast<float*>(fwd + 2 * FWD_CHUNKS * 64);
      for (int e = lane; e < FWD_BIAS; e += 64) {
        float v;
        if (e < 128) v = w.bb0[e];
        else if (e < 160) v = e - 128 < 16 ? w.bb1[e - 128] : 0.f;
        else if (e < 288) v = w.hb0[e - 160];
        else if (e < 416) v = w.hb1[e - 288];
        else v = e - 416 < 3 ? w.hb2[e - 416] : 0.f;
        bp[e] = v;
      }
    }
  } else if (c < FWD_CHUNKS + BWD_CHUNKS) {
    if (!bwd) return;
    const int cb = c - FWD_CHUNKS;
    u32x4 hi, lo;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      uint32_t a, b;
      split2(bwd_weight(w, cb, h, r, 2 * p), bwd_weight(w, cb, h, r, 2 * p + 1), a, b);
      hi[p] = a; lo[p] = b;
    }
    bwd[cb * 64 + lane] = hi;
    bwd[BWD_CHUNKS * 64 + cb * 64 + lane] = lo;
  }
}

template <bool SAVE, int PLANES /* PL_* (x3_device.h) */, bool PLANAR>
__global__ __launch_bounds__(512) void field_fwd_x3_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ feat, const int* __restrict__ ray_id,
    const float* __restrict__ sh, const float* __restrict__ emb, const int* __restrict__ cam_idx,
    const u32x4* __restrict__ image, float* __restrict__ sigma, float* __restrict__ rgb, float* __restrict__ h0_out,
    unsigned long long* __restrict__ masks /* [n][3][2] */, uint16_t* __restrict__ planes_hi, uint16_t* __restrict__ planes_lo,
    int64_t cap, int64_t feat_stride /* 0: feat is [n,32]; else level-major float2 [16][feat_stride] (gfn_hash_fwd_lm) */,
    const int32_t* __restrict__ list /* phased forward: decode only the samples list[0 .. *list_n) (ascending); the others' outputs and */,
    const int32_t* __restrict__ list_n /* plane rows are not touched; the launch does nothing when *list_n == 0 */,
    float dither /* PL_RNE: amplitude of the per-sample dither (plane_dither, x3_device.h) */) {
  if (list && *list_n == 0) return;  // grid-uniform
  if (list && *list_n < 0) list = nullptr;  // "every sample": no list was written
  extern __shared__ __attribute__((aligned(16))) u32x4 L[];
  for (int e = threadIdx.x; e < GFN_FIELD_X3_FWD_BYTES / 16; e += blockDim.x) L[e] = image[e];
  __syncthreads();
  const u32x4* Ihi = L;
  const u32x4* Ilo = L + FWD_CHUNKS * 64;
  const float* bias = reinterpret_cast<const float*>(L + 2 * FWD_CHUNKS * 64);
  // n = samples to decode: the live ones, or the entries of the list (tiles are then 32 consecutive ENTRIES: every tile full)
  int64_t n = list ? (int64_t)*list_n : gfn_live_n(n_max, n_dev);
  if (list && n > n_max) n = n_max;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t n_tiles = (n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 8) {
    const int64_t e = tile * 32 + j;
    const bool valid = e < n;
    const int64_t s = list ? (int64_t)list[valid ? e : n - 1] : e;  // the lane's sample
    const int64_t q = list ? s : (valid ? s : n - 1);
    const float pc = PLANES == PL_RNE ? 1.f + plane_dither(q, dither) : 1.f;
    // ---- k-step u of the feature row: columns 16u + 8h + i (invalid lanes rewrite the last live row with the same values)
    u32x4 xh[2], xl[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      float x[8];
      if (PLANAR) {  // columns 16u + 8h + i = levels 8u + 4h .. + 3 of sample q
        const float2* fq = reinterpret_cast<const float2*>(feat) + (int64_t)(4 * h) * feat_stride + q;  // per lane; the rest is uniform
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const float2 v = fq[(int64_t)(8 * u + m) * feat_stride];
          x[2 * m] = v.x; x[2 * m + 1] = v.y;
        }
      } else {
        load8(feat + q * 32 + 16 * u + 8 * h, x);
      }
      split8(x, xh[u], xl[u]);
      if (PLANES == PL_SPLIT) store_nat(planes_hi + cap * FP_FEAT, planes_lo + cap * FP_FEAT, 32, q, 16 * u, xh[u], xl[u], h);
      if (PLANES == PL_RNE) { u32x4 r; rne8(x, r); store_nat_rne(planes_hi + cap * FP_FEAT, 32, q, 16 * u, r, h); }  // (f16 numbers already: exact, no dither)
    }
    const int ray = ray_id[q];
    const DecoderOut o = decoder_tile<PLANES>(Ihi, Ilo, bias, lane, xh, xl, sh + (int64_t)ray * 16,
                                              emb ? emb + (int64_t)cam_idx[ray] * 32 : nullptr, planes_hi, planes_lo, cap, q, pc);
    if (valid) {
      if (h == 0) {
        sigma[s] = o.sigma;
        if (SAVE) h0_out[s] = o.h0;
        rgb[3 * s] = o.r; rgb[3 * s + 1] = o.g; rgb[3 * s + 2] = o.b;
      }
      if (SAVE) {
        unsigned long long* mp = masks + s * 6 + h;
        mp[0] = o.m_a1; mp[2] = o.m_c1; mp[4] = o.m_c2;
      }
    }
  }
}

// PLANES = 0: the focal stage (decoder frozen): d feat only, per-sample gradient scale.
// PLANES = 1: the global stage: also writes dz of every layer as split-f16 planes (for gfn_wgrad_x3) under ONE
//             power-of-two scale for the whole batch (gfn_grad_scale_of(*gmax)), and reduces the appearance-embedding
//             gradient over each ray's samples into demb.
// PLANES = 2: the same with ONE f16 plane per layer, dz rounded to nearest under the sample's dither (PL_RNE, x3_device.h), except
//             dza1, which keeps both halves (lo' at column BP_DZA1_LO of planes_hi's buffer); planes_lo is not touched.
// SCATTER: instead of writing d feat, scatter it straight into the hash table's gradient pool
// (Hash3DAnchoredFunction::backward, Hash3DAnchored_cuda.cu:81-155,198-239): the accumulator tile of d feat holds, on
// lane (h, j), sample j's gradient for the 8 levels {2h, 2h+1, 4+2h, 5+2h, 8+2h, 9+2h, 12+2h, 13+2h} -- each lane hashes
// those levels of its sample and issues the (run-length reduced) packed-f16 / f32 atomics itself.  The [N,32] d feat
// tensor is never written and the decoder's MFMAs run beside the other waves' atomics.
struct ScatterArgs {
  const float* pts; const int* anchors; const int* prim_pool; const float* bias_pool; int64_t n_volumes; int log2_T;
  void* grad_pool; int grad_f16; LevelScales sc;
  // record mode (f16 pool only): instead of issuing its atomics -- the L2 executes ~21 G scattered atomics/s whatever their
  // scope or footprint (tools/gather_bench.hip) -- every add becomes a (slot, packed f16x2) record appended to the stream of
  // (16 Ki-slot range, this workgroup): rec[(range * 256 + workgroup) * rec_cap + position], positions from LDS cursors;
  // csrc/grad_part.hip then accumulates every range in LDS (gfn_grad_records_reduce)
  uint2* rec; unsigned int* rec_count; unsigned int rec_cap; int n_ranges;
  // gate (gfn_launch_opts.record_merge = GFN_MERGE_DEVICE): the launch is one of a PAIR -- the plain record form and the run-merging one -- and only the kernel whose
  // `gate_want` equals [*gate_front >= gate_frac * *gate_n] does anything: which form a batch gets is decided on the device, from this step's own
  // counts (samples in front of the termination points, samples), so a replayed step and an eagerly launched one decide alike
  const int* gate_front; const int* gate_n; float gate_frac; int gate_want;
};

// sum v over the lanes [j, run_end] of this 32-lane half (runs = consecutive samples in one cell of one volume)
__device__ __forceinline__ float half_run_sum32(float v, int j, int run_end) {
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const float u = __shfl_down(v, off, 32);
    if (j + off <= run_end) v += u;
  }
  return v;
}

// MERGE: run-length reduction over consecutive samples in one cell.  It divides the number of adds (12.6 M instead of 17.3 M
// in a focal step), which is what the atomic forms pay for; the record form merges for free in the accumulate kernel, so
// it only keeps the reduction where the records themselves are the cost (global stage: 48 M).
template <bool GRAD_F16, bool RECORD = false, bool MERGE = true>
__device__ __forceinline__ void scatter_level(const ScatterArgs& S, const float* __restrict__ s_scale, int level, float px, float py, float pz,
                                              int64_t vol, float gx, float gy, bool valid, int h, int j, uint32_t T,
                                              const int (&prim)[3], unsigned int* s_cursor = nullptr) {
  const int64_t tr = (int64_t)level * S.n_volumes + vol;
  Corners c;
  uint32_t cell[3];
  hash_corners(px, py, pz, s_scale[level], prim, S.bias_pool ? S.bias_pool + 3 * tr : nullptr, T - 1u, c, cell);
  float g0 = gfn_round_f16(gx * 128.f);  // :219 grad_in = (grad * 128).half()
  float g1 = gfn_round_f16(gy * 128.f);
  const bool live = valid && (g0 != 0.f || g1 != 0.f);  // :149
  if (!live) { g0 = 0.f; g1 = 0.f; }
  const uint32_t pvol = __shfl_up((uint32_t)vol, 1, 32), pcx = __shfl_up(cell[0], 1, 32);
  const uint32_t pcy = __shfl_up(cell[1], 1, 32), pcz = __shfl_up(cell[2], 1, 32);
  const bool head = (j == 0) || pvol != (uint32_t)vol || pcx != cell[0] || pcy != cell[1] || pcz != cell[2];
  const unsigned long long heads64 = __ballot(head);
  float c0[8], c1[8];
#pragma unroll
  for (int d = 0; d < 8; d++) {  // :150 (half)(w0*ws[d]) -- each contribution quantised like the reference
    c0[d] = gfn_round_f16(g0 * c.w[d]);
    c1[d] = gfn_round_f16(g1 * c.w[d]);
  }
  bool issue = live;
  if (MERGE && __popcll(heads64) <= 40) {  // wave-uniform: runs are long enough to pay for the shuffles
    const uint32_t heads = (uint32_t)(heads64 >> (32 * h));
    const uint32_t later = (j == 31) ? 0u : (heads >> (j + 1));
    const int run_end = later ? j + __builtin_ctz(later) : 31;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      c0[d] = half_run_sum32(c0[d], j, run_end);
      c1[d] = half_run_sum32(c1[d], j, run_end);
    }
    issue = head;  // zero sums are skipped below
  }
  if (GRAD_F16 && RECORD) {  // compile-time: the atomic form below is not even compiled into the record kernels
    half2_t* baseh = reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(S.grad_pool) + (int64_t)level * T);
    const uint32_t slot0 = (uint32_t)level * (T >> 1);
    if (issue) {
#pragma unroll
      for (int d = 0; d < 8; d++) {
        if (c0[d] != 0.f || c1[d] != 0.f) {
          const half2_t v = {(_Float16)c0[d], (_Float16)c1[d]};
          const uint32_t slot = slot0 + c.pos[d];
          const uint32_t range = slot >> GFN_GRAD_RANGE_LOG2;
          const unsigned int pos = atomicAdd(s_cursor + range, 1u);  // LDS: this workgroup's cursor of the range
          if (pos < S.rec_cap) S.rec[(range * 256u + blockIdx.x) * S.rec_cap + pos] = make_uint2(slot, __builtin_bit_cast(uint32_t, v));  // < 2^32 records (host check)
          else __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*)(baseh + c.pos[d]), v);  // stream full
        }
      }
    }
    return;
  }
  if (issue) {
    if (GRAD_F16) {
      half2_t* baseh = reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(S.grad_pool) + (int64_t)level * T);
#pragma unroll
      for (int d = 0; d < 8; d++) {
        if (c0[d] != 0.f || c1[d] != 0.f) {
          const half2_t v = {(_Float16)c0[d], (_Float16)c1[d]};  // x128-scaled, like the reference's pool (:222,:151)
          __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*)(baseh + c.pos[d]), v);
        }
      }
    } else {
      float* basef = reinterpret_cast<float*>(S.grad_pool) + (int64_t)level * T;
#pragma unroll
      for (int d = 0; d < 8; d++) {
        if (c0[d] != 0.f || c1[d] != 0.f) {
          float* dstp = basef + (size_t)c.pos[d] * 2;
          atomicAdd(dstp, c0[d] * (1.f / 128.f));  // the reference's final /128 (:238), exact (power of two)
          atomicAdd(dstp + 1, c1[d] * (1.f / 128.f));
        }
      }
    }
  }
}

// Every form runs 512-thread workgroups = two waves per SIMD, so that one wave's hashing / record stores run beside the other's
// MFMAs (the PLANES + SCATTER form used to run 256: 0.68 ms where the MFMA and the store phases add up; it fits 255 registers
// when asked to) -- no build of these kernels may spill (tests/test_abi_and_host.py)
template <int PLANES, int SCATTER>
constexpr int bwd_threads() { return 512; }

template <int PLANES, int SCATTER /* 0 none, 1 f16 pool, 2 f32 pool, 3 f16 pool through records, 4 the same with runs of samples in one cell merged first */>
__global__ __launch_bounds__((bwd_threads<PLANES, SCATTER>())) void field_bwd_dx_x3_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ d_sigma, const float* __restrict__ d_rgb,
    const float* __restrict__ rgb, const float* __restrict__ h0, const unsigned long long* __restrict__ masks,
    const u32x4* __restrict__ image, float* __restrict__ dfeat, const float* __restrict__ gmax_dev,
    uint16_t* __restrict__ planes_hi, uint16_t* __restrict__ planes_lo, int64_t cap, const int* __restrict__ ray_id,
    const int* __restrict__ cam_idx, float* __restrict__ demb, ScatterArgs S,
    const int32_t* __restrict__ list /* (no planes) only the samples list[0 .. *list_n): the samples in front of their rays' */,
    const int32_t* __restrict__ list_n /* termination points -- every other sample's incoming gradient is zero */,
    float dither /* PL_RNE: amplitude of the per-sample dither (plane_dither, x3_device.h) */) {
#define BPL(half, C) (planes_##half + cap * (C))
  if (SCATTER >= 3 && S.gate_front) {  // kernel-uniform: the other kernel of the pair has this batch
    const int gn = *S.gate_n;
    const bool dense = gn > 0 && (float)*S.gate_front >= S.gate_frac * (float)gn;
    if ((int)dense != S.gate_want) return;
  }
  extern __shared__ __attribute__((aligned(16))) u32x4 L[];
  for (int e = threadIdx.x; e < GFN_FIELD_X3_BWD_BYTES / 16; e += blockDim.x) L[e] = image[e];
  float* s_scale = reinterpret_cast<float*>(L + GFN_FIELD_X3_BWD_BYTES / 16);
  __shared__ unsigned int s_rec_cursor[SCATTER >= 3 ? GFN_GRAD_MAX_RANGES : 1];
  if (SCATTER && threadIdx.x < GFN_N_LEVELS) s_scale[threadIdx.x] = S.sc.s[threadIdx.x];
  if (SCATTER >= 3)
    for (int e = threadIdx.x; e < GFN_GRAD_MAX_RANGES; e += blockDim.x) s_rec_cursor[e] = 0;
  __syncthreads();
  const u32x4* Ihi = L;
  const u32x4* Ilo = L + BWD_CHUNKS * 64;
  const bool listed = !PLANES && list != nullptr && *list_n >= 0;  // (a negative count: "every sample", no list was written)
  int64_t n = listed ? (int64_t)*list_n : gfn_live_n(n_max, n_dev);
  if (n > n_max) n = n_max;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t n_tiles = (n + 31) / 32;
  float g_sc = 1.f, g_inv = 1.f;
  if (PLANES) { g_sc = gfn_grad_scale_of(gmax_dev[0]); g_inv = 1.0f / g_sc; }
  constexpr int kWaves = bwd_threads<PLANES, SCATTER>() / 64;
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kWaves) {
    const int64_t e = tile * 32 + j;
    const bool valid = e < n;
    const int64_t s = listed ? (int64_t)list[valid ? e : n - 1] : e;  // the lane's sample
    const int64_t q = listed ? s : (valid ? s : n - 1);
    const float ds = valid ? d_sigma[q] : 0.f;
    float dz3[3];
    bool any_grad = ds != 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float y = rgb[3 * q + c];
      const float g = valid ? d_rgb[3 * q + c] : 0.f;
      dz3[c] = g * y * (1.f - y);  // sigmoid'
      any_grad = any_grad || g != 0.f;
    }
    float4* dst = SCATTER ? nullptr : reinterpret_cast<float4*>(dfeat + s * 32);
    if (!__any(any_grad)) {  // wave-uniform: the whole tile lies behind its rays' termination points
      if (valid) {
        if (!SCATTER) {
#pragma unroll
          for (int qd = 0; qd < 4; qd++) dst[2 * qd + h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (PLANES) {  // zero this sample's rows of the five gradient planes (lane half h: its half of each row)
          const u32x4 z = {0u, 0u, 0u, 0u};
          constexpr bool LO = PLANES == PL_SPLIT;
          if (PLANES == PL_RNE) {
#pragma unroll
            for (int e = 0; e < 8; e++) *reinterpret_cast<u32x4*>(BPL(hi, BP_DZA1_LO) + plane_off(s, 128, 8 * h + e)) = z;
          }
#pragma unroll
          for (int e = 0; e < 8; e++) {  // 128-wide rows = 16 octets, 8 per half
            const int64_t o = plane_off(s, 128, 8 * h + e);
            *reinterpret_cast<u32x4*>(BPL(hi, BP_DZ2) + o) = z; if (LO) *reinterpret_cast<u32x4*>(BPL(lo, BP_DZ2) + o) = z;
            *reinterpret_cast<u32x4*>(BPL(hi, BP_DZ1) + o) = z; if (LO) *reinterpret_cast<u32x4*>(BPL(lo, BP_DZ1) + o) = z;
            *reinterpret_cast<u32x4*>(BPL(hi, BP_DZA1) + o) = z; if (LO) *reinterpret_cast<u32x4*>(BPL(lo, BP_DZA1) + o) = z;
          }
          *reinterpret_cast<u32x4*>(BPL(hi, BP_DH) + plane_off(s, 16, h)) = z; if (LO) *reinterpret_cast<u32x4*>(BPL(lo, BP_DH) + plane_off(s, 16, h)) = z;
          if (h == 0) { *reinterpret_cast<u32x4*>(BPL(hi, BP_DZ3) + plane_off(s, 8, 0)) = z; if (LO) *reinterpret_cast<u32x4*>(BPL(lo, BP_DZ3) + plane_off(s, 8, 0)) = z; }
        }
      }
      continue;
    }
    // d h0 = d sigma * exp(clamp(h0+1, -15, 15))  (trunc_exp backward, activations.py:41)
    float xx = h0[q] + 1.0f;
    xx = fminf(fmaxf(xx, -15.f), 15.f);
    const float dh0 = ds * expf(xx);
    float sc, inv;
    if (PLANES) {
      sc = g_sc; inv = g_inv;
    } else {
      // per-sample power-of-two scale: gradients are ~1e-5 and below, under the f16 normal range.
      // Every quantity of this sample's column is linear in (dz3, dh0), so the scale is undone exactly at the end.
      const float mx = fmaxf(fmaxf(fabsf(dz3[0]), fabsf(dz3[1])), fmaxf(fabsf(dz3[2]), fabsf(dh0)));
      int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
      e = e < 27 ? 27 : (e > 227 ? 227 : e);
      sc = __uint_as_float((uint32_t)(254 - e) << 23);   // 2^(127-e): mx*sc in [1,2)
      inv = __uint_as_float((uint32_t)e << 23);          // 2^(e-127)
    }
    const unsigned long long* mp = masks + q * 6 + h;
    // (with planes every mask is loaded in front of the layer that reads it: behind the plane stores the compiler may not move the
    // loads there itself, and the four registers of the two masks not yet needed are what the record form of the one-plane kernel lacks)
    unsigned long long m_a1 = 0, m_c1 = 0;
    const unsigned long long m_c2 = mp[4];
    if (!PLANES) { m_a1 = mp[0]; m_c1 = mp[2]; }
    const float pc = PLANES == PL_RNE ? 1.f - plane_dither(q, dither) : 1.f;  // the inverse (to 2^-22) of the factor on this sample's activations
    // ---- d c2 = hW2^T dz3 : one k-step whose fragment is (dz3[0..2], 0...) on the h=0 lanes
    u32x4 zh, zl;
    {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (h == 0) { v[0] = dz3[0] * sc; v[1] = dz3[1] * sc; v[2] = dz3[2] * sc; }
      split8(v, zh, zl);
      if (PLANES == PL_SPLIT && valid && h == 0) {
        *reinterpret_cast<u32x4*>(BPL(hi, BP_DZ3) + plane_off(s, 8, 0)) = zh;
        *reinterpret_cast<u32x4*>(BPL(lo, BP_DZ3) + plane_off(s, 8, 0)) = zl;
      }
      if (PLANES == PL_RNE && valid && h == 0) { u32x4 r; rne8(v, r, pc); *reinterpret_cast<u32x4*>(BPL(hi, BP_DZ3) + plane_off(s, 8, 0)) = r; }
    }
    Acc2 gf;
    if (!PLANES) {
      // Focal stage: weight fragments are read one chunk ahead of their MFMAs (mma3p, x3_device.h), and the narrow products (d h, d feat) consume
      // each 32-row tile of their input as soon as it exists -- the same k-steps in the same order, the same bits -- so that three tiles of
      // fragments are never alive at once (what pays for the prefetch's registers).  The global stage's form (below) also keeps every tile for
      // the planes and d appearance; it is bound by its plane stores, not by the matrix pipe, and stays as it was.
      WFrag w = wfrag_load(Ihi, Ilo, D_G3, lane);
      Frag g2[4];
#pragma unroll
      for (int it = 0; it < 4; it++) {
        Acc2 a;
        acc_init_zero(a);
        mma3p(a, w, Ihi, Ilo, it < 3 ? D_G3 + it + 1 : D_G2, lane, zh, zl);
        grad_tile_to_frag<true>(a, g2[it], m_c2, it);
      }
      // ---- d c1 = hW1^T dz2, feeding d h: rows 1..15 = hW0[:, 16:31]^T dz1 (row 0 = d h0)
      Acc2 adh;
      acc_init_zero(adh);
#pragma unroll
      for (int it = 0; it < 4; it++) {
        Acc2 a;
        acc_init_zero(a);
#pragma unroll
        for (int u = 0; u < 8; u++) mma3p(a, w, Ihi, Ilo, u < 7 ? D_G2 + (u + 1) * 4 + it : D_G1 + 2 * it, lane, g2[u >> 1].hi[u & 1], g2[u >> 1].lo[u & 1]);
        Frag g1;
        grad_tile_to_frag<true>(a, g1, m_c1, it);
        mma3p(adh, w, Ihi, Ilo, D_G1 + 2 * it + 1, lane, g1.hi[0], g1.lo[0]);
        mma3p(adh, w, Ihi, Ilo, it < 3 ? D_G2 + it + 1 : D_GB1, lane, g1.hi[1], g1.lo[1]);
      }
      u32x4 dh_h, dh_l;
      {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = acc_get(adh, r);
        if (h == 0) v[0] = dh0 * sc;
        split8(v, dh_h, dh_l);
      }
      // ---- d a1 = bW1^T dh, feeding d feat = bW0^T dz
      acc_init_zero(gf);
#pragma unroll
      for (int it = 0; it < 4; it++) {
        Acc2 a;
        acc_init_zero(a);
        mma3p(a, w, Ihi, Ilo, D_GB0 + 2 * it, lane, dh_h, dh_l);
        Frag ga;
        grad_tile_to_frag<true>(a, ga, m_a1, it);
        mma3p(gf, w, Ihi, Ilo, D_GB0 + 2 * it + 1, lane, ga.hi[0], ga.lo[0]);
        mma3p(gf, w, Ihi, Ilo, it < 3 ? D_GB1 + it + 1 : -1, lane, ga.hi[1], ga.lo[1]);
      }
    } else {
      Frag g2[4];
#pragma unroll
      for (int it = 0; it < 4; it++) {
        Acc2 a;
        acc_init_zero(a);
        mma3(a, Ihi, Ilo, D_G3 + it, lane, zh, zl);
        const PlaneDst pd{BPL(hi, BP_DZ2), 128, s, 32 * it, h, pc, valid};
        grad_tile_to_frag<true, PLANES == PL_RNE>(a, g2[it], m_c2, it, &pd);
        if (PLANES == PL_SPLIT && valid) store_tile(BPL(hi, BP_DZ2), BPL(lo, BP_DZ2), 128, s, 32 * it, g2[it], h);
      }
      // ---- d c1 = hW1^T dz2
      Frag g1[4];
      m_c1 = mp[2];
#pragma unroll
      for (int it = 0; it < 4; it++) {
        Acc2 a;
        acc_init_zero(a);
#pragma unroll
        for (int u = 0; u < 8; u++) mma3(a, Ihi, Ilo, D_G2 + u * 4 + it, lane, g2[u >> 1].hi[u & 1], g2[u >> 1].lo[u & 1]);
        const PlaneDst pd{BPL(hi, BP_DZ1), 128, s, 32 * it, h, pc, valid};
        grad_tile_to_frag<true, PLANES == PL_RNE>(a, g1[it], m_c1, it, &pd);
        if (PLANES == PL_SPLIT && valid) store_tile(BPL(hi, BP_DZ1), BPL(lo, BP_DZ1), 128, s, 32 * it, g1[it], h);
      }
      // ---- d h: rows 1..15 = hW0[:, 16:31]^T dz1, row 0 = d h0
      u32x4 dh_h, dh_l;
      {
        Acc2 a;
        acc_init_zero(a);
#pragma unroll
        for (int u = 0; u < 8; u++) mma3(a, Ihi, Ilo, D_G1 + u, lane, g1[u >> 1].hi[u & 1], g1[u >> 1].lo[u & 1]);
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = acc_get(a, r);
        if (h == 0) v[0] = dh0 * sc;
        split8(v, dh_h, dh_l);
        if (PLANES == PL_SPLIT && valid) store_half_tile(BPL(hi, BP_DH), BPL(lo, BP_DH), 16, s, 0, dh_h, dh_l, h);
        if (PLANES == PL_RNE && valid) { u32x4 r; rne8(v, r, pc); store_half_tile_rne(BPL(hi, BP_DH), 16, s, 0, r, h); }
      }
      if (PLANES && demb) {
        // ---- d appearance = hW0[:, 31:63]^T dz1, summed over each ray's samples (the embedding row is per ray)
        Acc2 a;
        acc_init_zero(a);
#pragma unroll
        for (int u = 0; u < 8; u++) mma3(a, Ihi, Ilo, D_GE + u, lane, g1[u >> 1].hi[u & 1], g1[u >> 1].lo[u & 1]);
        const int ray = valid ? ray_id[q] : -1;
        const int prev = __shfl_up(ray, 1, 32);
        const bool head = (j == 0) || prev != ray;
        const unsigned long long heads64 = __ballot(head);
        const uint32_t heads = (uint32_t)(heads64 >> (32 * h));
        const uint32_t later = (j == 31) ? 0u : (heads >> (j + 1));
        const int run_end = later ? j + __builtin_ctz(later) : 31;
        float* erow = (head && valid) ? demb + (int64_t)cam_idx[ray] * 32 : nullptr;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float v = half_run_sum(valid ? acc_get(a, r) : 0.f, j, run_end);
          if (erow && v != 0.f) atomicAdd(erow + (r & 3) + 8 * (r >> 2) + 4 * h, v * inv);
        }
      }
      // ---- d a1 = bW1^T dh
      Frag ga[4];
      m_a1 = mp[0];
#pragma unroll
      for (int it = 0; it < 4; it++) {
        Acc2 a;
        acc_init_zero(a);
        mma3(a, Ihi, Ilo, D_GB1 + it, lane, dh_h, dh_l);
        grad_tile_to_frag<true>(a, ga[it], m_a1, it);
        if (PLANES && valid) store_tile(BPL(hi, BP_DZA1), PLANES == PL_RNE ? BPL(hi, BP_DZA1_LO) : BPL(lo, BP_DZA1), 128, s, 32 * it, ga[it], h);
      }
      // ---- d feat = bW0^T dz
      acc_init_zero(gf);
#pragma unroll
      for (int u = 0; u < 8; u++) mma3(gf, Ihi, Ilo, D_GB0 + u, lane, ga[u >> 1].hi[u & 1], ga[u >> 1].lo[u & 1]);
    }
    if (SCATTER) {
      // registers (2a, 2a+1) of the tile = channels (0, 1) of level 2h + (a & 1) + 4 (a >> 1) of sample j
      const uint32_t T = 1u << S.log2_T;
      int64_t qs = q;
      if (PLANES) asm volatile("" : "+v"(qs));  // the point and its volume are loaded HERE: hoisted to the top of the tile (as the compiler
                                                // does when it may) they sit in four registers through the whole decoder chain, which the
                                                // one-plane record form then spills
      const float px = S.pts[3 * qs], py = S.pts[3 * qs + 1], pz = S.pts[3 * qs + 2];
      const int64_t vol = (int64_t)S.anchors[3 * qs];
      // a level's primes are loaded BEFORE the level before it stores its records: the stores may alias anything as far as the
      // compiler knows, so a load written behind them stays behind them -- and on this architecture then waits for their
      // acknowledgement (one counter for loads and stores): 0.34 -> 0.31 ms for the focal backward
      int pr[3], nx[3];
      {
        const int* p0 = S.prim_pool + 3 * ((int64_t)(2 * h) * S.n_volumes + vol);
        pr[0] = p0[0]; pr[1] = p0[1]; pr[2] = p0[2];
      }
#pragma unroll
      for (int a = 0; a < 8; a++) {
        const int level = 2 * h + (a & 1) + 4 * (a >> 1);
        if (a < 7) {
          const int next = 2 * h + ((a + 1) & 1) + 4 * ((a + 1) >> 1);
          const int* pn = S.prim_pool + 3 * ((int64_t)next * S.n_volumes + vol);
          nx[0] = pn[0]; nx[1] = pn[1]; nx[2] = pn[2];
        }
        const float gx = acc_get(gf, 2 * a) * inv, gy = acc_get(gf, 2 * a + 1) * inv;
        if (SCATTER >= 3) scatter_level<true, true, (PLANES != 0 || SCATTER == 4)>(S, s_scale, level, px, py, pz, vol, gx, gy, valid, h, j, T, pr, s_rec_cursor);
        else if (SCATTER == 1) scatter_level<true>(S, s_scale, level, px, py, pz, vol, gx, gy, valid, h, j, T, pr);
        else scatter_level<false>(S, s_scale, level, px, py, pz, vol, gx, gy, valid, h, j, T, pr);
        pr[0] = nx[0]; pr[1] = nx[1]; pr[2] = nx[2];
      }
    } else if (valid) {
#pragma unroll
      for (int qd = 0; qd < 4; qd++)  // rows 8qd+4h .. +3
        dst[2 * qd + h] = make_float4(acc_get(gf, 4 * qd) * inv, acc_get(gf, 4 * qd + 1) * inv, acc_get(gf, 4 * qd + 2) * inv,
                                      acc_get(gf, 4 * qd + 3) * inv);
    }
  }
  if (SCATTER >= 3) {
    __syncthreads();
    for (int e = threadIdx.x; e < S.n_ranges; e += blockDim.x)
      S.rec_count[e * 256 + blockIdx.x] = s_rec_cursor[e] < S.rec_cap ? s_rec_cursor[e] : S.rec_cap;
  }
}

#undef BPL

// largest |gradient| entering the decoder: max over samples of |d rgb| and |d sigma * exp(clamp(h0+1))|, as f32 bits
// (non-negative floats order like unsigned integers) -- the batch-wide scale of the global-stage backward
__global__ __launch_bounds__(256) void grad_max_kernel(int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ d_sigma,
                                                       const float* __restrict__ d_rgb, const float* __restrict__ h0,
                                                       uint32_t* __restrict__ gmax) {
  const int64_t n = gfn_live_n(n_max, n_dev);
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float xx = h0[i] + 1.0f;
    xx = fminf(fmaxf(xx, -15.f), 15.f);
    const float a = fabsf(d_sigma[i] * expf(xx));
    const float b = fmaxf(fmaxf(fabsf(d_rgb[3 * i]), fabsf(d_rgb[3 * i + 1])), fabsf(d_rgb[3 * i + 2]));
    const float c = fmaxf(a, b);
    m = (c <= 3.0e38f) ? fmaxf(m, c) : m;  // ignore inf / nan
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  // ONE same-address atomic per workgroup: 4,096 waves each adding their own took 40 of the kernel's 50 us at the L2
  __shared__ float s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (m > 0.f) atomicMax(gmax, __float_as_uint(m));
  }
}

unsigned persistent_grid8(int64_t n_max, const gfn_launch_opts* opts) {
  int64_t tiles = (n_max + 31) / 32;
  int64_t blocks = (tiles + 7) / 8;
  const int cap = gfn_decoder_cu_cap(opts);
  if (blocks > cap) blocks = cap;  // one 512-thread workgroup per CU: the LDS image is copied once per workgroup
  return (unsigned)(blocks > 0 ? blocks : 1);
}

}  // namespace

extern "C" {

int gfn_field_pack_x3(const float* const* weights, void* fwd_image, void* bwd_image, void* stream) {
  GFN_CHECK_ARG(weights, "gfn_field_pack_x3: null weights");
  for (int i = 0; i < 10; i++) GFN_CHECK_ARG(weights[i], "gfn_field_pack_x3: null weight pointer %d", i);
  GFN_CHECK_ARG(fwd_image || bwd_image, "gfn_field_pack_x3: no output image");
  FieldWeights w{weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7], weights[8], weights[9]};
  const int threads = (FWD_CHUNKS + BWD_CHUNKS) * 64;
  field_pack_x3_kernel<<<threads / 256, 256, 0, gfn_stream(stream)>>>(w, reinterpret_cast<u32x4*>(fwd_image), reinterpret_cast<u32x4*>(bwd_image));
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_field_fwd_x3(const float* feat, const int32_t* ray_id, const float* sh, const float* emb, const int32_t* cam_idx,
                     const void* fwd_image, int64_t n_max, const int32_t* n_dev, float* sigma, float* rgb, float* h0,
                     uint64_t* masks, uint16_t* planes_hi, uint16_t* planes_lo, int64_t plane_cap, int64_t feat_stride,
                     const int32_t* list, const int32_t* list_n, const gfn_launch_opts* opts, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG((list == nullptr) == (list_n == nullptr), "gfn_field_fwd_x3: a sample list needs its count");
  GFN_CHECK_ARG(feat && ray_id && sh && fwd_image && sigma && rgb, "gfn_field_fwd_x3: null pointer");
  GFN_CHECK_ARG(feat_stride == 0 || feat_stride >= n_max, "gfn_field_fwd_x3: feat_stride=%lld must be 0 or >= n_max", (long long)feat_stride);
  GFN_CHECK_ARG(!planes_hi || (plane_cap >= n_max && plane_cap % 32 == 0), "gfn_field_fwd_x3: plane_cap=%lld must be >= n_max and a multiple of 32", (long long)plane_cap);
  GFN_CHECK_ARG(!emb || cam_idx, "gfn_field_fwd_x3: emb given without cam_idx");
  GFN_CHECK_ARG((h0 == nullptr) == (masks == nullptr), "gfn_field_fwd_x3: h0 and masks must be given together");
  GFN_CHECK_ARG((planes_hi || !planes_lo) && (!planes_hi || masks), "gfn_field_fwd_x3: planes need the hi half (alone: rounded to nearest) and the masks");
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31) / 32, "gfn_field_fwd_x3: n_max=%lld", (long long)n_max);
  GFN_CHECK_ARG(((reinterpret_cast<uintptr_t>(fwd_image) | reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(sh) |
                  reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(planes_hi) | reinterpret_cast<uintptr_t>(planes_lo)) & 15) == 0,
                "gfn_field_fwd_x3: image, feat, sh, emb and planes must be 16-byte aligned");
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
#define GFN_FWD_ALL(X) X(true, 1, false) X(true, 0, false) X(false, 0, false) X(true, 0, true) X(false, 0, true) X(true, 1, true) X(true, 2, false) X(true, 2, true)
  if (gfn_first_call_on_device(attr_seen)) {
#define GFN_SET(S, P, L) GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(field_fwd_x3_kernel<S, P, L>), hipFuncAttributeMaxDynamicSharedMemorySize, GFN_FIELD_X3_FWD_BYTES));
    GFN_FWD_ALL(GFN_SET)
#undef GFN_SET
  }
  const unsigned grid = persistent_grid8(n_max, opts);
  const u32x4* img = reinterpret_cast<const u32x4*>(fwd_image);
  hipStream_t st = gfn_stream(stream);
  unsigned long long* mk = reinterpret_cast<unsigned long long*>(masks);
  const bool save = masks != nullptr, planar = feat_stride != 0;
  const int planes = !planes_hi ? PL_NONE : (planes_lo ? PL_SPLIT : PL_RNE);
  if (!save) h0 = nullptr;
#define GFN_LAUNCH(S, P, L)                                                                                                         \
  if (save == S && planes == P && planar == L)                                                                                       \
    field_fwd_x3_kernel<S, P, L><<<grid, 512, GFN_FIELD_X3_FWD_BYTES, st>>>(n_max, n_dev, feat, ray_id, sh, emb, cam_idx, img, sigma, rgb, h0, mk, \
                                                                           planes_hi, planes_lo, plane_cap, feat_stride, list, list_n, gfn_plane_dither_amp(opts));
  GFN_FWD_ALL(GFN_LAUNCH)
#undef GFN_LAUNCH
#undef GFN_FWD_ALL
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

struct ScatterHost {  // optional: scatter d feat into a table gradient pool instead of writing it
  const float* pts = nullptr; const int32_t* anchors = nullptr; const int32_t* prim_pool = nullptr; const float* bias_pool = nullptr;
  int64_t n_volumes = 0; int log2_T = 0; void* grad_pool = nullptr; int grad_f16 = 0;
  void* rec = nullptr; unsigned int* rec_count = nullptr; int64_t rec_cap_total = 0;  // record mode (see ScatterArgs)
};

static int field_bwd_x3_impl(const char* fn, const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0,
                             const uint64_t* masks, const void* bwd_image, int64_t n_max, const int32_t* n_dev, float* dfeat,
                             const float* gmax_dev, uint16_t* planes_hi, uint16_t* planes_lo, int64_t plane_cap, const int32_t* ray_id,
                             const int32_t* cam_idx, float* demb, const ScatterHost& sh, const gfn_launch_opts* opts, void* stream,
                             const int32_t* list = nullptr, const int32_t* list_n = nullptr) {
  if (n_max == 0) return GFN_OK;
  const int merge = opts ? opts->record_merge : GFN_MERGE_OFF;
  GFN_CHECK_ARG(merge == GFN_MERGE_OFF || merge == GFN_MERGE_ON || merge == GFN_MERGE_DEVICE, "%s: record_merge=%d", fn, merge);
  GFN_CHECK_ARG(merge != GFN_MERGE_DEVICE || (opts->merge_front && opts->merge_n && opts->merge_frac >= 0.f),
                "%s: GFN_MERGE_DEVICE needs both device counts and merge_frac >= 0", fn);
  GFN_CHECK_ARG((list == nullptr) == (list_n == nullptr) && !(list && planes_hi), "%s: a sample list needs its count and excludes the planes", fn);
  GFN_CHECK_ARG(!list || sh.grad_pool, "%s: the sample list belongs to the scattering form", fn);
  const bool scatter = sh.grad_pool != nullptr;
  GFN_CHECK_ARG(d_sigma && d_rgb && rgb && h0 && masks && bwd_image && (dfeat || scatter), "%s: null pointer", fn);
  GFN_CHECK_ARG(!scatter || (sh.pts && sh.anchors && sh.prim_pool && sh.n_volumes > 0 && sh.n_volumes < (1ll << 26) && sh.log2_T >= 4 && sh.log2_T <= 26),
                "%s: scatter needs pts, anchors, primes, n_volumes and log2_T", fn);
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31) / 32, "%s: n_max=%lld", fn, (long long)n_max);
  GFN_CHECK_ARG(!planes_hi || (plane_cap >= n_max && plane_cap % 32 == 0), "%s: plane_cap=%lld must be >= n_max and a multiple of 32", fn, (long long)plane_cap);
  GFN_CHECK_ARG(((reinterpret_cast<uintptr_t>(bwd_image) | reinterpret_cast<uintptr_t>(dfeat) | reinterpret_cast<uintptr_t>(planes_hi) |
                  reinterpret_cast<uintptr_t>(planes_lo)) & 15) == 0, "%s: image, dfeat and planes must be 16-byte aligned", fn);
  constexpr int kLds = GFN_FIELD_X3_BWD_BYTES + 64;  // + the 16 level scales of the scatter epilogue
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
#define GFN_BWD_ALL(X) X(0, 0) X(0, 1) X(0, 2) X(0, 3) X(0, 4) X(1, 0) X(1, 1) X(1, 2) X(1, 3) X(2, 0) X(2, 1) X(2, 2) X(2, 3)
  if (gfn_first_call_on_device(attr_seen)) {
#define GFN_SET(P, SC) GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(field_bwd_dx_x3_kernel<P, SC>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    GFN_BWD_ALL(GFN_SET)
#undef GFN_SET
  }
  const unsigned grid = persistent_grid8(n_max, opts);
  const float dither = gfn_plane_dither_amp(opts);
  const u32x4* img = reinterpret_cast<const u32x4*>(bwd_image);
  const unsigned long long* mk = reinterpret_cast<const unsigned long long*>(masks);
  ScatterArgs S;
  S.pts = sh.pts; S.anchors = sh.anchors; S.prim_pool = sh.prim_pool; S.bias_pool = sh.bias_pool; S.n_volumes = sh.n_volumes; S.log2_T = sh.log2_T;
  S.grad_pool = sh.grad_pool; S.grad_f16 = sh.grad_f16;
  S.rec = reinterpret_cast<uint2*>(sh.rec); S.rec_count = sh.rec_count; S.rec_cap = 0; S.n_ranges = 0;
  S.gate_front = nullptr; S.gate_n = nullptr; S.gate_frac = 0.f; S.gate_want = 0;
  if (sh.rec) {
    const int64_t T = 1ll << sh.log2_T;
    const int64_t n_ranges = (8 * T + T / 2 + (1ll << GFN_GRAD_RANGE_LOG2) - 1) >> GFN_GRAD_RANGE_LOG2;
    GFN_CHECK_ARG(sh.grad_f16 && sh.rec_count && (reinterpret_cast<uintptr_t>(sh.rec) & 15) == 0 && n_ranges <= GFN_GRAD_MAX_RANGES,
                  "%s: records need the f16 pool, a count array, a 16-byte aligned buffer and log2_T <= 21", fn);
    const int64_t cap = (sh.rec_cap_total / (n_ranges * 256)) & ~1ll;  // 256 workgroup streams per range whatever the grid
    GFN_CHECK_ARG(cap >= 2, "%s: record buffer too small (%lld ranges x 256 streams)", fn, (long long)n_ranges);
    S.rec_cap = (unsigned int)(cap > 0x7ffffffe ? 0x7ffffffe : cap);
    GFN_CHECK_ARG(n_ranges * 256 * (int64_t)S.rec_cap < (1ll << 32), "%s: more than 2^32 records (the kernels index them in 32 bits)", fn);
    S.n_ranges = (int)n_ranges;
  }
  gfn_hash_level_scales_host(S.sc.s);
  const int planes = !planes_hi ? PL_NONE : (planes_lo ? PL_SPLIT : PL_RNE);
  const int mode = !scatter ? 0 : (sh.rec ? ((merge == GFN_MERGE_ON && !planes) ? 4 : 3) : (sh.grad_f16 ? 1 : 2));
  if (!planes) { gmax_dev = nullptr; plane_cap = 0; ray_id = nullptr; cam_idx = nullptr; demb = nullptr; }
#define GFN_LAUNCH(P, SC)                                                                                                        \
  if (planes == P && mode == SC)                                                                                                 \
    field_bwd_dx_x3_kernel<P, SC><<<grid, bwd_threads<P, SC>(), kLds, gfn_stream(stream)>>>(n_max, n_dev, d_sigma, d_rgb, rgb, h0, mk, img, dfeat, gmax_dev, planes_hi, \
                                                                          planes_lo, plane_cap, ray_id, cam_idx, demb, S, list, list_n, dither);
  if (mode == 3 && !planes && merge == GFN_MERGE_DEVICE) {  // the gated pair: the device picks the form (see ScatterArgs)
    S.gate_front = opts->merge_front; S.gate_n = opts->merge_n; S.gate_frac = opts->merge_frac;
    for (int want = 0; want < 2; want++) {
      S.gate_want = want;
      if (want == 0)
        field_bwd_dx_x3_kernel<0, 3><<<grid, bwd_threads<0, 3>(), kLds, gfn_stream(stream)>>>(n_max, n_dev, d_sigma, d_rgb, rgb, h0, mk, img, dfeat, gmax_dev, planes_hi,
                                                                                       planes_lo, plane_cap, ray_id, cam_idx, demb, S, list, list_n, dither);
      else
        field_bwd_dx_x3_kernel<0, 4><<<grid, bwd_threads<0, 4>(), kLds, gfn_stream(stream)>>>(n_max, n_dev, d_sigma, d_rgb, rgb, h0, mk, img, dfeat, gmax_dev, planes_hi,
                                                                                       planes_lo, plane_cap, ray_id, cam_idx, demb, S, list, list_n, dither);
      GFN_CHECK_LAUNCH();
    }
    return GFN_OK;
  }
  GFN_BWD_ALL(GFN_LAUNCH)
#undef GFN_LAUNCH
#undef GFN_BWD_ALL
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_field_bwd_dx_x3(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0, const uint64_t* masks,
                        const void* bwd_image, int64_t n_max, const int32_t* n_dev, float* dfeat, const gfn_launch_opts* opts, void* stream) {
  return field_bwd_x3_impl("gfn_field_bwd_dx_x3", d_sigma, d_rgb, rgb, h0, masks, bwd_image, n_max, n_dev, dfeat, nullptr, nullptr, nullptr, 0,
                           nullptr, nullptr, nullptr, ScatterHost{}, opts, stream);
}

int gfn_field_bwd_scatter_x3(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0, const uint64_t* masks,
                             const void* bwd_image, int64_t n_max, const int32_t* n_dev, const float* pts, const int32_t* anchors,
                             const int32_t* prim_pool, const float* bias_pool, int64_t n_volumes, int log2_T, void* grad_pool, int grad_f16,
                             const float* gmax_dev, uint16_t* planes_hi, uint16_t* planes_lo, int64_t plane_cap, const int32_t* ray_id,
                             const int32_t* cam_idx, float* demb, void* records, uint32_t* record_counts, int64_t record_capacity,
                             const int32_t* list, const int32_t* list_n, const gfn_launch_opts* opts, void* stream) {
  GFN_CHECK_ARG(n_max == 0 || grad_pool, "gfn_field_bwd_scatter_x3: null gradient pool");
  GFN_CHECK_ARG(n_max == 0 || planes_hi || !planes_lo, "gfn_field_bwd_scatter_x3: a lo' plane needs its hi plane");
  GFN_CHECK_ARG(n_max == 0 || !planes_hi || gmax_dev, "gfn_field_bwd_scatter_x3: planes need gmax");
  GFN_CHECK_ARG(n_max == 0 || !demb || (ray_id && cam_idx), "gfn_field_bwd_scatter_x3: demb needs ray_id and cam_idx");
  ScatterHost sh;
  sh.pts = pts; sh.anchors = anchors; sh.prim_pool = prim_pool; sh.bias_pool = bias_pool; sh.n_volumes = n_volumes; sh.log2_T = log2_T;
  sh.grad_pool = grad_pool; sh.grad_f16 = grad_f16;
  sh.rec = records; sh.rec_count = record_counts; sh.rec_cap_total = record_capacity;
  return field_bwd_x3_impl("gfn_field_bwd_scatter_x3", d_sigma, d_rgb, rgb, h0, masks, bwd_image, n_max, n_dev, nullptr, gmax_dev, planes_hi, planes_lo,
                           plane_cap, ray_id, cam_idx, demb, sh, opts, stream, list, list_n);
}

int gfn_field_bwd_x3(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0, const uint64_t* masks,
                     const void* bwd_image, int64_t n_max, const int32_t* n_dev, float* dfeat, const float* gmax_dev,
                     uint16_t* planes_hi, uint16_t* planes_lo, int64_t plane_cap, const int32_t* ray_id, const int32_t* cam_idx,
                     float* demb, const gfn_launch_opts* opts, void* stream) {
  GFN_CHECK_ARG(n_max == 0 || (gmax_dev && planes_hi), "gfn_field_bwd_x3: gmax and the planes (hi, or hi and lo') are required");
  GFN_CHECK_ARG(n_max == 0 || !demb || (ray_id && cam_idx), "gfn_field_bwd_x3: demb needs ray_id and cam_idx");
  return field_bwd_x3_impl("gfn_field_bwd_x3", d_sigma, d_rgb, rgb, h0, masks, bwd_image, n_max, n_dev, dfeat, gmax_dev, planes_hi, planes_lo,
                           plane_cap, ray_id, cam_idx, demb, ScatterHost{}, opts, stream);
}

int gfn_grad_max(const float* d_sigma, const float* d_rgb, const float* h0, int64_t n_max, const int32_t* n_dev, float* gmax_dev,
                 void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(d_sigma && d_rgb && h0 && gmax_dev, "gfn_grad_max: null pointer");
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31), "gfn_grad_max: n_max=%lld", (long long)n_max);
  int64_t blocks = (n_max + 255) / 256;
  if (blocks > 512) blocks = 512;
  grad_max_kernel<<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(n_max, n_dev, d_sigma, d_rgb, h0, reinterpret_cast<uint32_t*>(gmax_dev));
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

float gfn_grad_scale_host(float gmax) { return gfn_grad_scale_of(gmax); }

}  // extern "C"
