// fused_fwd.hip -- the forward of GFNeRFModel.get_outputs after the sampler, in ONE persistent kernel:
// Hash3DAnchored lookup (+ the focal stage's residual table) -> base_network -> density / head input ->
// mlp_head -> front-to-back compositing WITH early termination (gfnerf/nerfacto.py:531-547,
// nerfacto_field.py:412-591, rays.py:178-200, renderers.py:69-283).
//
// Why one kernel: the lookup is bound by the L2-miss request rate of the memory fabric (DESIGN.md section 4:
// 57 G requests/s whatever the kernel does), the decoder by the matrix pipe -- different resources, so the decoder's
// MFMAs run for free beside the gathers of the other waves of the CU; and a ray that has become opaque stops costing
// gathers: a wave walks ITS ray front to back in 32-sample tiles and leaves the ray as soon as the transmittance is
// below early_T (the separate kernels look up and decode every sample, then discard).  Nothing but sigma / rgb / h0 /
// ReLU masks (what the backward needs) leaves the chip: the [N,32] feature tensor is never written.
//
// A wave (64 lanes) owns a tile of 32 consecutive samples of one ray: lane (h, j) gathers 8 of sample j's 16 levels --
// exactly the 16 features that lane holds as B operand of the first layer (k-step u: features 16u + 8h + i =
// levels 8u + 4h + i/2) -- then runs decoder_tile (x3_device.h), then the lower half composites the 32 samples with a
// wave scan.  Rays are handed out through an atomic counter (rays differ 100x in length).  Samples behind the
// termination point get sigma = rgb = weight = alpha = 0 so that the unchanged backward kernels see no gradient there.
#include <stdlib.h>


#include "gfn_common.h"
#include "hash_device.h"
#include "x3_device.h"

namespace {

struct FusedArgs {
  const int* pts_offsets; int n_rays;
  const float* pts; const int* anchors; const float* dt; const float* ts;
  const float* rec; const int* counts;  // gfn_fused_render_rec: the samples where the counting march recorded them (slot k of ray r at (r * 1024 + k) * 8)
  const int* page_table;                // ... or, paged (gfn_sample_rays_record): at (page_table[r * 32 + k / 32] * 32 + k % 32) * 8
  const void* table_a; const void* table_b; const int* prim_a; const int* prim_b; const float* bias_pool;
  int64_t n_volumes; int log2_T;
  const u32x4* image; const float* sh; const float* emb; const int* cam_idx;
  float early_T;
  float* sigma; float* rgb; float* h0; unsigned long long* masks; float* weights; float* alphas; float* feat;
  uint16_t* planes_hi; uint16_t* planes_lo; int64_t cap; float dither;
  float* out_rgb; float* out_acc; float* out_depth; float* t_minmax;
  int* counter;
  int cam_uniform;  // gfn_fused_render(_rec): every ray of the launch has the camera cam_idx[0] (one image): its embedding is folded into the head's bias
  int exp;          // GFN_RENDER_EXP (experiments, tools/README.md round 6): bit 0 s_setprio around the decoder, bit 1 the younger half of the waves starts late
};

__device__ __forceinline__ float incl_scan64(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  return v;
}
__device__ __forceinline__ float sum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float nan_to_num_f(float x) {  // torch.nan_to_num defaults (rays.py:198)
  if (x != x) return 0.f;
  if (x == INFINITY) return 3.4028234663852886e38f;
  if (x == -INFINITY) return -3.4028234663852886e38f;
  return x;
}

constexpr int FUSED_LDS_BYTES = GFN_FIELD_X3_FWD_BYTES + 64;
constexpr int FUSED_THREADS = 256;  // SAVE forms: one wave per SIMD, the 512-entry register file holds a tile without spilling
constexpr int FUSED_THREADS_LEAN = 512;  // forms that keep nothing for a backward: 32-bit table / prime offsets fit them into 256 registers
template <bool SAVE>
constexpr int fused_threads() { return SAVE ? FUSED_THREADS : FUSED_THREADS_LEAN; }

// gather_level<true> with the level's base as a 32-bit pair index (same gathers, same left-to-right blend, same values).
// Shaped for the instruction count (the inference kernel is bound by VALU issue, profiles/r04_pmc_render_per_kernel.csv):
//  * the address is a 32-bit BYTE offset from the (uniform) table base: `global_load_dword v, v_off, s[base]` -- no 64-bit address arithmetic per gather;
//  * w * float(f16) is ONE v_fma_mix_f32 reading the f16 half in place (w * f + 0: the product rounded once, exactly what v_cvt_f32_f16 + v_mul_f32
//    give; only the sign of a zero product can differ, and zeros add and round alike) -- the compiler folds the builtin form back into cvt + mul;
//  * both channels are added with one packed add.  Products and sums stay separately rounded, in the reference's order (Hash3DAnchored_cuda.cu:73-77).
__device__ __forceinline__ float2 gather_pairs_u32(const half2_t* __restrict__ table, uint32_t slot0, const Corners& c) {
  using f2 = __attribute__((ext_vector_type(2))) float;
  uint32_t v[8];
#pragma unroll
  for (int d = 0; d < 8; d++) {
    const uint32_t off = (slot0 + c.pos[d]) << 2;  // < 2^31: 17 T slots of 4 bytes, log2_T <= 26 checked by the host
    v[d] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(table) + off);
  }
  f2 a;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    float t0, t1;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(t0) : "v"(c.w[d]), "v"(v[d]));
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(t1) : "v"(c.w[d]), "v"(v[d]));
    a = (d == 0) ? f2{t0, t1} : a + f2{t0, t1};
  }
  return make_float2(gfn_round_f16(a[0]), gfn_round_f16(a[1]));
}

// the same in two steps, for callers that keep several levels' gathers in flight before they blend any (fused_render_patch_kernel)
__device__ __forceinline__ void gather_issue_u32(const half2_t* __restrict__ table, uint32_t slot0, const uint32_t (&pos)[8], uint32_t (&v)[8]) {
#pragma unroll
  for (int d = 0; d < 8; d++) {
    const uint32_t off = (slot0 + pos[d]) << 2;
    v[d] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(table) + off);
  }
}
__device__ __forceinline__ float2 blend_pairs(const uint32_t (&v)[8], const float (&w)[8]) {
  using f2 = __attribute__((ext_vector_type(2))) float;
  f2 a;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    float t0, t1;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(t0) : "v"(w[d]), "v"(v[d]));
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(t1) : "v"(w[d]), "v"(v[d]));
    a = (d == 0) ? f2{t0, t1} : a + f2{t0, t1};
  }
  return make_float2(gfn_round_f16(a[0]), gfn_round_f16(a[1]));
}

// hash_corners / rehash_corners (hash_device.h) with xor and mask kept APART: an optimisation barrier between them keeps the compiler from fusing the two
// into ONE v_bitop3_b32.  Round 6: the inference kernel's instantiation with the launch's pointer tests compiled away (fused_render_patch_kernel<..., PLAIN>:
// one long basic block per gather group) rendered a handful of pixels of a frame differently from run to run -- same build, same inputs -- until these
// sixteen fused operations per level were split again; with them split two renders of a frame are the same bits (tools/README.md, round 6).  Whether the
// instruction, its distance to the quarter-rate multiplies feeding it, or the compiler's hazard table for gfx950 is at fault was not established: the
// forms that have been bit-exact against the oracle for five rounds keep hash_device.h's code, this instantiation uses the split one.  Same values.
__device__ __forceinline__ uint32_t xor_then_mask(uint32_t t, uint32_t mask) {
  asm("" : "+v"(t));
  return t & mask;
}
__device__ __forceinline__ void corner_slots_split(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1, uint32_t mask, uint32_t (&pos)[8]) {
  pos[0] = xor_then_mask(x0 ^ y0 ^ z0, mask); pos[1] = xor_then_mask(x0 ^ y0 ^ z1, mask);
  pos[2] = xor_then_mask(x0 ^ y1 ^ z0, mask); pos[3] = xor_then_mask(x0 ^ y1 ^ z1, mask);
  pos[4] = xor_then_mask(x1 ^ y0 ^ z0, mask); pos[5] = xor_then_mask(x1 ^ y0 ^ z1, mask);
  pos[6] = xor_then_mask(x1 ^ y1 ^ z0, mask); pos[7] = xor_then_mask(x1 ^ y1 ^ z1, mask);
}
// (no bias pool: PLAIN's launches have none)
__device__ __forceinline__ void hash_corners_split(float px, float py, float pz, float scale, const int* __restrict__ prim, uint32_t mask, Corners& c, uint32_t cell[3]) {
  px = px * scale; py = py * scale; pz = pz * scale;
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const uint32_t ux = sat_u32(fx), uy = sat_u32(fy), uz = sat_u32(fz);
  cell[0] = ux; cell[1] = uy; cell[2] = uz;
  const uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  const uint32_t x0 = ux * pa, y0 = uy * pb, z0 = uz * pc;
  corner_slots_split(x0, x0 + pa, y0, y0 + pb, z0, z0 + pc, mask, c.pos);
  const float a = px - fx, b = py - fy, cc = pz - fz;
  const float na = 1.f - a, nb = 1.f - b, nc = 1.f - cc;
  c.w[0] = (na * nb) * nc; c.w[1] = (na * nb) * cc; c.w[2] = (na * b) * nc; c.w[3] = (na * b) * cc;
  c.w[4] = (a * nb) * nc; c.w[5] = (a * nb) * cc; c.w[6] = (a * b) * nc; c.w[7] = (a * b) * cc;
}
__device__ __forceinline__ void rehash_corners_split(const uint32_t cell[3], const int* __restrict__ prim, uint32_t mask, Corners& c) {
  const uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  const uint32_t x0 = cell[0] * pa, y0 = cell[1] * pb, z0 = cell[2] * pc;
  corner_slots_split(x0, x0 + pa, y0, y0 + pb, z0, z0 + pc, mask, c.pos);
}

// TAB: 0 one table, 1 two tables with their own primes (the focal stage: global + residual block table).
// A compile-time choice on purpose: with a run-time `if` on the table mode inside the unrolled level loop the 512-thread build
// returned wrong values for whole 16-lane groups of the upper half-wave now and then (hipcc 7.2; the same source is fine at 256 threads).
template <int TAB, bool SAVE, int PLANES /* PL_* (x3_device.h) */>
__global__ __launch_bounds__((fused_threads<SAVE>())) void fused_fwd_kernel(FusedArgs A, LevelScales sc) {
  extern __shared__ __attribute__((aligned(16))) u32x4 L[];
  for (int e = threadIdx.x; e < GFN_FIELD_X3_FWD_BYTES / 16; e += blockDim.x) L[e] = A.image[e];
  float* s_scale = reinterpret_cast<float*>(L + GFN_FIELD_X3_FWD_BYTES / 16);
  if (threadIdx.x < GFN_N_LEVELS) s_scale[threadIdx.x] = sc.s[threadIdx.x];
  __syncthreads();
  const u32x4* Ihi = L;
  const u32x4* Ilo = L + FWD_CHUNKS * 64;
  const float* bias = reinterpret_cast<const float*>(L + 2 * FWD_CHUNKS * 64);
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const uint32_t T = 1u << A.log2_T;
  const float early_T = A.early_T;

  // Rays are handed out through 8 atomic counters, one per eighth of the batch; a workgroup starts on the eighth of its
  // XCD group (blockIdx % 8 labels the workgroups that share an XCD and its L2) and steals from the others when that runs
  // dry.  Training batches are random rays, where it makes no difference; a render tile is a block of neighbouring pixels
  // whose rays cross the same cells, and this keeps their table lines in ONE L2 instead of eight.
  // Bounded loops (<= n_rays + 8 fetches per wave, <= 1024 / 32 tiles per ray): the grid always drains.
  const int seg_len = (A.n_rays + 7) / 8;
  int seg_k = 0, fetches = 0;
  // t range over every ray of this wave, kept in LDS (the register file is full): ONE pair of same-address atomics at the end
  __shared__ float s_tmin[fused_threads<SAVE>() / 64], s_tmax[fused_threads<SAVE>() / 64];
  const int wave_id = threadIdx.x >> 6;
  if (lane == 0) { s_tmin[wave_id] = INFINITY; s_tmax[wave_id] = 0.f; }
  for (;;) {
    if (seg_k >= 8 || fetches > A.n_rays + 8) break;
    const int seg = ((int)(blockIdx.x & 7) + seg_k) & 7;
    const int seg_lo = seg * seg_len;
    const int seg_n = min(seg_len, A.n_rays - seg_lo);
    int idx = 0;
    if (lane == 0) idx = seg_n > 0 ? atomicAdd(A.counter + seg, 1) : 0;
    idx = __shfl(idx, 0, 64);
    fetches++;
    if (seg_n <= 0 || idx >= seg_n) { seg_k++; continue; }  // wave-uniform
    const int ray = seg_lo + idx;
    const int s = A.pts_offsets[ray], e = A.pts_offsets[ray + 1];
    const float* sh_row = A.sh + (int64_t)ray * 16;
    const float* emb_row = A.emb ? A.emb + (int64_t)A.cam_idx[ray] * 32 : nullptr;
    float carry = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, ca = 0.f, cd = 0.f, tmin = INFINITY, tmax = 0.f;
    int base = s;
    for (int tile = 0; tile < GFN_MAX_SAMPLE_PER_RAY / 32 + 1 && base < e; tile++, base += 32) {
      const int i = base + j;
      const bool valid = i < e;
      const int64_t q = valid ? i : e - 1;  // lanes past the ray's end redo its last sample and store nothing
      // ---- lookup: this lane's 8 levels of sample q
      const float px = A.pts[3 * q], py = A.pts[3 * q + 1], pz = A.pts[3 * q + 2];
      const int vol = A.anchors[3 * q];
      u32x4 xh[2], xl[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        float x[8];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const int level = 8 * u + 4 * h + m;
          const int tr3 = 3 * (level * (int)A.n_volumes + vol);
          const uint32_t slot0 = (uint32_t)level << (A.log2_T - 1);
          Corners c;
          uint32_t cell[3];
          hash_corners<(TAB != 1 || SAVE)>(px, py, pz, s_scale[level], A.prim_a + tr3, A.bias_pool ? A.bias_pool + tr3 : nullptr, T - 1u, c, cell);
          float2 r = gather_pairs_u32(reinterpret_cast<const half2_t*>(A.table_a), slot0, c);
          if (TAB == 1) {  // hash_feats += residual_hash_feats (nerfacto_field.py:486), through the block table's own primes
            if (A.prim_b) rehash_corners(cell, A.prim_b + tr3, T - 1u, c);  // same cell, same weights: only the slots change
            const float2 r2 = gather_pairs_u32(reinterpret_cast<const half2_t*>(A.table_b), slot0, c);
            r.x = r.x + r2.x; r.y = r.y + r2.y;
          }
          x[2 * m] = r.x; x[2 * m + 1] = r.y;
        }
        if (A.feat && valid) {
          float4* fp = reinterpret_cast<float4*>(A.feat + q * 32 + 16 * u + 8 * h);
          fp[0] = make_float4(x[0], x[1], x[2], x[3]); fp[1] = make_float4(x[4], x[5], x[6], x[7]);
        }
        split8(x, xh[u], xl[u]);
        if (PLANES == PL_SPLIT) store_nat(A.planes_hi + A.cap * FP_FEAT, A.planes_lo + A.cap * FP_FEAT, 32, q, 16 * u, xh[u], xl[u], h);
        if (PLANES == PL_RNE) { u32x4 r; rne8(x, r); store_nat_rne(A.planes_hi + A.cap * FP_FEAT, 32, q, 16 * u, r, h); }  // (f16 numbers already: exact, no dither)
      }
      // ---- decoder
      const DecoderOut o = decoder_tile<PLANES>(Ihi, Ilo, bias, lane, xh, xl, sh_row, emb_row, A.planes_hi, A.planes_lo, A.cap, q,
                                                PLANES == PL_RNE ? 1.f + plane_dither(q, A.dither) : 1.f);
      if (valid) {
        if (h == 0) {
          A.sigma[i] = o.sigma;
          if (SAVE) A.h0[i] = o.h0;
          A.rgb[3 * (int64_t)i] = o.r; A.rgb[3 * (int64_t)i + 1] = o.g; A.rgb[3 * (int64_t)i + 2] = o.b;
        }
        if (SAVE) {
          unsigned long long* mp = A.masks + (int64_t)i * 6 + h;
          mp[0] = o.m_a1; mp[2] = o.m_c1; mp[4] = o.m_c2;
        }
      }
      // ---- compositing of these 32 samples (get_weights_f2nerf + renderers), lower half of the wave
      const bool mine = valid && h == 0;
      const float ti = mine ? A.ts[i] : 0.f;
      const float ds = mine ? A.dt[i] * o.sigma : 0.f;
      const float incl = incl_scan64(ds, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 0.f;
      const float Tr = expf(-(carry + excl));
      const float alpha = 1.f - expf(-ds);
      float w = nan_to_num_f(alpha * Tr);
      if (early_T > 0.f && Tr < early_T) w = 0.f;
      if (mine) {
        if (A.weights) A.weights[i] = w;
        if (A.alphas) A.alphas[i] = alpha;
        cr += w * o.r; cg += w * o.g; cb += w * o.b; ca += w; cd += w * ti;
        tmin = fminf(tmin, ti); tmax = fmaxf(tmax, ti);
      }
      carry += __shfl(incl, 31, 64);
      if (early_T > 0.f && expf(-carry) < early_T) { base += 32; break; }  // the ray is opaque: leave it
    }
    // samples behind the termination point: no lookup, no decoder; zero what the backward reads
    for (int i = base + lane; i < e; i += 64) {
      A.sigma[i] = 0.f;
      A.rgb[3 * (int64_t)i] = 0.f; A.rgb[3 * (int64_t)i + 1] = 0.f; A.rgb[3 * (int64_t)i + 2] = 0.f;
      if (A.weights) A.weights[i] = 0.f;
      if (A.alphas) A.alphas[i] = 0.f;
      const float ti = A.ts[i];
      tmin = fminf(tmin, ti); tmax = fmaxf(tmax, ti);
    }
    cr = sum64(cr); cg = sum64(cg); cb = sum64(cb); ca = sum64(ca); cd = sum64(cd);
    if (A.t_minmax) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
      }
      if (e - s < GFN_MAX_SAMPLE_PER_RAY) tmin = 0.f;  // the reference's dense zero padding joins the min
      if (lane == 0) { s_tmin[wave_id] = fminf(s_tmin[wave_id], tmin); s_tmax[wave_id] = fmaxf(s_tmax[wave_id], tmax); }
    }
    if (lane == 0) {
      A.out_rgb[3 * ray] = cr; A.out_rgb[3 * ray + 1] = cg; A.out_rgb[3 * ray + 2] = cb;
      if (A.out_acc) A.out_acc[ray] = ca;
      if (A.out_depth) A.out_depth[ray] = cd / (ca + 1e-10f);  // renderers.py:279
    }
  }
  if (A.t_minmax && lane == 0 && s_tmin[wave_id] <= s_tmax[wave_id]) {  // non-negative floats order like their bit patterns
    atomicMin(reinterpret_cast<unsigned*>(A.t_minmax), __float_as_uint(s_tmin[wave_id]));
    atomicMax(reinterpret_cast<unsigned*>(A.t_minmax + 1), __float_as_uint(s_tmax[wave_id]));
  }
}

// ---- inference: the same tile of work, dealt the other way round ---------------------------------------------------------------
// fused_fwd_kernel gives a wave 32 consecutive samples of ONE ray: from level 9 on each of them sits in a cell of its own, 32
// distinct cells per gather instruction.  A render tile is a block of neighbouring pixels, and the k-th samples of an 8 x 4 patch
// of neighbouring pixels' rays sit in 1.1 (level 0) ... 10 (level 15) distinct cells -- 0.09 distinct cells per sample and level
// against 0.73 (tools/exp_patch_sharing.py) -- and lanes of one gather instruction that want the same address cost what ONE
// lane costs (tools/gather_share.hip: the rate counts distinct addresses).  So here lane (h, j) owns RAY j of the wave's patch
// for the whole kernel and the wave walks the 32 rays in lockstep, sample index by sample index: lookup (8 of the 16 levels
// per lane, as above) -> decoder_tile on the 32 rays' k-th samples -> every ray composites its own sample front to back in
// its lane (no scan) and drops out once its transmittance is below early_T.  Per-sample outputs are not written (inference).
// tile_w > 0: the rays are a tile_w-wide block of pixels, row major, and a patch is patch_w x (32 / patch_w) pixels;
// tile_w == 0: a patch is 32 consecutive rays.
constexpr int kRenderStrip = 4;      // patch columns per strip of the hand-out order (below)
constexpr int RENDER_THREADS = 512;  // two waves per SIMD: with 32-bit table / prime offsets the kernel needs 254 registers, no spills
// RENDER_LEVELS: levels whose gathers (16 each with two tables) a lane keeps in flight together (2 or 4; GFN_RENDER_LEVELS in the environment
// picks the instantiation -- measured: tools/README.md, round 4)
// REC: the samples come from the march's records (32 bytes each, gfn_sample_rays_record) instead of the packed arrays a fill pass made of them
// FOLD: the appearance embedding is the same for every ray (A.cam_uniform, or there is none): folded into head layer 1's bias once per workgroup
// (decoder_tile<., EMB_FOLDED>)
// X2: a measurement (GFN_RENDER_X2=<layer mask>, x3_device.h X2_*; VERDICT r05 #2's report): those layers' activation operands as one f16 number each, two MFMAs per product sum
// PLAIN: what the launch's pointers say is known at compile time -- no bias pool (rand_bias off) and, with two tables, the block table's own primes: the
// per-level `pointer ? ... : ...` branches (wave-uniform, but each one ends a scheduling region and the slot copies for the shared-primes case are made
// whether they are used or not) fold away: 4,423 -> 4,248 instructions in the kernel, 62 -> 38 basic blocks; its hashes keep xor and mask apart
// (hash_corners_split above: with the two fused into v_bitop3_b32 this instantiation was not deterministic)
template <int TAB, int RENDER_LEVELS, bool REC, bool FOLD, int X2 = 0, bool PLAIN = false>
__global__ __launch_bounds__(RENDER_THREADS) void fused_render_patch_kernel(FusedArgs A, LevelScales sc, int tile_w, int patch_w) {
  extern __shared__ __attribute__((aligned(16))) u32x4 L[];
  for (int e = threadIdx.x; e < GFN_FIELD_X3_FWD_BYTES / 16; e += blockDim.x) L[e] = A.image[e];
  float* s_scale = reinterpret_cast<float*>(L + GFN_FIELD_X3_FWD_BYTES / 16);
  if (threadIdx.x < GFN_N_LEVELS) s_scale[threadIdx.x] = sc.s[threadIdx.x];
  __syncthreads();
  if (FOLD && A.emb) {
    fold_embedding_into_bias(L, A.emb + (int64_t)A.cam_idx[0] * 32);
    __syncthreads();
  }
  const u32x4* Ihi = L;
  const u32x4* Ilo = L + FWD_CHUNKS * 64;
  const float* bias = reinterpret_cast<const float*>(L + 2 * FWD_CHUNKS * 64);
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const uint32_t T = 1u << A.log2_T;
  const float early_T = A.early_T;
  const int n_vol = (int)A.n_volumes;    // 16 * n_volumes * 3 < 2^31 (checked by the host)
  const int n_patches = A.n_rays / 32;  // the host guarantees whole patches
  const int patches_per_row = tile_w > 0 ? tile_w / patch_w : 1, patch_h = tile_w > 0 ? 32 / patch_w : 1;
  const int pw_shift = 31 - __builtin_clz((unsigned)(patch_w > 0 ? patch_w : 1));  // patch_w divides 32: a power of two (checked by the host)
  // patches are handed out like fused_fwd_kernel's rays: 8 counters, a workgroup starts on the eighth of its XCD group
  const int seg_len = (n_patches + 7) / 8;
  int seg_k = 0, fetches = 0;
  __shared__ float s_tmin[RENDER_THREADS / 64], s_tmax[RENDER_THREADS / 64];
  const int wave_id = threadIdx.x >> 6;
  if (lane == 0) { s_tmin[wave_id] = INFINITY; s_tmax[wave_id] = 0.f; }
  if ((A.exp & 2) && wave_id >= RENDER_THREADS / 128)  // experiment: the second wave of every SIMD starts ~(exp >> 8) x 8 k cycles late (lookup beside decoder?)
    for (int i = 0; i < ((A.exp >> 8) & 0xff); i++) __builtin_amdgcn_s_sleep(127);
  for (;;) {
    if (seg_k >= 8 || fetches > n_patches + 8) break;
    const int seg = ((int)(blockIdx.x & 7) + seg_k) & 7;
    const int seg_lo = seg * seg_len;
    const int seg_n = min(seg_len, n_patches - seg_lo);
    int idx = 0;
    if (lane == 0) idx = seg_n > 0 ? atomicAdd(A.counter + seg, 1) : 0;
    idx = __shfl(idx, 0, 64);
    fetches++;
    if (seg_n <= 0 || idx >= seg_n) { seg_k++; continue; }  // wave-uniform
    const int patch = __builtin_amdgcn_readfirstlane(seg_lo + idx);  // wave-uniform: its row / column arithmetic stays on the scalar unit
    // the order patches are handed out in: vertical strips of kRenderStrip patch columns, row major inside a strip -- the ~220 patches an XCD has
    // in flight are then a 32 x 220 pixel block instead of a 384 x 18 pixel band of the tile, their rays closer together in the volume and
    // more of their table lines shared in the XCD's L2 (40.4 -> 39.7 ms/Mpix; strips of 2 ... 6 columns alike, 12 ... 24 less)
    int prow = patch / patches_per_row, pcol = patch - prow * patches_per_row;
    if (tile_w > 0 && patches_per_row % kRenderStrip == 0) {
      const int per_strip = kRenderStrip * (n_patches / patches_per_row);
      const int strip = patch / per_strip, rem = patch - strip * per_strip;
      prow = rem / kRenderStrip; pcol = strip * kRenderStrip + rem % kRenderStrip;
    }
    const int ray = tile_w > 0 ? (prow * patch_h + (j >> pw_shift)) * tile_w + pcol * patch_w + (j & (patch_w - 1)) : patch * 32 + j;
    const int cnt = REC ? A.counts[ray] : A.pts_offsets[ray + 1] - A.pts_offsets[ray];
    int64_t s = REC ? (int64_t)ray * GFN_MAX_SAMPLE_PER_RAY : (int64_t)A.pts_offsets[ray];
    const bool paged = REC && A.page_table != nullptr;  // wave-uniform
    int kmax = cnt;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
    const int cam = (!FOLD && A.emb) ? A.cam_idx[ray] : 0;
    float carry = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, ca = 0.f, cd = 0.f;
    bool done = false;
    // a lane's ray is fixed for the whole walk: its 8 SH values are split once (with the embedding folded away the registers are there)
    // (the four-levels-in-flight instantiation has no registers to spare: it splits per sample like before)
    constexpr bool KEEP_SH = FOLD && RENDER_LEVELS == 2;
    u32x4 sh_split[2];
    if (KEEP_SH) {
      float x[8];
      load8(A.sh + (int64_t)ray * 16 + 8 * h, x);
      if (X2 & X2_H0) { rne8(x, sh_split[0]); sh_split[1] = u32x4{0u, 0u, 0u, 0u}; } else split8(x, sh_split[0], sh_split[1]);
    }
    for (int k = 0; k < kmax; k++) {
      const bool act = k < cnt && !done;
      if (__ballot(act) == 0ull) break;  // every ray of the patch has ended or become opaque
      if (paged && (k & (GFN_REC_PAGE - 1)) == 0 && k < cnt) {  // a new page of this ray's records: s such that slot k is still s + k
        int ray_p = ray;
        asm volatile("" : "+v"(ray_p));
        s = ((int64_t)A.page_table[(int64_t)ray_p * (GFN_MAX_SAMPLE_PER_RAY / GFN_REC_PAGE) + (k >> GFN_REC_PAGE_LOG2)] << GFN_REC_PAGE_LOG2) - k;
      }
      const int64_t q = s + k;
      // (rebuilt every sample from two 32-bit values the compiler cannot see through: as 64-bit pointers -- or their hoisted parts -- kept
      // across the loops they were what spilled)
      int ray_k = ray, cam_k = cam;
      asm volatile("" : "+v"(ray_k), "+v"(cam_k));
      const float* sh_row = A.sh + (int64_t)ray_k * 16;
      const float* emb_row = (!FOLD && A.emb) ? A.emb + (int64_t)cam_k * 32 : nullptr;
      // ---- lookup: this lane's 8 levels of its ray's k-th sample (rays that have left gather nothing)
      u32x4 xh[2], xl[2];
      float x[2][8];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int m = 0; m < 8; m++) x[u][m] = 0.f;
      if (act) {
        float px, py, pz;
        int vol;
        if (REC) {  // {t, dist, leaf, transform | warped x, y, z, block}: the table-space point as the fill pass makes it (sampler.hip)
          const float4 r1 = *reinterpret_cast<const float4*>(A.rec + q * GFN_MARCH_REC_FLOATS + 4);
          const float inv3 = 1.0f / 3.0f;
          px = (r1.x + 1.5f) * inv3; py = (r1.y + 1.5f) * inv3; pz = (r1.z + 1.5f) * inv3;
          vol = __float_as_int(A.rec[q * GFN_MARCH_REC_FLOATS + 3]);
        } else {
          px = A.pts[3 * q]; py = A.pts[3 * q + 1]; pz = A.pts[3 * q + 2];
          vol = A.anchors[3 * q];
        }
        // The lookup of a lane's 8 levels is a chain of round trips (point -> primes -> slots) with a handful of instructions between them,
        // and with two waves per SIMD there is little else to run meanwhile: taken level by level it was 1 + 2 x 8 x 2 dependent trips per
        // sample.  So, per half of the lane's levels: its 8 prime triples first (one trip), then the levels in groups of RENDER_LEVELS -- hash every
        // level of the group for both tables, issue ALL its gathers (16 per level), then blend -- 1 + 2 + 8 / RENDER_LEVELS trips.  Registers are there: the
        // decoder's fragments are dead during the lookup.  32-bit offsets throughout (64-bit per-level bases cost 2 x 8 x 4 registers).
#pragma unroll
        for (int uu = 0; uu < 2; uu++) {
        int pra[4][3], prb[4][3];  // the prime triples of this half's 4 levels, both tables: one trip
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const int level = 8 * uu + 4 * h + m;
          const int tr3 = 3 * (level * n_vol + vol);
          pra[m][0] = A.prim_a[tr3]; pra[m][1] = A.prim_a[tr3 + 1]; pra[m][2] = A.prim_a[tr3 + 2];
          if (TAB == 1 && (PLAIN || A.prim_b)) { prb[m][0] = A.prim_b[tr3]; prb[m][1] = A.prim_b[tr3 + 1]; prb[m][2] = A.prim_b[tr3 + 2]; }
        }
#pragma unroll
        for (int g0 = 4 * uu; g0 < 4 * uu + 4; g0 += RENDER_LEVELS) {
          Corners c[RENDER_LEVELS];
          uint32_t posb[RENDER_LEVELS][8], va[RENDER_LEVELS][8], vb[RENDER_LEVELS][8];
#pragma unroll
          for (int l = 0; l < RENDER_LEVELS; l++) {
            const int g = g0 + l, level = 8 * (g >> 2) + 4 * h + (g & 3);
            uint32_t cell[3];
            if (PLAIN) hash_corners_split(px, py, pz, s_scale[level], pra[g & 3], T - 1u, c[l], cell);
            else hash_corners(px, py, pz, s_scale[level], pra[g & 3], A.bias_pool ? A.bias_pool + 3 * (level * n_vol + vol) : nullptr, T - 1u, c[l], cell);
            if (TAB == 1) {
              Corners cb;
              if (PLAIN) rehash_corners_split(cell, prb[g & 3], T - 1u, cb);
              else if (A.prim_b) rehash_corners(cell, prb[g & 3], T - 1u, cb);  // same cell, same weights: only the slots change
#pragma unroll
              for (int d = 0; d < 8; d++) posb[l][d] = (PLAIN || A.prim_b) ? cb.pos[d] : c[l].pos[d];
            }
          }
#pragma unroll
          for (int l = 0; l < RENDER_LEVELS; l++) {
            const int g = g0 + l, level = 8 * (g >> 2) + 4 * h + (g & 3);
            const uint32_t slot0 = (uint32_t)level << (A.log2_T - 1);  // level * T halfs = level * T / 2 pairs
            gather_issue_u32(reinterpret_cast<const half2_t*>(A.table_a), slot0, c[l].pos, va[l]);
            if (TAB == 1) gather_issue_u32(reinterpret_cast<const half2_t*>(A.table_b), slot0, posb[l], vb[l]);
          }
#pragma unroll
          for (int l = 0; l < RENDER_LEVELS; l++) {
            const int g = g0 + l;
            float2 r = blend_pairs(va[l], c[l].w);
            if (TAB == 1) {  // hash_feats += residual_hash_feats (nerfacto_field.py:486)
              const float2 r2 = blend_pairs(vb[l], c[l].w);
              r.x = r.x + r2.x; r.y = r.y + r2.y;
            }
            x[g >> 2][2 * (g & 3)] = r.x; x[g >> 2][2 * (g & 3) + 1] = r.y;
          }
        }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (X2 & X2_B0) { rne8(x[u], xh[u]); xl[u] = u32x4{0u, 0u, 0u, 0u}; }  // (to NEAREST: split8's hi half is rounded toward zero, its lo' makes up for it)
        else split8(x[u], xh[u], xl[u]);
      }
      // ---- decoder: the 32 rays' k-th samples are the tile's 32 rows
      if (A.exp & 1) __builtin_amdgcn_s_setprio(2);  // experiment: the wave in its matrix phase wins the issue arbitration
      const DecoderOut o = decoder_tile<PL_NONE, FOLD, X2>(Ihi, Ilo, bias, lane, xh, xl, sh_row, emb_row, nullptr, nullptr, 0, q, 1.f, KEEP_SH ? sh_split : nullptr);
      if (A.exp & 1) __builtin_amdgcn_s_setprio(0);
      // ---- compositing (get_weights_f2nerf + renderers): each ray in its own lane, front to back
      if (act && h == 0) {
        const float ti = REC ? A.rec[q * GFN_MARCH_REC_FLOATS] : A.ts[q];
        const float ds = (REC ? A.rec[q * GFN_MARCH_REC_FLOATS + 1] : A.dt[q]) * o.sigma;
        const float Tr = expf(-carry);
        const float alpha = 1.f - expf(-ds);
        float w = nan_to_num_f(alpha * Tr);
        if (early_T > 0.f && Tr < early_T) w = 0.f;
        cr += w * o.r; cg += w * o.g; cb += w * o.b; ca += w; cd += w * ti;
        carry += ds;
        if (early_T > 0.f && expf(-carry) < early_T) done = true;  // the ray is opaque: it gathers nothing from here on
      }
      done = __shfl((int)done, j, 64) != 0;  // the upper half (the ray's other 8 levels) follows its ray
    }
    if (h == 0) {
      A.out_rgb[3 * ray] = cr; A.out_rgb[3 * ray + 1] = cg; A.out_rgb[3 * ray + 2] = cb;
      if (A.out_acc) A.out_acc[ray] = ca;
      if (A.out_depth) A.out_depth[ray] = cd / (ca + 1e-10f);  // renderers.py:279
    }
    if (A.t_minmax) {  // t grows along a ray: its range is its first and its last sample, terminated or not
      float tmin = INFINITY, tmax = 0.f;
      if (cnt > 0) {
        int64_t first = s, last = s + cnt - 1;
        if (paged) {
          const int* pt = A.page_table + (int64_t)ray * (GFN_MAX_SAMPLE_PER_RAY / GFN_REC_PAGE);
          first = (int64_t)pt[0] << GFN_REC_PAGE_LOG2;
          last = ((int64_t)pt[(cnt - 1) >> GFN_REC_PAGE_LOG2] << GFN_REC_PAGE_LOG2) + ((cnt - 1) & (GFN_REC_PAGE - 1));
        }
        tmin = REC ? A.rec[first * GFN_MARCH_REC_FLOATS] : A.ts[first];
        tmax = REC ? A.rec[last * GFN_MARCH_REC_FLOATS] : A.ts[last];
      }
      if (cnt < GFN_MAX_SAMPLE_PER_RAY) tmin = 0.f;  // the reference's dense zero padding joins the min
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
      }
      if (lane == 0) { s_tmin[wave_id] = fminf(s_tmin[wave_id], tmin); s_tmax[wave_id] = fmaxf(s_tmax[wave_id], tmax); }
    }
  }
  if (A.t_minmax && lane == 0 && s_tmin[wave_id] <= s_tmax[wave_id]) {
    atomicMin(reinterpret_cast<unsigned*>(A.t_minmax), __float_as_uint(s_tmin[wave_id]));
    atomicMax(reinterpret_cast<unsigned*>(A.t_minmax + 1), __float_as_uint(s_tmax[wave_id]));
  }
}

__global__ void fused_init_kernel(int* counter, float* t_minmax) {
  for (int i = 0; i < 8; i++) counter[i] = 0;
  if (t_minmax) { t_minmax[0] = INFINITY; t_minmax[1] = 0.f; }
}

}  // namespace

extern "C" {

int gfn_fused_forward(const int32_t* pts_offsets, int64_t n_rays, const float* pts, const int32_t* anchors, const float* dt,
                      const float* ts, const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool,
                      const int32_t* prim_pool2, const float* bias_pool, int64_t n_volumes, int log2_T, const void* fwd_image,
                      const float* sh, const float* emb, const int32_t* cam_idx, float early_T, float* sigma, float* rgb,
                      float* h0, uint64_t* masks, float* weights, float* alphas, float* feat, uint16_t* planes_hi,
                      uint16_t* planes_lo, int64_t plane_cap, float* out_rgb, float* out_acc, float* out_depth,
                      float* t_minmax, int32_t* work_counter, const gfn_launch_opts* opts, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(pts_offsets && pts && anchors && dt && ts && table_f16 && prim_pool && fwd_image && sh && sigma && rgb && out_rgb &&
                    work_counter, "gfn_fused_forward: null pointer");
  GFN_CHECK_ARG(n_rays > 0 && n_rays < (1ll << 30), "gfn_fused_forward: n_rays=%lld", (long long)n_rays);
  GFN_CHECK_ARG(n_volumes > 0 && n_volumes < (1ll << 25) && log2_T >= 4 && log2_T <= 26, "gfn_fused_forward: n_volumes / log2_T out of range");  // 3 * 16 * n_volumes and 16 * T / 2 are 32-bit offsets in the kernel
  GFN_CHECK_ARG(!emb || cam_idx, "gfn_fused_forward: emb given without cam_idx");
  GFN_CHECK_ARG((h0 == nullptr) == (masks == nullptr), "gfn_fused_forward: h0 and masks must be given together");
  GFN_CHECK_ARG((planes_hi || !planes_lo) && (!planes_hi || (masks && plane_cap > 0 && plane_cap % 32 == 0)),
                "gfn_fused_forward: planes need the hi half (alone: rounded to nearest), the masks and a capacity that is a multiple of 32");
  GFN_CHECK_ARG(((reinterpret_cast<uintptr_t>(fwd_image) | reinterpret_cast<uintptr_t>(sh) | reinterpret_cast<uintptr_t>(emb) |
                  reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(planes_hi) | reinterpret_cast<uintptr_t>(planes_lo)) & 15) == 0,
                "gfn_fused_forward: image, sh, emb, feat and planes must be 16-byte aligned");
  FusedArgs A;
  A.pts_offsets = pts_offsets; A.n_rays = (int)n_rays; A.pts = pts; A.anchors = anchors; A.dt = dt; A.ts = ts;
  A.table_a = table_f16; A.table_b = table2_f16; A.prim_a = prim_pool; A.prim_b = table2_f16 ? prim_pool2 : nullptr; A.bias_pool = bias_pool;
  A.n_volumes = n_volumes; A.log2_T = log2_T;
  A.image = reinterpret_cast<const u32x4*>(fwd_image); A.sh = sh; A.emb = emb; A.cam_idx = cam_idx; A.early_T = early_T;
  A.sigma = sigma; A.rgb = rgb; A.h0 = h0; A.masks = reinterpret_cast<unsigned long long*>(masks); A.weights = weights; A.alphas = alphas;
  A.feat = feat; A.planes_hi = planes_hi; A.planes_lo = planes_lo; A.cap = plane_cap; A.dither = gfn_plane_dither_amp(opts);
  A.out_rgb = out_rgb; A.out_acc = out_acc; A.out_depth = out_depth; A.t_minmax = t_minmax; A.counter = work_counter;
  LevelScales sc;
  gfn_hash_level_scales_host(sc.s);
  hipStream_t st = gfn_stream(stream);
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
#define GFN_FUSED_ALL(X) X(0, false, 0) X(0, true, 0) X(0, true, 1) X(0, true, 2) X(1, false, 0) X(1, true, 0) X(1, true, 1) X(1, true, 2)
  if (gfn_first_call_on_device(attr_seen)) {
#define GFN_SET(D, S, P) \
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<D, S, P>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    GFN_FUSED_ALL(GFN_SET)
#undef GFN_SET
  }
  fused_init_kernel<<<1, 1, 0, st>>>(work_counter, t_minmax);
  GFN_CHECK_LAUNCH();
  int64_t waves = n_rays;  // one ray at a time per wave
  const int tab = table2_f16 ? 1 : 0;
  const bool save = masks != nullptr;
  const int planes = !planes_hi ? PL_NONE : (planes_lo ? PL_SPLIT : PL_RNE);
  const int threads = save ? FUSED_THREADS : FUSED_THREADS_LEAN;
  int64_t blocks = (waves + threads / 64 - 1) / (threads / 64);
  const int cu_cap = gfn_decoder_cu_cap(opts);  // leave CUs to kernels of another stream
  if (blocks > cu_cap) blocks = cu_cap;
#define GFN_LAUNCH(D, S, P) \
  if (tab == D && save == S && planes == P) fused_fwd_kernel<D, S, P><<<(unsigned)blocks, fused_threads<S>(), FUSED_LDS_BYTES, st>>>(A, sc);
  GFN_FUSED_ALL(GFN_LAUNCH)
#undef GFN_LAUNCH
#undef GFN_FUSED_ALL
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"

// the launch of gfn_fused_render / gfn_fused_render_rec (A holds the sample source of either)
static int fused_render_launch(FusedArgs& A, const char* who, int64_t n_rays, const uint16_t* table_f16, const uint16_t* table2_f16,
                               const int32_t* prim_pool, const int32_t* prim_pool2, const float* bias_pool, int64_t n_volumes, int log2_T,
                               const void* fwd_image, const float* sh, const float* emb, const int32_t* cam_idx, int cam_uniform, float early_T, int tile_width,
                               int patch_width, float* out_rgb, float* out_acc, float* out_depth, float* t_minmax, int32_t* work_counter,
                               const gfn_launch_opts* opts, void* stream) {
  GFN_CHECK_ARG(table_f16 && prim_pool && fwd_image && sh && out_rgb && work_counter, "%s: null pointer", who);
  GFN_CHECK_ARG(n_rays > 0 && n_rays < (1ll << 30) && n_rays % 32 == 0, "%s: n_rays=%lld must be a positive multiple of 32", who, (long long)n_rays);
  GFN_CHECK_ARG(n_volumes > 0 && n_volumes < (1ll << 25) && log2_T >= 4 && log2_T <= 26, "%s: n_volumes / log2_T out of range", who);
  GFN_CHECK_ARG(!emb || cam_idx, "%s: emb given without cam_idx", who);
  GFN_CHECK_ARG(tile_width == 0 || (tile_width > 0 && patch_width > 0 && 32 % patch_width == 0 && tile_width % patch_width == 0 &&
                                    n_rays % tile_width == 0 && (n_rays / tile_width) % (32 / patch_width) == 0),
                "%s: a %d-wide tile of %lld rays does not divide into %d x %d patches", who, tile_width, (long long)n_rays, patch_width,
                patch_width > 0 ? 32 / patch_width : 0);
  GFN_CHECK_ARG(((reinterpret_cast<uintptr_t>(fwd_image) | reinterpret_cast<uintptr_t>(sh) | reinterpret_cast<uintptr_t>(emb)) & 15) == 0,
                "%s: image, sh and emb must be 16-byte aligned", who);
  A.n_rays = (int)n_rays;
  A.table_a = table_f16; A.table_b = table2_f16; A.prim_a = prim_pool; A.prim_b = table2_f16 ? prim_pool2 : nullptr; A.bias_pool = bias_pool;
  A.n_volumes = n_volumes; A.log2_T = log2_T;
  A.image = reinterpret_cast<const u32x4*>(fwd_image); A.sh = sh; A.emb = emb; A.cam_idx = cam_idx; A.early_T = early_T;
  A.out_rgb = out_rgb; A.out_acc = out_acc; A.out_depth = out_depth; A.t_minmax = t_minmax; A.counter = work_counter;
  static const int exp_bits = [] { const char* e = getenv("GFN_RENDER_EXP"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
  A.cam_uniform = cam_uniform != 0; A.exp = exp_bits;
  const bool fold = !emb || cam_uniform != 0;  // nothing to add, or the same 128 numbers for every sample of the launch
  LevelScales sc;
  gfn_hash_level_scales_host(sc.s);
  hipStream_t st = gfn_stream(stream);
  static unsigned long long attr_seen = 0;
#define GFN_RENDER_ALL(X) X(0, 2, false, false) X(1, 2, false, false) X(0, 4, false, false) X(1, 4, false, false) X(0, 2, true, false) X(1, 2, true, false) X(0, 4, true, false) X(1, 4, true, false) \
                          X(0, 2, false, true) X(1, 2, false, true) X(0, 4, false, true) X(1, 4, false, true) X(0, 2, true, true) X(1, 2, true, true) X(0, 4, true, true) X(1, 4, true, true)
  if (gfn_first_call_on_device(attr_seen)) {
#define GFN_SET(T, L, R, F) \
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_render_patch_kernel<T, L, R, F>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    GFN_RENDER_ALL(GFN_SET)
#undef GFN_SET
  }
  static const int levels = [] { const char* e = getenv("GFN_RENDER_LEVELS"); return (e && atoi(e) == 4) ? 4 : 2; }();
  // The layers of the decoder whose activation operand is one f16 number (x3_device.h X2_*): gfn_launch_opts.colour_x2 asks for the colour head's layer 2
  // (128 -> 128: half of the decoder's MFMAs; measured, tools/README.md round 6: 7-8 % of the frame time for 5.5e-5 ... 1.3e-4 on trained frames, where layers 2 AND 3 cost
  // twice the distance for 2 % more); GFN_RENDER_X2=<mask> overrides it (a measurement, read per call: tools/exp_render_x2.py renders ONE engine every way in one process).  The
  // forms exist for the tile renderer's instantiation (PLAIN) only: other launches run the x3 form whatever is asked
  const char* x2_env = getenv("GFN_RENDER_X2");
  const int x2 = x2_env ? atoi(x2_env) : ((opts && opts->colour_x2) ? X2_H1 : 0);
#define GFN_X2_MASKS(X) X(1) X(2) X(4) X(8) X(16) X(24) X(28) X(31)
  if (x2 != 0 && !bias_pool && table2_f16 && prim_pool2 && A.rec && fold && levels == 2) {
    bool known = false;
#define GFN_X2_KNOWN(M) known = known || x2 == M;
    GFN_X2_MASKS(GFN_X2_KNOWN)
#undef GFN_X2_KNOWN
    GFN_CHECK_ARG(known, "%s: GFN_RENDER_X2=%d is not one of the instantiated layer masks (1 2 4 8 16 24 28 31)", who, x2);
    static unsigned long long x2_seen = 0;
    if (gfn_first_call_on_device(x2_seen)) {
#define GFN_X2_SET(M) \
      GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_render_patch_kernel<1, 2, true, true, M, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
      GFN_X2_MASKS(GFN_X2_SET)
#undef GFN_X2_SET
    }
    fused_init_kernel<<<1, 1, 0, st>>>(work_counter, t_minmax);
    GFN_CHECK_LAUNCH();
    int64_t xb = (n_rays / 32 + RENDER_THREADS / 64 - 1) / (RENDER_THREADS / 64);
    if (xb > gfn_decoder_cu_cap(opts)) xb = gfn_decoder_cu_cap(opts);
#define GFN_X2_LAUNCH(M) \
    if (x2 == M) fused_render_patch_kernel<1, 2, true, true, M, true><<<(unsigned)xb, RENDER_THREADS, FUSED_LDS_BYTES, st>>>(A, sc, tile_width, patch_width);
    GFN_X2_MASKS(GFN_X2_LAUNCH)
#undef GFN_X2_LAUNCH
    GFN_CHECK_LAUNCH();
    return GFN_OK;
  }
#undef GFN_X2_MASKS
  fused_init_kernel<<<1, 1, 0, st>>>(work_counter, t_minmax);
  GFN_CHECK_LAUNCH();
  int64_t blocks = (n_rays / 32 + RENDER_THREADS / 64 - 1) / (RENDER_THREADS / 64);  // one patch at a time per wave
  const int cu_cap = gfn_decoder_cu_cap(opts);
  if (blocks > cu_cap) blocks = cu_cap;
  const int tab = table2_f16 ? 1 : 0;
  const bool rec = A.rec != nullptr;
  const int threads = (exp_bits & 4) ? RENDER_THREADS / 2 : RENDER_THREADS;  // experiment (GFN_RENDER_EXP bit 2): ONE wave per SIMD -- a wave's tile time without a partner
  // no bias pool and (two tables) the second table's own primes, the record form, the fold, two levels in flight: the instantiation with those facts compiled in
  // (two tables only: the one-table form gains nothing to speak of and spills a register that way; GFN_RENDER_EXP bit 3: the generic form, for A/B)
  const bool plain = !bias_pool && table2_f16 && prim_pool2 && rec && fold && levels == 2 && !(exp_bits & 8);
  if (plain) {
    static unsigned long long plain_seen = 0;
    if (gfn_first_call_on_device(plain_seen)) {
      GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_render_patch_kernel<1, 2, true, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    }
    fused_render_patch_kernel<1, 2, true, true, 0, true><<<(unsigned)blocks, threads, FUSED_LDS_BYTES, st>>>(A, sc, tile_width, patch_width);
    GFN_CHECK_LAUNCH();
    return GFN_OK;
  }
#define GFN_LAUNCH(T, L, R, F) \
  if (tab == T && levels == L && rec == R && fold == F) fused_render_patch_kernel<T, L, R, F><<<(unsigned)blocks, threads, FUSED_LDS_BYTES, st>>>(A, sc, tile_width, patch_width);
  GFN_RENDER_ALL(GFN_LAUNCH)
#undef GFN_LAUNCH
#undef GFN_RENDER_ALL
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

extern "C" {

int gfn_fused_render(const int32_t* pts_offsets, int64_t n_rays, const float* pts, const int32_t* anchors, const float* dt,
                     const float* ts, const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool,
                     const int32_t* prim_pool2, const float* bias_pool, int64_t n_volumes, int log2_T, const void* fwd_image,
                     const float* sh, const float* emb, const int32_t* cam_idx, int cam_uniform, float early_T, int tile_width, int patch_width,
                     float* out_rgb, float* out_acc, float* out_depth, float* t_minmax, int32_t* work_counter, const gfn_launch_opts* opts,
                     void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(pts_offsets && pts && anchors && dt && ts, "gfn_fused_render: null sample pointer");
  FusedArgs A{};
  A.pts_offsets = pts_offsets; A.pts = pts; A.anchors = anchors; A.dt = dt; A.ts = ts;
  return fused_render_launch(A, "gfn_fused_render", n_rays, table_f16, table2_f16, prim_pool, prim_pool2, bias_pool, n_volumes, log2_T, fwd_image, sh,
                             emb, cam_idx, cam_uniform, early_T, tile_width, patch_width, out_rgb, out_acc, out_depth, t_minmax, work_counter, opts, stream);
}

int gfn_fused_render_rec(const int32_t* counts, int64_t n_rays, const float* march_rec, const int32_t* page_table, const uint16_t* table_f16, const uint16_t* table2_f16,
                         const int32_t* prim_pool, const int32_t* prim_pool2, const float* bias_pool, int64_t n_volumes, int log2_T,
                         const void* fwd_image, const float* sh, const float* emb, const int32_t* cam_idx, int cam_uniform, float early_T, int tile_width,
                         int patch_width, float* out_rgb, float* out_acc, float* out_depth, float* t_minmax, int32_t* work_counter,
                         const gfn_launch_opts* opts, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(counts && march_rec && (reinterpret_cast<uintptr_t>(march_rec) & 15) == 0, "gfn_fused_render_rec: counts / march_rec null or not 16-byte aligned");
  FusedArgs A{};
  A.counts = counts; A.rec = march_rec; A.page_table = page_table;
  return fused_render_launch(A, "gfn_fused_render_rec", n_rays, table_f16, table2_f16, prim_pool, prim_pool2, bias_pool, n_volumes, log2_T, fwd_image,
                             sh, emb, cam_idx, cam_uniform, early_T, tile_width, patch_width, out_rgb, out_acc, out_depth, t_minmax, work_counter, opts, stream);
}

}  // extern "C"
