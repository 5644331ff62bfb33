// batch.hip -- batch assembly on the device (the half upstream of ray generation):
//   PixelSampler.sample_method / collate_image_dataset_batch  nerfstudio/data/pixel_samplers.py:47-117
//   ErrorPixelSampler.sample_method / weighted_choice_multinomial  pixel_samplers.py:636-723
//   CacheDataloader._update_error_map  nerfstudio/data/utils/dataloaders.py:140-142, fed by gfnerf/gf_pipeline.py:180-185
//
// The reference keeps the cached images on the host, draws pixel indices with Python's `random` / torch.multinomial
// (16 M exponentials per 2^24-pixel chunk, every step) and gathers on the CPU.  Here the images, the error map and a
// two-level CDF of it (f64 partial sums per 4096 pixels, kept current by the update kernel) live in HBM; a weighted
// draw is a binary search over the partial sums + one wave scanning 4096 weights, and "without replacement" is
// duplicate rejection in an LDS hash set (sequential sampling without replacement == redraw-on-duplicate).
// Random BITS are an input, so the result is a pure function of its arguments (oracle/batch.py restates it).
#include "gfn_common.h"

namespace {

constexpr int64_t kBlk = GFN_ERRMAP_BLOCK;
constexpr int64_t kChunk = GFN_PIXEL_CHUNK;
constexpr int kAttempts = GFN_PIXEL_ATTEMPTS;

// ------------------------------------------------------------------------------------------------ uniform sampler
// indices = floor(rand(B,3) * tensor([n, H, W])).long()  (:63-67): the product is an f32 product (float * int64 tensor
// promotes to float).  The reference indexes out of range if it rounds up to the extent; clamped here.
__global__ void pixel_uniform_kernel(const float* __restrict__ u, int64_t B, int n_images, int H, int W, int64_t* __restrict__ lin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  int c = (int)floorf(u[3 * i + 0] * (float)n_images);
  int y = (int)floorf(u[3 * i + 1] * (float)H);
  int x = (int)floorf(u[3 * i + 2] * (float)W);
  c = min(max(c, 0), n_images - 1);
  y = min(max(y, 0), H - 1);
  x = min(max(x, 0), W - 1);
  lin[i] = ((int64_t)c * H + y) * W + x;
}

// collate_image_dataset_batch (:73-103): value[c, y, x] for the image, camera-index fix-up, and the pixel-centre
// coordinates RayGenerator reads from image_coords (ray_generators.py:52, cameras.py get_image_coords: +0.5).
template <bool U8>
__global__ void pixel_collate_kernel(const int64_t* __restrict__ lin, int64_t B, int n_images, int H, int W, const void* __restrict__ images,
                                     const int32_t* __restrict__ image_idx, const int32_t* __restrict__ rel_cam_idx,
                                     int32_t* __restrict__ indices, int32_t* __restrict__ slot, int32_t* __restrict__ rel_cam,
                                     float* __restrict__ gt, float* __restrict__ coords) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  int64_t p = lin[i];
  const int64_t n_pix = (int64_t)n_images * H * W;
  p = p < 0 ? 0 : (p >= n_pix ? n_pix - 1 : p);
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const int c = (int)(p / ((int64_t)W * H));
  if (indices) {
    indices[3 * i + 0] = image_idx ? image_idx[c] : c;
    indices[3 * i + 1] = y;
    indices[3 * i + 2] = x;
  }
  if (slot) slot[i] = c;
  if (rel_cam) rel_cam[i] = rel_cam_idx ? rel_cam_idx[c] : c;
  if (gt && images) {
    if (U8) {  // uint8 image / 255.0 in f32 (data_utils / base_dataset.get_image :95-115)
      const uint8_t* q = (const uint8_t*)images + 3 * p;
      for (int k = 0; k < 3; k++) gt[3 * i + k] = (float)q[k] / 255.0f;
    } else {
      const float* q = (const float*)images + 3 * p;
      for (int k = 0; k < 3; k++) gt[3 * i + k] = q[k];
    }
  }
  if (coords) {
    coords[2 * i + 0] = (float)y + 0.5f;
    coords[2 * i + 1] = (float)x + 0.5f;
  }
}

// ------------------------------------------------------------------------------------------------ error-map CDF
// f64 sum of one 4096-pixel block, in a fixed order: lane l adds elements l, l+256, ... then a halving tree.
__global__ __launch_bounds__(256) void errmap_block_sums_kernel(const float* __restrict__ w, int64_t n, double* __restrict__ sums) {
  __shared__ double red[256];
  const int64_t base = (int64_t)blockIdx.x * kBlk;
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < kBlk; k += 256) {
    const int64_t p = base + k;
    if (p < n) acc += (double)w[p];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

// error = sum_c |gt - pred| (gf_pipeline.py:181-184); error_map[c, y, x] = error (dataloaders.py:142).  The old value
// leaves by atomic exchange so that duplicate indices still telescope into the block sum.
__global__ void errmap_update_kernel(const int32_t* __restrict__ idx, int64_t B, const float* __restrict__ pred, const float* __restrict__ gt,
                                     int n_images, int H, int W, float* __restrict__ error_map, double* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int c = idx[3 * i], y = idx[3 * i + 1], x = idx[3 * i + 2];
  if (c < 0 || c >= n_images || y < 0 || y >= H || x < 0 || x >= W) return;
  const float e = (fabsf(gt[3 * i] - pred[3 * i]) + fabsf(gt[3 * i + 1] - pred[3 * i + 1])) + fabsf(gt[3 * i + 2] - pred[3 * i + 2]);
  const int64_t p = ((int64_t)c * H + y) * W + x;
  const float old = atomicExch(error_map + p, e);
  if (sums) atomicAdd(sums + p / kBlk, (double)e - (double)old);
}

// exclusive prefix of the block sums, one workgroup (n_blocks <= a few 1e5): prefix[0] = 0, prefix[b+1] = prefix[b] + sums[b].
// Thread t owns a contiguous segment; the 1024 segment totals are scanned with wave shuffles (a serial loop over them in
// one thread cost 40 us).  f64 throughout: exact, hence order independent, whenever the weights are dyadic (the parity tests).
__device__ __forceinline__ double shfl_up_f64(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_up(lo, off, 64); hi = __shfl_up(hi, off, 64);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(1024) void errmap_prefix_kernel(const double* __restrict__ sums, int64_t nb, double* __restrict__ prefix) {
  __shared__ double wave_tot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t a = min((int64_t)t * per, nb), b = min(a + per, nb);
  double acc = 0.0;
  for (int64_t k = a; k < b; k++) acc += sums[k];
  double incl = acc;  // inclusive scan over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double u = shfl_up_f64(incl, off);
    if (lane >= off) incl += u;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wave; w++) base += wave_tot[w];
  double run = base + (incl - acc);  // exclusive prefix of this thread's segment
  if (t == 0) prefix[0] = 0.0;
  for (int64_t k = a; k < b; k++) { run += sums[k]; prefix[k + 1] = run; }
}

// Range of pixels draw d of the weighted group samples from, following weighted_choice_multinomial (:636-674) with its
// quirks: size // num_chunks draws per FULL 2^24 chunk; the size % num_chunks left-over draws come from the tail
// beyond the last full chunk when n % num_chunks != 0 (sic), else uniformly from all pixels; a tail never sees a
// weighted draw when size divides evenly.
struct DrawRange { int64_t lo, hi; bool weighted; };
__host__ __device__ inline DrawRange draw_range(int64_t d, int64_t n_weighted, int64_t n) {
  if (d >= n_weighted) return {0, n, false};  // step 2: random.sample(range(n_pixels), k) (:705-706)
  const int64_t nc = n / kChunk;
  if (nc == 0) return {0, n, true};
  const int64_t per = n_weighted / nc;
  if (d < per * nc) { const int64_t c = d / per; return {c * kChunk, (c + 1) * kChunk, true}; }
  if (n % nc != 0) return {nc * kChunk, n, true};
  return {0, n, false};
}

__device__ __forceinline__ int64_t uniform_below(uint64_t bits, int64_t range) { return (int64_t)__umul64hi(bits, (uint64_t)range); }

// Attempt a of draw d uses the 64 bits splitmix64(bits[d] + a * golden): any number of attempts from one input word.
__host__ __device__ inline uint64_t attempt_bits(uint64_t seed, int a) {
  uint64_t z = seed + (uint64_t)(a + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Candidate pixel of one draw, computed by a whole wave (same value on every lane).  Weighted: target = P[lo] + u * (P[hi] -
// P[lo]) in f64, u = top 53 bits; block = last b with P[b] <= target; pixel = first i of the block whose running sum
// exceeds target - P[b].  *zero_total: the weighted range carries no weight (torch.multinomial raises; uniform here).
__device__ int64_t draw_candidate(const float* __restrict__ w, int64_t n, const double* __restrict__ prefix, uint64_t r64, int64_t d,
                                  int64_t n_weighted, int lane, bool* zero_total) {
  const DrawRange rg = draw_range(d, n_weighted, n);
  const int64_t b0 = rg.lo / kBlk, b1 = (rg.hi + kBlk - 1) / kBlk;
  const double p0 = prefix[b0], total = prefix[b1] - p0;
  *zero_total = rg.weighted && !(total > 0.0);
  if (!rg.weighted || !(total > 0.0)) return rg.lo + uniform_below(r64, rg.hi - rg.lo);
  const double u = (double)(r64 >> 11) * 0x1p-53;
  const double target = p0 + u * total;
  int64_t lo = b0, hi = b1;  // invariant: prefix[lo] <= target
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (prefix[mid] <= target) lo = mid; else hi = mid;
  }
  while (lo > b0 && !(prefix[lo + 1] > prefix[lo])) lo--;  // rounding put the target on an empty block
  while (lo + 1 < b1 && !(prefix[lo + 1] > prefix[lo])) lo++;
  const double r = target - prefix[lo];
  const int64_t base = lo * kBlk;
  // lane l owns elements [64 l, 64 l + 64); two passes over them (the second hits L1) instead of 64 live registers
  const int64_t e0 = base + 64 * lane;
  auto elem = [&](int k) -> float {
    const int64_t p = e0 + k;
    return (p < rg.hi && p >= rg.lo) ? w[p] : 0.f;
  };
  double s = 0.0;
#pragma unroll 4
  for (int k = 0; k < 64; k++) s += (double)elem(k);
  double excl = 0.0;  // exclusive scan over lanes, in lane order
  for (int l = 0; l < 63; l++) {
    const double sl = __shfl(s, l);
    if (lane > l) excl += sl;
  }
  int pick = -1, last_pos = -1;  // first element whose inclusive running sum exceeds r; else the last positive one
  double run = excl;
#pragma unroll 4
  for (int k = 0; k < 64; k++) {
    const float v = elem(k);
    run += (double)v;
    if (v > 0.f) {
      last_pos = 64 * lane + k;
      if (pick < 0 && run > r) pick = 64 * lane + k;
    }
  }
  int best = pick < 0 ? 1 << 30 : pick, lastp = last_pos;
  for (int o = 32; o > 0; o >>= 1) {
    best = min(best, __shfl_xor(best, o));
    lastp = max(lastp, __shfl_xor(lastp, o));
  }
  return base + (best < (1 << 30) ? best : max(lastp, 0));
}

// attempt 0 of every draw, one wave each
__global__ __launch_bounds__(256) void errmap_draw_kernel(const float* __restrict__ w, int64_t n, const double* __restrict__ prefix,
                                                          const int64_t* __restrict__ bits, int64_t B, int64_t n_weighted,
                                                          int64_t* __restrict__ cand, int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= B) return;
  bool zero;
  const int64_t c = draw_candidate(w, n, prefix, attempt_bits((uint64_t)bits[d], 0), d, n_weighted, lane, &zero);
  if (lane == 0) {
    cand[d] = c;
    if (zero) atomicAdd(status, 1);
  }
}

// Without replacement: draw d keeps its attempt-a candidate unless a lower draw of the same attempt, or any draw of an
// earlier attempt, holds the same pixel; losers draw again (attempt a + 1) -- sequential sampling without replacement.
// The weighted group and the uniform group are independent samples (the reference concatenates them, :709).  One
// workgroup; the set lives in LDS when it fits, else in the workspace.  After kAttempts a duplicate is kept and counted.
struct Slot { unsigned long long key; unsigned int owner; unsigned int pad; };
constexpr unsigned long long kEmpty = ~0ull;

__global__ __launch_bounds__(1024) void errmap_dedupe_kernel(const float* __restrict__ w, int64_t n, const double* __restrict__ prefix,
                                                             const int64_t* __restrict__ bits, int64_t* __restrict__ cand, int64_t B,
                                                             int64_t n_weighted, Slot* __restrict__ g_table, int64_t slots, int use_lds,
                                                             int32_t* __restrict__ pending_ws, int64_t* __restrict__ lin,
                                                             int32_t* __restrict__ status) {
  extern __shared__ Slot l_table[];
  __shared__ int n_pending;
  Slot* table = use_lds ? l_table : g_table;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int group = 0; group < 2; group++) {
    const int64_t g0 = group == 0 ? 0 : n_weighted, g1 = group == 0 ? n_weighted : B;
    if (g1 <= g0) continue;
    int64_t gs = 1024;  // this group's share of the table: a power of two >= twice its draws
    while (gs < 2 * (g1 - g0) && gs < slots) gs <<= 1;
    const unsigned long long mask = (unsigned long long)gs - 1;
    for (int64_t k = t; k < gs; k += 1024) { table[k].key = kEmpty; table[k].owner = 0xffffffffu; }
    for (int64_t d = g0 + t; d < g1; d += 1024) lin[d] = -1;  // -1 = pending; each draw belongs to one thread throughout
    __syncthreads();
    for (int a = 0; a < kAttempts; a++) {
      for (int64_t d = g0 + t; d < g1; d += 1024) {
        if (lin[d] >= 0) continue;
        const unsigned long long key = (unsigned long long)__hip_atomic_load(cand + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> 20;
        for (;;) {
          Slot* s = table + (h & mask);
          const unsigned long long prev = atomicCAS(&s->key, kEmpty, key);
          if (prev == kEmpty || prev == key) { atomicMin(&s->owner, ((unsigned)a << 24) | (unsigned)(d - g0)); break; }
          h++;
        }
      }
      if (t == 0) n_pending = 0;
      __syncthreads();
      for (int64_t d = g0 + t; d < g1; d += 1024) {
        if (lin[d] >= 0) continue;
        const unsigned long long key = (unsigned long long)__hip_atomic_load(cand + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> 20;
        for (;;) {
          Slot* s = table + (h & mask);  // agent-scope loads: the set was written with atomics, which bypass this CU's L1
          if (__hip_atomic_load(&s->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key) {
            if (__hip_atomic_load(&s->owner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (((unsigned)a << 24) | (unsigned)(d - g0)))
              lin[d] = (int64_t)key;
            else
              pending_ws[atomicAdd(&n_pending, 1)] = (int)(d - g0);
            break;
          }
          h++;
        }
      }
      __syncthreads();
      const int np = n_pending;
      if (np == 0) break;  // uniform: every thread reads the same count after the barrier
      if (a + 1 < kAttempts) {
        for (int j = wave; j < np; j += 16) {  // a wave per pending draw: its next candidate
          const int64_t d = g0 + __hip_atomic_load(pending_ws + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool zero;
          const int64_t c = draw_candidate(w, n, prefix, attempt_bits((uint64_t)bits[d], a + 1), d, n_weighted, lane, &zero);
          if (lane == 0) __hip_atomic_store(cand + d, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
    }
    for (int64_t d = g0 + t; d < g1; d += 1024)
      if (lin[d] < 0) {  // still a duplicate after every attempt: keep it and say so
        lin[d] = __hip_atomic_load(cand + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(status + 1, 1);
      }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int gfn_pixel_uniform(const float* u, int64_t n, int32_t n_images, int32_t height, int32_t width, int64_t* lin, void* stream) {
  GFN_CHECK_ARG(n >= 0 && n_images > 0 && height > 0 && width > 0, "gfn_pixel_uniform: bad extents");
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(u && lin, "gfn_pixel_uniform: null pointer");
  pixel_uniform_kernel<<<(unsigned)((n + 255) / 256), 256, 0, gfn_stream(stream)>>>(u, n, n_images, height, width, lin);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_pixel_collate(const int64_t* lin, int64_t n, int32_t n_images, int32_t height, int32_t width, const void* images, int32_t images_u8,
                      const int32_t* image_idx, const int32_t* rel_cam_idx, int32_t* indices, int32_t* slot, int32_t* rel_cam, float* gt,
                      float* coords, void* stream) {
  GFN_CHECK_ARG(n >= 0 && n_images > 0 && height > 0 && width > 0, "gfn_pixel_collate: bad extents");
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(lin, "gfn_pixel_collate: null pointer");
  GFN_CHECK_ARG(!gt || images, "gfn_pixel_collate: gt asked for without images");
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (images_u8)
    pixel_collate_kernel<true><<<grid, 256, 0, gfn_stream(stream)>>>(lin, n, n_images, height, width, images, image_idx, rel_cam_idx, indices, slot,
                                                                     rel_cam, gt, coords);
  else
    pixel_collate_kernel<false><<<grid, 256, 0, gfn_stream(stream)>>>(lin, n, n_images, height, width, images, image_idx, rel_cam_idx, indices, slot,
                                                                      rel_cam, gt, coords);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_errmap_block_sums(const float* error_map, int64_t n_pixels, double* block_sums, void* stream) {
  GFN_CHECK_ARG(n_pixels >= 0, "gfn_errmap_block_sums: bad size");
  if (n_pixels == 0) return GFN_OK;
  GFN_CHECK_ARG(error_map && block_sums, "gfn_errmap_block_sums: null pointer");
  const int64_t nb = (n_pixels + kBlk - 1) / kBlk;
  GFN_CHECK_ARG(nb < (1ll << 31), "gfn_errmap_block_sums: too many pixels");
  errmap_block_sums_kernel<<<(unsigned)nb, 256, 0, gfn_stream(stream)>>>(error_map, n_pixels, block_sums);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_errmap_update(const int32_t* indices, int64_t n, const float* pred, const float* gt, int32_t n_images, int32_t height, int32_t width,
                      float* error_map, double* block_sums, void* stream) {
  GFN_CHECK_ARG(n >= 0 && n_images > 0 && height > 0 && width > 0, "gfn_errmap_update: bad extents");
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(indices && pred && gt && error_map, "gfn_errmap_update: null pointer");
  errmap_update_kernel<<<(unsigned)((n + 255) / 256), 256, 0, gfn_stream(stream)>>>(indices, n, pred, gt, n_images, height, width, error_map, block_sums);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int64_t gfn_errmap_table_slots(int64_t n) {
  int64_t s = 1024;
  while (s < 2 * n) s <<= 1;
  return s;
}

int gfn_errmap_sample(const float* error_map, int64_t n_pixels, const double* block_sums, const int64_t* bits, int64_t n, int64_t n_weighted,
                      double* prefix_ws, int64_t* cand_ws, int32_t* pending_ws, void* table_ws, int64_t* lin, int32_t* status, void* stream) {
  GFN_CHECK_ARG(n >= 0 && n_weighted >= 0 && n_weighted <= n && n_pixels > 0, "gfn_errmap_sample: bad sizes");
  GFN_CHECK_ARG(n < (1 << 24), "gfn_errmap_sample: at most 2^24 - 1 draws");
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(error_map && block_sums && bits && prefix_ws && cand_ws && pending_ws && lin && status, "gfn_errmap_sample: null pointer");
  GFN_CHECK_ARG(n <= n_pixels, "gfn_errmap_sample: %lld draws without replacement from %lld pixels", (long long)n, (long long)n_pixels);
  hipStream_t st = gfn_stream(stream);
  const int64_t nb = (n_pixels + kBlk - 1) / kBlk;
  GFN_CHECK_HIP(gfn_zero_async(status, 2 * sizeof(int32_t), st));
  errmap_prefix_kernel<<<1, 1024, 0, st>>>(block_sums, nb, prefix_ws);
  GFN_CHECK_LAUNCH();
  errmap_draw_kernel<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(error_map, n_pixels, prefix_ws, bits, n, n_weighted, cand_ws, status);
  GFN_CHECK_LAUNCH();
  const int64_t group = n_weighted > n - n_weighted ? n_weighted : n - n_weighted;
  const int64_t slots = gfn_errmap_table_slots(group);
  const int use_lds = slots * (int64_t)sizeof(Slot) <= 128 * 1024;
  GFN_CHECK_ARG(use_lds || table_ws, "gfn_errmap_sample: %lld draws need the table workspace", (long long)n);
  const size_t lds = use_lds ? (size_t)slots * sizeof(Slot) : 0;
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen)) {
    GFN_CHECK_HIP(hipFuncSetAttribute((const void*)errmap_dedupe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  }
  errmap_dedupe_kernel<<<1, 1024, lds, st>>>(error_map, n_pixels, prefix_ws, bits, cand_ws, n, n_weighted, (Slot*)table_ws, slots, use_lds, pending_ws,
                                            lin, status);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
