// lm_lookup.hip -- Hash3DAnchored lookup (Hash3DAnchored_cuda.cu:11-79) in LEVEL-MAJOR order.
//
// The direct lookup walks all 16 levels of one sample before the next sample, so the whole chip's working set is every
// level of every table at once (142 MB of f16 slots): each corner of a fine level is an L2 miss, and the memory fabric
// serves only ~57 G such 64-byte requests per second whatever the instruction form (tools/gather_bench.hip).  The same
// benchmark gathers 262 G/s from a table that fits the 4 MB L2 of an XCD and 121 G/s from an 8 MB one.  So here every
// workgroup owns a contiguous run of samples (thread t keeps samples t, t + 1024, ... in registers for the whole
// kernel) and ALL workgroups sweep the table together, one launch per table:
//     for level: for address part h (a half = 4 MB of f16 pairs at T = 2^21): every sample's corners that fall into it
// Corners of the earlier parts wait in LDS; the last part completes the level: blend (same order as the direct kernel,
// so the features are bit-identical), round to f16, store (the second table's launch adds to the first's result).
// Level l's upper half and level l+1's lower half are the same addresses (the reference's level-overlap quirk), so
// half of the passes find their part warm (with every part warm the kernel takes 0.346 instead of 0.399 ms).  What was tried
// on the pass structure (tools/README.md, round 2): quarters instead of halves, with and without pulling the next cold quarter
// into the L2 by one load per line during the pass before it (no gain in the step, see gfn_hash_fwd_lm below), two 512-thread
// workgroups per CU (0.47 ms: their passes drift apart), the workgroup barrier after the gathers are issued instead of after
// they are consumed (0.42), two samples' gathers in flight per thread (0.44).  Coarse levels (< halves_from) are not split; there the first lane of a run of
// samples in one cell gathers for the whole run.
#include "gfn_common.h"
#include "hash_device.h"

namespace {

constexpr int kLmThreads = 1024;

struct LmArgs {
  int64_t n_max; const int32_t* n_dev;
  const float* pts; const int* anchors; int anchor_stride;
  const void* table; const int* prim; const float* bias;
  int64_t n_volumes; int log2_T; int halves_from;  // levels >= halves_from are gathered one address part at a time ...
  int split_log2;                                   // ... of 2^split_log2 parts per level (1: halves, 2: quarters)
  int cap;  // samples of one workgroup per round (LDS capacity)
  float* out;  // f32 [n,32] row-major, or level-major float2 [16][n_max]
  // phased (termination-aware) forward: look up only the samples named by list[0 .. *list_n) (ascending sample indices; the other
  // samples' feature slots are left as they are); the whole launch does nothing when *list_n == 0 -- decided on the device
  const int32_t* list; const int32_t* list_n;
  LevelScales sc;
};

// a gather as (uniform 64-bit base in SGPRs) + (32-bit byte offset per lane): one address register per corner instead of two
__device__ __forceinline__ half2_t ld_pair(const half2_t* __restrict__ base, uint32_t pos) {
  return *reinterpret_cast<const half2_t*>(reinterpret_cast<const char*>(base) + (size_t)(uint32_t)(pos << 2));
}

constexpr int kLmK = 4;  // samples per thread: a workgroup run holds up to 4096 samples

// Thread t of a workgroup owns samples t, t + 1024, ... of the run for the whole kernel: their points stay in
// registers; LDS only parks the lower-half corner values (32 B per sample) between the two halves of a level.
// ACC: out += features (the second table of the focal stage; one table per launch keeps level l's upper half warm for
// level l + 1's lower half -- alternating tables inside a level measured 25 % slower).
template <bool PLANAR, bool ACC>
__global__ __launch_bounds__(kLmThreads) void lm_lookup_kernel(LmArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_val[];  // [8][cap]
  __shared__ float s_scale[GFN_N_LEVELS];
  const int tid = threadIdx.x;
  if (tid < GFN_N_LEVELS) s_scale[tid] = A.sc.s[tid];
  // n = entries to look up: the live samples, or the entries of the list (every workgroup then owns a run of the LIST: the same
  // number of wanted samples each, every wave of every pass full)
  const int32_t* list = A.list;  // (a local: writing to the argument struct would move all of it out of the kernel-argument segment)
  int64_t n = list ? (int64_t)*A.list_n : gfn_live_n(A.n_max, A.n_dev);
  if (list && n < 0) { list = nullptr; n = gfn_live_n(A.n_max, A.n_dev); }  // "every sample": no list was written
  if (list && n > A.n_max) n = A.n_max;
  const int64_t per_round = (int64_t)A.cap * gridDim.x;
  const uint32_t mask = (1u << A.log2_T) - 1u;
  for (int64_t r0 = 0; r0 < n; r0 += per_round) {
    // this round's samples, dealt in equal contiguous runs so that every workgroup does the same work per pass
    const int64_t left = n - r0 < per_round ? n - r0 : per_round;
    const int64_t per = (left + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = r0 + (int64_t)blockIdx.x * per;
    int64_t cnt64 = r0 + left - b0;
    const int cnt = (int)(cnt64 < 0 ? 0 : (cnt64 > per ? per : cnt64));
    float px[kLmK], py[kLmK], pz[kLmK];
    int vol[kLmK];
    int smp[kLmK];  // the sample in the slot
#pragma unroll
    for (int k = 0; k < kLmK; k++) {
      const int s = tid + k * kLmThreads;
      const int64_t e = s < cnt ? b0 + s : r0;  // idle slots read a valid entry (r0 < n): b0 itself may lie beyond the buffers
      const int64_t i = list ? (int64_t)list[e] : e;
      smp[k] = (int)i;
      px[k] = A.pts[3 * i]; py[k] = A.pts[3 * i + 1]; pz[k] = A.pts[3 * i + 2];
      vol[k] = A.anchors[i * A.anchor_stride];
    }
    __syncthreads();
    for (int level = 0; level < GFN_N_LEVELS; level++) {
      const bool split = level >= A.halves_from;
      const float scale = s_scale[level];
      {
        const half2_t* base = reinterpret_cast<const half2_t*>(reinterpret_cast<const _Float16*>(A.table) + (int64_t)level * (1ll << A.log2_T));
        int prim[kLmK][3];  // this level's primes of the thread's samples: one round trip, shared by both halves
#pragma unroll
        for (int k = 0; k < kLmK; k++) {
          const int* p = A.prim + 3 * ((int64_t)level * A.n_volumes + vol[k]);
          prim[k][0] = p[0]; prim[k][1] = p[1]; prim[k][2] = p[2];
        }
        if (!split) {
          // coarse levels: consecutive samples of a ray sit in the same cell of the same volume and need the same 8 slots.
          // Only the first lane of such a run gathers; the others take its values through the LDS crossbar (ds_bpermute),
          // which costs the address path nothing -- and the address path is what bounds this kernel.
          const int lane = tid & 63;
#pragma unroll
          for (int k = 0; k < kLmK; k++) {
            const int s = tid + k * kLmThreads;
            const bool valid = s < cnt;
            Corners c;
            uint32_t cell[3];
            const int64_t tr = (int64_t)level * A.n_volumes + vol[k];
            hash_corners(px[k], py[k], pz[k], scale, prim[k], A.bias ? A.bias + 3 * tr : nullptr, mask, c, cell);
            const int pvol = __shfl_up(vol[k], 1, 64);
            const uint32_t pcx = __shfl_up(cell[0], 1, 64), pcy = __shfl_up(cell[1], 1, 64), pcz = __shfl_up(cell[2], 1, 64);
            const bool head = lane == 0 || pvol != vol[k] || pcx != cell[0] || pcy != cell[1] || pcz != cell[2];
            const unsigned long long heads = __ballot(head);
            half2_t v[8];
            if (__popcll(heads) <= 40) {  // wave-uniform: enough lanes saved to pay for the shuffles
              if (head) {
#pragma unroll
                for (int d = 0; d < 8; d++) v[d] = ld_pair(base, c.pos[d]);
              }
              const int src = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));  // the run's first lane
#pragma unroll
              for (int d = 0; d < 8; d++) v[d] = __builtin_bit_cast(half2_t, __shfl(__builtin_bit_cast(int, v[d]), src, 64));
            } else {
#pragma unroll
              for (int d = 0; d < 8; d++) v[d] = ld_pair(base, c.pos[d]);
            }
            if (valid) {
              float2* dst = PLANAR ? reinterpret_cast<float2*>(A.out) + (int64_t)level * A.n_max + smp[k]
                                   : reinterpret_cast<float2*>(A.out + (int64_t)smp[k] * 32 + 2 * level);
              float a0 = 0.f, a1 = 0.f;
#pragma unroll
              for (int d = 0; d < 8; d++) {
                const float t0 = c.w[d] * (float)v[d][0], t1 = c.w[d] * (float)v[d][1];  // :74-77, left-to-right sum (as gather_level)
                a0 = (d == 0) ? t0 : a0 + t0;
                a1 = (d == 0) ? t1 : a1 + t1;
              }
              const float2 rr = make_float2(gfn_round_f16(a0), gfn_round_f16(a1));
              if (ACC) { const float2 prev = *dst; *dst = make_float2(prev.x + rr.x, prev.y + rr.y); }
              else *dst = rr;
            }
          }
          __syncthreads();
          continue;
        }
        const int pshift = A.log2_T - A.split_log2;  // a slot's part = its top split_log2 bits
        const int n_parts = 1 << A.split_log2;
        for (int h = 0; h < n_parts; h++) {
          const bool last = h == n_parts - 1;
#pragma unroll
          for (int k = 0; k < kLmK; k++) {
            const int s = tid + k * kLmThreads;
            if (s < cnt) {
              Corners c;
              uint32_t cell[3];
              const int64_t tr = (int64_t)level * A.n_volumes + vol[k];
              hash_corners(px[k], py[k], pz[k], scale, prim[k], A.bias ? A.bias + 3 * tr : nullptr, mask, c, cell);
              if (!last) {  // an earlier part of the slot range: fetch and park
#pragma unroll
                for (int d = 0; d < 8; d++)
                  if ((int)(c.pos[d] >> pshift) == h) s_val[d * A.cap + s] = __builtin_bit_cast(uint32_t, ld_pair(base, c.pos[d]));
              } else {
                float2* dst = PLANAR ? reinterpret_cast<float2*>(A.out) + (int64_t)level * A.n_max + smp[k]
                                     : reinterpret_cast<float2*>(A.out + (int64_t)smp[k] * 32 + 2 * level);
                float2 prev = make_float2(0.f, 0.f);
                if (ACC) prev = *dst;
                half2_t v[8];
#pragma unroll
                for (int d = 0; d < 8; d++) {
                  if ((int)(c.pos[d] >> pshift) == h) v[d] = ld_pair(base, c.pos[d]);
                  else v[d] = __builtin_bit_cast(half2_t, s_val[d * A.cap + s]);
                }
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int d = 0; d < 8; d++) {
                  const float t0 = c.w[d] * (float)v[d][0], t1 = c.w[d] * (float)v[d][1];  // :74-77, left-to-right sum (as gather_level)
                  a0 = (d == 0) ? t0 : a0 + t0;
                  a1 = (d == 0) ? t1 : a1 + t1;
                }
                const float2 rr = make_float2(gfn_round_f16(a0), gfn_round_f16(a1));
                *dst = ACC ? make_float2(prev.x + rr.x, prev.y + rr.y) : rr;  // feat(table) + feat(table2), as the direct kernel adds them
              }
            }
          }
          __syncthreads();  // keep the workgroup's waves in the same pass: they share the part's lines in the XCD's L2
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int gfn_hash_fwd_lm(const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool, const int32_t* prim_pool2,
                    const float* bias_pool, const float* pts, const int32_t* anchors, int anchor_stride, int64_t n_max, const int32_t* n_dev,
                    int64_t n_volumes, int log2_T, int halves_from, int split_log2, int planar, int accumulate, float* out,
                    const int32_t* list, const int32_t* list_n, const gfn_launch_opts* opts, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG((list == nullptr) == (list_n == nullptr), "gfn_hash_fwd_lm: a sample list needs its count");
  GFN_CHECK_ARG(table_f16 && prim_pool && pts && anchors && out, "gfn_hash_fwd_lm: null pointer");
  GFN_CHECK_ARG(n_max > 0 && n_volumes > 0 && log2_T >= 4 && log2_T <= 24 && split_log2 >= 0 && split_log2 <= 3, "gfn_hash_fwd_lm: bad sizes");
  LmArgs A;
  A.n_max = n_max; A.n_dev = n_dev; A.pts = pts; A.anchors = anchors; A.anchor_stride = anchor_stride;
  A.bias = bias_pool; A.n_volumes = n_volumes; A.log2_T = log2_T; A.list = list; A.list_n = list_n;
  A.halves_from = halves_from < 0 ? 0 : halves_from;
  // 0: halves.  Quarters (2 MB parts at T = 2^21) win 4 % when the lookup runs back to back on an otherwise idle chip (0.371
  // vs 0.385 ms: the table then sits in the memory-side cache and the cold parts come from there) and lose 2-5 % inside the
  // training step or behind a 1 GiB stream (0.383 vs 0.376; in the step 0.389 vs 0.370); eighths pay for their passes (0.53)
  A.split_log2 = split_log2 > 0 ? split_log2 : 1;
  A.out = out;
  gfn_hash_level_scales_host(A.sc.s);
  const int grid = gfn_decoder_cu_cap(opts);  // one workgroup per CU
  int64_t cap = (n_max + grid - 1) / grid;
  cap = (cap + 63) & ~63ll;
  if (cap > kLmK * kLmThreads) cap = kLmK * kLmThreads;  // 32 B of LDS and a register slot per sample
  A.cap = (int)cap;
  const size_t lds = (size_t)cap * 32;
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen)) {
    const int max_lds = kLmK * kLmThreads * 32;
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lm_lookup_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lm_lookup_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lm_lookup_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lm_lookup_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
  }
  hipStream_t st = gfn_stream(stream);
  for (int tb = 0; tb < (table2_f16 ? 2 : 1); tb++) {
    A.table = tb ? table2_f16 : table_f16;
    A.prim = (tb && prim_pool2) ? prim_pool2 : prim_pool;
    const bool acc = tb != 0 || accumulate != 0;
    if (planar) {
      if (acc) lm_lookup_kernel<true, true><<<grid, kLmThreads, lds, st>>>(A); else lm_lookup_kernel<true, false><<<grid, kLmThreads, lds, st>>>(A);
    } else {
      if (acc) lm_lookup_kernel<false, true><<<grid, kLmThreads, lds, st>>>(A); else lm_lookup_kernel<false, false><<<grid, kLmThreads, lds, st>>>(A);
    }
    GFN_CHECK_LAUNCH();
  }
  return GFN_OK;
}

}  // extern "C"
