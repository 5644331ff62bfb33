// lm_lookup.hip -- Hash3DAnchored lookup (Hash3DAnchored_cuda.cu:11-79) in LEVEL-MAJOR order.
//
// The direct lookup walks all 16 levels of one sample before the next sample, so the whole chip's working set is every
// level of every table at once (142 MB of f16 slots): each corner of a fine level is an L2 miss, and the memory fabric
// serves only ~57 G such 64-byte requests per second whatever the instruction form (tools/gather_bench.hip).  The same
// benchmark gathers 262 G/s from a table that fits the 4 MB L2 of an XCD and 121 G/s from an 8 MB one.  So here every
// workgroup owns a contiguous run of samples (thread t keeps samples t, t + 1024, ... in registers for the whole
// kernel) and ALL workgroups sweep the table together, one launch per table:
//     for level: for address half h (4 MB of f16 pairs): every sample's corners that fall into that half
// Corners of the first half wait in LDS; the second half completes the level: blend (same order as the direct kernel,
// so the features are bit-identical), round to f16, store (the second table's launch adds to the first's result).
// Level l's upper half and level l+1's lower half are the same addresses (the reference's level-overlap quirk), so
// consecutive passes share a warm L2.  Coarse levels (< halves_from) are not split; there the first lane of a run of
// samples in one cell gathers for the whole run.
#include "gfn_common.h"
#include "hash_device.h"

namespace {

constexpr int kLmThreads = 1024;

struct LmArgs {
  int64_t n_max; const int32_t* n_dev;
  const float* pts; const int* anchors; int anchor_stride;
  const void* table; const int* prim; const float* bias;
  int64_t n_volumes; int log2_T; int halves_from;  // levels >= halves_from are split into address halves
  int cap;  // samples of one workgroup per round (LDS capacity)
  float* out;  // f32 [n,32] row-major, or level-major float2 [16][n_max]
  LevelScales sc;
};

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
  const int64_t n = gfn_live_n(A.n_max, A.n_dev);
  const int64_t per_round = (int64_t)A.cap * gridDim.x;
  const uint32_t mask = (1u << A.log2_T) - 1u;
  const int top = A.log2_T - 1;
  for (int64_t r0 = 0; r0 < n; r0 += per_round) {
    // this round's samples, dealt in equal contiguous runs so that every workgroup does the same work per pass
    const int64_t left = n - r0 < per_round ? n - r0 : per_round;
    const int64_t per = (left + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = r0 + (int64_t)blockIdx.x * per;
    int64_t cnt64 = r0 + left - b0;
    const int cnt = (int)(cnt64 < 0 ? 0 : (cnt64 > per ? per : cnt64));
    float px[kLmK], py[kLmK], pz[kLmK];
    int vol[kLmK];
#pragma unroll
    for (int k = 0; k < kLmK; k++) {
      const int s = tid + k * kLmThreads;
      const int64_t i = s < cnt ? b0 + s : r0;  // idle slots read a valid sample (r0 < n): b0 itself may lie beyond the buffers
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
                for (int d = 0; d < 8; d++) v[d] = base[c.pos[d]];
              }
              const int src = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));  // the run's first lane
#pragma unroll
              for (int d = 0; d < 8; d++) v[d] = __builtin_bit_cast(half2_t, __shfl(__builtin_bit_cast(int, v[d]), src, 64));
            } else {
#pragma unroll
              for (int d = 0; d < 8; d++) v[d] = base[c.pos[d]];
            }
            if (valid) {
              float2* dst = PLANAR ? reinterpret_cast<float2*>(A.out) + (int64_t)level * A.n_max + b0 + s
                                   : reinterpret_cast<float2*>(A.out + (b0 + s) * 32 + 2 * level);
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
        for (int h = 0; h < 2; h++) {
#pragma unroll
          for (int k = 0; k < kLmK; k++) {
            const int s = tid + k * kLmThreads;
            if (s < cnt) {
              Corners c;
              uint32_t cell[3];
              const int64_t tr = (int64_t)level * A.n_volumes + vol[k];
              hash_corners(px[k], py[k], pz[k], scale, prim[k], A.bias ? A.bias + 3 * tr : nullptr, mask, c, cell);
              if (h == 0) {  // lower half of the slot range: fetch and park
#pragma unroll
                for (int d = 0; d < 8; d++)
                  if (!(c.pos[d] >> top)) s_val[d * A.cap + s] = __builtin_bit_cast(uint32_t, base[c.pos[d]]);
              } else {
                float2* dst = PLANAR ? reinterpret_cast<float2*>(A.out) + (int64_t)level * A.n_max + b0 + s
                                     : reinterpret_cast<float2*>(A.out + (b0 + s) * 32 + 2 * level);
                float2 prev = make_float2(0.f, 0.f);
                if (ACC) prev = *dst;
                half2_t v[8];
#pragma unroll
                for (int d = 0; d < 8; d++) {
                  if (c.pos[d] >> top) v[d] = base[c.pos[d]];
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
          __syncthreads();  // keep the workgroup's waves in the same pass
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int gfn_hash_fwd_lm(const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool, const int32_t* prim_pool2,
                    const float* bias_pool, const float* pts, const int32_t* anchors, int anchor_stride, int64_t n_max, const int32_t* n_dev,
                    int64_t n_volumes, int log2_T, int halves_from, int planar, int accumulate, float* out, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(table_f16 && prim_pool && pts && anchors && out, "gfn_hash_fwd_lm: null pointer");
  GFN_CHECK_ARG(n_max > 0 && n_volumes > 0 && log2_T >= 4 && log2_T <= 24, "gfn_hash_fwd_lm: bad sizes");
  LmArgs A;
  A.n_max = n_max; A.n_dev = n_dev; A.pts = pts; A.anchors = anchors; A.anchor_stride = anchor_stride;
  A.bias = bias_pool; A.n_volumes = n_volumes; A.log2_T = log2_T;
  A.halves_from = halves_from < 0 ? 0 : halves_from;
  A.out = out;
  gfn_hash_level_scales_host(A.sc.s);
  const int grid = gfn_decoder_cu_cap();  // one workgroup per CU
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
