// EXPERIMENT, not built into the product (round 5): "a level pair per XCD" for the level-major lookup.
//
// lm_lookup_kernel (csrc/lm_lookup.hip) lets every workgroup sweep all 16 levels for 1/256 of the samples, so each of the 8 XCD L2s
// pulls all 71 MB of live slots once per launch (0.57 GB of the launch's 1.04 GB of fabric traffic).  The proposal: workgroup i runs on
// XCD i % 8 (tools/xcc_map.hip); give XCD x the levels 2x, 2x + 1 for ALL samples, so that its L2 only ever sees three 4 MB parts.
// A level needs two parts (its lower and its upper address half), an XCD's L2 holds one, and the corner values of the part gathered
// first must wait somewhere until the second part completes the blend (the sum over the 8 corners has a fixed order: bit-identity).
// In the product they wait in LDS (32 B per sample: 4096 samples per CU).  Here an XCD sees 8 x the samples -- 786 k, 25 MB of
// parked values per level -- so the sweep must be `for part: for round of samples`, re-reading the points once per part (3 x) and
// parking in GLOBAL memory (2 x 25 MB written and read back per XCD: 0.8 GB per launch, more than the table fetch it saves).
//
// This file measures the UPPER BOUND of the scheme before any of that is built: the same hashing, the same gathers from the same
// parts in the same order, the blend and the feature stores -- but nothing is parked (the corners of the part that is not being
// swept count as zero: the features are WRONG, the time is a lower bound of the real kernel's).  If even this is not clearly faster
// than the product kernel on the same samples, the scheme is dead.  exp_xcd_lookup.py runs both.
#include "../gf-nerf_amd/csrc/gfn_common.h"
#include "../gf-nerf_amd/csrc/hash_device.h"

namespace {

constexpr int kThreads = 1024, kK = 4;

struct XArgs {
  int64_t n_max; const int32_t* n_dev;
  const float* pts; const int* anchors; int anchor_stride;
  const void* table; const int* prim; int64_t n_volumes; int log2_T;
  float* out;  // level-major float2 [16][n_max]
  int nt_points;  // read the points with the non-temporal policy (they are streamed three times per XCD)
  LevelScales sc;
};

__device__ __forceinline__ half2_t ld_pair(const half2_t* __restrict__ base, uint32_t pos) {
  return *reinterpret_cast<const half2_t*>(reinterpret_cast<const char*>(base) + (size_t)(uint32_t)(pos << 2));
}

__global__ __launch_bounds__(kThreads) void xcd_lookup_kernel(XArgs A) {
  __shared__ float s_scale[GFN_N_LEVELS];
  const int tid = threadIdx.x;
  if (tid < GFN_N_LEVELS) s_scale[tid] = A.sc.s[tid];
  __syncthreads();
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, nw = gridDim.x >> 3;
  const int64_t n = gfn_live_n(A.n_max, A.n_dev);
  const int64_t per = (n + nw - 1) / nw;
  const int64_t b0 = (int64_t)w * per;
  const int64_t mine = b0 >= n ? 0 : (n - b0 < per ? n - b0 : per);
  const uint32_t mask = (1u << A.log2_T) - 1u;
  const int pshift = A.log2_T - 1;
  for (int sweep = 0; sweep < 3; sweep++) {  // part 2 xcd + sweep: level 2x's lower half | 2x's upper + (2x+1)'s lower | (2x+1)'s upper
    for (int64_t r0 = 0; r0 < mine; r0 += kK * kThreads) {
      float px[kK], py[kK], pz[kK];
      int vol[kK];
      int64_t smp[kK];
#pragma unroll
      for (int k = 0; k < kK; k++) {
        const int64_t s = r0 + tid + k * kThreads;
        const int64_t i = s < mine ? b0 + s : b0;
        smp[k] = s < mine ? i : -1;
        if (A.nt_points) {
          px[k] = __builtin_nontemporal_load(A.pts + 3 * i); py[k] = __builtin_nontemporal_load(A.pts + 3 * i + 1);
          pz[k] = __builtin_nontemporal_load(A.pts + 3 * i + 2);
          vol[k] = __builtin_nontemporal_load(A.anchors + i * A.anchor_stride);
        } else {
          px[k] = A.pts[3 * i]; py[k] = A.pts[3 * i + 1]; pz[k] = A.pts[3 * i + 2];
          vol[k] = A.anchors[i * A.anchor_stride];
        }
      }
#pragma unroll
      for (int which = 0; which < 2; which++) {
        const int half = which == 0 ? sweep : sweep - 1;  // the address half of level 2x + which that this sweep's part is
        if (half < 0 || half > 1) continue;
        const int level = 2 * xcd + which;
        const float scale = s_scale[level];
        const half2_t* base = reinterpret_cast<const half2_t*>(reinterpret_cast<const _Float16*>(A.table) + (int64_t)level * (1ll << A.log2_T));
        int prim[kK][3];
#pragma unroll
        for (int k = 0; k < kK; k++) {
          const int* p = A.prim + 3 * ((int64_t)level * A.n_volumes + vol[k]);
          prim[k][0] = p[0]; prim[k][1] = p[1]; prim[k][2] = p[2];
        }
#pragma unroll
        for (int k = 0; k < kK; k++) {
          if (smp[k] < 0) continue;
          Corners c;
          uint32_t cell[3];
          hash_corners(px[k], py[k], pz[k], scale, prim[k], nullptr, mask, c, cell);
          half2_t v[8];
#pragma unroll
          for (int d = 0; d < 8; d++) {
            v[d] = half2_t{(_Float16)0.f, (_Float16)0.f};  // (the real kernel: the value parked by the sweep of the other half)
            if ((int)(c.pos[d] >> pshift) == half) v[d] = ld_pair(base, c.pos[d]);
          }
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int d = 0; d < 8; d++) {
            const float t0 = c.w[d] * (float)v[d][0], t1 = c.w[d] * (float)v[d][1];
            a0 = (d == 0) ? t0 : a0 + t0;
            a1 = (d == 0) ? t1 : a1 + t1;
          }
          float2* dst = reinterpret_cast<float2*>(A.out) + (int64_t)level * A.n_max + smp[k];
          if (half == 1) *dst = make_float2(gfn_round_f16(a0), gfn_round_f16(a1));  // the upper half completes the level
          else if (a0 == 12345.f) *dst = make_float2(a0, a1);                       // (keeps the lower half's gathers alive)
        }
      }
      __syncthreads();  // keep the workgroup's waves in the same round
    }
  }
}

}  // namespace

void gfn_set_error(const char*, ...) {}

extern "C" int xcd_lookup(const uint16_t* table_f16, const int32_t* prim_pool, const float* pts, const int32_t* anchors, int anchor_stride,
                          int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T, float* out, int nt_points, void* stream) {
  XArgs A;
  A.n_max = n_max; A.n_dev = n_dev; A.pts = pts; A.anchors = anchors; A.anchor_stride = anchor_stride; A.table = table_f16; A.prim = prim_pool;
  A.n_volumes = n_volumes; A.log2_T = log2_T; A.out = out; A.nt_points = nt_points;
  gfn_hash_level_scales_host(A.sc.s);
  xcd_lookup_kernel<<<256, kThreads, 0, reinterpret_cast<hipStream_t>(stream)>>>(A);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
