// EXPERIMENT, not built into the product (round 1): radix-partitioned lookup -- requests sorted by owning CU, served from LDS.
// Bit-identical to the direct kernel but 0.85 ms per table against 0.36 ms for the level-major kernel (tools/README.md).
// part_lookup.hip -- Hash3DAnchored lookup (Hash3DAnchored_cuda.cu:11-79) as a radix-partitioned join.
//
// The direct lookup is bound by the memory fabric's request rate: every corner of the fine levels is its own 64-byte
// L2 miss for 4 useful bytes (57 G requests/s).  Here no lane ever touches the table in HBM at
// random.  Per round of 8 levels (two rounds per table):
//   A  partition : a workgroup takes 512 samples, hashes their 64 corners, and sorts the slot requests by the CU that
//                  OWNS the slot's address range (256 owners; counting sort in LDS), then writes one contiguous segment
//                  per owner (2 bytes per request) and, in sample order, each request's rank inside its segment;
//   B  serve     : owner CU b copies its 144 KiB slice of the f16 table into LDS once and answers its whole request
//                  stream from LDS (2 bytes in, 4 bytes out per request, both streamed);
//   C  combine   : the 512-sample workgroup reads its 256 answer segments back (contiguous), and every sample picks its
//                  64 values out of LDS by (owner, rank) and blends them exactly like the direct kernel.
// 16 streamed bytes per corner instead of one 64-byte random request; the values and the blend order are the direct
// kernel's, so the features are bit-identical.
#include "gfn_common.h"
#include "hash_device.h"

namespace {

constexpr int kChunk = 512;               // samples per partition workgroup (256 measured slower: shorter segments)
constexpr int kLevelsPerRound = 8;
constexpr int kReqPerSample = kLevelsPerRound * 8;
constexpr int kReqPerChunk = kChunk * kReqPerSample;  // 32768
constexpr int kBins = 256;

struct PartArgs {
  int64_t n_max; const int32_t* n_dev;
  const float* pts; const int* anchors; int anchor_stride;
  const int* prim; const float* bias; int64_t n_volumes; int log2_T;
  int round;          // levels [8 round, 8 round + 8)
  uint32_t bin_size;  // u32 slots per owner = 9 T / 512
  int64_t cap;        // capacity of one owner's request stream
  LevelScales sc;
};

// slot pair index inside the round's address range and its owner
__device__ __forceinline__ void locate(uint32_t pos, int lvl_in_round, int log2_T, uint32_t bin_size, uint32_t& bin, uint32_t& slot) {
  const uint32_t p = ((uint32_t)lvl_in_round << (log2_T - 1)) + pos;  // (l - 8r) * T/2 + pos, in u32 (f16 pair) units
  // bin_size = 9 * 2^(log2_T - 9): p / bin_size = (p >> (log2_T - 9)) / 9, the quotient by 9 through a multiply (x < 2^13)
  const uint32_t x = p >> (log2_T - 9);
  bin = (uint32_t)(((uint64_t)x * 954437177ull) >> 33);
  slot = p - bin * bin_size;
}

// all 8 levels' primes of one sample at once (independent loads: one round trip instead of eight)
struct Primes8 { int p[kLevelsPerRound][3]; };
__device__ __forceinline__ void load_primes(const PartArgs& A, int64_t vol, Primes8& P) {
#pragma unroll
  for (int lvl = 0; lvl < kLevelsPerRound; lvl++) {
    const int64_t tr = (int64_t)(A.round * kLevelsPerRound + lvl) * A.n_volumes + vol;
    P.p[lvl][0] = A.prim[3 * tr]; P.p[lvl][1] = A.prim[3 * tr + 1]; P.p[lvl][2] = A.prim[3 * tr + 2];
  }
}
__device__ __forceinline__ void level_corners(const PartArgs& A, const float* __restrict__ s_scale, float px, float py, float pz, int64_t vol,
                                              const Primes8& P, int lvl, Corners& c) {
  const int level = A.round * kLevelsPerRound + lvl;
  const int64_t tr = (int64_t)level * A.n_volumes + vol;
  uint32_t cell[3];
  hash_corners(px, py, pz, s_scale[level], P.p[lvl], A.bias ? A.bias + 3 * tr : nullptr, (1u << A.log2_T) - 1u, c, cell);
}

// block-wide exclusive scan of one value per thread (kThreads = 256 bins live on threads 0..255)
__device__ __forceinline__ unsigned int scan256(unsigned int cnt, int tid, unsigned int* wave_tot /*[4]*/) {
  unsigned int v = cnt;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (tid < kBins && lane == 63) wave_tot[wave] = v;
  __syncthreads();
  unsigned int add = 0;
  for (int w = 0; w < wave && w < 4; w++) add += wave_tot[w];
  return v - cnt + add;
}

constexpr int kThreads = kChunk;  // one sample per thread

// ---- A: partition ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void part_a_kernel(PartArgs A, unsigned int* __restrict__ bin_cursor, uint16_t* __restrict__ ranks,
                                                         uint16_t* __restrict__ req, uint2* __restrict__ chunk_meta, int* __restrict__ overflow) {
  __shared__ unsigned int hist[kBins], lds_off[kBins], gbase[kBins], wave_tot[4];
  __shared__ float s_scale[GFN_N_LEVELS];
  extern __shared__ __attribute__((aligned(16))) uint16_t stage[];  // [kReqPerChunk]
  const int tid = threadIdx.x;
  if (tid < kBins) hist[tid] = 0;
  if (tid < GFN_N_LEVELS) s_scale[tid] = A.sc.s[tid];
  __syncthreads();
  const int64_t n = gfn_live_n(A.n_max, A.n_dev);
  const int64_t chunk = blockIdx.x;
  const int64_t s0 = chunk * kChunk;
  if (s0 >= n) {  // chunk beyond the live samples: empty segments
    if (tid < kBins) chunk_meta[chunk * kBins + tid] = make_uint2(0u, 0u);
    return;
  }
  const int64_t i = s0 + tid;
  const bool live = i < n;
  // this sample's 64 requests: (owner << 16 | slot) and the rank inside the (chunk, owner) segment, kept in registers
  uint32_t where[kReqPerSample];
  uint32_t rk[kReqPerSample / 2];
  if (live) {
    const float px = A.pts[3 * i], py = A.pts[3 * i + 1], pz = A.pts[3 * i + 2];
    const int64_t vol = (int64_t)A.anchors[i * A.anchor_stride];
    Primes8 P;
    load_primes(A, vol, P);
#pragma unroll
    for (int lvl = 0; lvl < kLevelsPerRound; lvl++) {
      Corners c;
      level_corners(A, s_scale, px, py, pz, vol, P, lvl, c);
#pragma unroll
      for (int d = 0; d < 8; d++) {
        uint32_t bin, slot;
        locate(c.pos[d], lvl, A.log2_T, A.bin_size, bin, slot);
        where[lvl * 8 + d] = (bin << 16) | slot;
        const uint32_t r = atomicAdd(&hist[bin], 1u);
        if (d & 1) rk[(lvl * 8 + d) >> 1] |= r << 16; else rk[(lvl * 8 + d) >> 1] = r;
      }
    }
    uint4* rp = reinterpret_cast<uint4*>(ranks + i * kReqPerSample);
#pragma unroll
    for (int v4 = 0; v4 < kReqPerSample / 8; v4++) rp[v4] = make_uint4(rk[4 * v4], rk[4 * v4 + 1], rk[4 * v4 + 2], rk[4 * v4 + 3]);
  }
  __syncthreads();
  // reserve this chunk's segment in every owner's stream; exclusive scan of the histogram = the LDS staging order
  unsigned int cnt = 0;
  if (tid < kBins) {
    cnt = hist[tid];
    unsigned int base = cnt ? atomicAdd(bin_cursor + tid, cnt) : 0u;
    if (cnt && (int64_t)base + cnt > A.cap) { atomicOr(overflow, 8); base = 0; }  // results invalid; the caller falls back
    gbase[tid] = base;
    chunk_meta[chunk * kBins + tid] = make_uint2(base, cnt);
  }
  const unsigned int off = scan256(cnt, tid, wave_tot);
  if (tid < kBins) lds_off[tid] = off;
  __syncthreads();
  if (live) {
#pragma unroll
    for (int q = 0; q < kReqPerSample; q++) {
      const uint32_t w = where[q];
      const uint32_t rank = (rk[q >> 1] >> (16 * (q & 1))) & 0xffffu;
      stage[lds_off[w >> 16] + rank] = (uint16_t)(w & 0xffffu);
    }
  }
  __syncthreads();
  // copy out: one wave per owner at a time, contiguous 2-byte elements
  const int lane = tid & 63, wave = tid >> 6;
  for (int b = wave; b < kBins; b += kThreads / 64) {
    const unsigned int c = hist[b], src = lds_off[b];
    uint16_t* dst = req + (int64_t)b * A.cap + gbase[b];
    for (unsigned int e = lane; e < c; e += 64) dst[e] = stage[src + e];
  }
}

// ---- B: serve ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void part_b_kernel(const uint32_t* __restrict__ table_u32, int64_t round_base_u32, uint32_t bin_size,
                                                     int64_t table_u32_len, const unsigned int* __restrict__ bin_cursor, int64_t cap,
                                                     const uint16_t* __restrict__ req, uint32_t* __restrict__ val) {
  extern __shared__ __attribute__((aligned(16))) uint32_t slice[];  // [bin_size]
  const int b = blockIdx.x;
  const int64_t g0 = round_base_u32 + (int64_t)b * bin_size;
  if (g0 + bin_size <= table_u32_len && (g0 & 3) == 0 && (bin_size & 3) == 0) {
    const uint4* src = reinterpret_cast<const uint4*>(table_u32 + g0);
    for (uint32_t e = threadIdx.x; e < bin_size / 4; e += blockDim.x) reinterpret_cast<uint4*>(slice)[e] = src[e];
  } else {
    for (uint32_t e = threadIdx.x; e < bin_size; e += blockDim.x) slice[e] = (g0 + e < table_u32_len) ? table_u32[g0 + e] : 0u;
  }
  __syncthreads();
  int64_t n = bin_cursor[b];
  if (n > cap) n = cap;
  const uint16_t* rq = req + (int64_t)b * cap;
  uint32_t* vl = val + (int64_t)b * cap;
  // 8 requests per thread per step: one 16-byte read, eight LDS reads, two 16-byte writes (cap is a multiple of 64: aligned)
  const int64_t n8 = (n + 7) / 8;
  for (int64_t g = threadIdx.x; g < n8; g += blockDim.x) {
    const uint4 r = reinterpret_cast<const uint4*>(rq)[g];  // may read up to 7 stale slots past n: they are < 65536, clamp below
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint32_t sidx = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
      if (sidx >= bin_size) sidx = 0;
      v[k] = slice[sidx];
    }
    reinterpret_cast<uint4*>(vl)[2 * g] = make_uint4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<uint4*>(vl)[2 * g + 1] = make_uint4(v[4], v[5], v[6], v[7]);
  }
}

// ---- C: combine --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void part_c_kernel(PartArgs A, const uint16_t* __restrict__ ranks, const uint32_t* __restrict__ val,
                                                         const uint2* __restrict__ chunk_meta, int accumulate, float* __restrict__ out) {
  __shared__ unsigned int lds_off[kBins], wave_tot[4];
  __shared__ uint2 s_meta[kBins];
  __shared__ float s_scale[GFN_N_LEVELS];
  extern __shared__ __attribute__((aligned(16))) uint32_t vals[];  // [kReqPerChunk]
  const int tid = threadIdx.x;
  if (tid < GFN_N_LEVELS) s_scale[tid] = A.sc.s[tid];
  const int64_t n = gfn_live_n(A.n_max, A.n_dev);
  const int64_t chunk = blockIdx.x;
  const int64_t s0 = chunk * kChunk;
  if (s0 >= n) return;
  const int64_t i = s0 + tid;
  const bool live = i < n;
  // start this sample's own loads early: point, volume, primes, ranks
  float px = 0.f, py = 0.f, pz = 0.f;
  int64_t vol = 0;
  Primes8 P;
  uint4 rk4[kReqPerSample / 8];
  if (live) {
    px = A.pts[3 * i]; py = A.pts[3 * i + 1]; pz = A.pts[3 * i + 2];
    vol = (int64_t)A.anchors[i * A.anchor_stride];
    const uint4* rp = reinterpret_cast<const uint4*>(ranks + i * kReqPerSample);
#pragma unroll
    for (int v4 = 0; v4 < kReqPerSample / 8; v4++) rk4[v4] = rp[v4];
    load_primes(A, vol, P);
  }
  uint2 meta = make_uint2(0u, 0u);
  if (tid < kBins) { meta = chunk_meta[chunk * kBins + tid]; s_meta[tid] = meta; }
  const unsigned int off = scan256(meta.y, tid, wave_tot);
  if (tid < kBins) lds_off[tid] = off;
  __syncthreads();
  {  // answer segments -> LDS, 8 owners' loads in flight per wave
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kThreads / 64;
    for (int g = wave * 8; g < kBins; g += kWaves * 8) {
      uint32_t r[8][4];
#pragma unroll
      for (int bb = 0; bb < 8; bb++) {
        const uint2 m = s_meta[g + bb];
        const uint32_t* src = val + (int64_t)(g + bb) * A.cap + m.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned int e = lane + 64 * k;
          r[bb][k] = e < m.y ? src[e] : 0u;
        }
      }
#pragma unroll
      for (int bb = 0; bb < 8; bb++) {
        const uint2 m = s_meta[g + bb];
        const unsigned int dst = lds_off[g + bb];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned int e = lane + 64 * k;
          if (e < m.y) vals[dst + e] = r[bb][k];
        }
        if (m.y > 256) {  // rare: a segment longer than 256 answers
          const uint32_t* src = val + (int64_t)(g + bb) * A.cap + m.x;
          for (unsigned int e = lane + 256; e < m.y; e += 64) vals[dst + e] = src[e];
        }
      }
    }
  }
  __syncthreads();
  if (!live) return;
  float o[2 * kLevelsPerRound];
#pragma unroll
  for (int lvl = 0; lvl < kLevelsPerRound; lvl++) {
    Corners c;
    level_corners(A, s_scale, px, py, pz, vol, P, lvl, c);
    const uint4 pk = rk4[lvl];
    const uint32_t rr[4] = {pk.x, pk.y, pk.z, pk.w};
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      uint32_t bin, slot;
      locate(c.pos[d], lvl, A.log2_T, A.bin_size, bin, slot);
      const uint32_t rank = (rr[d >> 1] >> (16 * (d & 1))) & 0xffffu;
      const half2_t v = __builtin_bit_cast(half2_t, vals[lds_off[bin] + rank]);
      const float t0 = c.w[d] * (float)v[0], t1 = c.w[d] * (float)v[1];  // :74-77, left-to-right sum (as gather_level)
      a0 = (d == 0) ? t0 : a0 + t0;
      a1 = (d == 0) ? t1 : a1 + t1;
    }
    o[2 * lvl] = gfn_round_f16(a0); o[2 * lvl + 1] = gfn_round_f16(a1);
  }
  float4* dst = reinterpret_cast<float4*>(out + i * 32 + A.round * 2 * kLevelsPerRound);
#pragma unroll
  for (int g = 0; g < 4; g++) {
    float4 v = make_float4(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
    if (accumulate) { const float4 p = dst[g]; v.x = p.x + v.x; v.y = p.y + v.y; v.z = p.z + v.z; v.w = p.w + v.w; }
    dst[g] = v;
  }
}

}  // namespace

extern "C" {

// workspace bytes for n_max samples (ranks + request / answer streams + per-chunk segment table + cursors)
int64_t gfn_hash_fwd_part_workspace(int64_t n_max) {
  const int64_t chunks = (n_max + kChunk - 1) / kChunk;
  const int64_t cap = ((2 * n_max * kReqPerSample / kBins + 4096) + 63) & ~63ll;
  return n_max * kReqPerSample * 2 + kBins * cap * (2 + 4) + chunks * kBins * 8 + kBins * 4 + 256;
}

int gfn_hash_fwd_part(const uint16_t* table_f16, const int32_t* prim_pool, const float* bias_pool, const float* pts, const int32_t* anchors,
                      int anchor_stride, int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T, int accumulate, float* out,
                      void* workspace, int64_t workspace_bytes, int32_t* overflow_dev, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(table_f16 && prim_pool && pts && anchors && out && workspace && overflow_dev, "gfn_hash_fwd_part: null pointer");
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 25), "gfn_hash_fwd_part: n_max=%lld", (long long)n_max);
  GFN_CHECK_ARG(n_volumes > 0 && n_volumes < (1ll << 26) && log2_T >= 12 && log2_T <= 21, "gfn_hash_fwd_part: log2_T=%d must be in [12,21]", log2_T);
  GFN_CHECK_ARG(workspace_bytes >= gfn_hash_fwd_part_workspace(n_max) && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                "gfn_hash_fwd_part: workspace too small or not 256-byte aligned");
  const int64_t T = 1ll << log2_T;
  const uint32_t bin_size = (uint32_t)(9 * T / 512);  // u32 slots per owner: the round spans 9T f16 = 4.5T u32 over 256 owners
  GFN_CHECK_ARG(bin_size <= 65536 && (size_t)bin_size * 4 <= 160 * 1024 - 1024, "gfn_hash_fwd_part: slice does not fit LDS");
  const int64_t chunks = (n_max + kChunk - 1) / kChunk;
  const int64_t cap = ((2 * n_max * kReqPerSample / kBins + 4096) + 63) & ~63ll;
  uint8_t* w = static_cast<uint8_t*>(workspace);
  unsigned int* bin_cursor = reinterpret_cast<unsigned int*>(w);
  w += (kBins * 4 + 255) & ~255;
  uint16_t* ranks = reinterpret_cast<uint16_t*>(w); w += n_max * kReqPerSample * 2;
  uint16_t* req = reinterpret_cast<uint16_t*>(w); w += kBins * cap * 2;
  uint32_t* val = reinterpret_cast<uint32_t*>(w); w += kBins * cap * 4;
  uint2* chunk_meta = reinterpret_cast<uint2*>(w);
  hipStream_t st = gfn_stream(stream);
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen)) {
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(part_a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kReqPerChunk * 2));
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(part_b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(part_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kReqPerChunk * 4));
  }
  PartArgs A;
  A.n_max = n_max; A.n_dev = n_dev; A.pts = pts; A.anchors = anchors; A.anchor_stride = anchor_stride; A.prim = prim_pool; A.bias = bias_pool;
  A.n_volumes = n_volumes; A.log2_T = log2_T; A.bin_size = bin_size; A.cap = cap;
  gfn_hash_level_scales_host(A.sc.s);
  const uint32_t* table_u32 = reinterpret_cast<const uint32_t*>(table_f16);
  const int64_t table_u32_len = 16 * T;  // 16 T x 2 f16
  for (int round = 0; round < 2; round++) {
    A.round = round;
    GFN_CHECK_HIP(hipMemsetAsync(bin_cursor, 0, kBins * 4, st));
    part_a_kernel<<<(unsigned)chunks, kThreads, kReqPerChunk * 2, st>>>(A, bin_cursor, ranks, req, chunk_meta, overflow_dev);
    GFN_CHECK_LAUNCH();
    part_b_kernel<<<kBins, 1024, (size_t)bin_size * 4, st>>>(table_u32, (int64_t)round * kLevelsPerRound * T / 2, bin_size, table_u32_len, bin_cursor,
                                                            cap, req, val);
    GFN_CHECK_LAUNCH();
    part_c_kernel<<<(unsigned)chunks, kThreads, kReqPerChunk * 4, st>>>(A, ranks, val, chunk_meta, accumulate, out);
    GFN_CHECK_LAUNCH();
  }
  return GFN_OK;
}

}  // extern "C"
