// grad_part.hip -- the hash-table gradient (Hash3DAnchoredFunction::backward, Hash3DAnchored_cuda.cu:81-155,198-239)
// accumulated in LDS instead of by scattered global atomics.
//
// The L2 executes ~21 G scattered atomics per second whatever their scope, type or footprint (tools/gather_bench.hip,
// tools/README.md): the 12.8 M packed-f16 adds of a focal-stage step cost 0.6 ms that way, 30 M+ in the global stage.
// The decoder backward (csrc/field_x3.hip, record mode) therefore only WRITES its adds, as (slot, packed f16x2) records
// appended to the stream of (32 Ki-slot range, producing workgroup) -- positions come from per-range cursors in the
// producer's LDS, so nothing is sorted afterwards.  Here the workgroup that owns a range adds its 256 streams into a
// 128 KiB LDS image with ds_pk_add_f16 (the f16 accumulation of the reference's half-precision pool, :151,:222) and adds
// the image to the pool with plain coalesced stores: nobody else touches those slots.  16 streamed bytes per add instead
// of one L2 atomic; a stream that overflows falls back to the atomic in the producer.
#include "gfn_common.h"
#include "hash_device.h"

namespace {

constexpr int kSlice = 1 << GFN_GRAD_RANGE_LOG2;  // slots (f16 pairs) per range: 32 Ki x 4 B = 128 KiB of LDS

__global__ __launch_bounds__(1024) void grad_accumulate_kernel(const uint2* __restrict__ rec, unsigned int* __restrict__ rec_count,
                                                              unsigned int rec_cap, int n_ranges, int64_t pool_slots,
                                                              half2_t* __restrict__ pool, unsigned int* __restrict__ counts_seen) {
  extern __shared__ __attribute__((aligned(16))) half2_t image[];  // [kSlice]
  __shared__ unsigned int s_cnt[256];
  __shared__ unsigned int s_any;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = blockIdx.x; b < n_ranges; b += gridDim.x) {
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (tid < 256) {
      const unsigned int c = rec_count[b * 256 + tid];
      rec_count[b * 256 + tid] = 0;  // consumed: the next backward finds its counters zero without a fill in between
      if (counts_seen) counts_seen[b * 256 + tid] = c;
      s_cnt[tid] = c < rec_cap ? c : rec_cap;
      if (c) s_any = 1;  // benign race: every writer stores 1
    }
    for (int i = tid; i < kSlice / 4; i += 1024) reinterpret_cast<uint4*>(image)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if (!s_any) continue;  // workgroup-uniform
    // a wave takes 16 of the 256 producer streams, four at a time so that their (short, dependent) loads overlap; a
    // stream is contiguous: 16 bytes (two records) per lane and load
    auto add = [&](uint32_t slot, uint32_t val) {
      __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(image + (slot & (kSlice - 1))), __builtin_bit_cast(half2_t, val));
    };
    for (int w0 = wave * 16; w0 < wave * 16 + 16; w0 += 4) {
      unsigned int n[4];
      const uint2* src[4];
      uint4 r[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        n[q] = s_cnt[w0 + q];
        src[q] = rec + ((size_t)b * 256 + w0 + q) * rec_cap;
        r[q] = make_uint4(0u, 0u, 0u, 0u);
        if (2u * lane < n[q]) r[q] = *reinterpret_cast<const uint4*>(src[q] + 2 * lane);  // rec_cap is even, the buffer 16-byte aligned
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (2u * lane < n[q]) {
          add(r[q].x, r[q].y);
          if (2u * lane + 1 < n[q]) add(r[q].z, r[q].w);
        }
        for (unsigned int i = 2 * lane + 128; i < n[q]; i += 128) {  // streams longer than 128 records
          const uint4 t = *reinterpret_cast<const uint4*>(src[q] + i);
          add(t.x, t.y);
          if (i + 1 < n[q]) add(t.z, t.w);
        }
      }
    }
    __syncthreads();
    // image += pool range, 16 bytes (4 slots) per lane, all eight loads of a thread in flight before the first add; this
    // workgroup owns the range: no atomics (overflow atomics were issued by the producer, an earlier kernel)
    const int64_t s0 = (int64_t)b << GFN_GRAD_RANGE_LOG2;
    const bool whole = s0 + kSlice <= pool_slots;  // workgroup-uniform; the last range may be partial
    if (whole) {
      uint4* dst = reinterpret_cast<uint4*>(pool + s0);
      uint4 p[8];
#pragma unroll
      for (int k = 0; k < 8; k++) p[k] = dst[tid + 1024 * k];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint4 v = reinterpret_cast<const uint4*>(image)[tid + 1024 * k];
        if (v.x | v.y | v.z | v.w) {
          uint4 o;
          o.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].x) + __builtin_bit_cast(half2_t, v.x));
          o.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].y) + __builtin_bit_cast(half2_t, v.y));
          o.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].z) + __builtin_bit_cast(half2_t, v.z));
          o.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].w) + __builtin_bit_cast(half2_t, v.w));
          dst[tid + 1024 * k] = o;
        }
      }
    } else {
      for (int i = tid; i < kSlice; i += 1024) {
        const int64_t s = s0 + i;
        const half2_t v = image[i];
        if (s < pool_slots && (v[0] != (_Float16)0.f || v[1] != (_Float16)0.f)) pool[s] = pool[s] + v;
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int gfn_grad_record_ranges(int log2_T) {
  const int64_t T = 1ll << log2_T;
  return (int)((8 * T + T / 2 + kSlice - 1) >> GFN_GRAD_RANGE_LOG2);  // live f16 pairs [0, 17T/2): the level-overlap quirk
}

int gfn_grad_records_reduce(const void* records, uint32_t* record_counts, int64_t record_capacity, int log2_T, void* grad_pool_f16,
                            uint32_t* counts_seen, void* stream) {
  GFN_CHECK_ARG(records && record_counts && grad_pool_f16, "gfn_grad_records_reduce: null pointer");
  GFN_CHECK_ARG(log2_T >= 4 && log2_T <= 21, "gfn_grad_records_reduce: log2_T=%d must be in [4,21]", log2_T);
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(records) & 15) == 0, "gfn_grad_records_reduce: records must be 16-byte aligned");
  const int64_t T = 1ll << log2_T;
  const int64_t pool_slots = 8 * T + T / 2;
  const int n_ranges = gfn_grad_record_ranges(log2_T);
  GFN_CHECK_ARG(n_ranges >= 1 && n_ranges <= GFN_GRAD_MAX_RANGES, "gfn_grad_records_reduce: %d slot ranges", n_ranges);
  int64_t cap = (record_capacity / ((int64_t)n_ranges * 256)) & ~1ll;
  GFN_CHECK_ARG(cap >= 2, "gfn_grad_records_reduce: record buffer too small");
  if (cap > 0x7ffffffe) cap = 0x7ffffffe;
  static bool attr_set = false;
  if (!attr_set) {
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(grad_accumulate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSlice * 4));
    attr_set = true;
  }
  grad_accumulate_kernel<<<n_ranges, 1024, kSlice * 4, gfn_stream(stream)>>>(reinterpret_cast<const uint2*>(records), record_counts, (unsigned int)cap,
                                                                           n_ranges, pool_slots, reinterpret_cast<half2_t*>(grad_pool_f16), counts_seen);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
