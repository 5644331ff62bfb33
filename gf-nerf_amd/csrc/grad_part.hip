// grad_part.hip -- the hash-table gradient (Hash3DAnchoredFunction::backward, Hash3DAnchored_cuda.cu:81-155,198-239)
// accumulated in LDS instead of by scattered global atomics.
//
// The L2 executes ~21 G scattered atomics per second whatever their scope, type or footprint (tools/gather_bench.hip,
// tools/README.md): the 17 M packed-f16 adds of a focal-stage step cost 0.8 ms that way, 30 M+ in the global stage.
// The decoder backward (csrc/field_x3.hip, record mode) therefore only WRITES its adds, as (slot, packed f16x2) records
// appended to the stream of (16 Ki-slot range, producing workgroup) -- positions come from per-range cursors in the
// producer's LDS, so nothing is sorted afterwards.  Here the workgroup that owns a range adds its 256 streams into an LDS
// image and adds the image to the pool with plain coalesced stores: nobody else touches those slots.  16 streamed bytes
// per add instead of one L2 atomic; a stream that overflows falls back to the atomic in the producer.
//
// The image.  LDS executes FLOATING-POINT atomics (ds_pk_add_f16, ds_add_f32) at one lane per ~3.2 clocks per CU and INTEGER
// ones (ds_add_u32 / ds_add_u64) at one per ~0.5 clocks, the rate of plain stores (tools/lds_atom_bench.hip).  Every f16
// value is an integer multiple of 2^-24, so the two halves of a record go into ONE ds_add_u64 of (c1 << 32) + c0 in units
// of 2^-24: the sum is EXACT and independent of the order of the adds, and is rounded to f16 once at the end (the
// reference's atomicAdd(half2) rounds after every add, :151; its result depends on the order the hardware took them).  A
// 32-bit field holds |sum| < 128 (x128-scaled, i.e. a gradient of 1 per slot and step).  Every lane therefore also sums
// the magnitudes of what it adds: if the whole range's sum of |component| stays below 2^31 units no field can have
// left its 32 bits at any moment and the image is valid.  Otherwise the range is done again in COARSER units, 2^(k-24) with k
// chosen from that sum so that it fits now (round 5; rounds 1-4 fell back to f16 adds here -- a focal step's ~16 k records per
// range at |value| ~ 1e-2 sit right at the limit, so that which ranges were exact depended on the state): every addend is rounded
// to the coarser grid once (2^(k-25) of a unit whose sums are >= 128: below f16's own rounding of the result by 2^10 and more),
// the sum is again order independent, rounded to f16 once.  Only a NaN / inf among the addends (or f16_accumulate) takes the
// reference's way, ds_pk_add_f16 into an f16 image, which carries it into the pool like atomicAdd(half2) does -- decided per
// range, workgroup-uniform.
#include "gfn_common.h"
#include "hash_device.h"

namespace {

constexpr int kSlice = 1 << GFN_GRAD_RANGE_LOG2;  // slots (f16 pairs) per range: 16 Ki x 8 B = 128 KiB of LDS
constexpr float kFixed = 16777216.f;               // 2^24: the f16 grid
constexpr int kQ = 2;                              // producer streams a wave reads together

__device__ __forceinline__ uint32_t fixed_to_half2(unsigned long long S, float unit = 1.f / kFixed) {  // unit: value of one count (2^(k-24))
  const int lo = (int)(unsigned int)(S & 0xffffffffull);            // the low field, signed
  const int hi = (int)(((long long)S - (long long)lo) >> 32);       // what is left once its borrow is taken back
  const half2_t v = {(_Float16)((float)lo * unit), (_Float16)((float)hi * unit)};
  return __builtin_bit_cast(uint32_t, v);
}

// mode 0: exact fixed-point accumulation wherever it provably fits; 1: always the reference's f16 accumulation
// A wave reads kQ = 2 of its 16 producer streams at a time (four: 0.115 instead of 0.099 ms per focal step, and 128 registers
// instead of 64); the register bound (at most 96, so that waves of another kernel -- the table's Adam a chunk of ranges behind, the
// next batch's sampler -- fit beside a workgroup) costs nothing at that size.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5))) void grad_accumulate_kernel(
    const uint2* __restrict__ rec, unsigned int* __restrict__ rec_count, unsigned int rec_cap, int range_begin, int n_ranges, int64_t pool_slots,
    half2_t* __restrict__ pool, unsigned int* __restrict__ counts_seen, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long img64[];  // [kSlice] fixed-point pairs, or [kSlice] half2 in its first half
  half2_t* image = reinterpret_cast<half2_t*>(img64);
  __shared__ unsigned int s_cnt[256];
  __shared__ unsigned int s_any, s_poison;
  __shared__ unsigned long long s_mag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = range_begin + blockIdx.x; b < n_ranges; b += gridDim.x) {  // n_ranges: end of this launch's ranges
    if (tid == 0) { s_any = 0; s_mag = 0ull; s_poison = 0; }
    __syncthreads();
    if (tid < 256) {
      const unsigned int c = rec_count[b * 256 + tid];
      rec_count[b * 256 + tid] = 0;  // consumed: the next backward finds its counters zero without a fill in between
      if (counts_seen) counts_seen[b * 256 + tid] = c;
      s_cnt[tid] = c < rec_cap ? c : rec_cap;
      if (c) s_any = 1;  // benign race: every writer stores 1
    }
    __syncthreads();
    if (!s_any) continue;  // workgroup-uniform
    bool exact = mode == 0;  // fixed point (units of 2^(shift - 24)) | the reference's f16 adds
    int shift = 0;
    float to_fixed = kFixed, unit = 1.f / kFixed;
    for (int attempt = 0; attempt < 3; attempt++) {
      for (int i = tid; i < (exact ? kSlice / 2 : kSlice / 4); i += 1024) reinterpret_cast<uint4*>(img64)[i] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
      float mag = 0.f;  // this lane's sum of |c0| + |c1| in 2^-24 units: an upper bound (by up to 2x) of what any field saw, and -- unlike
                        // max(|c0|, |c1|), which returns the non-NaN operand -- NaN or inf in EITHER half poisons it, so such a range
                        // is redone with the reference's f16 adds, which carry the NaN into the pool like atomicAdd(half2) does
      // a wave takes 16 of the 256 producer streams, four at a time so that their (short, dependent) loads overlap; a
      // stream is contiguous: 16 bytes (two records) per lane and load
      auto add = [&](uint32_t slot, uint32_t val) {
        const uint32_t local = slot & (kSlice - 1);
        const half2_t v = __builtin_bit_cast(half2_t, val);
        if (exact) {
          const float f0 = (float)v[0] * to_fixed, f1 = (float)v[1] * to_fixed;  // shift = 0: exact, an integer below 2^40
          mag += fabsf(f0) + fabsf(f1);
          const int c0 = __float2int_rn(f0), c1 = __float2int_rn(f1);  // (shift = 0: saturates beyond 2^31, such a range is done again below;
                                                                       // shift > 0: the addend's one rounding, to the coarser grid)
          atomicAdd(img64 + local, ((unsigned long long)(unsigned int)c1 << 32) + (unsigned long long)(long long)c0);
        } else {
          __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(image + local), v);
        }
      };
      for (int w0 = wave * 16; w0 < wave * 16 + 16; w0 += kQ) {
        unsigned int n[kQ];
        const uint2* src[kQ];
        uint4 r[kQ];
#pragma unroll
        for (int q = 0; q < kQ; q++) {
          n[q] = s_cnt[w0 + q];
          src[q] = rec + ((size_t)b * 256 + w0 + q) * rec_cap;
          r[q] = make_uint4(0u, 0u, 0u, 0u);
          if (2u * lane < n[q]) r[q] = *reinterpret_cast<const uint4*>(src[q] + 2 * lane);  // rec_cap is even, the buffer 16-byte aligned
        }
#pragma unroll
        for (int q = 0; q < kQ; q++) {
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
      if (exact && shift == 0) {  // f32 sums of <= a few thousand terms: relative error < 1e-3, hence the margin below; NaN / inf -> poison
        if (mag < 1.0995e12f) { if (mag > 0.f) atomicAdd(&s_mag, (unsigned long long)mag); }
        else s_poison = 1;  // (benign race)
      }
      __syncthreads();
      if (!exact || shift > 0) break;                 // the f16 image, or coarser units chosen to fit: done
      if (s_poison) { exact = false; continue; }      // workgroup-uniform: a NaN / inf among the addends -- the reference's way carries it
      if (s_mag < 2140000000ull) break;               // no field can have left its 32 bits (2^31 = 2147483648)
      shift = 64 - __clzll(s_mag) - 30;               // sum of |addend| >> shift < 2^30 (+ half a count of rounding per addend, <= 2^17 of them)
      to_fixed = __uint_as_float((uint32_t)(127 + 24 - shift) << 23);   // 2^(24 - shift)
      unit = __uint_as_float((uint32_t)(127 - 24 + shift) << 23);       // 2^(shift - 24)
    }
    // pool range += image, 16 bytes (4 slots) per lane, all loads of a thread in flight before the first add; this
    // workgroup owns the range: no atomics (overflow atomics were issued by the producer, an earlier kernel)
    const int64_t s0 = (int64_t)b << GFN_GRAD_RANGE_LOG2;
    const bool whole = s0 + kSlice <= pool_slots;  // workgroup-uniform; the last range may be partial
    if (whole) {
      uint4* dst = reinterpret_cast<uint4*>(pool + s0);
      constexpr int kIt = kSlice / 4 / 1024;
      uint4 p[kIt];
#pragma unroll
      for (int k = 0; k < kIt; k++) p[k] = dst[tid + 1024 * k];
#pragma unroll
      for (int k = 0; k < kIt; k++) {
        const int i4 = tid + 1024 * k;
        uint4 v;
        if (exact) {
          const uint4 a = reinterpret_cast<const uint4*>(img64)[2 * i4], c = reinterpret_cast<const uint4*>(img64)[2 * i4 + 1];
          v.x = fixed_to_half2(((unsigned long long)a.y << 32) | a.x, unit); v.y = fixed_to_half2(((unsigned long long)a.w << 32) | a.z, unit);
          v.z = fixed_to_half2(((unsigned long long)c.y << 32) | c.x, unit); v.w = fixed_to_half2(((unsigned long long)c.w << 32) | c.z, unit);
        } else {
          v = reinterpret_cast<const uint4*>(image)[i4];
        }
        if ((v.x | v.y | v.z | v.w) & 0x7fff7fffu) {
          uint4 o;
          o.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].x) + __builtin_bit_cast(half2_t, v.x));
          o.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].y) + __builtin_bit_cast(half2_t, v.y));
          o.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].z) + __builtin_bit_cast(half2_t, v.z));
          o.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, p[k].w) + __builtin_bit_cast(half2_t, v.w));
          dst[i4] = o;
        }
      }
    } else {
      for (int i = tid; i < kSlice; i += 1024) {
        const int64_t s = s0 + i;
        const half2_t v = exact ? __builtin_bit_cast(half2_t, fixed_to_half2(img64[i], unit)) : image[i];
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
                            uint32_t* counts_seen, int f16_accumulate, int range_begin, int range_end, void* stream) {
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
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen)) {
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(grad_accumulate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSlice * 8));
  }
  if (range_end <= 0 || range_end > n_ranges) range_end = n_ranges;  // 0, 0: every range
  GFN_CHECK_ARG(range_begin >= 0 && range_begin <= range_end, "gfn_grad_records_reduce: ranges [%d, %d)", range_begin, range_end);
  if (range_begin == range_end) return GFN_OK;
  grad_accumulate_kernel<<<range_end - range_begin, 1024, kSlice * 8, gfn_stream(stream)>>>(
      reinterpret_cast<const uint2*>(records), record_counts, (unsigned int)cap, range_begin, range_end, pool_slots,
      reinterpret_cast<half2_t*>(grad_pool_f16), counts_seen, f16_accumulate ? 1 : 0);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
