// What does FETCH_SIZE count for SCATTERED 4-byte reads on gfx950?  (is the raw counter, or twice it, the traffic?)
//
// The guide calibrates the counter for wide coalesced reads only (128-byte fabric requests tallied at 64 bytes: double it) and
// says other access widths are uncalibrated.  This tool reads ONE dword per touched line from a 1 GiB table (4 x the Infinity
// Cache, so nothing is served on-die) in five known patterns; run it under `rocprofv3 --pmc FETCH_SIZE` (and, separately,
// `--pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B`) and compare the counter per kernel with the bytes each pattern must move:
//   touch_stride<64>   every 64-byte line once, neighbouring lanes on neighbouring lines       1 GiB of lines touched
//   touch_stride<128>  every second 64-byte line (= every 128-byte line once)                   the same number of 128-byte lines, half the 64-byte ones
//   touch_stride<256>  every fourth 64-byte line                                                 a quarter of the 64-byte lines
//   touch_random       one dword at 16 Mi pseudo-random line-aligned addresses (the lookup's pattern)
//   stream16           the whole table, 16 bytes per lane coalesced (the guide's calibrated case: counter = bytes / 2)
// If requests were 128 bytes wide and tallied as 64, stride 64 and stride 128 would report the SAME count (each 128-byte line is
// fetched once either way); if requests are 64 bytes wide and tallied as 64, stride 128 reports half of stride 64.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib.bin fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int STRIDE>
__global__ __launch_bounds__(256) void touch_stride(const uint32_t* __restrict__ table, size_t n_touch, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_touch; i += gridDim.x * 256ull) acc += table[i * (STRIDE / 4)];
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void touch_random(const uint32_t* __restrict__ table, size_t n_lines, size_t n_touch, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_touch; i += gridDim.x * 256ull) {
    uint64_t h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    acc += table[(h % n_lines) * 16];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ table, size_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) { uint4 v = table[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = 1ull << 30;
  uint32_t *table, *out;
  CK(hipMalloc(&table, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(table, 1, bytes));
  CK(hipDeviceSynchronize());
  const int blocks = 4096;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0)); touch_stride<64><<<blocks, 256>>>(table, bytes / 64, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("touch_stride<64>   %8.3f ms   64-B lines touched %7.1f MB   128-B lines touched %7.1f MB\n", ms, bytes / 1e6, bytes / 1e6);
    CK(hipEventRecord(e0)); touch_stride<128><<<blocks, 256>>>(table, bytes / 128, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("touch_stride<128>  %8.3f ms   64-B lines touched %7.1f MB   128-B lines touched %7.1f MB\n", ms, bytes / 2e6, bytes / 1e6);
    CK(hipEventRecord(e0)); touch_stride<256><<<blocks, 256>>>(table, bytes / 256, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("touch_stride<256>  %8.3f ms   64-B lines touched %7.1f MB   128-B lines touched %7.1f MB\n", ms, bytes / 4e6, bytes / 2e6);
    const size_t n_rand = 16ull << 20;
    CK(hipEventRecord(e0)); touch_random<<<blocks, 256>>>(table, bytes / 64, n_rand, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("touch_random       %8.3f ms   64-B lines touched <= %5.1f MB  128-B lines touched <= %5.1f MB  (16 Mi touches, few collide in 16 Mi lines of 1 GiB)\n", ms, n_rand * 64 / 1e6, n_rand * 128 / 1e6);
    CK(hipEventRecord(e0)); stream16<<<blocks, 256>>>((const uint4*)table, bytes / 16, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream16           %8.3f ms   bytes read %7.1f MB\n", ms, bytes / 1e6);
  }
  return 0;
}
