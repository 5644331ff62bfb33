// Scattered plain stores: does a 16-byte store cost more than an 8-byte one?  (The backward appends 8-byte records to 278 k
// streams; pairing records of one stream into 16-byte stores would halve the store count.)  Every lane writes to its own random
// 128-byte line of a 1 GiB buffer; 4 / 8 / 16 bytes per store.
//   hipcc --offload-arch=gfx950 -O3 -o store_bench.bin store_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int BYTES>
__global__ __launch_bounds__(256) void scatter(uint8_t* __restrict__ buf, uint32_t line_mask, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const size_t line = mix(tid * 977u + it * 131071u + k * 7919u) & line_mask;
      uint8_t* p = buf + line * 128;
      if (BYTES == 4) *reinterpret_cast<uint32_t*>(p) = tid;
      else if (BYTES == 8) *reinterpret_cast<uint2*>(p) = make_uint2(tid, it);
      else *reinterpret_cast<uint4*>(p) = make_uint4(tid, it, k, 1);
    }
  }
}

template <int BYTES>
int run(uint8_t* buf, uint32_t line_mask, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 24 * 2048 / blocks;
  scatter<BYTES><<<blocks, 256>>>(buf, line_mask, iters);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0)); scatter<BYTES><<<blocks, 256>>>(buf, line_mask, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double n = (double)blocks * 256 * iters * 8;
  printf("%2d-byte stores, %4u MB of lines, %4d blocks: %7.3f ms  %6.1f G stores/s  %6.2f TB/s of payload\n", BYTES, (line_mask + 1) / 8192, blocks, best,
         n / best / 1e6, n * BYTES / best / 1e9);
  return 0;
}

int main() {
  uint8_t* buf;
  CK(hipMalloc(&buf, 1ull << 30)); CK(hipMemset(buf, 0, 1ull << 30));
  for (uint32_t mb : {2u, 4u, 8u, 16u, 32u, 64u, 1024u}) {
    const uint32_t mask = mb * 8192u - 1u;
    for (int blocks : {768, 2048}) { run<8>(buf, mask, blocks); if (mb == 32u || mb == 1024u) { run<4>(buf, mask, blocks); run<16>(buf, mask, blocks); } }
  }
  return 0;
}
