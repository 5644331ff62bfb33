// What does a scattered 4-byte gather cost when G lanes of the wave want the SAME address (neighbouring pixels' rays in one
// table cell)?  buffer_load_dword, 8 per lane in flight, groups of G = 1 .. 32 consecutive (or G lanes strided by 64 / G) lanes
// share an address; tables of 4 MB (one XCD's L2), 64 MB (beyond it).
//   hipcc --offload-arch=gfx950 -O3 -o gather_share.bin gather_share.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int G, bool STRIDED>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t grp = wave * 64 + (STRIDED ? lane % (64 / G) : lane / G);  // lanes of one group share their addresses
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, 0x7fffffff, 0x00020000);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (mix(grp * 977u + it * 131071u + k * 7919u) & mask) * 4u, 0, 0);
#pragma unroll
    for (int k = 0; k < 8; k++) acc += v[k];
  }
  out[tid] = acc;
}

template <int G, bool STRIDED>
int run(const uint32_t* table, uint32_t mask, uint32_t* out) {
  const int blocks = 2048, iters = 24;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gather<G, STRIDED><<<blocks, 256>>>(table, mask, iters, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    gather<G, STRIDED><<<blocks, 256>>>(table, mask, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double n = (double)blocks * 256 * iters * 8;
  printf("G %2d %-11s table %3u MB: %7.3f ms  %7.1f G lane-gathers/s  %6.1f G distinct addresses/s\n", G, STRIDED ? "strided" : "consecutive",
         (mask + 1) / 262144, best, n / best / 1e6, n / G / best / 1e6);
  return 0;
}

int main() {
  uint32_t* table; uint32_t* out;
  CK(hipMalloc(&table, 64u << 20)); CK(hipMemset(table, 1, 64u << 20));
  CK(hipMalloc(&out, 64u << 20));
  for (uint32_t mb : {4u, 64u}) {
    const uint32_t mask = mb * 262144u - 1u;
    run<1, false>(table, mask, out); run<2, false>(table, mask, out); run<4, false>(table, mask, out); run<8, false>(table, mask, out);
    run<16, false>(table, mask, out); run<32, false>(table, mask, out);
    run<2, true>(table, mask, out); run<4, true>(table, mask, out); run<8, true>(table, mask, out); run<32, true>(table, mask, out);
  }
  return 0;
}
