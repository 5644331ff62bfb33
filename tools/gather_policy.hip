// Do the cache-policy bits of a load change what a scattered 4-byte gather costs?  (The level-major lookup is bound by L2 -> L1
// line fills: 128 bytes moved for 4 wanted.  A load that bypasses the L1 might move less.)
// buffer_load_dword with aux = sc0 (1), nt (2), sc1 (16) and their combinations, from a 4 MB (one XCD's L2) and a 64 MB table.
//   hipcc --offload-arch=gfx950 -O3 -o gather_policy.bin gather_policy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int AUX, int K>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, 0x7fffffff, 0x00020000);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (mix(tid * 977u + it * 131071u + k * 7919u) & mask) * 4u, 0, AUX);
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k];
  }
  out[tid] = acc;
}

template <int AUX>
int run(const char* name, const uint32_t* table, uint32_t mask, uint32_t* out) {
  const int blocks = 2048, iters = 24, K = 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gather<AUX, K><<<blocks, 256>>>(table, mask, iters, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    gather<AUX, K><<<blocks, 256>>>(table, mask, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double n = (double)blocks * 256 * iters * K;
  printf("aux %2d %-12s table %4u MB: %7.3f ms  %7.1f G gathers/s\n", AUX, name, (mask + 1) / 262144, best, n / best / 1e6);
  return 0;
}

int main() {
  uint32_t* table; uint32_t* out;
  CK(hipMalloc(&table, 64u << 20)); CK(hipMemset(table, 1, 64u << 20));
  CK(hipMalloc(&out, 64u << 20));
  for (uint32_t mb : {2u, 4u, 64u}) {
    const uint32_t mask = mb * 262144u - 1u;
    run<0>("default", table, mask, out);
    run<1>("sc0", table, mask, out);
    run<2>("nt", table, mask, out);
    run<3>("sc0 nt", table, mask, out);
    run<16>("sc1", table, mask, out);
    run<17>("sc0 sc1", table, mask, out);
    run<18>("sc1 nt", table, mask, out);
    run<19>("sc0 sc1 nt", table, mask, out);
  }
  return 0;
}
