#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SCOPE>
__global__ __launch_bounds__(256) void scatter(float* __restrict__ table, uint32_t mask, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t idx = mix(tid * 977u + it * 131071u + k * 7919u) & mask;
      if (SCOPE == 0) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (SCOPE == 1) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (SCOPE == 2) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else if (SCOPE == 3) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      else if (SCOPE == 4) __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(table) + idx, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // integer
      else if (SCOPE == 5) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(table) + (idx >> 1), 0x300000005ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else table[idx] = 1.0f;  // plain scattered 4-byte store
    }
  }
}
template <int SCOPE> int run(const char* name, float* table, uint32_t mask) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  scatter<SCOPE><<<2048, 256>>>(table, mask, 24); CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); scatter<SCOPE><<<2048, 256>>>(table, mask, 24); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  printf("%-10s table %4u MB: %7.3f ms %7.1f G atomics/s\n", name, (mask + 1) / 262144, best, 2048.0 * 256 * 24 * 8 / best / 1e6);
  return 0;
}
int main() {
  float* table; CK(hipMalloc(&table, 256u << 20)); CK(hipMemset(table, 0, 256u << 20));
  for (uint32_t mb : {4u, 64u}) {
    const uint32_t mask = mb * 262144u - 1u;
    run<0>("agent", table, mask); run<1>("workgroup", table, mask); run<3>("wavefront", table, mask); run<2>("system", table, mask);
    run<4>("u32 add", table, mask); run<5>("u64 add", table, mask); run<6>("store", table, mask);
  }
  // sanity: did every add land?  (sum of the table after one known launch)
  return 0;
}
