// Scratch micro-benchmark: random 4-/8-byte gathers from a table (what bounds Hash3DAnchored's lookup?).
// hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// KIND 0: global_load_dword (64-bit vaddr), 1: buffer_load_dword (32-bit voffset), 2: buffer dwordx2
template <int KIND, int K>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, 0x7fffffff, 0x00020000);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t idx[K];
#pragma unroll
    for (int k = 0; k < K; k++) idx[k] = mix(tid * 977u + it * 131071u + k * 7919u) & mask;
    uint32_t v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      if (KIND == 0) v[k] = table[idx[k]];
      else if (KIND == 1) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, idx[k] * 4u, 0, 0);
      else { auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, (idx[k] & ~1u) * 4u, 0, 0); v[k] = t[0] ^ t[1]; }
    }
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k];
  }
  out[tid] = acc;
}

// random packed-f16 atomic adds (what the hash backward issues), no return value
template <int K>
__global__ __launch_bounds__(256) void scatter(uint32_t* __restrict__ table, uint32_t mask, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 one = {(_Float16)1.0f, (_Float16)0.5f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t idx = mix(tid * 977u + it * 131071u + k * 7919u) & mask;
      __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(table + idx), one);
    }
  }
}

int run_scatter(uint32_t* table, uint32_t mask, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  scatter<8><<<blocks, 256>>>(table, mask, iters);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    scatter<8><<<blocks, 256>>>(table, mask, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double n = (double)blocks * 256 * iters * 8;
  printf("%-34s table %4u MB blocks %5d K  8: %7.3f ms  %7.1f G atomics/s\n", "global_atomic_pk_add_f16", (mask + 1) / 262144, blocks, best, n / best / 1e6);
  return 0;
}

template <int KIND, int K>
int run(const char* name, const uint32_t* table, uint32_t mask, int blocks, int iters, uint32_t* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gather<KIND, K><<<blocks, 256>>>(table, mask, iters, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    gather<KIND, K><<<blocks, 256>>>(table, mask, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double n = (double)blocks * 256 * iters * K;
  printf("%-34s table %4u MB blocks %5d K %2d: %7.3f ms  %7.1f G gathers/s\n", name, (mask + 1) / 262144, blocks, K, best, n / best / 1e6);
  return 0;
}

int main() {
  uint32_t* table; uint32_t* out;
  const size_t tbytes = 512u << 20;
  CK(hipMalloc(&table, tbytes)); CK(hipMemset(table, 1, tbytes));
  CK(hipMalloc(&out, 64u << 20));
  for (uint32_t mb : {4u, 64u, 128u, 512u}) {
    const uint32_t mask = mb * 262144u - 1u;
    for (int blocks : {2048, 4096}) {
      const int iters = 2048 * 24 / blocks;
      run<0, 8>("global dword", table, mask, blocks, iters, out);
      run<1, 8>("buffer dword", table, mask, blocks, iters, out);
      run<1, 16>("buffer dword", table, mask, blocks, iters / 2, out);
      run<1, 32>("buffer dword", table, mask, blocks, iters / 4, out);
      run<2, 8>("buffer dwordx2", table, mask, blocks, iters, out);
      run<2, 16>("buffer dwordx2", table, mask, blocks, iters / 2, out);
    }
  }
  // table-size sweep around the 4 MB L2 of one XCD, and the scattered atomics of the backward
  for (uint32_t mb : {1u, 2u, 4u, 8u, 16u, 32u}) run<1, 8>("buffer dword (size sweep)", table, mb * 262144u - 1u, 2048, 24, out);
  for (uint32_t mb : {4u, 8u, 64u, 128u}) run_scatter(table, mb * 262144u - 1u, 2048, 24);
  // how many CUs does it take to saturate the memory-side atomic rate?  (256 threads per block, one block per CU up to 256)
  for (int blocks : {16, 32, 64, 128, 256, 512, 1024}) run_scatter(table, 64u * 262144u - 1u, blocks, 24 * 2048 / blocks / 4);
  return 0;
}
