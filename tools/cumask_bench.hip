// Scratch: random-gather rate vs. number of CUs enabled through hipExtStreamCreateWithCUMask, and whether an MFMA-bound
// kernel on the complementary CUs runs concurrently without slowing the gather.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out, int total_threads) {
  uint32_t acc = 0;
  for (uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; tid < (uint32_t)total_threads; tid += gridDim.x * blockDim.x) {
    for (int it = 0; it < iters; it++) {
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = table[mix(tid * 977u + it * 131071u + k * 7919u) & mask];
#pragma unroll
      for (int k = 0; k < 8; k++) acc += v[k];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void mfma_burn(int iters, float* out) {
  f32x16 a = {}; h8 x, y;
  for (int i = 0; i < 8; i++) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (_Float16)(i * 0.5f); }
  f32x16 b = {}, c = {}, d = {};
  for (int i = 0; i < iters; i++) {
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, b, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, d, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] + b[1] + c[2] + d[3];
}
int main() {
  uint32_t* table; uint32_t* out; float* fout;
  CK(hipMalloc(&table, 256u << 20)); CK(hipMemset(table, 1, 256u << 20));
  CK(hipMalloc(&out, 64u << 20)); CK(hipMalloc(&fout, 64u << 20));
  const uint32_t mask = 128u * 262144u - 1u;  // 128 MB
  const int total_threads = 2048 * 256, iters = 24;
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  for (int ncu : {256, 224, 192, 160, 128, 96, 64, 32}) {
    // bit i of the mask = CU i; take CUs round-robin so every XCD loses the same share (assume bit order interleaves XCDs)
    std::vector<uint32_t> gm(8, 0), mm(8, 0);
    for (int i = 0; i < 256; i++) {
      const bool g = (i * ncu / 256) != ((i + 1) * ncu / 256) || false;
      // choose: CU i belongs to the gather set if (i % 8 ... ) keep it simple: first ncu of each group of 256 by stride pattern
      (void)g;
    }
    int cnt = 0;
    for (int i = 0; i < 256; i++) {
      const bool in_g = ((int64_t)(i + 1) * ncu / 256) > ((int64_t)i * ncu / 256);  // evenly spread
      if (in_g) { gm[i / 32] |= 1u << (i % 32); cnt++; } else mm[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sg, sm;
    CK(hipExtStreamCreateWithCUMask(&sg, 8, gm.data()));
    const bool have_m = ncu < 256;
    if (have_m) CK(hipExtStreamCreateWithCUMask(&sm, 8, mm.data()));
    const int blocks = cnt * 8;
    gather<<<blocks, 256, 0, sg>>>(table, mask, iters, out, total_threads);
    CK(hipDeviceSynchronize());
    float alone = 1e9, together = 1e9, mf = 0, mf_alone = 0;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0, sg)); gather<<<blocks, 256, 0, sg>>>(table, mask, iters, out, total_threads); CK(hipEventRecord(e1, sg));
      CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < alone) alone = ms;
    }
    if (have_m) {
      const int mblocks = (256 - cnt) * 2, miters = 20000;
      CK(hipEventRecord(f0, sm)); mfma_burn<<<mblocks, 256, 0, sm>>>(miters, fout); CK(hipEventRecord(f1, sm)); CK(hipEventSynchronize(f1));
      CK(hipEventElapsedTime(&mf_alone, f0, f1));
      for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(f0, sm)); mfma_burn<<<mblocks, 256, 0, sm>>>(miters, fout); CK(hipEventRecord(f1, sm));
        CK(hipEventRecord(e0, sg)); gather<<<blocks, 256, 0, sg>>>(table, mask, iters, out, total_threads); CK(hipEventRecord(e1, sg));
        CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < together) together = ms;
        CK(hipEventElapsedTime(&mf, f0, f1));
      }
    }
    const double n = (double)total_threads * iters * 8;
    printf("gather on %3d CUs: alone %6.3f ms (%5.1f G/s)   with MFMA burn on the other %3d CUs: %6.3f ms (%5.1f G/s); burn alone %6.3f ms, overlapped %6.3f ms\n",
           cnt, alone, n / alone / 1e6, 256 - cnt, have_m ? together : 0.f, have_m ? n / together / 1e6 : 0.0, mf_alone, mf);
    CK(hipStreamDestroy(sg)); if (have_m) CK(hipStreamDestroy(sm));
  }
  return 0;
}
