// Read-only streaming bandwidth on MI355X: what can a kernel that only READS reach?  (wgrad_x3 streams 1.85 GB of planes.)
//   read_regs<U>  : grid-stride, U independent 16-byte loads per thread in flight, xor-reduced
//   read_glds<U>  : the same bytes by global_load_lds_dwordx4 into the wave's LDS window (U x 1 KB in flight per wave)
//   hipcc --offload-arch=gfx950 -O3 -o read_bw.bin read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int U>
__global__ __launch_bounds__(256) void read_regs(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i + (U - 1) * stride < n16; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// each workgroup walks CONTIGUOUS chunks of U KB per wave (like a slab), chunks dealt round-robin over workgroups
template <int U>
__global__ __launch_bounds__(256) void read_regs_chunk(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  const size_t chunk = 256ull * U;  // 16-byte units per workgroup step
  for (size_t c = blockIdx.x; (c + 1) * chunk <= n16; c += gridDim.x) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = src[c * chunk + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int U, int NBUF>
__global__ __launch_bounds__(256) void read_glds(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];  // [NBUF][U][256]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t chunk = 256ull * U;
  uint32_t acc = 0;
  int buf = 0;
  for (size_t c = blockIdx.x; (c + 1) * chunk <= n16; c += gridDim.x, buf = (buf + 1) % NBUF) {
#pragma unroll
    for (int u = 0; u < U; u++)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + c * chunk + u * 256 + threadIdx.x), (lptr_t)(lds + (buf * U + u) * 256 + wave * 64), 16, 0, 0);
    if (NBUF == 1) { __syncthreads(); acc += lds[threadIdx.x].x; }
    else { __syncthreads(); acc += lds[((buf + NBUF - 1) % NBUF) * U * 256 + threadIdx.x].x; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// as read_glds, but every 8 KB block (512 pieces) is fetched in wgrad_x3's order: LDS piece e = (tile t, sample s, quarter c) comes
// from HBM piece (4t + c) * 32 + s  -> per wave-instruction four 256-byte runs, 512 bytes apart
template <int U, int NBUF>
__global__ __launch_bounds__(256) void read_glds_tile(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int wave = threadIdx.x >> 6;
  const size_t chunk = 256ull * U;
  uint32_t acc = 0;
  int buf = 0;
  for (size_t c = blockIdx.x; (c + 1) * chunk <= n16; c += gridDim.x, buf = (buf + 1) % NBUF) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int e = (u & 1) * 256 + threadIdx.x;  // piece of the 8 KB block (two instructions per block)
      const int t = e >> 7, smp = (e >> 2) & 31, q = e & 3;
      const size_t srcp = c * chunk + (u >> 1) * 512 + (4 * t + q) * 32 + smp;
      __builtin_amdgcn_global_load_lds((gptr_t)(src + srcp), (lptr_t)(lds + (buf * U + u) * 256 + wave * 64), 16, 0, 0);
    }
    __syncthreads();
    acc += lds[((buf + NBUF - 1) % NBUF) * U * 256 + threadIdx.x].x;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = 2ull << 30, n16 = bytes / 16;
  uint4* src; uint32_t* out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 1, bytes)); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define T(name, lds, ...)                                                                                          \
  for (int blocks : {256, 512, 1024, 2048, 4096}) {                                                                 \
    float best = 1e9f;                                                                                              \
    for (int r = 0; r < 4; r++) {                                                                                   \
      CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                     \
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;                                     \
    }                                                                                                               \
    CK(hipGetLastError());                                                                                          \
    printf("%-28s blocks %5d: %7.3f ms  %6.2f TB/s\n", name, blocks, best, bytes / best / 1e9);                     \
  }
  T("read_regs<1>", 0, read_regs<1><<<blocks, 256>>>(src, n16, out))
  T("read_regs<4>", 0, read_regs<4><<<blocks, 256>>>(src, n16, out))
  T("read_regs<8>", 0, read_regs<8><<<blocks, 256>>>(src, n16, out))
  T("read_regs_chunk<4>", 0, read_regs_chunk<4><<<blocks, 256>>>(src, n16, out))
  T("read_regs_chunk<8>", 0, read_regs_chunk<8><<<blocks, 256>>>(src, n16, out))
  T("read_regs_chunk<16>", 0, read_regs_chunk<16><<<blocks, 256>>>(src, n16, out))
  T("read_glds<4,2> 32KB", 0, read_glds<4, 2><<<blocks, 256, 2 * 4 * 4096>>>(src, n16, out))
  T("read_glds<8,2> 64KB", 0, read_glds<8, 2><<<blocks, 256, 2 * 8 * 4096>>>(src, n16, out))
  T("read_glds<4,3> 48KB", 0, read_glds<4, 3><<<blocks, 256, 3 * 4 * 4096>>>(src, n16, out))
  T("read_glds_tile<4,2> 32KB", 0, read_glds_tile<4, 2><<<blocks, 256, 2 * 4 * 4096>>>(src, n16, out))
  T("read_glds_tile<8,2> 64KB", 0, read_glds_tile<8, 2><<<blocks, 256, 2 * 8 * 4096>>>(src, n16, out))
  T("read_glds<8,1> 32KB", 0, read_glds<8, 1><<<blocks, 256, 8 * 4096>>>(src, n16, out))
  return 0;
}
