// lds_atom_bench.hip -- how fast does a CU execute scattered LDS atomics of each kind?  (what bounds grad_accumulate_kernel)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/lds_atom_bench tools/lds_atom_bench.hip && gpurun_out/lds_atom_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
constexpr int kSlots = 32768;  // 128 KiB image, like the accumulate kernel
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ idx, int per_thread, uint32_t* __restrict__ out) {
  extern __shared__ uint32_t image[];
  for (int i = threadIdx.x; i < kSlots; i += 1024) image[i] = 0;
  __syncthreads();
  const uint32_t* p = idx + ((size_t)blockIdx.x * 1024 + threadIdx.x) * per_thread;
  for (int i = 0; i < per_thread; i += 4) {
    const uint4 s = *reinterpret_cast<const uint4*>(p + i);
    const uint32_t a[4] = {s.x & (kSlots - 1), s.y & (kSlots - 1), s.z & (kSlots - 1), s.w & (kSlots - 1)};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (MODE == 0) image[a[q]] = s.x;                                                                   // plain store
      if (MODE == 1) __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(image + a[q]), half2_t{(_Float16)1.f, (_Float16)0.5f});
      if (MODE == 2) atomicAdd(reinterpret_cast<float*>(image) + a[q], 1.0f);                              // ds_add_f32
      if (MODE == 3) atomicAdd(image + a[q], 3u);                                                          // ds_add_u32
      if (MODE == 4) { atomicAdd(image + (a[q] & ~1u), 3u); atomicAdd(image + (a[q] | 1u), 5u); }          // two ds_add_u32 (an i32 pair per slot)
      if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(image) + (a[q] >> 1), 0x0000000300000005ull);  // ds_add_u64
    }
  }
  __syncthreads();
  uint32_t acc = 0;
  for (int i = threadIdx.x; i < kSlots; i += 1024) acc += image[i];
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
template <int MODE>
void run(const char* name, const uint32_t* idx, int per_thread, uint32_t* out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<512, 1024, kSlots * 4>>>(idx, per_thread, out);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) k<MODE><<<512, 1024, kSlots * 4>>>(idx, per_thread, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double n = 512.0 * 1024 * per_thread;
  printf("%-34s %8.3f ms  %7.1f G adds/s  (%.2f clocks per lane per CU at 2.4 GHz)\n", name, ms, n / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / n);
}
int main() {
  const int per_thread = 64;  // 512 workgroups x 1024 threads x 64 = 33.5 M adds (a focal step has 17.3 M)
  const size_t n = (size_t)512 * 1024 * per_thread;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 20); }
  uint32_t *d, *out; hipMalloc(&d, n * 4); hipMalloc(&out, 4096); hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  run<0>("plain ds_write_b32", d, per_thread, out);
  run<1>("ds_pk_add_f16", d, per_thread, out);
  run<2>("ds_add_f32", d, per_thread, out);
  run<3>("ds_add_u32", d, per_thread, out);
  run<4>("2 x ds_add_u32", d, per_thread, out);
  run<5>("ds_add_u64", d, per_thread, out);
  return 0;
}
