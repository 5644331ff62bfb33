// runtime.hip -- error reporting and device probing for the C ABI.
#include <stdarg.h>
#include <string.h>

#include "gfn_common.h"

namespace {
thread_local char g_err[512] = "";

// Local step of the direct all-reduce (gfnerf_amd/dist.py): recv holds n_shards copies of this rank's shard, one per
// peer; out = (sum of the copies, in f32, in rank order) * scale.  8 elements per lane and copy, 16-byte (f16) / 32-byte
// (f32) loads; HBM stream.
template <typename T>
__global__ __launch_bounds__(256) void reduce_shards_kernel(const T* __restrict__ recv, int n_shards, int64_t shard, float scale,
                                                            T* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= shard) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < n_shards; s++) {
    T v[8];
    __builtin_memcpy(v, recv + (int64_t)s * shard + i, sizeof(v));
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] += (float)v[k];
  }
  T r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) r[k] = (T)(acc[k] * scale);
  __builtin_memcpy(out + i, r, sizeof(r));
}

// gfn_zero_async: 16-byte stores over the aligned body, single bytes at the ragged ends
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* __restrict__ p, size_t head, size_t body16, size_t tail) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  uint4* b = reinterpret_cast<uint4*>(p + head);
  for (size_t i = tid; i < body16; i += stride) b[i] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < head) p[tid] = 0;
  if (tid < tail) p[head + body16 * 16 + tid] = 0;
}

// Several small device-to-device copies in ONE launch (the per-step batch hand-over of a captured training step: four tensors of
// 16 - 48 KB; as four memcpy nodes they cost the stream 4 x 6 us of launch latency in front of every replay).
constexpr int kMaxCopies = 8;
struct CopyList {
  const uint32_t* src[kMaxCopies];
  uint32_t* dst[kMaxCopies];
  int64_t words[kMaxCopies];
  int n;
};
__global__ __launch_bounds__(256) void copy_list_kernel(CopyList L) {
  const int which = blockIdx.y;
  if (which >= L.n) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L.words[which]; i += (int64_t)gridDim.x * 256) L.dst[which][i] = L.src[which][i];
}

// Philox4x32-10 (Salmon et al., SC'11): counter (index lo, index hi, call, stream), key from the seed.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// The random draws of one training step in one launch: march noise (u + add) * mul, u in [0, 1) with 24 random bits like torch.rand
// (PersSampler_cuda.cu:380-389), and 63-bit sort keys whose argsort is a uniform random permutation (S3IM's randperm, losses.py:780-789).
// state[0] = number of calls so far, advanced by the last workgroup to finish (every workgroup has read it by then); state[1] = the
// workgroups that have finished.  A captured step replays this launch: the call number, read from memory, is what moves on.
__global__ __launch_bounds__(256) void step_draws_kernel(uint64_t seed, unsigned long long* __restrict__ state, float* __restrict__ noise,
                                                         int64_t n_noise, float add, float mul, long long* __restrict__ keys, int64_t n_keys) {
  const unsigned long long call = state[0];
  const int64_t quads = (n_noise + 3) / 4, pairs = (n_keys + 1) / 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads + pairs; q += (int64_t)gridDim.x * 256) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)call, (uint32_t)(call >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    if (q < quads) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (4 * q + k < n_noise) noise[4 * q + k] = ((float)(c[k] >> 8) * (1.0f / 16777216.0f) + add) * mul;
    } else {
      const int64_t j = 2 * (q - quads);
      keys[j] = (long long)((((uint64_t)c[0] << 32) | c[1]) >> 1);
      if (j + 1 < n_keys) keys[j + 1] = (long long)((((uint64_t)c[2] << 32) | c[3]) >> 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&state[1], 1ull) == (unsigned long long)gridDim.x - 1) { state[0] = call + 1; state[1] = 0; }
  }
}
}

void gfn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

hipError_t gfn_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  unsigned char* c = static_cast<unsigned char*>(p);
  size_t head = (16 - (reinterpret_cast<uintptr_t>(c) & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t body16 = (bytes - head) / 16, tail = bytes - head - body16 * 16;
  size_t blocks = (body16 + 1023) / 1024;  // four stores per lane
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  zero_bytes_kernel<<<(unsigned)blocks, 256, 0, stream>>>(c, head, body16, tail);
  return hipGetLastError();
}

extern "C" {

const char* gfn_last_error(void) { return g_err; }

int gfn_version(void) { return GFN_ABI_VERSION; }

int gfn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

int gfn_reduce_shards(const void* recv, int n_shards, int64_t shard_elems, int is_f16, float scale, void* out, void* stream) {
  if (shard_elems == 0) return GFN_OK;
  GFN_CHECK_ARG(recv && out && n_shards > 0 && shard_elems > 0 && shard_elems % 8 == 0,
                "gfn_reduce_shards: shard_elems=%lld must be a positive multiple of 8", (long long)shard_elems);
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(recv) & 31) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "gfn_reduce_shards: buffers must be 32-byte (recv) / 16-byte (out) aligned");
  const unsigned grid = (unsigned)((shard_elems / 8 + 255) / 256);
  if (is_f16)
    reduce_shards_kernel<_Float16><<<grid, 256, 0, gfn_stream(stream)>>>(reinterpret_cast<const _Float16*>(recv), n_shards, shard_elems, scale,
                                                                         reinterpret_cast<_Float16*>(out));
  else
    reduce_shards_kernel<float><<<grid, 256, 0, gfn_stream(stream)>>>(reinterpret_cast<const float*>(recv), n_shards, shard_elems, scale,
                                                                      reinterpret_cast<float*>(out));
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_step_draws(uint64_t seed, uint64_t* state, float* noise, int64_t n_noise, float add, float mul, int64_t* keys, int64_t n_keys, void* stream) {
  GFN_CHECK_ARG(state && n_noise >= 0 && n_keys >= 0 && (noise || n_noise == 0) && (keys || n_keys == 0), "gfn_step_draws: bad arguments");
  const int64_t work = (n_noise + 3) / 4 + (n_keys + 1) / 2;
  if (work == 0) return GFN_OK;
  int64_t blocks = (work + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  step_draws_kernel<<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(seed, reinterpret_cast<unsigned long long*>(state), noise, n_noise, add, mul,
                                                                    reinterpret_cast<long long*>(keys), n_keys);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_copy_list(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream) {
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(src && dst && bytes && n > 0 && n <= kMaxCopies, "gfn_copy_list: 1..%d copies", kMaxCopies);
  CopyList L{};
  int64_t most = 0;
  for (int i = 0; i < n; i++) {
    GFN_CHECK_ARG(src[i] && dst[i] && bytes[i] >= 0 && bytes[i] % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(src[i]) | reinterpret_cast<uintptr_t>(dst[i])) & 3) == 0,
                  "gfn_copy_list: copy %d needs non-null 4-byte aligned pointers and a multiple of 4 bytes", i);
    L.src[i] = static_cast<const uint32_t*>(src[i]); L.dst[i] = static_cast<uint32_t*>(dst[i]); L.words[i] = bytes[i] / 4;
    if (L.words[i] > most) most = L.words[i];
  }
  L.n = n;
  int64_t bx = (most + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  copy_list_kernel<<<dim3((unsigned)bx, (unsigned)n), 256, 0, gfn_stream(stream)>>>(L);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
