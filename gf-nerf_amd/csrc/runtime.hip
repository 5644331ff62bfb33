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
}

void gfn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* gfn_last_error(void) { return g_err; }

int gfn_version(void) { return 100; /* 0.1.0 */ }

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
