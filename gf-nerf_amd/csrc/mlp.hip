// mlp.hip -- the tiny MLPs (gfnerf/mlp.py:35-57: nn.Linear + ReLU / Sigmoid) on the
// gfx950 matrix cores, exact f32: v_mfma_f32_32x32x2_f32 is bit-for-bit a
// k-ordered fmaf chain (157 TFLOP/s dense peak), so results stay within 1e-6 of
// the fp32 cuBLAS/ATen path the reference runs (nerfacto_field.py:455,555).
//
// Tiling is for 64-wide waves: a wave owns 32 SAMPLES (MFMA column j = lane&31)
// and walks the output features in 32-row tiles (MFMA row i); the weight image
// lives in LDS as Wl[k][out] so the A operand is one conflict-free ds_read_b32,
// and each lane keeps HALF of its sample's input row in registers (lane half
// h = lane>>5 holds k in [h*K/2, (h+1)*K/2)), which makes the B operand a plain
// register: MFMA step s pairs k = s (h=0 lanes) with k = K/2+s (h=1 lanes).
#include "gfn_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}
__device__ __forceinline__ float act_bwd(float dy, float y, int act) {
  if (act == ACT_RELU) return y > 0.f ? dy : 0.f;
  if (act == ACT_SIGMOID) return dy * y * (1.f - y);
  return dy;
}

// y = act(x Wl + b) for 32-sample tiles.  Wl[k][o] = wsrc[o*w_so + k*w_sk]:
//   forward      : Wl = W^T      (w_so = in_dim, w_sk = 1)
//   input grad   : x := dz = dy*act'(yprev) (in_act / yin), Wl = W (w_so = 1, w_sk = ld)
template <int KP>  // in_dim rounded up to a multiple of 8
__global__ __launch_bounds__(256) void linear_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ x, int ldx,
    const float* __restrict__ yin, int ldyin, int in_act, const float* __restrict__ wsrc, int w_so,
    int w_sk, const float* __restrict__ bias, int in_dim, int out_dim, int act, float* __restrict__ y,
    int ldy) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];  // [KP][OP]
  const int OP = (out_dim + 31) & ~31;
  for (int e = threadIdx.x; e < KP * OP; e += blockDim.x) {
    const int k = e / OP, o = e - k * OP;
    Wl[e] = (k < in_dim && o < out_dim) ? wsrc[(int64_t)o * w_so + (int64_t)k * w_sk] : 0.f;
  }
  __syncthreads();
  const int64_t n = gfn_live_n(n_max, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, j = lane & 31;
  constexpr int KH = KP / 2;
  const int64_t n_tiles = (n + 31) / 32;
  // rows may be wider than in_dim (e.g. the 63-wide head input stored with ld 64): columns in
  // [in_dim, KP) meet zero weights in Wl, so reading them is harmless as long as they are finite
  const bool vec_ok = (ldx % 4 == 0) && (ldx >= KP) && (!yin || (ldyin % 4 == 0 && ldyin >= KP)) &&
                      ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (!yin || (reinterpret_cast<uintptr_t>(yin) & 15) == 0);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t s = tile * 32 + j;
    const bool valid = s < n;
    float xr[KH];
    if (valid && vec_ok) {
      const float4* xp = reinterpret_cast<const float4*>(x + s * ldx + h * KH);
#pragma unroll
      for (int c = 0; c < KH / 4; c++) {
        float4 v = xp[c];
        xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
      }
      if (yin) {
        const float4* yp = reinterpret_cast<const float4*>(yin + s * ldyin + h * KH);
#pragma unroll
        for (int c = 0; c < KH / 4; c++) {
          float4 v = yp[c];
          xr[4 * c] = act_bwd(xr[4 * c], v.x, in_act); xr[4 * c + 1] = act_bwd(xr[4 * c + 1], v.y, in_act);
          xr[4 * c + 2] = act_bwd(xr[4 * c + 2], v.z, in_act); xr[4 * c + 3] = act_bwd(xr[4 * c + 3], v.w, in_act);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < KH; c++) {
        const int k = h * KH + c;
        float v = 0.f;
        if (valid && k < in_dim) {
          v = x[s * ldx + k];
          if (yin) v = act_bwd(v, yin[s * ldyin + k], in_act);
        }
        xr[c] = v;
      }
    }
    for (int ot = 0; ot < OP / 32; ot++) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        acc[r] = (bias && o < out_dim) ? bias[o] : 0.f;
      }
      const float* wl = Wl + (h * KH) * OP + ot * 32 + j;
#pragma unroll
      for (int c = 0; c < KH; c++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[c * OP], xr[c], acc, 0, 0, 0);
      }
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int o0 = ot * 32 + 8 * q + 4 * h;
          float4 v = make_float4(act_fwd(acc[4 * q], act), act_fwd(acc[4 * q + 1], act),
                                 act_fwd(acc[4 * q + 2], act), act_fwd(acc[4 * q + 3], act));
          float* dst = y + s * ldy + o0;
          if (o0 + 3 < out_dim && (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0)) {
            *reinterpret_cast<float4*>(dst) = v;
          } else {
            if (o0 < out_dim) dst[0] = v.x;
            if (o0 + 1 < out_dim) dst[1] = v.y;
            if (o0 + 2 < out_dim) dst[2] = v.z;
            if (o0 + 3 < out_dim) dst[3] = v.w;
          }
        }
      }
    }
  }
}

// dW[o][k] += sum_s dz[s][o] x[s][k],  db[o] += sum_s dz[s][o],  dz = dy*act'(y).
// The contraction runs over SAMPLES: 64-sample tiles of dz and x are staged once
// in LDS; wave w owns output tiles t = w, w+4, ... of the (OP/32 x KP/32) grid.
constexpr int kTS = 64;
__global__ __launch_bounds__(256) void wgrad_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ x, int ldx,
    const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy, int in_dim,
    int out_dim, int act, float* __restrict__ dW, float* __restrict__ db) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int OP = (out_dim + 31) & ~31, KP = (in_dim + 31) & ~31;
  float* dzs = lds;             // [kTS][OP]
  float* xs = lds + kTS * OP;   // [kTS][KP]
  const int64_t n = gfn_live_n(n_max, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int n_ot = OP / 32, n_kt = KP / 32, n_out_tiles = n_ot * n_kt;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
  float bsum = 0.f;
  const int64_t n_tiles = (n + kTS - 1) / kTS;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t s0 = tile * kTS;
    for (int e = threadIdx.x; e < kTS * OP; e += blockDim.x) {
      const int r = e / OP, o = e - r * OP;
      const int64_t s = s0 + r;
      float v = 0.f;
      if (s < n && o < out_dim) v = act_bwd(dy[s * lddy + o], y[s * ldy + o], act);
      dzs[e] = v;
    }
    for (int e = threadIdx.x; e < kTS * KP; e += blockDim.x) {
      const int r = e / KP, k = e - r * KP;
      const int64_t s = s0 + r;
      xs[e] = (s < n && k < in_dim) ? x[s * ldx + k] : 0.f;
    }
    __syncthreads();
    if ((int)threadIdx.x < OP) {
      float a = 0.f;
      for (int r = 0; r < kTS; r++) a += dzs[r * OP + threadIdx.x];
      bsum += a;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int ti = wave + 4 * t;
      if (ti < n_out_tiles) {
        const int ot = ti / n_kt, kt = ti - ot * n_kt;
        const float* ap = dzs + h * OP + ot * 32 + j;  // A[i=o][kk=sample]
        const float* bp = xs + h * KP + kt * 32 + j;   // B[kk=sample][j=k]
#pragma unroll 8
        for (int p = 0; p < kTS / 2; p++) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * p * OP], bp[2 * p * KP], acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (db && (int)threadIdx.x < out_dim) atomicAdd(db + threadIdx.x, bsum);
  if (dW) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int ti = wave + 4 * t;
      if (ti < n_out_tiles) {
        const int ot = ti / n_kt, kt = ti - ot * n_kt;
        const int k = kt * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (o < out_dim && k < in_dim) atomicAdd(dW + (int64_t)o * in_dim + k, acc[t][r]);
        }
      }
    }
  }
}

template <int KP>
int launch_linear(int64_t n_max, const int32_t* n_dev, const float* x, int ldx, const float* yin, int ldyin,
                  int in_act, const float* wsrc, int w_so, int w_sk, const float* bias, int in_dim, int out_dim,
                  int act, float* y, int ldy, hipStream_t st) {
  const int OP = (out_dim + 31) & ~31;
  const size_t lds_bytes = (size_t)KP * OP * sizeof(float);
  int64_t tiles = (n_max + 31) / 32;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 512) blocks = 512;  // persistent: the LDS weight image is built once per workgroup
  if (blocks < 1) blocks = 1;
  linear_kernel<KP><<<(unsigned)blocks, 256, lds_bytes, st>>>(n_max, n_dev, x, ldx, yin, ldyin, in_act, wsrc, w_so, w_sk,
                                                             bias, in_dim, out_dim, act, y, ldy);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int dispatch_linear(int64_t n_max, const int32_t* n_dev, const float* x, int ldx, const float* yin, int ldyin,
                    int in_act, const float* wsrc, int w_so, int w_sk, const float* bias, int in_dim, int out_dim,
                    int act, float* y, int ldy, hipStream_t st) {
#define GFN_LIN(KP)                                                                                              \
  return launch_linear<KP>(n_max, n_dev, x, ldx, yin, ldyin, in_act, wsrc, w_so, w_sk, bias, in_dim, out_dim, act, \
                           y, ldy, st)
  if (in_dim <= 8) GFN_LIN(8);
  if (in_dim <= 16) GFN_LIN(16);
  if (in_dim <= 32) GFN_LIN(32);
  if (in_dim <= 64) GFN_LIN(64);
  GFN_LIN(128);
#undef GFN_LIN
}

}  // namespace

extern "C" {

int gfn_linear_fwd(const float* x, int ldx, const float* W, const float* b, int64_t n_max,
                   const int32_t* n_dev, int in_dim, int out_dim, int act, float* y, int ldy,
                   void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(x && W && y, "gfn_linear_fwd: null pointer");
  GFN_CHECK_ARG(in_dim >= 1 && in_dim <= 128 && out_dim >= 1 && out_dim <= 128,
                "gfn_linear_fwd: dims %dx%d unsupported (1..128)", in_dim, out_dim);
  GFN_CHECK_ARG(ldx >= in_dim && ldy >= out_dim && act >= 0 && act <= 2, "gfn_linear_fwd: bad ld/act");
  GFN_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31), "gfn_linear_fwd: n_max=%lld", (long long)n_max);
  if (n_max == 0) return GFN_OK;
  return dispatch_linear(n_max, n_dev, x, ldx, nullptr, 0, 0, W, in_dim, 1, b, in_dim, out_dim, act, y, ldy,
                         gfn_stream(stream));
}

int gfn_linear_bwd(const float* x, int ldx, const float* W, const float* y, int ldy,
                   const float* dy, int lddy, int64_t n_max, const int32_t* n_dev, int in_dim,
                   int out_dim, int act, float* dx, int lddx, float* dW, float* db, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(x && W && y && dy, "gfn_linear_bwd: null pointer");
  GFN_CHECK_ARG(in_dim >= 1 && in_dim <= 128 && out_dim >= 1 && out_dim <= 128,
                "gfn_linear_bwd: dims %dx%d unsupported (1..128)", in_dim, out_dim);
  GFN_CHECK_ARG(ldx >= in_dim && ldy >= out_dim && lddy >= out_dim && (!dx || lddx >= in_dim) && act >= 0 && act <= 2,
                "gfn_linear_bwd: bad ld/act");
  GFN_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31), "gfn_linear_bwd: n_max=%lld", (long long)n_max);
  if (n_max == 0) return GFN_OK;
  hipStream_t st = gfn_stream(stream);
  if (dx) {
    // dx = dz W : "input" dz[n,out_dim] (dy with act'(y) applied on load), weight image Wl[o][k] = W[o][k]
    int rc = dispatch_linear(n_max, n_dev, dy, lddy, y, ldy, act, W, 1, in_dim, nullptr, out_dim, in_dim, ACT_NONE,
                             dx, lddx, st);
    if (rc) return rc;
  }
  if (dW || db) {
    const int OP = (out_dim + 31) & ~31, KP = (in_dim + 31) & ~31;
    const size_t lds_bytes = (size_t)kTS * (OP + KP) * sizeof(float);
    int64_t blocks = (n_max + kTS - 1) / kTS;
    if (blocks > 512) blocks = 512;
    wgrad_kernel<<<(unsigned)blocks, 256, lds_bytes, st>>>(n_max, n_dev, x, ldx, y, ldy, dy, lddy, in_dim, out_dim, act,
                                                          dW, db);
    GFN_CHECK_LAUNCH();
  }
  return GFN_OK;
}

}  // extern "C"
