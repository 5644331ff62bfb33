// field.hip -- the whole decoder of GFNeRFField in one kernel per direction.
//
// Forward  (nerfacto_field.py:455-555):  hash features -> base_network (32->128->16) ->
//   sigma = trunc_exp(h0+1) and [SH(16) | geo(15) | appearance(32)] -> mlp_head
//   (63->128->128->3, sigmoid) -> rgb.   Nothing but sigma, rgb, h0 and three 128-bit ReLU
//   masks per sample leaves the chip (the reference materialises every activation,
//   the concat and the boolean-mask copies in HBM at 1024 padded samples per ray).
// Backward w.r.t. the hash features (the focal stage: decoder frozen, nerfacto_field.py:
//   460-461,549): (d sigma, d rgb) -> d feat through the transposed layers, using the masks.
//
// Exact f32 on v_mfma_f32_32x32x2_f32.  A wave owns 32 samples (MFMA column j = lane&31);
// weights are the A operand from LDS images (146 KiB forward, 106 KiB backward: one
// 256-thread workgroup per CU, one wave per SIMD with the full 512-entry register file --
// a dependent chain of this MFMA already issues back to back), and every layer's 32x32 accumulator
// tile IS the next layer's B operand: lane (h = lane>>5, j) holds rows
// (r&3) + 8(r>>2) + 4h of sample j, so MFMA step r pairs row f0(r) (h=0 lanes) with
// f0(r)+4 (h=1 lanes) and the weight image is simply read at that row -- no LDS round
// trip, no transposes, activations never leave registers.
#include "gfn_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ int feat_row(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// head layer-1 input column fed by MFMA step s on lane half h (see header comment):
//   s in [0,8)  : geo rows of the base-network tile, row = (s&3)+8(s>>2)+4h, column 15+row (row 0 = sigma channel: none)
//   s in [8,16) : SH component h*8 + (s-8)
//   s in [16,32): appearance channel h*16 + (s-16), column 31 + that
__device__ __forceinline__ int head_col(int s, int h) {
  if (s < 8) {
    const int row = (s & 3) + 8 * (s >> 2) + 4 * h;
    return row == 0 ? -1 : 15 + row;
  }
  if (s < 16) return h * 8 + (s - 8);
  return 31 + h * 16 + (s - 16);
}

struct FieldWeights {
  const float* bW0; const float* bb0;  // [128,32], [128]
  const float* bW1; const float* bb1;  // [16,128], [16]
  const float* hW0; const float* hb0;  // [128,63], [128]
  const float* hW1; const float* hb1;  // [128,128], [128]
  const float* hW2; const float* hb2;  // [3,128], [3]
};

// LDS layout of the forward kernel (floats)
constexpr int F_WB0 = 0;                      // [32][128]   Wb0img[k][o] = bW0[o][k]
constexpr int F_WB1 = F_WB0 + 32 * 128;       // [128][32]   Wb1img[k][o] = bW1[o][k] (o < 16)
constexpr int F_WH0 = F_WB1 + 128 * 32;       // [32][2][128] step/half image of hW0
constexpr int F_WH1 = F_WH0 + 64 * 128;       // [128][128]  Wh1img[k][o] = hW1[o][k]
constexpr int F_WH2 = F_WH1 + 128 * 128;      // [128][32]   Wh2img[k][o] = hW2[o][k] (o < 3)
constexpr int F_BIAS = F_WH2 + 128 * 32;      // bb0[128] bb1[32] hb0[128] hb1[128] hb2[32]
constexpr int F_TOTAL = F_BIAS + 128 + 32 + 128 + 128 + 32;
static_assert(F_TOTAL * 4 <= 160 * 1024, "forward LDS image must fit one CU");

template <bool SAVE>
__global__ __launch_bounds__(256) void field_fwd_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ feat,
    const int* __restrict__ ray_id, const float* __restrict__ sh, const float* __restrict__ emb,
    const int* __restrict__ cam_idx, FieldWeights w, float* __restrict__ sigma, float* __restrict__ rgb,
    float* __restrict__ h0_out, unsigned long long* __restrict__ masks /* [n][3][2] */) {
  extern __shared__ __attribute__((aligned(16))) float L[];
  for (int e = threadIdx.x; e < 32 * 128; e += blockDim.x) { const int k = e >> 7, o = e & 127; L[F_WB0 + e] = w.bW0[o * 32 + k]; }
  for (int e = threadIdx.x; e < 128 * 32; e += blockDim.x) { const int k = e >> 5, o = e & 31; L[F_WB1 + e] = o < 16 ? w.bW1[o * 128 + k] : 0.f; }
  for (int e = threadIdx.x; e < 64 * 128; e += blockDim.x) {
    const int sh_ = e >> 7, o = e & 127;
    const int col = head_col(sh_ >> 1, sh_ & 1);
    L[F_WH0 + e] = col >= 0 ? w.hW0[o * 63 + col] : 0.f;
  }
  for (int e = threadIdx.x; e < 128 * 128; e += blockDim.x) { const int k = e >> 7, o = e & 127; L[F_WH1 + e] = w.hW1[o * 128 + k]; }
  for (int e = threadIdx.x; e < 128 * 32; e += blockDim.x) { const int k = e >> 5, o = e & 31; L[F_WH2 + e] = o < 3 ? w.hW2[o * 128 + k] : 0.f; }
  for (int e = threadIdx.x; e < 128; e += blockDim.x) {
    L[F_BIAS + e] = w.bb0[e];
    L[F_BIAS + 160 + e] = w.hb0[e];
    L[F_BIAS + 288 + e] = w.hb1[e];
    if (e < 32) { L[F_BIAS + 128 + e] = e < 16 ? w.bb1[e] : 0.f; L[F_BIAS + 416 + e] = e < 3 ? w.hb2[e] : 0.f; }
  }
  __syncthreads();
  const float* Bb0 = L + F_BIAS; const float* Bb1 = L + F_BIAS + 128;
  const float* Bh0 = L + F_BIAS + 160; const float* Bh1 = L + F_BIAS + 288; const float* Bh2 = L + F_BIAS + 416;

  const int64_t n = gfn_live_n(n_max, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t n_tiles = (n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t s = tile * 32 + j;
    const bool valid = s < n;
    const int64_t q = valid ? s : n - 1;
    // ---- base layer 1: 32 -> 128, ReLU.  B = this lane's half of the feature row
    float xr[16];
    {
      const float4* xp = reinterpret_cast<const float4*>(feat + q * 32 + h * 16);
#pragma unroll
      for (int c = 0; c < 4; c++) { const float4 v = xp[c]; xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w; }
    }
    f32x16 a1[4];
    unsigned long long m_a1 = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) a1[t][r] = Bb0[feat_row(t, r, h)];
      const float* wl = L + F_WB0 + (h * 16) * 128 + t * 32 + j;
#pragma unroll
      for (int c = 0; c < 16; c++) a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[c * 128], xr[c], a1[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const bool on = a1[t][r] > 0.f;
        a1[t][r] = on ? a1[t][r] : 0.f;
        m_a1 |= (unsigned long long)on << (t * 16 + r);
      }
    }
    // ---- base layer 2: 128 -> 16 (tile padded to 32 rows), no activation
    f32x16 hh;
#pragma unroll
    for (int r = 0; r < 16; r++) hh[r] = Bb1[feat_row(0, r, h)];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        hh = __builtin_amdgcn_mfma_f32_32x32x2f32(L[F_WB1 + feat_row(t, r, h) * 32 + j], a1[t][r], hh, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // sigma = trunc_exp(h0 + 1) (nerfacto_field.py:499): row 0 lives on the h=0 lanes, register 0
    if (valid && h == 0) {
      sigma[s] = expf(hh[0] + 1.0f);
      if (SAVE) h0_out[s] = hh[0];
    }
    // ---- head layer 1: [geo | SH | appearance] (63) -> 128, ReLU
    const int ray = ray_id[q];
    float xs[8], xe[16];
    {
      const float4* sp = reinterpret_cast<const float4*>(sh + (int64_t)ray * 16 + h * 8);
      const float4 s0 = sp[0], s1 = sp[1];
      xs[0] = s0.x; xs[1] = s0.y; xs[2] = s0.z; xs[3] = s0.w; xs[4] = s1.x; xs[5] = s1.y; xs[6] = s1.z; xs[7] = s1.w;
      if (emb) {
        const float4* ep = reinterpret_cast<const float4*>(emb + (int64_t)cam_idx[ray] * 32 + h * 16);
#pragma unroll
        for (int c = 0; c < 4; c++) { const float4 v = ep[c]; xe[4 * c] = v.x; xe[4 * c + 1] = v.y; xe[4 * c + 2] = v.z; xe[4 * c + 3] = v.w; }
      } else {
#pragma unroll
        for (int c = 0; c < 16; c++) xe[c] = 0.f;
      }
    }
    f32x16 c1[4];
    unsigned long long m_c1 = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) c1[t][r] = Bh0[feat_row(t, r, h)];
      const float* wl = L + F_WH0 + h * 128 + t * 32 + j;  // image row (s*2+h)
#pragma unroll
      for (int g = 0; g < 8; g++) c1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(g * 2) * 128], hh[g], c1[t], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 8; g++) c1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[((8 + g) * 2) * 128], xs[g], c1[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 16; g++) c1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[((16 + g) * 2) * 128], xe[g], c1[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const bool on = c1[t][r] > 0.f;
        c1[t][r] = on ? c1[t][r] : 0.f;
        m_c1 |= (unsigned long long)on << (t * 16 + r);
      }
    }
    // ---- head layer 2: 128 -> 128, ReLU
    f32x16 c2[4];
    unsigned long long m_c2 = 0;
#pragma unroll
    for (int ot = 0; ot < 4; ot++) {
#pragma unroll
      for (int r = 0; r < 16; r++) c2[ot][r] = Bh1[feat_row(ot, r, h)];
#pragma unroll
      for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
          c2[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(L[F_WH1 + feat_row(t, r, h) * 128 + ot * 32 + j], c1[t][r], c2[ot], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const bool on = c2[ot][r] > 0.f;
        c2[ot][r] = on ? c2[ot][r] : 0.f;
        m_c2 |= (unsigned long long)on << (ot * 16 + r);
      }
    }
    // ---- head layer 3: 128 -> 3 (padded to 32), sigmoid
    f32x16 o3;
#pragma unroll
    for (int r = 0; r < 16; r++) o3[r] = Bh2[feat_row(0, r, h)];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        o3 = __builtin_amdgcn_mfma_f32_32x32x2f32(L[F_WH2 + feat_row(t, r, h) * 32 + j], c2[t][r], o3, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (valid) {
      if (h == 0) {  // rows 0..2 = registers 0..2 of the h=0 lanes
        rgb[3 * s] = 1.f / (1.f + expf(-o3[0]));
        rgb[3 * s + 1] = 1.f / (1.f + expf(-o3[1]));
        rgb[3 * s + 2] = 1.f / (1.f + expf(-o3[2]));
      }
      if (SAVE) {
        unsigned long long* mp = masks + s * 6 + h;
        mp[0] = m_a1; mp[2] = m_c1; mp[4] = m_c2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward w.r.t. the hash features (decoder frozen).  LDS images are the weights in their NATURAL
// row-major [out][in] layout: for dX = W^T dZ the contraction index is `out`, which is exactly the
// row index an accumulator tile of dZ presents as B operand.
constexpr int B_W3 = 0;                       // [4][128]   hW2 rows 0..2 (+ zero row)
constexpr int B_W2 = B_W3 + 4 * 128;          // [128][128] hW1
constexpr int B_W1 = B_W2 + 128 * 128;        // [128][32]  hW0[:, 0:32]  (columns 16..30 = geo)
constexpr int B_WB1 = B_W1 + 128 * 32;        // [16][128]  bW1
constexpr int B_WB0 = B_WB1 + 16 * 128;       // [128][32]  bW0
constexpr int B_TOTAL = B_WB0 + 128 * 32;
static_assert(B_TOTAL * 4 <= 160 * 1024, "backward LDS image must fit one CU");

__global__ __launch_bounds__(256) void field_bwd_dx_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ d_sigma,
    const float* __restrict__ d_rgb, const float* __restrict__ rgb, const float* __restrict__ h0,
    const unsigned long long* __restrict__ masks, FieldWeights w, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) float L[];
  for (int e = threadIdx.x; e < 4 * 128; e += blockDim.x) L[B_W3 + e] = e < 3 * 128 ? w.hW2[e] : 0.f;
  for (int e = threadIdx.x; e < 128 * 128; e += blockDim.x) L[B_W2 + e] = w.hW1[e];
  for (int e = threadIdx.x; e < 128 * 32; e += blockDim.x) { const int o = e >> 5, c = e & 31; L[B_W1 + e] = w.hW0[o * 63 + c]; }
  for (int e = threadIdx.x; e < 16 * 128; e += blockDim.x) L[B_WB1 + e] = w.bW1[e];
  for (int e = threadIdx.x; e < 128 * 32; e += blockDim.x) L[B_WB0 + e] = w.bW0[e];
  __syncthreads();
  const int64_t n = gfn_live_n(n_max, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t n_tiles = (n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t s = tile * 32 + j;
    const bool valid = s < n;
    const int64_t q = valid ? s : n - 1;
    float ds = valid ? d_sigma[q] : 0.f;
    float dz3[3];
    bool any_grad = ds != 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float y = rgb[3 * q + c];
      const float g = valid ? d_rgb[3 * q + c] : 0.f;
      dz3[c] = g * y * (1.f - y);  // sigmoid'
      any_grad = any_grad || g != 0.f;
    }
    float4* dst = reinterpret_cast<float4*>(dfeat + s * 32);
    if (!__any(any_grad)) {  // wave-uniform: every sample of the tile lies behind its ray's termination point
      if (valid) {
#pragma unroll
        for (int qd = 0; qd < 4; qd++) dst[2 * qd + h] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    const unsigned long long* mp = masks + q * 6 + h;
    const unsigned long long m_a1 = mp[0], m_c1 = mp[2], m_c2 = mp[4];
    // ---- d c2 = hW2^T dz3   (contraction over 3 outputs, padded to 4 = two k-steps)
    const float b0 = h == 0 ? dz3[0] : dz3[1];
    const float b1 = h == 0 ? dz3[2] : 0.f;
    f32x16 g2[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) g2[t][r] = 0.f;
      g2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(L[B_W3 + h * 128 + t * 32 + j], b0, g2[t], 0, 0, 0);
      g2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(L[B_W3 + (2 + h) * 128 + t * 32 + j], b1, g2[t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; r++) g2[t][r] = ((m_c2 >> (t * 16 + r)) & 1ull) ? g2[t][r] : 0.f;  // ReLU'
    }
    // ---- d c1 = hW1^T dz2
    f32x16 g1[4];
#pragma unroll
    for (int ot = 0; ot < 4; ot++) {
#pragma unroll
      for (int r = 0; r < 16; r++) g1[ot][r] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
          g1[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(L[B_W2 + feat_row(t, r, h) * 128 + ot * 32 + j], g2[t][r], g1[ot], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) g1[ot][r] = ((m_c1 >> (ot * 16 + r)) & 1ull) ? g1[ot][r] : 0.f;
    }
    // ---- d xh[:, 0:32] = hW0[:, 0:32]^T dz1 ; only columns 16..30 (geo) continue into the base network
    f32x16 gx;
#pragma unroll
    for (int r = 0; r < 16; r++) gx[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        gx = __builtin_amdgcn_mfma_f32_32x32x2f32(L[B_W1 + feat_row(t, r, h) * 32 + j], g1[t][r], gx, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // d h: row 0 = d sigma * exp(clamp(h0+1)) (trunc_exp backward, activations.py:41); rows 1..15 = columns 16..30.
    // this lane holds columns 16+4h..19+4h (registers 8..11) and 24+4h..27+4h (registers 12..15)
    float xx = h0[q] + 1.0f;
    xx = fminf(fmaxf(xx, -15.f), 15.f);
    const float dh0 = ds * expf(xx);
    // ---- d a1 = bW1^T dh : contraction over the 16 rows of h as 8 k-steps (h=0 lane row, h=1 lane row):
    //   u<4: (1+u, 5+u)   u=4..6: (9+(u-4), 13+(u-4))   u=7: (12, 0)
    f32x16 ga[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) ga[t][r] = 0.f;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int row0 = u < 4 ? 1 + u : 9 + (u - 4);
        const int row1 = u < 4 ? 5 + u : (u < 7 ? 13 + (u - 4) : 0);
        const float bv = (h == 1 && u == 7) ? dh0 : gx[8 + u];
        const int row = h == 0 ? row0 : row1;
        ga[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(L[B_WB1 + row * 128 + t * 32 + j], bv, ga[t], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) ga[t][r] = ((m_a1 >> (t * 16 + r)) & 1ull) ? ga[t][r] : 0.f;
    }
    // ---- d feat = bW0^T dz  (32 outputs = one tile)
    f32x16 gf;
#pragma unroll
    for (int r = 0; r < 16; r++) gf[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        gf = __builtin_amdgcn_mfma_f32_32x32x2f32(L[B_WB0 + feat_row(t, r, h) * 32 + j], ga[t][r], gf, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (valid) {
#pragma unroll
      for (int qd = 0; qd < 4; qd++)  // rows 8qd+4h .. +3
        dst[2 * qd + h] = make_float4(gf[4 * qd], gf[4 * qd + 1], gf[4 * qd + 2], gf[4 * qd + 3]);
    }
  }
}

int check_weights(const char* fn, const FieldWeights& w) {
  GFN_CHECK_ARG(w.bW0 && w.bb0 && w.bW1 && w.bb1 && w.hW0 && w.hb0 && w.hW1 && w.hb1 && w.hW2 && w.hb2, "%s: null weight pointer", fn);
  return GFN_OK;
}

unsigned persistent_grid(int64_t n_max) {
  int64_t tiles = (n_max + 31) / 32;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 256) blocks = 256;  // one 256-thread workgroup per CU (one wave per SIMD, the whole 512-register file
                                   // each); the LDS images are built once per workgroup
  return (unsigned)(blocks > 0 ? blocks : 1);
}

}  // namespace

extern "C" {

int gfn_field_fwd(const float* feat, const int32_t* ray_id, const float* sh, const float* emb,
                  const int32_t* cam_idx, const float* const* weights /*10 pointers*/, int64_t n_max,
                  const int32_t* n_dev, float* sigma, float* rgb, float* h0, uint64_t* masks, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(feat && ray_id && sh && weights && sigma && rgb, "gfn_field_fwd: null pointer");
  GFN_CHECK_ARG(!emb || cam_idx, "gfn_field_fwd: emb given without cam_idx");
  GFN_CHECK_ARG((h0 == nullptr) == (masks == nullptr), "gfn_field_fwd: h0 and masks must be given together");
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31) / 32, "gfn_field_fwd: n_max=%lld", (long long)n_max);
  FieldWeights w{weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7], weights[8], weights[9]};
  if (int rc = check_weights("gfn_field_fwd", w)) return rc;
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen)) {
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(field_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, F_TOTAL * 4));
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(field_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, F_TOTAL * 4));
  }
  const unsigned grid = persistent_grid(n_max);
  if (masks)
    field_fwd_kernel<true><<<grid, 256, F_TOTAL * 4, gfn_stream(stream)>>>(n_max, n_dev, feat, ray_id, sh, emb, cam_idx, w, sigma, rgb, h0,
                                                                          reinterpret_cast<unsigned long long*>(masks));
  else
    field_fwd_kernel<false><<<grid, 256, F_TOTAL * 4, gfn_stream(stream)>>>(n_max, n_dev, feat, ray_id, sh, emb, cam_idx, w, sigma, rgb,
                                                                           nullptr, nullptr);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_field_bwd_dx(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0,
                     const uint64_t* masks, const float* const* weights, int64_t n_max,
                     const int32_t* n_dev, float* dfeat, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(d_sigma && d_rgb && rgb && h0 && masks && weights && dfeat, "gfn_field_bwd_dx: null pointer");
  GFN_CHECK_ARG(n_max > 0 && n_max < (1ll << 31) / 32, "gfn_field_bwd_dx: n_max=%lld", (long long)n_max);
  FieldWeights w{weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7], weights[8], weights[9]};
  if (int rc = check_weights("gfn_field_bwd_dx", w)) return rc;
  static unsigned long long attr_seen = 0;  // one bit per device: the attribute belongs to the (function, device) pair
  if (gfn_first_call_on_device(attr_seen)) {
    GFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(field_bwd_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, B_TOTAL * 4));
  }
  field_bwd_dx_kernel<<<persistent_grid(n_max), 256, B_TOTAL * 4, gfn_stream(stream)>>>(
      n_max, n_dev, d_sigma, d_rgb, rgb, h0, reinterpret_cast<const unsigned long long*>(masks), w, dfeat);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
