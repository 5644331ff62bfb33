// render.hip -- everything between the MLPs and the pixel: SH(4) view encoding,
// head-input assembly + density activation, front-to-back compositing with early
// termination (forward and backward), the Charbonnier loss and Adam.
//
// Replaces (reference file:line):
//   tcnn SphericalHarmonics(4) call      gfnerf/nerfacto_field.py:152-158,519-521
//   trunc_exp / concat glue              gfnerf/nerfacto_field.py:491-503,540-547
//   RaySamples.get_weights_f2nerf        nerfstudio/cameras/rays.py:178-200
//   RGB/Accumulation/Depth renderers     nerfstudio/model_components/renderers.py:97-98,216,279-281
//   CharbonnierLoss                      nerfstudio/model_components/losses.py:73-84
//   torch.optim.Adam (config.py:130)     one fused elementwise kernel
#include <math.h>

#include "gfn_common.h"

namespace {

// ---- SH degree 4, tiny-cuda-nn convention ---------------------------------
__global__ void sh4_kernel(const float* __restrict__ dirs, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // get_normalized_directions (nerfacto_field.py:64-70) then tcnn maps [0,1] -> [-1,1]
  float x = (dirs[3 * i] + 1.0f) * 0.5f, y = (dirs[3 * i + 1] + 1.0f) * 0.5f, z = (dirs[3 * i + 2] + 1.0f) * 0.5f;
  x = x * 2.f - 1.f; y = y * 2.f - 1.f; z = z * 2.f - 1.f;
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  float o[16];
  o[0] = 0.28209479177387814f;
  o[1] = -0.4886025119029199f * y;
  o[2] = 0.4886025119029199f * z;
  o[3] = -0.4886025119029199f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.9461746957575601f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.5462742152960396f * x2 - 0.5462742152960396f * y2;
  o[9] = -0.5900435899266435f * y * (3.f * x2 - y2);
  o[10] = 2.890611442640554f * xy * z;
  o[11] = -0.4570457994644658f * y * (5.f * z2 - 1.f);
  o[12] = 0.3731763325901154f * z * (5.f * z2 - 3.f);
  o[13] = -0.4570457994644658f * x * (5.f * z2 - 1.f);
  o[14] = 1.445305721320277f * z * (x2 - y2);
  o[15] = -0.5900435899266435f * x * (x2 - 3.f * y2);
#pragma unroll
  for (int k = 0; k < 16; k++) out[16 * i + k] = gfn_round_f16(o[k]);  // tcnn emits __half
}

// ---- head input assembly: one wave per sample row (64 columns) -------------
__global__ __launch_bounds__(256) void head_pack_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ h,
    const float* __restrict__ sh, const float* __restrict__ emb, const int* __restrict__ cam_idx,
    const int* __restrict__ ray_id, float* __restrict__ sigma, float* __restrict__ xh) {
  const int64_t n = gfn_live_n(n_max, n_dev);
  const int col = threadIdx.x & 63;
  for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n; s += (int64_t)gridDim.x * 4) {
    const int ray = ray_id[s];
    float v;
    if (col < 16) v = sh[(int64_t)ray * 16 + col];
    else if (col < 31) v = h[s * 16 + (col - 15)];
    else if (col < 63) v = emb ? emb[(int64_t)cam_idx[ray] * 32 + (col - 31)] : 0.f;
    else v = 0.f;
    xh[s * 64 + col] = v;
    if (col == 0) sigma[s] = expf(h[s * 16] + 1.0f);  // trunc_exp(h0 + 1), nerfacto_field.py:499
  }
}

// backward of head_pack, sample-parallel part: 16 lanes per sample write dh[s, 0:16]
__global__ __launch_bounds__(256) void head_unpack_bwd_kernel(
    int64_t n_max, const int32_t* __restrict__ n_dev, const float* __restrict__ h,
    const float* __restrict__ dsigma, const float* __restrict__ dxh, float* __restrict__ dh) {
  const int64_t n = gfn_live_n(n_max, n_dev);
  const int c = threadIdx.x & 15;
  for (int64_t s = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); s < n; s += (int64_t)gridDim.x * 16) {
    float v;
    if (c == 0) {
      float xx = h[s * 16] + 1.0f;
      xx = fminf(fmaxf(xx, -15.f), 15.f);  // activations.py:41
      v = dsigma[s] * expf(xx);
    } else {
      v = dxh[s * 64 + 15 + c];
    }
    dh[s * 16 + c] = v;
  }
}

// embedding gradient: the 32 appearance channels are per-RAY constants, so reduce over the
// ray's samples first (one wave per ray, half-waves take alternate samples) and issue one
// 128-B atomic row per ray instead of one per sample.
__global__ __launch_bounds__(256) void emb_grad_kernel(const int* __restrict__ pts_offsets, int n_rays,
                                                       const float* __restrict__ dxh,
                                                       const int* __restrict__ cam_idx,
                                                       float* __restrict__ demb) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int s = pts_offsets[ray], e = pts_offsets[ray + 1];
  const int c = lane & 31, hh = lane >> 5;
  float acc = 0.f;
  for (int i = s + hh; i < e; i += 2) acc += dxh[(int64_t)i * 64 + 31 + c];
  acc += __shfl_xor(acc, 32, 64);
  if (hh == 0 && e > s) atomicAdd(demb + (int64_t)cam_idx[ray] * 32 + c, acc);
}

// ---- compositing -------------------------------------------------------------
__device__ __forceinline__ float nan_to_num(float v) {
  if (isnan(v)) return 0.f;
  if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    float u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void tminmax_init_kernel(float* t_minmax) {
  t_minmax[0] = INFINITY;
  t_minmax[1] = 0.f;  // t >= 0 always (global_near = 0.01)
}

// The termination-aware (phased) training forward: phase A looks up and decodes only the first K samples of every ray, composites
// them, and finds the rays that are still transparent behind sample K; phase B evaluates the rest of those rays only (nothing, as a
// rule); a planning kernel then picks the next step's K from where this batch's rays ended.  Everything is decided on the device.
struct PhaseArgs {
  const uint8_t* ray_gate;  // second pass: composite only the rays with a non-zero byte here (the others keep their results)
  const int* gate;          // ... and nothing at all when *gate == 0
  const int* k_split;       // first pass (non-null): K
  uint8_t* need_more;       // [R] out (first pass): the ray is longer than K samples and not yet terminated behind them
  int* term;                // [R] out (either pass): samples in front of the ray's termination point (all of them if it never ends)
};

// one wave per ray; lanes sweep the ray's samples 64 at a time (front to back)
template <bool PHASED>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const int* __restrict__ pts_offsets, int n_rays, const float* __restrict__ sigma,
    const float* __restrict__ dt, const float* __restrict__ t, const float* __restrict__ rgb,
    float early_T, float* __restrict__ weights, float* __restrict__ alphas,
    float* __restrict__ out_rgb, float* __restrict__ out_acc, float* __restrict__ out_depth,
    float* __restrict__ t_minmax, PhaseArgs P) {
  __shared__ float s_tmin[4], s_tmax[4];
  if (PHASED && P.gate && *P.gate == 0) return;  // grid-uniform
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wave;
  bool has_ray = ray < n_rays;  // wave-uniform; idle waves still join the workgroup's min / max
  if (PHASED && P.ray_gate && has_ray && P.ray_gate[ray] == 0) has_ray = false;
  const int s = has_ray ? pts_offsets[ray] : 0, e = has_ray ? pts_offsets[ray + 1] : 0;
  int n_front = 0;    // samples in front of the termination point
  float carry = 0.f;  // sum of dt*sigma of all earlier chunks
  float cr = 0.f, cg = 0.f, cb = 0.f, ca = 0.f, cd = 0.f;
  float tmin = INFINITY, tmax = 0.f;
  bool dead = false;  // wave-uniform: transmittance already below early_T
  // the chunk's six inputs are requested together and one chunk ahead: the loop is a chain of dependent round trips otherwise
  struct In { float t, dt, sg, r, g, b; };
  auto fetch = [&](int base) {
    In x = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int i = base + lane;
    if (i < e) { x.t = t[i]; x.dt = dt[i]; x.sg = sigma[i]; x.r = rgb[3 * i]; x.g = rgb[3 * i + 1]; x.b = rgb[3 * i + 2]; }
    return x;
  };
  In nxt = fetch(s);
  for (int base = s; base < e; base += 64) {
    const int i = base + lane;
    const bool v = i < e;
    const In x = nxt;
    if (base + 64 < e) nxt = fetch(base + 64);
    const float ti = x.t;
    if (v) { tmin = fminf(tmin, ti); tmax = fmaxf(tmax, ti); }
    if (dead) {
      if (v) { if (weights) weights[i] = 0.f; if (alphas) alphas[i] = 1.f - expf(-(x.dt * x.sg)); }
      continue;
    }
    const float ds = v ? x.dt * x.sg : 0.f;
    const float incl = wave_incl_scan(ds, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.f;
    const float T = expf(-(carry + excl));
    const float alpha = 1.f - expf(-ds);
    float w = nan_to_num(alpha * T);
    if (early_T > 0.f && T < early_T) w = 0.f;
    if (PHASED) n_front += __popcll(__ballot(v && !(early_T > 0.f && T < early_T)));
    if (v) {
      if (weights) weights[i] = w;
      if (alphas) alphas[i] = alpha;
      cr += w * x.r; cg += w * x.g; cb += w * x.b;
      ca += w;
      cd += w * ti;
    }
    carry += __shfl(incl, 63, 64);
    if (early_T > 0.f && expf(-carry) < early_T) dead = true;
  }
  cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); ca = wave_sum(ca); cd = wave_sum(cd);
  if (t_minmax) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      tmin = fminf(tmin, __shfl_xor(tmin, off, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
    }
    if (lane == 0) {
      if (has_ray && e - s < GFN_MAX_SAMPLE_PER_RAY) tmin = 0.f;  // the reference's dense zero padding joins the min
      s_tmin[wave] = tmin; s_tmax[wave] = tmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one pair of same-address atomics per workgroup, not per ray: they serialise at the memory side
      tmin = fminf(fminf(s_tmin[0], s_tmin[1]), fminf(s_tmin[2], s_tmin[3]));
      tmax = fmaxf(fmaxf(s_tmax[0], s_tmax[1]), fmaxf(s_tmax[2], s_tmax[3]));
      // non-negative floats order like their bit patterns
      atomicMin(reinterpret_cast<unsigned*>(t_minmax), __float_as_uint(tmin));
      atomicMax(reinterpret_cast<unsigned*>(t_minmax + 1), __float_as_uint(tmax));
    }
  }
  if (lane == 0 && has_ray) {
    out_rgb[3 * ray] = cr; out_rgb[3 * ray + 1] = cg; out_rgb[3 * ray + 2] = cb;
    if (out_acc) out_acc[ray] = ca;
    if (out_depth) out_depth[ray] = cd / (ca + 1e-10f);  // renderers.py:279
  }
  if (PHASED && has_ray) {
    if (P.term && lane == 0) P.term[ray] = n_front;
    if (P.k_split) {
      const int K = *P.k_split;
      const bool need = !dead && (e - s) > K;  // wave-uniform: still transparent behind its first K samples, and there are more
      if (lane == 0) P.need_more[ray] = need ? 1 : 0;
    }
  }
}

// The sample list of a phase: ray r contributes c_r consecutive samples from its sample `first` on --
//   phase A (need_more == null): first = 0, c_r = min(K, n_r);   phase B: first = K, c_r = need_more[r] ? n_r - K : 0;
//   the samples in front of the termination points (term != null; the backward's list): first = 0, c_r = min(term[r], n_r).
// scan: one workgroup, exclusive prefix sum of c_r over the rays -> coff i32 [R + 1]; stats[base] = entries, stats[base + 1] = rays
// with c_r > 0 (base 0 / 2 / 4: stats = {A samples, A rays, B samples, B rays, front samples, front rays}).  fill: one wave per ray.
__global__ __launch_bounds__(1024) void phase_scan_kernel(const int* __restrict__ pts_offsets, int n_rays, const int* __restrict__ k_split,
                                                          const uint8_t* __restrict__ need_more, const int* __restrict__ term,
                                                          int* __restrict__ coff, int* __restrict__ stats) {
  __shared__ int s_wave[16], s_rays[16];
  __shared__ int s_carry, s_nr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = *k_split;
  if (tid == 0) { s_carry = 0; s_nr = 0; }
  __syncthreads();
  for (int r0 = 0; r0 < n_rays; r0 += 1024) {
    const int r = r0 + tid;
    int c = 0;
    if (r < n_rays) {
      const int n = pts_offsets[r + 1] - pts_offsets[r];
      if (term) c = term[r] < n ? term[r] : n;
      else c = need_more ? ((need_more[r] && n > K) ? n - K : 0) : (n < K ? n : K);
      c = c < 0 ? 0 : c;
    }
    int v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    const int nz = __popcll(__ballot(c > 0));
    if (lane == 63) { s_wave[wave] = v; s_rays[wave] = nz; }
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    if (r < n_rays) coff[r] = before + v - c;
    __syncthreads();
    if (tid == 0) {
      int tot = 0, nr = 0;
      for (int w = 0; w < 16; w++) { tot += s_wave[w]; nr += s_rays[w]; }
      s_carry += tot; s_nr += nr;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const bool all = !need_more && !term && K >= 1024;  // phase A of a one-phase step: every sample -- no list is needed (entries = -1)
    coff[n_rays] = all ? 0 : s_carry;
    const int base = term ? 4 : (need_more ? 2 : 0);
    stats[base] = all ? -1 : s_carry; stats[base + 1] = s_nr;
  }
}

// zero_sigma / zero_rgb (phase A, optional): the samples of the ray that are NOT listed get sigma = 0 and rgb = 0 -- empty space
// with a finite colour for the compositing and the backward, which read every sample of a ray
__global__ __launch_bounds__(256) void phase_fill_kernel(const int* __restrict__ pts_offsets, int n_rays, const int* __restrict__ k_split,
                                                         int phase_b, const int* __restrict__ coff, int* __restrict__ list,
                                                         float* __restrict__ zero_sigma, float* __restrict__ zero_rgb) {
  if (coff[n_rays] == 0) return;  // grid-uniform: an empty phase, or phase A of a one-phase step (nothing to list, nothing to clear)
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int c = coff[ray + 1] - coff[ray];
  const int s0 = pts_offsets[ray], n = pts_offsets[ray + 1] - s0;
  const int first = s0 + (phase_b ? *k_split : 0);
  int* dst = list + coff[ray];
  for (int i = lane; i < c; i += 64) dst[i] = first + i;
  if (zero_sigma && !phase_b) {
    for (int i = c + lane; i < n; i += 64) {
      zero_sigma[s0 + i] = 0.f;
      zero_rgb[3 * (s0 + i)] = 0.f; zero_rgb[3 * (s0 + i) + 1] = 0.f; zero_rgb[3 * (s0 + i) + 2] = 0.f;
    }
  }
}

// Both steps in ONE launch for batches of up to kPhaseOneLaunchRays rays (a training batch: 4096): every workgroup of 16 waves = 16 rays
// sums the entries of the rays in front of its own by itself (the 16 KB of offsets come from L2; ~2 us) instead of waiting for a
// one-workgroup scan kernel -- these lists are built three times per step on the critical path, between launches that are empty as a rule.
constexpr int kPhaseOneLaunchRays = 16384;
__device__ __forceinline__ int phase_entries(const int* __restrict__ pts_offsets, int r, int K, const uint8_t* __restrict__ need_more,
                                             const int* __restrict__ term) {
  const int n = pts_offsets[r + 1] - pts_offsets[r];
  int c;
  if (term) c = term[r] < n ? term[r] : n;
  else c = need_more ? ((need_more[r] && n > K) ? n - K : 0) : (n < K ? n : K);
  return c < 0 ? 0 : c;
}

__global__ __launch_bounds__(1024) void phase_list_kernel(const int* __restrict__ pts_offsets, int n_rays, const int* __restrict__ k_split,
                                                          const uint8_t* __restrict__ need_more, const int* __restrict__ term,
                                                          int* __restrict__ coff, int* __restrict__ list, int* __restrict__ stats,
                                                          float* __restrict__ zero_sigma, float* __restrict__ zero_rgb) {
  __shared__ int s_before[16], s_total[16], s_rays[16], s_c[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = *k_split;
  const int base = blockIdx.x * 16;  // this workgroup's rays: base .. base + 15, one per wave
  int before = 0, total = 0, rays = 0;
  for (int r = tid; r < n_rays; r += 1024) {
    const int c = phase_entries(pts_offsets, r, K, need_more, term);
    total += c;
    rays += c > 0;
    if (r < base) before += c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    before += __shfl_down(before, off, 64);
    total += __shfl_down(total, off, 64);
    rays += __shfl_down(rays, off, 64);
  }
  if (lane == 0) { s_before[wave] = before; s_total[wave] = total; s_rays[wave] = rays; }
  if (lane == 1) s_c[wave] = base + wave < n_rays ? phase_entries(pts_offsets, base + wave, K, need_more, term) : 0;
  __syncthreads();
  before = 0; total = 0; rays = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { before += s_before[w]; total += s_total[w]; rays += s_rays[w]; }
  for (int w = 0; w < wave; w++) before += s_c[w];
  const bool all = !need_more && !term && K >= 1024;  // phase A of a one-phase step: every sample -- no list is needed (entries = -1)
  if (blockIdx.x == 0 && tid == 0) {
    coff[n_rays] = all ? 0 : total;
    const int sb = term ? 4 : (need_more ? 2 : 0);
    stats[sb] = all ? -1 : total; stats[sb + 1] = rays;
  }
  const int ray = base + wave;
  if (ray >= n_rays) return;
  if (lane == 0) coff[ray] = before;
  if (all || total == 0) return;  // an empty phase, or phase A of a one-phase step (nothing to list, nothing to clear)
  const int c = s_c[wave];
  const int s0 = pts_offsets[ray], n = pts_offsets[ray + 1] - s0;
  const int first = s0 + (need_more ? K : 0);
  int* dst = list + before;
  for (int i = lane; i < c; i += 64) dst[i] = first + i;
  if (zero_sigma && !need_more) {
    for (int i = c + lane; i < n; i += 64) {
      zero_sigma[s0 + i] = 0.f;
      zero_rgb[3 * (s0 + i)] = 0.f; zero_rgb[3 * (s0 + i) + 1] = 0.f; zero_rgb[3 * (s0 + i) + 2] = 0.f;
    }
  }
}

// The next step's K from this batch: with n_r = samples of ray r and f_r = samples in front of its termination point,
//     cost(K) = sum_r min(K, n_r)  +  [some f_r > K] * (fixed + sum_{r: f_r > K} (n_r - K))        (in samples)
// (phase A evaluates the first K samples of every ray; phase B, if any ray needs it, costs its fixed sweeps of the tables plus the
// rest of those rays).  K = argmin + a margin for the batch-to-batch scatter of the longest ray; one workgroup, three histograms.
__global__ __launch_bounds__(1024) void two_phase_plan_kernel(const int* __restrict__ pts_offsets, int n_rays, const int* __restrict__ term,
                                                              int* __restrict__ k_split, int* __restrict__ stats, int* __restrict__ stats_last,
                                                              float fixed_cost, float save_enter, float save_stay, int enabled) {
  // bins 0 .. 1024 (a ray has at most 1024 samples); thread k owns bin k, thread 1023 also carries bin 1024 into the totals
  __shared__ int hN[1025], hC[1025], hS[1025];
  __shared__ int pN[1024], pKN[1024], pC[1024], pS[1024];  // inclusive prefix sums over bins 0 .. k
  __shared__ int s_wave[16][4];
  __shared__ float s_best[2];
  __shared__ int s_arg[2];
  __shared__ int s_fmax, s_f99;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { s_fmax = 0; s_f99 = 1024; }
  for (int k = tid; k < 1025; k += 1024) { hN[k] = 0; hC[k] = 0; hS[k] = 0; }
  const int k_used = *k_split;
  __syncthreads();
  for (int r = tid; r < n_rays; r += 1024) {
    int n = pts_offsets[r + 1] - pts_offsets[r];
    n = n < 0 ? 0 : (n > 1024 ? 1024 : n);
    int f = term[r];
    f = f < 0 ? 0 : (f > n ? n : f);
    atomicAdd(&hN[n], 1); atomicAdd(&hC[f], 1); atomicAdd(&hS[f], n);
  }
  __syncthreads();
  int v[4] = {hN[tid], tid * hN[tid], hC[tid], hS[tid]};
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int u = __shfl_up(v[c], off, 64);
      if (lane >= off) v[c] += u;
    }
  }
  if (lane == 63) { s_wave[wave][0] = v[0]; s_wave[wave][1] = v[1]; s_wave[wave][2] = v[2]; s_wave[wave][3] = v[3]; }
  __syncthreads();
  for (int w = 0; w < wave; w++) { v[0] += s_wave[w][0]; v[1] += s_wave[w][1]; v[2] += s_wave[w][2]; v[3] += s_wave[w][3]; }
  pN[tid] = v[0]; pKN[tid] = v[1]; pC[tid] = v[2]; pS[tid] = v[3];
  __syncthreads();
  const int totN = pN[1023] + hN[1024], totC = pC[1023] + hC[1024], totS = pS[1023] + hS[1024];
  const int totKN = pKN[1023] + 1024 * hN[1024];
  // the latest termination point of the batch and the 99th percentile (every thread looks at its own bin: no serial scan)
  if (hC[tid] > 0) atomicMax(&s_fmax, tid);
  if (tid == 1023 && hC[1024] > 0) atomicMax(&s_fmax, 1024);
  if (pC[tid] >= totC - totC / 100) atomicMin(&s_f99, tid);
  // candidates K = 8 * (tid + 1), tid < 128:  A(K) = sum_r min(K, n_r);  rays / samples of phase B = those with f_r > K
  float cost = 3.0e38f;
  const int K = 8 * (tid + 1);
  if (tid < 128) {
    float a, b;
    int nb;
    if (K >= 1024) { a = (float)totKN; nb = 0; b = 0.f; }
    else {
      a = (float)pKN[K] + (float)K * (float)(totN - pN[K]);
      nb = totC - pC[K];
      b = (float)(totS - pS[K]) - (float)K * (float)nb;
    }
    cost = a + (nb > 0 ? fixed_cost + b : 0.f);
  }
  int arg = K;
  if (tid < 128) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float oc = __shfl_xor(cost, off, 64);
      const int oa = __shfl_xor(arg, off, 64);
      if (oc < cost || (oc == cost && oa < arg)) { cost = oc; arg = oa; }
    }
    if (lane == 0) { s_best[wave] = cost; s_arg[wave] = arg; }
  }
  __syncthreads();
  if (tid == 0) {
    if (s_best[1] < cost || (s_best[1] == cost && s_arg[1] < arg)) { cost = s_best[1]; arg = s_arg[1]; }
    // one phase expected (the cheapest K covers every ray of this batch): K = the latest termination point + a margin as wide as
    // the tail of this batch's distribution (latest - 99th percentile), for the next batch's rays are not this batch's;
    // two phases expected: the cheapest K as it is
    int next = arg;
    if (arg >= 1024 || totC - pC[arg] == 0) {
      const int fmax = s_fmax;
      int f99 = s_f99;
      if (f99 > fmax) f99 = fmax;
      const int margin = fmax - f99 > 2 ? fmax - f99 : 2;
      next = fmax + margin;
    }
    if (!enabled || next > 1024) next = 1024;
    // a split has to save samples to pay for itself (list indirection, the extra small launches; with activation planes also their
    // unaligned rows): a new one at least save_enter of them, a running one save_stay (no flip-flopping around the break-even
    // point: a captured step is captured again when the mode changes)
    if (next < 1024) {
      const float kept = ((float)pKN[next] + (float)next * (float)(totN - pN[next])) / fmaxf((float)totKN, 1.f);
      if (1.f - kept < (k_used >= 1024 ? save_enter : save_stay)) next = 1024;
    }
    // this step's counters (samples of phase A; rays, samples of phase B; the K used) for whoever wants to look
    stats_last[0] = stats[0] < 0 ? totKN : stats[0]; stats_last[1] = stats[3]; stats_last[2] = stats[2]; stats_last[3] = k_used;
    *k_split = next;
  }
}

__global__ void depth_finalize_kernel(float* depth, int64_t n, const float* __restrict__ t_minmax, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = depth[i];
  d = fminf(fmaxf(d, t_minmax[0]), t_minmax[1]);  // renderers.py:281 (batch-global clip)
  depth[i] = d * (1.0f / scale);                  // nerfacto.py:546: tensor / python scalar = a * (1/b) on device
}

// d(rgb,acc)/d(sigma,rgb): w_i = alpha_i T_i;  dw_i/dsigma_i = dt_i T_{i+1};
// dw_k/dsigma_i = -dt_i w_k (k>i).  One wave per ray, chunks walked BACK to front so
// the suffix sum  sum_{k>i} G_k w_k  is built from later terms only (no cancellation).
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const int* __restrict__ pts_offsets, int n_rays, const float* __restrict__ sigma,
    const float* __restrict__ dt, const float* __restrict__ rgb, const float* __restrict__ g_rgb,
    const float* __restrict__ g_acc, float early_T, float* __restrict__ d_sigma,
    float* __restrict__ d_rgb) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int s = pts_offsets[ray], e = pts_offsets[ray + 1];
  const int n_chunks = (e - s + 63) / 64;  // <= 16
  // pass 1: carry-in of dt*sigma for every chunk (lane c keeps chunk c's carry)
  float my_carry = 0.f;
  {
    float carry = 0.f;
    for (int c = 0; c < n_chunks; c++) {
      const int i = s + c * 64 + lane;
      const float ds = (i < e) ? dt[i] * sigma[i] : 0.f;
      if (lane == c) my_carry = carry;
      carry += wave_sum(ds);
    }
  }
  const float gr = g_rgb[3 * ray], gg = g_rgb[3 * ray + 1], gb = g_rgb[3 * ray + 2];
  const float ga = g_acc ? g_acc[ray] : 0.f;
  float suffix = 0.f;  // sum over later chunks of G_k w_k
  for (int c = n_chunks - 1; c >= 0; c--) {
    const int i = s + c * 64 + lane;
    const bool v = i < e;
    const float carry = __shfl(my_carry, c, 64);
    const float ds = v ? dt[i] * sigma[i] : 0.f;
    const float incl = wave_incl_scan(ds, lane);
    const float T = expf(-(carry + (incl - ds)));
    const float Tn = expf(-(carry + incl));
    float w = (1.f - expf(-ds)) * T;
    const bool cut = early_T > 0.f && T < early_T;
    if (cut || !v) w = 0.f;
    float G = 0.f;
    if (v) G = gr * rgb[3 * i] + gg * rgb[3 * i + 1] + gb * rgb[3 * i + 2] + ga;
    const float gw = G * w;
    // reverse inclusive scan within the chunk: rs = sum_{k>=lane} gw_k
    float rs = gw;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      float u = __shfl_down(rs, off, 64);
      if (lane + off < 64) rs += u;
    }
    const float later = suffix + (rs - gw);
    if (v) {
      d_sigma[i] = cut ? 0.f : dt[i] * (G * Tn - later);
      if (d_rgb) { d_rgb[3 * i] = w * gr; d_rgb[3 * i + 1] = w * gg; d_rgb[3 * i + 2] = w * gb; }
    }
    suffix += __shfl(rs, 0, 64);
  }
}

// ---- loss -------------------------------------------------------------------
__global__ __launch_bounds__(256) void charbonnier_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ gt, int64_t n_elems,
                                                          float inv_rows, float eps, float* loss,
                                                          float* __restrict__ g_pred) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - gt[i];
    const float q = sqrtf(d * d + eps * eps);
    acc += q;
    if (g_pred) g_pred[i] = d / q * inv_rows;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && loss) atomicAdd(loss, acc * inv_rows);
}

// Everything a training iteration of the plugin surface runs between the compositing and the backward, in ONE launch (a batch is a few thousand
// rays: 12 k elements): the Charbonnier loss + gradient (the arithmetic of charbonnier_kernel), the PSNR of get_metrics_dict (nerfacto.py:300,624), a
// zeroed buffer for gfn_s3im's scatter, and the block stage's error-map write-back (gf_pipeline.py:180-185 = errmap_update_kernel).  The sums are taken
// in a fixed order -- thread-strided partial sums, wave shuffles, the workgroup's four wave totals, then the workgroups' partials by the LAST workgroup
// to finish (a ticket in the caller's workspace, left at zero again) -- so the same bits every run, and the outputs are WRITTEN, not accumulated:
// nothing to zero beforehand.  ws == NULL: one workgroup does it all (no workspace; ~30 us at 4096 rays, where 48 workgroups take 6).
constexpr int kLossThreads = 256, kLossMaxBlocks = 64;
__global__ __launch_bounds__(kLossThreads) void train_losses_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int64_t rows, float eps,
                                                                   float* __restrict__ out3, float* __restrict__ g_char, float* __restrict__ g_zero,
                                                                   const int32_t* __restrict__ idx, const int32_t* __restrict__ slot, int n_images, int H, int W,
                                                                   float* __restrict__ error_map, double* __restrict__ sums, int err_block,
                                                                   float* __restrict__ ws) {
  const int64_t n = rows * 3;
  const float inv_rows = 1.0f / (float)rows;
  const int64_t stride = (int64_t)gridDim.x * kLossThreads;
  float acc = 0.f, sq = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < n; i += stride) {
    const float d = pred[i] - gt[i];
    const float q = sqrtf(d * d + eps * eps);
    acc += q;
    sq += d * d;
    if (g_char) g_char[i] = d / q * inv_rows;
    if (g_zero) g_zero[i] = 0.f;
  }
  if (error_map && idx) {
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < rows; i += stride) {
      const int c = slot ? slot[i] : idx[3 * i], y = idx[3 * i + 1], x = idx[3 * i + 2];
      if (c < 0 || c >= n_images || y < 0 || y >= H || x < 0 || x >= W) continue;
      const float e = (fabsf(gt[3 * i] - pred[3 * i]) + fabsf(gt[3 * i + 1] - pred[3 * i + 1])) + fabsf(gt[3 * i + 2] - pred[3 * i + 2]);
      const int64_t p = ((int64_t)c * H + y) * W + x;
      const float old = atomicExch(error_map + p, e);
      if (sums) atomicAdd(sums + p / err_block, (double)e - (double)old);
    }
  }
  acc = wave_sum(acc); sq = wave_sum(sq);
  __shared__ float s_acc[kLossThreads / 64], s_sq[kLossThreads / 64];
  if ((threadIdx.x & 63) == 0) { s_acc[threadIdx.x >> 6] = acc; s_sq[threadIdx.x >> 6] = sq; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float a = 0.f, b = 0.f;
  for (int w = 0; w < kLossThreads / 64; w++) { a += s_acc[w]; b += s_sq[w]; }
  if (gridDim.x > 1) {  // this workgroup's partials; the last one to arrive adds them all, in workgroup order
    ws[2 * blockIdx.x] = a; ws[2 * blockIdx.x + 1] = b;
    __threadfence();
    unsigned int* ticket = reinterpret_cast<unsigned int*>(ws + 2 * kLossMaxBlocks);
    if (atomicAdd(ticket, 1u) != gridDim.x - 1) return;
    __threadfence();
    a = 0.f; b = 0.f;
    for (unsigned int k = 0; k < gridDim.x; k++) { a += __builtin_nontemporal_load(ws + 2 * k); b += __builtin_nontemporal_load(ws + 2 * k + 1); }
    *ticket = 0u;
  }
  out3[0] = a * inv_rows;
  out3[1] = 0.f;
  out3[2] = -10.0f * log10f(b / (float)n);
}

// out = g_a * w_a[0] + g_b * w_b[0]: the gradients of two losses of one prediction under the (device-resident) gradients of the two loss values --
// what autograd hands the backward of `loss = rgb_loss + s3im_loss` (trainer.py:391) -- in one launch instead of two scalings and an add
__global__ __launch_bounds__(256) void loss_grad_combine_kernel(const float* __restrict__ g_a, const float* __restrict__ w_a, const float* __restrict__ g_b,
                                                                const float* __restrict__ w_b, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = w_a ? g_a[i] * w_a[0] : 0.f;
  if (g_b && w_b) v += g_b[i] * w_b[0];
  out[i] = v;
}

// ---- Adam ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float one_m_b1, float beta2, float one_m_b2, float step_size,
                                                   float bc2_sqrt, float eps, const float* __restrict__ hyper) {
  if (hyper) { step_size = hyper[0]; bc2_sqrt = hyper[1]; }  // (lr / bias_correction1, sqrt(bias_correction2)) of this step, from the device
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * one_m_b1;
    const float vi = v[i] * beta2 + (one_m_b2 * gi) * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// Adam over a hash table: f32 or x128-scaled f16 gradient, optional f16 shadow refresh
template <bool G16>
__global__ __launch_bounds__(256) void adam_ex_kernel(float* __restrict__ p, const void* __restrict__ g,
                                                      float g_scale, float* __restrict__ m, float* __restrict__ v,
                                                      int64_t n, float one_m_b1, float beta2, float one_m_b2,
                                                      float step_size, float bc2_sqrt, float eps,
                                                      _Float16* __restrict__ shadow, const float* __restrict__ hyper) {
  if (hyper) { step_size = hyper[0]; bc2_sqrt = hyper[1]; }
  // non-temporal accesses: 1 GB streams through once per step and must not evict what the level-major lookup of the next
  // step keeps in L2 when this kernel runs beside it (HotPathConfig.async_adam)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = G16 ? (float)__builtin_nontemporal_load(reinterpret_cast<const _Float16*>(g) + i) : __builtin_nontemporal_load(reinterpret_cast<const float*>(g) + i);
    gi = gi * g_scale;
    const float m0 = __builtin_nontemporal_load(m + i), v0 = __builtin_nontemporal_load(v + i), p0 = __builtin_nontemporal_load(p + i);
    const float mi = m0 + (gi - m0) * one_m_b1;
    const float vi = v0 * beta2 + (one_m_b2 * gi) * gi;
    __builtin_nontemporal_store(mi, m + i);
    __builtin_nontemporal_store(vi, v + i);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p0 - step_size * (mi / denom);
    __builtin_nontemporal_store(pi, p + i);
    if (shadow) shadow[i] = (_Float16)pi;
  }
}

// Cameras.generate_rays for perspective cameras without distortion (nerfstudio/cameras/cameras.py:560-734, GF's
// lookat_directions :704): pixel -> (origin, unit direction, pixel area, |d|).  Either explicit (camera, (y, x)) pairs
// (what the pixel samplers hand to RayGenerator) or a raster run of one camera's image (render.py:109), pixel p ->
// (y, x) = (p / width + 0.5, p % width + 0.5) (get_image_coords, pixel_offset 0.5).
__device__ __forceinline__ float cam_dir(const float* __restrict__ M, float cxn, float cyn, float out[3]) {
  float v[3];
#pragma unroll
  for (int i = 0; i < 3; i++) v[i] = (cxn * M[4 * i] + cyn * M[4 * i + 1]) + (-1.0f) * M[4 * i + 2];  // sum(dir * rotation, -1), :697-699
  float nrm = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  nrm = fmaxf(nrm, 8.8817841970012523e-16f);  // normalize_with_norm, camera_utils.py:28,251
#pragma unroll
  for (int i = 0; i < 3; i++) out[i] = v[i] / nrm;
  return nrm;
}

__global__ __launch_bounds__(256) void generate_rays_kernel(const float* __restrict__ c2w, const float* __restrict__ intr,
                                                           const int* __restrict__ cam_idx, int cam0, const float* __restrict__ coords,
                                                           int64_t first_pixel, int width, int64_t n, float* __restrict__ origins,
                                                           float* __restrict__ directions, float* __restrict__ lookat,
                                                           float* __restrict__ pixel_area, float* __restrict__ dir_norm) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int cam = cam_idx ? cam_idx[i] : cam0;
  const float* M = c2w + 12 * (int64_t)cam;
  const float fx = intr[4 * cam], fy = intr[4 * cam + 1], cx = intr[4 * cam + 2], cy = intr[4 * cam + 3];
  float y, x;
  if (coords) { y = coords[2 * i]; x = coords[2 * i + 1]; }
  else { const int64_t p = first_pixel + i; y = (float)(p / width) + 0.5f; x = (float)(p % width) + 0.5f; }
  const float c0x = (x - cx) / fx, c0y = -(y - cy) / fy;                        // :606
  const float c1x = (x - cx + 1.0f) / fx, c2y = -(y - cy + 1.0f) / fy;          // :607-608 (offsets for dx, dy)
  float d0[3], d1[3], d2[3];
  const float nrm = cam_dir(M, c0x, c0y, d0);
  cam_dir(M, c1x, c0y, d1);
  cam_dir(M, c0x, c2y, d2);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (origins) origins[3 * i + k] = M[4 * k + 3];
    if (lookat) lookat[3 * i + k] = M[4 * k + 2];
    directions[3 * i + k] = d0[k];
  }
  if (dir_norm) dir_norm[i] = nrm;
  if (pixel_area) {  // :712-716
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) { const float a = d0[k] - d1[k], b = d0[k] - d2[k]; sx += a * a; sy += b * b; }
    pixel_area[i] = sqrtf(sx) * sqrtf(sy);
  }
}

// S3IM (nerfstudio/model_components/losses.py:713-794; gfnerf/config.py:96-100: kernel 4, stride 4, 10 repeats,
// virtual patch height 32): the batch's rays, repeated under random permutations, are laid out as a [3, H, W] image and
// compared with a strided SSIM.  Stride == kernel size, so every virtual pixel belongs to at most one window: one thread
// per (channel, window) evaluates the window and scatters d loss / d pred to its 16 rays.
struct S3imWin { float w[4][4]; };
__global__ __launch_bounds__(256) void s3im_kernel(const float* __restrict__ gt, const float* __restrict__ pred, const int* __restrict__ index,
                                                  int64_t H, int64_t W, int64_t OH, int64_t OW, S3imWin win, float mult,
                                                  float* __restrict__ loss_dev, float* __restrict__ g_pred) {
  const int64_t cells = 3 * OH * OW;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float ssim = 0.f;
  if (id < cells) {
    const int c = (int)(id / (OH * OW));
    const int64_t rem = id - (int64_t)c * OH * OW, oy = rem / OW, ox = rem - oy * OW;
    float x1[16], x2[16], ww[16];
    int ray[16];
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int64_t y = oy * 4 - 1 + i, x = ox * 4 - 1 + j;  // padding (4-1)//2 = 1 (:741)
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const int r = in ? index[y * W + x] : -1;
        ray[4 * i + j] = r;
        const float a = in ? gt[3 * (int64_t)r + c] : 0.f, b = in ? pred[3 * (int64_t)r + c] : 0.f, wt = win.w[i][j];
        x1[4 * i + j] = a; x2[4 * i + j] = b; ww[4 * i + j] = wt;
        mu1 += wt * a; mu2 += wt * b; e11 += wt * a * a; e22 += wt * b * b; e12 += wt * a * b;
      }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
    const float inv = 1.f / (B1 * B2);
    ssim = A1 * A2 * inv;
    if (g_pred) {
      const float k = -mult / (float)cells;  // loss = mult * (1 - mean(ssim_map))
#pragma unroll
      for (int p = 0; p < 16; p++) {
        if (ray[p] < 0) continue;
        const float dA1 = 2.f * mu1 * ww[p], dA2 = 2.f * ww[p] * (x1[p] - mu1), dB1 = 2.f * mu2 * ww[p], dB2 = 2.f * ww[p] * (x2[p] - mu2);
        const float ds = (dA1 * A2 + A1 * dA2) * inv - ssim * (dB1 * B2 + B1 * dB2) * inv;
        atomicAdd(g_pred + 3 * (int64_t)ray[p] + c, k * ds);
      }
    }
  }
  // block sum of ssim -> one atomic per workgroup
  __shared__ float part[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ssim += __shfl_xor(ssim, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ssim;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (part[0] + part[1]) + (part[2] + part[3]);
    float contrib = -mult * tot / (float)cells;
    if (blockIdx.x == 0) contrib += mult;  // the "1 -" term, once
    atomicAdd(loss_dev, contrib);
  }
}

}  // namespace

extern "C" {

int gfn_sh4(const float* dirs, int64_t n_rays, float* out, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(dirs && out && n_rays > 0, "gfn_sh4: bad arguments");
  sh4_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, gfn_stream(stream)>>>(dirs, n_rays, out);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

static unsigned rows_grid(int64_t n_max, int rows_per_block) {
  int64_t b = (n_max + rows_per_block - 1) / rows_per_block;
  if (b > 256 * 64) b = 256 * 64;
  return (unsigned)(b > 0 ? b : 1);
}

int gfn_head_pack(const float* h, const float* sh, const float* emb, const int32_t* cam_idx,
                  const int32_t* ray_id, int64_t n_max, const int32_t* n_dev, float* sigma,
                  float* xh, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(h && sh && ray_id && sigma && xh, "gfn_head_pack: null pointer");
  GFN_CHECK_ARG(!emb || cam_idx, "gfn_head_pack: emb given without cam_idx");
  GFN_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31) / 64, "gfn_head_pack: n_max=%lld", (long long)n_max);
  if (n_max == 0) return GFN_OK;
  head_pack_kernel<<<rows_grid(n_max, 4), 256, 0, gfn_stream(stream)>>>(n_max, n_dev, h, sh, emb, cam_idx, ray_id, sigma, xh);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_head_unpack_bwd(const float* h, const float* dsigma, const float* dxh,
                        const int32_t* cam_idx, const int32_t* pts_offsets, int64_t n_rays,
                        int64_t n_max, const int32_t* n_dev, float* dh, float* demb, void* stream) {
  if (n_max == 0) return GFN_OK;
  GFN_CHECK_ARG(h && dsigma && dxh && dh, "gfn_head_unpack_bwd: null pointer");
  GFN_CHECK_ARG(!demb || (cam_idx && pts_offsets), "gfn_head_unpack_bwd: demb given without cam_idx/pts_offsets");
  GFN_CHECK_ARG(n_max >= 0 && n_max < (1ll << 31) / 64 && n_rays >= 0, "gfn_head_unpack_bwd: n_max=%lld", (long long)n_max);
  if (n_max == 0) return GFN_OK;
  hipStream_t st = gfn_stream(stream);
  head_unpack_bwd_kernel<<<rows_grid(n_max, 16), 256, 0, st>>>(n_max, n_dev, h, dsigma, dxh, dh);
  GFN_CHECK_LAUNCH();
  if (demb && n_rays > 0) {
    emb_grad_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, st>>>(pts_offsets, (int)n_rays, dxh, cam_idx, demb);
    GFN_CHECK_LAUNCH();
  }
  return GFN_OK;
}

int gfn_composite_fwd(const int32_t* pts_offsets, int64_t n_rays, const float* sigma,
                      const float* dt, const float* t, const float* rgb, float early_T,
                      float* weights, float* alphas, float* out_rgb, float* out_acc,
                      float* out_depth, float* t_minmax, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(pts_offsets && out_rgb, "gfn_composite_fwd: null pointer");  // sample arrays may be null when every ray is empty
  GFN_CHECK_ARG(n_rays >= 0 && n_rays < (1ll << 30), "gfn_composite_fwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return GFN_OK;
  hipStream_t st = gfn_stream(stream);
  if (t_minmax) {
    tminmax_init_kernel<<<1, 1, 0, st>>>(t_minmax);
    GFN_CHECK_LAUNCH();
  }
  composite_fwd_kernel<false><<<(unsigned)((n_rays + 3) / 4), 256, 0, st>>>(pts_offsets, (int)n_rays, sigma, dt, t, rgb, early_T,
                                                                           weights, alphas, out_rgb, out_acc, out_depth, t_minmax, PhaseArgs{});
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_composite_fwd_phased(const int32_t* pts_offsets, int64_t n_rays, const float* sigma, const float* dt, const float* t,
                             const float* rgb, float early_T, float* weights, float* alphas, float* out_rgb, float* out_acc,
                             float* out_depth, float* t_minmax, const uint8_t* ray_gate, const int32_t* gate, const int32_t* k_split,
                             uint8_t* need_more, int32_t* term, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(pts_offsets && out_rgb && term, "gfn_composite_fwd_phased: null pointer");
  GFN_CHECK_ARG(n_rays > 0 && n_rays < (1ll << 30), "gfn_composite_fwd_phased: n_rays=%lld", (long long)n_rays);
  GFN_CHECK_ARG(!k_split || need_more, "gfn_composite_fwd_phased: the first pass needs need_more");
  GFN_CHECK_ARG(!(ray_gate && t_minmax), "gfn_composite_fwd_phased: the t range belongs to the pass over all rays");
  hipStream_t st = gfn_stream(stream);
  if (t_minmax) {
    tminmax_init_kernel<<<1, 1, 0, st>>>(t_minmax);
    GFN_CHECK_LAUNCH();
  }
  PhaseArgs P{ray_gate, gate, k_split, need_more, term};
  composite_fwd_kernel<true><<<(unsigned)((n_rays + 3) / 4), 256, 0, st>>>(pts_offsets, (int)n_rays, sigma, dt, t, rgb, early_T,
                                                                          weights, alphas, out_rgb, out_acc, out_depth, t_minmax, P);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_phase_list(const int32_t* pts_offsets, int64_t n_rays, const int32_t* k_split, const uint8_t* need_more, const int32_t* term,
                   int32_t* coff, int32_t* list, int32_t* stats, float* zero_sigma, float* zero_rgb, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(pts_offsets && k_split && coff && list && stats && n_rays > 0 && n_rays < (1ll << 24), "gfn_phase_list: bad arguments");
  GFN_CHECK_ARG((zero_sigma == nullptr) == (zero_rgb == nullptr) && !(zero_sigma && (need_more || term)), "gfn_phase_list: sigma and rgb are cleared together, in phase A");
  GFN_CHECK_ARG(!(need_more && term), "gfn_phase_list: one kind of list per call");
  hipStream_t st = gfn_stream(stream);
  if (n_rays <= kPhaseOneLaunchRays) {
    phase_list_kernel<<<(unsigned)((n_rays + 15) / 16), 1024, 0, st>>>(pts_offsets, (int)n_rays, k_split, need_more, term, coff, list, stats, zero_sigma, zero_rgb);
    GFN_CHECK_LAUNCH();
    return GFN_OK;
  }
  phase_scan_kernel<<<1, 1024, 0, st>>>(pts_offsets, (int)n_rays, k_split, need_more, term, coff, stats);
  GFN_CHECK_LAUNCH();
  phase_fill_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, st>>>(pts_offsets, (int)n_rays, k_split, need_more ? 1 : 0, coff, list, zero_sigma, zero_rgb);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_two_phase_plan(const int32_t* pts_offsets, int64_t n_rays, const int32_t* term, int32_t* k_split, int32_t* stats,
                       int32_t* stats_last, float fixed_cost_samples, float save_enter, float save_stay, int enabled, void* stream) {
  GFN_CHECK_ARG(pts_offsets && term && k_split && stats && stats_last && n_rays >= 0 && n_rays < (1ll << 20), "gfn_two_phase_plan: bad arguments");
  two_phase_plan_kernel<<<1, 1024, 0, gfn_stream(stream)>>>(pts_offsets, (int)n_rays, term, k_split, stats, stats_last, fixed_cost_samples, save_enter, save_stay, enabled);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_depth_finalize(float* depth, int64_t n_rays, const float* t_minmax, float scale_factor, void* stream) {
  if (n_rays == 0) return GFN_OK;  // empty batch: pointers may be null
  GFN_CHECK_ARG(depth && t_minmax && n_rays > 0 && scale_factor != 0.f, "gfn_depth_finalize: bad arguments");
  depth_finalize_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, gfn_stream(stream)>>>(depth, n_rays, t_minmax, scale_factor);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_composite_bwd(const int32_t* pts_offsets, int64_t n_rays, const float* sigma,
                      const float* dt, const float* rgb, const float* g_rgb, const float* g_acc,
                      float early_T, float* d_sigma, float* d_rgb, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(pts_offsets && g_rgb, "gfn_composite_bwd: null pointer");
  GFN_CHECK_ARG(n_rays >= 0 && n_rays < (1ll << 30), "gfn_composite_bwd: n_rays=%lld", (long long)n_rays);
  if (n_rays == 0) return GFN_OK;
  composite_bwd_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, gfn_stream(stream)>>>(pts_offsets, (int)n_rays, sigma, dt, rgb,
                                                                                    g_rgb, g_acc, early_T, d_sigma, d_rgb);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_charbonnier(const float* pred, const float* gt, int64_t rows, float eps, float* loss_dev,
                    float* g_pred, void* stream) {
  if (rows == 0) return GFN_OK;
  GFN_CHECK_ARG(pred && gt && rows > 0, "gfn_charbonnier: bad arguments");
  const int64_t n = rows * 3;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  charbonnier_kernel<<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(pred, gt, n, 1.0f / (float)rows, eps, loss_dev, g_pred);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_train_losses(const float* pred, const float* gt, int64_t rows, float eps, float* out3, float* g_char, float* g_zero,
                     const int32_t* indices, const int32_t* slot, int32_t n_images, int32_t height, int32_t width, float* error_map,
                     double* block_sums, float* workspace, void* stream) {
  if (rows == 0) return GFN_OK;
  GFN_CHECK_ARG(pred && gt && out3 && rows > 0 && rows <= (1ll << 22), "gfn_train_losses: pred, gt, out3 and 1 <= rows <= 2^22 (a training batch)");
  GFN_CHECK_ARG(!error_map || (indices && n_images > 0 && height > 0 && width > 0), "gfn_train_losses: the error map needs the rays' (c, y, x) rows and its extents");
  int64_t blocks = workspace ? (rows * 3 + kLossThreads - 1) / kLossThreads : 1;
  if (blocks > kLossMaxBlocks) blocks = kLossMaxBlocks;
  train_losses_kernel<<<(unsigned)blocks, kLossThreads, 0, gfn_stream(stream)>>>(pred, gt, rows, eps, out3, g_char, g_zero, indices, slot, n_images, height, width,
                                                                                error_map, block_sums, GFN_ERRMAP_BLOCK, workspace);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_loss_grad_combine(const float* g_a, const float* w_a, const float* g_b, const float* w_b, int64_t n, float* out, void* stream) {
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(out && n > 0 && (g_a || !w_a) && (g_b || !w_b), "gfn_loss_grad_combine: a weight without its gradient, or no output");
  loss_grad_combine_kernel<<<(unsigned)((n + 255) / 256), 256, 0, gfn_stream(stream)>>>(g_a, w_a, g_b, w_b, n, out);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

void gfn_adam_hyper_host(int step, float lr, float beta1, float beta2, float* out2) {
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  out2[0] = (float)(lr / bc1);
  out2[1] = (float)sqrt(bc2);
}

int gfn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr,
                  float beta1, float beta2, float eps, const float* hyper_dev, void* stream) {
  GFN_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "gfn_adam_step: bad arguments");
  if (n == 0) return GFN_OK;
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  adam_kernel<<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(p, g, m, v, n, 1.f - beta1, beta2, 1.f - beta2,
                                                               (float)(lr / bc1), (float)sqrt(bc2), eps, hyper_dev);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_adam_step_ex(float* p, const void* g, int g_f16, float g_scale, float* m, float* v, int64_t n,
                     int step, float lr, float beta1, float beta2, float eps, uint16_t* shadow_f16,
                     const float* hyper_dev, void* stream) {
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "gfn_adam_step_ex: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  _Float16* sh = reinterpret_cast<_Float16*>(shadow_f16);
  if (g_f16)
    adam_ex_kernel<true><<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(p, g, g_scale, m, v, n, 1.f - beta1, beta2, 1.f - beta2,
                                                                          (float)(lr / bc1), (float)sqrt(bc2), eps, sh, hyper_dev);
  else
    adam_ex_kernel<false><<<(unsigned)blocks, 256, 0, gfn_stream(stream)>>>(p, g, g_scale, m, v, n, 1.f - beta1, beta2, 1.f - beta2,
                                                                           (float)(lr / bc1), (float)sqrt(bc2), eps, sh, hyper_dev);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_fill_zero(void* p, int64_t bytes, void* stream) {
  GFN_CHECK_ARG(p && bytes >= 0, "gfn_fill_zero: bad arguments");
  if (bytes == 0) return GFN_OK;
  GFN_CHECK_HIP(gfn_zero_async(p, (size_t)bytes, gfn_stream(stream)));
  return GFN_OK;
}

int gfn_generate_rays(const float* c2w, const float* intr, const int32_t* cam_idx, int32_t cam0, const float* coords,
                      int64_t first_pixel, int32_t width, int64_t n, float* origins, float* directions, float* lookat,
                      float* pixel_area, float* dir_norm, void* stream) {
  if (n == 0) return GFN_OK;
  GFN_CHECK_ARG(c2w && intr && directions, "gfn_generate_rays: null pointer");
  GFN_CHECK_ARG(n > 0 && n < (1ll << 40), "gfn_generate_rays: n=%lld", (long long)n);
  GFN_CHECK_ARG(coords || (width > 0 && first_pixel >= 0), "gfn_generate_rays: raster mode needs width > 0 and first_pixel >= 0");
  GFN_CHECK_ARG(cam_idx || cam0 >= 0, "gfn_generate_rays: cam0=%d", (int)cam0);
  generate_rays_kernel<<<(unsigned)((n + 255) / 256), 256, 0, gfn_stream(stream)>>>(c2w, intr, cam_idx, cam0, coords, first_pixel, width, n,
                                                                                   origins, directions, lookat, pixel_area, dir_norm);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_s3im(const float* gt, const float* pred, const int32_t* index, int64_t n_idx, int patch_h, float mult, float* loss_dev,
             float* g_pred, void* stream) {
  GFN_CHECK_ARG(gt && pred && index && loss_dev, "gfn_s3im: null pointer");
  GFN_CHECK_ARG(patch_h >= 4 && n_idx >= 4 * (int64_t)patch_h && n_idx % patch_h == 0, "gfn_s3im: n_idx=%lld must be a multiple of patch_h=%d (>= 4 columns)",
                (long long)n_idx, patch_h);
  const int64_t H = patch_h, W = n_idx / patch_h;
  const int64_t OH = (H + 2 - 4) / 4 + 1, OW = (W + 2 - 4) / 4 + 1;
  S3imWin win;
  {  // gaussian(4, 1.5) as torch builds it: f32 tensor of exp values divided by its f32 sum, outer product (:726-733)
    float g[4], sum = 0.f;
    for (int x = 0; x < 4; x++) { g[x] = (float)exp(-(double)((x - 2) * (x - 2)) / (2.0 * 1.5 * 1.5)); sum += g[x]; }
    for (int x = 0; x < 4; x++) g[x] = g[x] / sum;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) win.w[i][j] = g[i] * g[j];
  }
  const int64_t cells = 3 * OH * OW;
  s3im_kernel<<<(unsigned)((cells + 255) / 256), 256, 0, gfn_stream(stream)>>>(gt, pred, index, H, W, OH, OW, win, mult, loss_dev, g_pred);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
