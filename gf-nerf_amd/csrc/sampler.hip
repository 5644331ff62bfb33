// sampler.hip -- PersSampler on gfx950: octree traversal (K3), perspective-warp
// ray march (K4), dense export, node statistics (K5/K6).
//
// Replaces gfnerf/bindings/PtsSampler/PersSampler_cuda.cu:21-318 (K3/K4),
// :321-477 (GetSamples host logic: here count -> device scan -> fill with NO
// host synchronisation and PACKED outputs) and :518-665 (UpdateOctNodes).
//
// Arithmetic order is kept identical to oracle/gfn_oracle.c (itself a
// restatement of the .cu, Eigen reductions spelled as halving trees) and the
// file is compiled with -ffp-contract=off, so leaf lists, sample counts and
// every f32 produced here are bit-exact against the oracle.
#include "gfn_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int kRayBlock = 64;   // one wave per workgroup: 4096 rays -> 64 WGs spread over CUs
constexpr int kMaxDepth = 24;   // MAX_STACK_SIZE 48 int64 = 24 (node, child_ptr) pairs (:7,:70)

struct NodeView {
  const uint8_t* base;
  __device__ __forceinline__ float4 box(int u) const {  // center xyz + side_len (@0..15)
    return *reinterpret_cast<const float4*>(base + (size_t)u * GFN_TREE_NODE_BYTES);
  }
  __device__ __forceinline__ int child(int u, int k) const {  // low word of int64 childs[k] (@24)
    return *reinterpret_cast<const int*>(base + (size_t)u * GFN_TREE_NODE_BYTES + 24 + 8 * k);
  }
  __device__ __forceinline__ int trans_idx(int u) const {
    return *reinterpret_cast<const int*>(base + (size_t)u * GFN_TREE_NODE_BYTES + 96);
  }
  __device__ __forceinline__ int block_idx(int u) const {
    return *reinterpret_cast<const int*>(base + (size_t)u * GFN_TREE_NODE_BYTES + 104);
  }
};

// PersSampler_cuda.cu:21-51
__device__ __forceinline__ void get_intersection(const float o[3], const float d[3], float4 box,
                                                 float& near, float& far) {
  const float c[3] = {box.x, box.y, box.z};
  const float hf = box.w * .5f;
  float t0[3], t1[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (d[i] < 1e-6f && d[i] > -1e-6f) {
      if (o[i] > c[i] - hf && o[i] < c[i] + hf) {
        t0[i] = -1e6f; t1[i] = 1e6f;
      } else {
        t0[i] = 1e6f; t1[i] = -1e6f;
      }
    } else if (d[i] > 0) {
      t0[i] = ((c[i] - hf) - o[i]) / d[i];
      t1[i] = ((c[i] + hf) - o[i]) / d[i];
    } else {
      t0[i] = ((c[i] + hf) - o[i]) / d[i];
      t1[i] = ((c[i] - hf) - o[i]) / d[i];
    }
  }
  near = fmaxf(near, fmaxf(t0[0], fmaxf(t0[1], t0[2])));
  far = fminf(far, fminf(t1[0], fminf(t1[1], t1[2])));
}

// child visit order: PersSampler.cpp:138-151 sorts 0..7 by a total order whose
// closed form is bitrev3(k) ^ (~st & 7) (checked against the sort in tests).
__device__ __forceinline__ int search_child(int ray_st, int k) {
  const int br = ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1);
  return br ^ (~ray_st & 7);
}

// K3: PersSampler_cuda.cu:53-152.  FILL=false counts, FILL=true writes.
template <bool FILL>
__global__ __launch_bounds__(kRayBlock) void trace_kernel(
    int n_rays, NodeView nodes, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    float near_bound, float far_bound, int max_hits, int* __restrict__ counts,
    const int* __restrict__ offsets, int* __restrict__ hit_idx, float2* __restrict__ hit_nf,
    int64_t hit_cap, int* __restrict__ overflow) {
  __shared__ int s_node[kMaxDepth][kRayBlock];
  __shared__ signed char s_child[kMaxDepth][kRayBlock];
  const int lane = threadIdx.x;
  const int ray = blockIdx.x * kRayBlock + lane;
  if (ray >= n_rays) return;
  const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
  const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
  int max_cnt = max_hits;
  int64_t out_base = 0;
  if (FILL) {
    out_base = offsets[ray];
    int64_t end = offsets[ray + 1];
    if (end > hit_cap) { end = hit_cap; if (overflow) atomicOr(overflow, 1); }
    max_cnt = (int)(end > out_base ? end - out_base : 0);  // :76
  }
  const int ray_st = ((d[0] > 0.f) << 2) | ((d[1] > 0.f) << 1) | (d[2] > 0.f);  // :87
  int sp = 0, cnt = 0;
  s_node[0][lane] = 0;
  s_child[0][lane] = -1;
  while (sp >= 0 && cnt < max_cnt) {
    const int u = s_node[sp][lane];
    int child_ptr = s_child[sp][lane];
    bool descend = true;
    if (child_ptr == -1) {
      float cur_near = near_bound, cur_far = far_bound;
      get_intersection(o, d, nodes.box(u), cur_near, cur_far);
      if (cur_near < cur_far) {
        child_ptr = 0;
        while (child_ptr < 8 && nodes.child(u, search_child(ray_st, child_ptr)) < 0) child_ptr++;
        if (child_ptr >= 8) {  // leaf
          if (nodes.trans_idx(u) >= 0) {  // :110
            if (FILL) {
              hit_idx[out_base + cnt] = u;
              hit_nf[out_base + cnt] = make_float2(cur_near, cur_far);
            }
            cnt++;
          }
          descend = false;
        }
      } else {
        descend = false;
      }
    } else {
      child_ptr = child_ptr + 1;
      while (child_ptr < 8 && nodes.child(u, search_child(ray_st, child_ptr)) < 0) child_ptr++;
      if (child_ptr >= 8) descend = false;
    }
    if (descend) {
      s_child[sp][lane] = (signed char)child_ptr;
      if (sp + 1 < kMaxDepth) {
        sp++;
        s_node[sp][lane] = nodes.child(u, search_child(ray_st, child_ptr));
        s_child[sp][lane] = -1;
      } else if (overflow) {
        atomicOr(overflow, 2);  // deeper than the reference's own 24-entry stack
      }
    } else {
      sp--;
    }
  }
  if (!FILL) counts[ray] = cnt;
}

// Exclusive scan of counts[n] -> offsets[n+1] in one workgroup (n is a ray count).
// Exclusive prefix sum of the per-ray counts (offsets[n] = total).  Two launches of ceil(n / 1024) workgroups: (1) every workgroup
// scans its 1,024 counts (coalesced loads, wave scans + one LDS hop) and publishes its total; (2) every workgroup adds the totals
// of the workgroups before it.  The single-workgroup form this replaces read its counts strided (thread t walked a private chunk)
// and took 237 us on a 138 k-ray render tile; integers: the result is the same in any order.
__global__ __launch_bounds__(1024) void scan_local_kernel(const int* __restrict__ counts, int n, int* __restrict__ offsets,
                                                          int* __restrict__ block_sums) {
  __shared__ int s_wave[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 1024 + tid;
  const int c = i < n ? counts[i] : 0;
  int v = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (lane == 63) s_wave[wave] = v;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += s_wave[w];
  if (i < n) offsets[i] = base + v - c;  // exclusive, local to the workgroup
  if (tid == 1023) block_sums[blockIdx.x] = base + v;
}
__global__ __launch_bounds__(1024) void scan_add_kernel(int n, int* __restrict__ offsets, const int* __restrict__ block_sums) {
  __shared__ int s_base;
  if (threadIdx.x < 64) {  // first wave: sum of the workgroups before this one (and, in the last workgroup, the grand total)
    int acc = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 64) acc += block_sums[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (threadIdx.x == 0) {
      s_base = acc;
      if (blockIdx.x == gridDim.x - 1) offsets[n] = acc + block_sums[blockIdx.x];
    }
  }
  __syncthreads();
  const int i = blockIdx.x * 1024 + threadIdx.x;
  if (i < n) offsets[i] += s_base;
}
// A training batch (n <= 8192 rays): one workgroup, one launch -- the scan sits between the march and the fill on the sampler's
// branch, which ends the step since the backward got shorter; coalesced loads, 1,024 counts per round with a running carry.
__global__ __launch_bounds__(1024) void scan_small_kernel(const int* __restrict__ counts, int n, int* __restrict__ offsets) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const int c = i < n ? counts[i] : 0;
    int v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    int base = s_carry;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    if (i < n) offsets[i] = base + v - c;
    __syncthreads();
    if (tid == 1023) s_carry = base + v;
    __syncthreads();
  }
  if (tid == 0) offsets[n] = s_carry;
}
// block_sums: scratch of ceil(n / 1024) ints
inline void launch_scan(const int* counts, int n, int* offsets, int* block_sums, hipStream_t st) {
  if (n <= 8192) {
    scan_small_kernel<<<1, 1024, 0, st>>>(counts, n, offsets);
    return;
  }
  const int blocks = (n + 1023) / 1024;
  scan_local_kernel<<<blocks, 1024, 0, st>>>(counts, n, offsets, block_sums);
  scan_add_kernel<<<blocks, 1024, 0, st>>>(n, offsets, block_sums);
}

struct TransView {
  const float* p;  // 144 floats: w2xz[12][2][4] @0, weight[3][12] @96, center @132, side @135, dis @136
  __device__ __forceinline__ float w2xz(int i, int r, int c) const { return p[i * 8 + r * 4 + c]; }
  __device__ __forceinline__ float weight(int r, int i) const { return p[96 + r * 12 + i]; }
};

__device__ __forceinline__ float sum12(const float* q) {
  return (gfn_sum3(q[0], q[1], q[2]) + gfn_sum3(q[3], q[4], q[5])) +
         (gfn_sum3(q[6], q[7], q[8]) + gfn_sum3(q[9], q[10], q[11]));
}

// PersSampler_cuda.cu:172-188 followed by `jac * rays_d` and its norm (:268-271)
__device__ __forceinline__ float warp_jac_dir_norm(TransView tr, const float xyz[3], const float d[3]) {
  float tj[GFN_N_PROS][3];
#pragma unroll
  for (int i = 0; i < GFN_N_PROS; i++) {
    const float4 m0 = *reinterpret_cast<const float4*>(tr.p + i * 8);
    const float4 m1 = *reinterpret_cast<const float4*>(tr.p + i * 8 + 4);
    const float x0 = gfn_sum4(m0.x * xyz[0], m0.y * xyz[1], m0.z * xyz[2], m0.w * 1.f);
    const float x1 = gfn_sum4(m1.x * xyz[0], m1.y * xyz[1], m1.z * xyz[2], m1.w * 1.f);
    const float dv0 = 1.f / x1;
    const float dv1 = -x0 / (x1 * x1);
    tj[i][0] = dv0 * m0.x + dv1 * m1.x;
    tj[i][1] = dv0 * m0.y + dv1 * m1.y;
    tj[i][2] = dv0 * m0.z + dv1 * m1.z;
  }
  float pj[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float jr[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float q[GFN_N_PROS];
#pragma unroll
      for (int i = 0; i < GFN_N_PROS; i++) q[i] = tr.weight(r, i) * tj[i][c];
      jr[c] = sum12(q);
    }
    pj[r] = gfn_sum3(jr[0] * d[0], jr[1] * d[1], jr[2] * d[2]);
  }
  return sqrtf(gfn_sum3(pj[0] * pj[0], pj[1] * pj[1], pj[2] * pj[2]));
}

// PersSampler_cuda.cu:155-170
__device__ __forceinline__ void warp_point(TransView tr, const float xyz[3], float out[3]) {
  float v[GFN_N_PROS];
#pragma unroll
  for (int i = 0; i < GFN_N_PROS; i++) {
    const float4 m0 = *reinterpret_cast<const float4*>(tr.p + i * 8);
    const float4 m1 = *reinterpret_cast<const float4*>(tr.p + i * 8 + 4);
    const float x0 = gfn_sum4(m0.x * xyz[0], m0.y * xyz[1], m0.z * xyz[2], m0.w * 1.f);
    const float x1 = gfn_sum4(m1.x * xyz[0], m1.y * xyz[1], m1.z * xyz[2], m1.w * 1.f);
    v[i] = x0 / x1;
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float q[GFN_N_PROS];
#pragma unroll
    for (int i = 0; i < GFN_N_PROS; i++) q[i] = tr.weight(r, i) * v[i];
    out[r] = sum12(q);
  }
}

// K4: PersSampler_cuda.cu:190-318.
template <bool FILL>
__global__ __launch_bounds__(kRayBlock) void march_kernel(
    int n_rays, float sample_l, int scale_by_dis, int table_space, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ noise_all,
    const int* __restrict__ oct_offsets, const int* __restrict__ hit_idx,
    const float2* __restrict__ hit_nf, NodeView nodes, const float* __restrict__ transes,
    int* __restrict__ counts, const int* __restrict__ pts_offsets, float* __restrict__ world_pts,
    float* __restrict__ warped_pts, float* __restrict__ dirs, float* __restrict__ dists,
    float* __restrict__ ts, int* __restrict__ anchors, int* __restrict__ ray_id,
    float* __restrict__ first_oct_dis, int64_t pts_cap, int* __restrict__ overflow) {
  const int ray = blockIdx.x * kRayBlock + threadIdx.x;
  if (ray >= n_rays) return;
  const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
  const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
  const float* noise = noise_all + ray;  // :204 sliding window over one shared array
  const int oct_s = oct_offsets[ray];
  const int n_oct = oct_offsets[ray + 1] - oct_s;
  hit_idx += oct_s;
  hit_nf += oct_s;
  int max_n = GFN_MAX_SAMPLE_PER_RAY;
  int64_t base = 0;
  if (FILL) {
    base = pts_offsets[ray];
    int64_t end = pts_offsets[ray + 1];
    if (end > pts_cap) { end = pts_cap; if (overflow) atomicOr(overflow, 4); }
    max_n = (int)(end > base ? end - base : 0);
    if (max_n > GFN_MAX_SAMPLE_PER_RAY) max_n = GFN_MAX_SAMPLE_PER_RAY;  // :237
    if (first_oct_dis) first_oct_dis[ray] = n_oct > 0 ? hit_nf[0].x : 1e9f;  // :229-234
  }
  int pts_ptr = 0;
  if (max_n > 0 && n_oct > 0) {
    int oct_ptr = 0;
    int cur_oct_idx = hit_idx[0];
    float cur_march_step = 0.f, exp_march_step = 0.f;
    float2 nf = hit_nf[0];
    float cur_t = nf.x;
    float cur_far = nf.y;
    float cur_near = nf.x;
    float xyz[3] = {o[0] + d[0] * cur_t, o[1] + d[1] * cur_t, o[2] + d[2] * cur_t};
    bool first = true;
    while (pts_ptr < max_n && oct_ptr < n_oct) {
      const int trans_idx = nodes.trans_idx(cur_oct_idx);
      TransView tr{transes + (size_t)trans_idx * (GFN_TRANS_INFO_BYTES / 4)};
      const float dx = o[0] - tr.p[132], dy = o[1] - tr.p[133], dz = o[2] - tr.p[134];
      const float cur_radius = sqrtf(gfn_sum3(dx * dx, dy * dy, dz * dz)) / tr.p[136];  // :266
      const float cur_radius_clip = fmaxf(cur_radius, 1.f);
      const float pnorm = warp_jac_dir_norm(tr, xyz, d);
      const float warp_step = sample_l * noise[pts_ptr];  // :270
      exp_march_step = warp_step / (pnorm + 1e-6f);
      if (scale_by_dis) exp_march_step *= cur_radius_clip;
      cur_march_step = exp_march_step;
      if (FILL && !first) {  // :278-290
        const int64_t p = base + pts_ptr;
        float w[3];
        warp_point(tr, xyz, w);
        if (table_space) {
          // (sampled_pts + 1.5) / 3.0 (nerfacto_field.py:431).  On the device the reference runs on,
          // torch divides a tensor by a Python scalar as a * (1/b) in f32
          // (ATen/native/cuda/BinaryDivTrueKernel.cu, is_cpu_scalar branch) -- restated as such.
          const float inv3 = 1.0f / 3.0f;
          w[0] = (w[0] + 1.5f) * inv3; w[1] = (w[1] + 1.5f) * inv3; w[2] = (w[2] + 1.5f) * inv3;
        }
        if (world_pts) { world_pts[3 * p] = xyz[0]; world_pts[3 * p + 1] = xyz[1]; world_pts[3 * p + 2] = xyz[2]; }
        if (dirs) { dirs[3 * p] = d[0]; dirs[3 * p + 1] = d[1]; dirs[3 * p + 2] = d[2]; }
        warped_pts[3 * p] = w[0]; warped_pts[3 * p + 1] = w[1]; warped_pts[3 * p + 2] = w[2];
        ts[p] = cur_t;
        dists[p] = exp_march_step * (pnorm + 1e-6f);
        anchors[3 * p] = trans_idx;
        anchors[3 * p + 1] = cur_oct_idx;
        anchors[3 * p + 2] = nodes.block_idx(cur_oct_idx);
        if (ray_id) ray_id[p] = ray;
      }
      if (!first) pts_ptr += 1;
      while (cur_t + cur_march_step > cur_far) {  // :295-305
        oct_ptr++;
        if (oct_ptr >= n_oct) break;
        cur_oct_idx = hit_idx[oct_ptr];
        nf = hit_nf[oct_ptr];
        cur_near = nf.x;
        cur_far = nf.y;
        const int64_t ex = (int64_t)ceilf(fmaxf((cur_near - cur_t) / exp_march_step, 1.f));
        cur_march_step = exp_march_step * (float)ex;
      }
      cur_t += cur_march_step;
      xyz[0] = o[0] + d[0] * cur_t; xyz[1] = o[1] + d[1] * cur_t; xyz[2] = o[2] + d[2] * cur_t;
      first = false;
    }
  }
  if (!FILL) counts[ray] = pts_ptr;
}

// ============================================================================
// Fast path used by the training / render loop (gfn_sample_rays): same arithmetic,
// same outputs, laid out for the machine.
//   * nodes are repacked to ONE 64-byte line each (box, 8 child ids, trans/block idx),
//     so a DFS visit is a single cache-line fetch instead of up to ten dependent loads
//     from the 128-byte checkpoint struct;
//   * the traversal runs once, writing each ray's leaves to its own dense slot
//     (R x max_hits is small) -- no count pass, no scan, no host sync -- with an 8-lane
//     group per ray that tests all children of a node in one step;
//   * the march gives every ray a group of lanes of a wave (MarchGroup below): the 12
//     perspective projections are shared among them, the 3x12 contractions are DPP
//     trees in exactly the association of sum12(), and the leaf's TransInfo stays in
//     registers until the ray crosses into the next leaf.  16 lanes per ray for small
//     batches (4096 rays = 1024 waves instead of 64), 4 for render tiles.
// ============================================================================
struct __attribute__((aligned(64))) CompactNode {
  float cx, cy, cz, side;
  int child[8];
  int trans_idx, block_idx, pad0, pad1;
};
static_assert(sizeof(CompactNode) == 64, "one cache line per node");

__global__ void compact_nodes_kernel(NodeView nodes, int64_t n_nodes, CompactNode* __restrict__ out) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_nodes) return;
  CompactNode c;
  const float4 b = nodes.box((int)u);
  c.cx = b.x; c.cy = b.y; c.cz = b.z; c.side = b.w;
#pragma unroll
  for (int k = 0; k < 8; k++) c.child[k] = nodes.child((int)u, k);
  c.trans_idx = nodes.trans_idx((int)u);
  c.block_idx = nodes.block_idx((int)u);
  c.pad0 = 0; c.pad1 = 0;
  out[u] = c;
}

// K3 with an 8-lane group per ray: popping an internal node tests its 8 children at once (lane k
// takes the k-th child in the ray's visiting order, each child is one 64-byte line), the survivors
// are pushed in reverse so the DFS order -- and with it the emitted leaf list, near/far bits and the
// max_hits cut -- is exactly that of the reference's one-child-at-a-time loop (:89-140).
constexpr int kTG = 8;                      // lanes per ray
constexpr int kTraceRays = 256 / kTG;       // 32 rays per workgroup

// stack: entries per ray, 7 * (depth of the tree) + 2 -- the 24 levels the reference allows make 65 KB of stacks per
// workgroup, i.e. two workgroups per CU; a tree 12 levels deep needs half of that and runs twice as many rays per CU, which is what
// bounds the traversal when it has to make do with the few CUs a persistent kernel beside it leaves (render: 1.8 ms per tile)
__global__ __launch_bounds__(256) void trace_group_kernel(
    int n_rays, const CompactNode* __restrict__ nodes, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, float near_bound, float far_bound, int max_hits,
    int* __restrict__ counts, int* __restrict__ hit_idx, float2* __restrict__ hit_nf, int* __restrict__ overflow, int stack) {
  extern __shared__ __attribute__((aligned(16))) int s_trace[];
  int* s_id = s_trace + (threadIdx.x / kTG) * stack;                                         // >= 0: internal node to expand; < 0: ~leaf id to emit
  float2* s_nf = reinterpret_cast<float2*>(s_trace + kTraceRays * stack) + (threadIdx.x / kTG) * stack;  // (kTraceRays * stack * 4 bytes: a multiple of 8)
  const int k = threadIdx.x & (kTG - 1);
  const int slot = threadIdx.x / kTG;
  const int ray = blockIdx.x * kTraceRays + slot;
  const bool in_range = ray < n_rays;
  const int r = in_range ? ray : 0;
  const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const int64_t out_base = (int64_t)r * max_hits;
  const int ray_st = ((d[0] > 0.f) << 2) | ((d[1] > 0.f) << 1) | (d[2] > 0.f);
  const int my_child_slot = search_child(ray_st, k);
  const int gshift = (threadIdx.x & 63) & ~(kTG - 1);  // bit position of this group inside the wave ballot
  int sp = 0, cnt = 0;
  if (in_range) {  // root (:84-86)
    const int4* np = reinterpret_cast<const int4*>(nodes);
    const int4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
    float nr = near_bound, fr = far_bound;
    get_intersection(o, d, make_float4(__int_as_float(w0.x), __int_as_float(w0.y), __int_as_float(w0.z), __int_as_float(w0.w)), nr, fr);
    if (nr < fr) {
      const bool leaf = w1.x < 0 && w1.y < 0 && w1.z < 0 && w1.w < 0 && w2.x < 0 && w2.y < 0 && w2.z < 0 && w2.w < 0;
      if (!leaf) {
        if (k == 0) { s_id[0] = 0; s_nf[0] = make_float2(nr, fr); }
        sp = 1;
      } else if (w3.x >= 0) {
        if (k == 0) { s_id[0] = ~0; s_nf[0] = make_float2(nr, fr); }
        sp = 1;
      }
    }
  }
  bool active = in_range && sp > 0 && cnt < max_hits;
  while (__any(active)) {
    bool keep = false, leaf_child = false;
    int ck = -1;
    float cn = 0.f, cf = 0.f;
    if (active) {
      const int e = s_id[sp - 1];
      const float2 enf = s_nf[sp - 1];
      sp--;
      if (e < 0) {  // leaf with a valid transform: emit (:110-117)
        if (k == 0) {
          hit_idx[out_base + cnt] = ~e;
          hit_nf[out_base + cnt] = enf;
        }
        cnt++;
      } else {
        ck = nodes[e].child[my_child_slot];
        if (ck >= 0) {
          const int4* np = reinterpret_cast<const int4*>(nodes + ck);
          const int4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
          cn = near_bound; cf = far_bound;
          get_intersection(o, d, make_float4(__int_as_float(w0.x), __int_as_float(w0.y), __int_as_float(w0.z), __int_as_float(w0.w)), cn, cf);
          leaf_child = w1.x < 0 && w1.y < 0 && w1.z < 0 && w1.w < 0 && w2.x < 0 && w2.y < 0 && w2.z < 0 && w2.w < 0;
          keep = (cn < cf) && (!leaf_child || w3.x >= 0);
        }
      }
    }
    const unsigned mask = (unsigned)((__ballot(keep) >> gshift) & 0xFFull);  // wave-uniform call
    if (active && mask) {
      const int n_push = __popc(mask);
      if (sp + n_push <= stack) {
        if (keep) {
          const int pos = sp + __popc(mask >> (k + 1));  // reverse order: first child ends on top
          s_id[pos] = leaf_child ? ~ck : ck;
          s_nf[pos] = make_float2(cn, cf);
        }
        sp += n_push;
      } else if (overflow && k == 0) {
        atomicOr(overflow, 2);
      }
    }
    active = in_range && sp > 0 && cnt < max_hits;
  }
  if (in_range && k == 0) {
    counts[ray] = cnt;
    // the traversal ended at max_hits with nodes still on the stack: the reference's own truncation when max_hits is its
    // max_oct_intersect_per_ray (1024) -- a caller that passed a SMALLER cap to keep its hit lists small reads bit 8 and asks again
    if (sp > 0 && cnt >= max_hits && overflow) atomicOr(overflow, 8);
  }
}

// ---- the 12 projections of a step, shared by the L lanes of a ray --------------------------------------------------------------
// Sum over the projections in the association of sum12(): ((q0+(q1+q2)) + (q3+(q4+q5))) + ((q6+(q7+q8)) + (q9+(q10+q11))).
// lane i of a 16-lane row reads lane i+N / lane i-N of the same row (0 beyond the row): DPP row_shl / row_shr, no LDS traffic
template <int N>
__device__ __forceinline__ float row_down(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ float row_up(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xF, 0xF, true));
}
// lane i of a quad (4 neighbouring lanes) reads lane i ^ 1 / lane i ^ 2 of the same quad: DPP quad_perm
__device__ __forceinline__ float quad_x1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_x2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true));
}

// How the L lanes of a ray share the work of a step.
//   L = 16: one projection per lane (lanes 0 .. 11), the tree is 4 DPP steps and its result lands on lane 0 -- a step's ~300 instructions serve
//           the 4 rays of a wave, and 4096 training rays are 1024 waves, one per SIMD: the shortest serial chain, for batches that leave
//           the machine empty anyway (a wave stalls on every leaf change of ANY of its rays: few rays per wave is what counts there);
//   L = 4:  three projections per lane -- the innermost bracket of sum12() is lane-local, two quad exchanges finish the tree (a + b and
//           b + a are the same float) and every lane of the quad holds the result -- the same ~300 instructions serve 16 rays: a quarter of
//           the vector work per ray, for render tiles, where the march is bound by instruction issue (a 138 k-ray tile: 2.2 -> 0.76 ms).
template <int L> struct MarchGroup;
template <> struct MarchGroup<16> {
  static constexpr int P = 1, NZ = 1, THREADS = 256;
  static __device__ __forceinline__ float sum12(const float (&q)[1]) {
    const float a = q[0] + row_down<1>(q[0]);  // lanes 3k+1: q[3k+1] + q[3k+2]
    const float b = q[0] + row_down<1>(a);     // lanes 3k  : q[3k] + (q[3k+1] + q[3k+2])
    const float c = b + row_down<3>(b);        // lanes 0, 6: s0 + s1, s2 + s3
    return c + row_down<6>(c);                 // lane 0
  }
  static __device__ __forceinline__ float to_all(float v) { return __shfl(v, 0, 16); }         // lane 0's value in every lane of the row
  template <int N> static __device__ __forceinline__ float to_lane(float v) { return row_up<N>(v); }  // lane 0's value in lane N
};
template <> struct MarchGroup<4> {
  static constexpr int P = 3, NZ = 4, THREADS = 64;
  static __device__ __forceinline__ float sum12(const float (&q)[3]) {
    const float s = q[0] + (q[1] + q[2]);
    const float t = s + quad_x1(s);
    return t + quad_x2(t);                     // every lane of the quad
  }
  static __device__ __forceinline__ float to_all(float v) { return v; }
  template <int N> static __device__ __forceinline__ float to_lane(float v) { return v; }
};

// Where the counting march keeps its records when they are what the inference kernel reads (gfn_sample_rays_record).  Dense: slot k of
// ray r at (r * 1024 + k) -- 32 KB per ray whatever its length (17 GB for a 518 k-ray tile, of which the rays' ~180 samples fill 18 %).
// Paged (table != NULL): pages of GFN_REC_PAGE slots from a pool, handed out as the rays need them by one atomic counter; page p of ray r
// is table[r * 32 + p].  The pool is sized by the SAMPLES (n_pages: ~224 slots per ray), not by the cap; a pool that runs out is flagged
// in the overflow word (bit 16) and the caller samples again into a larger one.
struct RecPages {
  int* table; int* counter; int n_pages;
};

// K4 with L lanes per ray.  Hits come from the dense slots of trace_group_kernel.
template <int L, bool FILL>
__global__ __launch_bounds__(MarchGroup<L>::THREADS) void march_group_kernel(
    int n_rays, float sample_l, int scale_by_dis, int table_space, int max_hits,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ noise_all,
    const int* __restrict__ hit_counts, const int* __restrict__ hit_idx, const float2* __restrict__ hit_nf,
    const CompactNode* __restrict__ nodes, const float* __restrict__ transes, int* __restrict__ counts,
    const int* __restrict__ pts_offsets, float* __restrict__ world_pts, float* __restrict__ warped_pts,
    float* __restrict__ dirs, float* __restrict__ dists, float* __restrict__ ts, int* __restrict__ anchors,
    int* __restrict__ ray_id, float* __restrict__ first_oct_dis, int64_t pts_cap, int* __restrict__ overflow,
    float* __restrict__ rec /* count pass only: per ray 1024 x {t, dist, leaf, warped point} for fill_from_rec_kernel, or NULL */,
    RecPages pages /* table != NULL: the records live in pages of GFN_REC_PAGE slots handed out as the rays need them (see RecPages) */) {
  using G = MarchGroup<L>;
  constexpr int P = G::P, NZ = G::NZ;
  const int g = threadIdx.x & (L - 1);
  const int ray = blockIdx.x * (G::THREADS / L) + (threadIdx.x / L);
  const bool in_range = ray < n_rays;
  const int r = in_range ? ray : 0;
  const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const float* noise = noise_all + r;
  const int n_oct = in_range ? hit_counts[r] : 0;
  hit_idx += (int64_t)r * max_hits;
  hit_nf += (int64_t)r * max_hits;
  if (rec && !pages.table) rec += (int64_t)r * GFN_MAX_SAMPLE_PER_RAY * GFN_MARCH_REC_FLOATS;
  int page_slot0 = 0;  // (paged records) first slot of the page this ray is filling
  int max_n = GFN_MAX_SAMPLE_PER_RAY;
  int64_t base = 0;
  if (FILL) {
    base = pts_offsets[r];
    int64_t end = pts_offsets[r + 1];
    if (end > pts_cap) { end = pts_cap; if (overflow && g == 0) atomicOr(overflow, 4); }
    max_n = (int)(end > base ? end - base : 0);
    if (max_n > GFN_MAX_SAMPLE_PER_RAY) max_n = GFN_MAX_SAMPLE_PER_RAY;
  }
  if ((FILL || rec) && in_range && g == 0 && first_oct_dis) first_oct_dis[r] = n_oct > 0 ? hit_nf[0].x : 1e9f;  // (:229-234; the count pass of gfn_sample_rays_record)
  bool active = in_range && max_n > 0 && n_oct > 0;
  int pts_ptr = 0, oct_ptr = 0;
  int nz_base = -1;
  float nz[NZ];
#pragma unroll
  for (int m = 0; m < NZ; m++) nz[m] = 0.f;
  int cur_oct_idx = active ? hit_idx[0] : 0;
  float2 nf = active ? hit_nf[0] : make_float2(0.f, 0.f);
  float cur_t = nf.x, cur_far = nf.y, cur_near = nf.x;
  float cur_march_step = 0.f, exp_march_step = 0.f;
  float xyz[3] = {o[0] + d[0] * cur_t, o[1] + d[1] * cur_t, o[2] + d[2] * cur_t};
  bool first = true;
  // this lane's projections of the current leaf's TransInfo: rows proj0 .. proj0 + P - 1 of w2xz and the matching weight columns
  const int proj0 = P * g < GFN_N_PROS ? P * g : 0;  // (L = 16: lanes 12 .. 15 repeat projection 0, nothing reads their sums)
  int loaded_oct = -1, trans_idx = 0, block_idx = 0;
  float4 m0[P], m1[P];
  float w[3][P];  // [output row][projection of this lane]
#pragma unroll
  for (int p = 0; p < P; p++) {
    m0[p] = make_float4(0, 0, 0, 0); m1[p] = make_float4(0, 0, 0, 1);
    w[0][p] = 0.f; w[1][p] = 0.f; w[2][p] = 0.f;
  }
  float radius_clip = 1.f;
  while (__any(active)) {
    if (active && loaded_oct != cur_oct_idx) {  // uniform over the ray's lanes: reload on leaf change only
      const CompactNode* nd = nodes + cur_oct_idx;
      trans_idx = nd->trans_idx;
      block_idx = nd->block_idx;
      const float* tp = transes + (size_t)trans_idx * (GFN_TRANS_INFO_BYTES / 4);
#pragma unroll
      for (int p = 0; p < P; p++) {
        m0[p] = *reinterpret_cast<const float4*>(tp + (proj0 + p) * 8);
        m1[p] = *reinterpret_cast<const float4*>(tp + (proj0 + p) * 8 + 4);
#pragma unroll
        for (int rr = 0; rr < 3; rr++) w[rr][p] = tp[96 + 12 * rr + proj0 + p];
      }
      loaded_oct = cur_oct_idx;
      // the ray's distance to the leaf's cameras, in units of their distance summary: constant while the ray stays in the leaf
      const float dx = o[0] - tp[132], dy = o[1] - tp[133], dz = o[2] - tp[134];
      radius_clip = fmaxf(sqrtf(gfn_sum3(dx * dx, dy * dy, dz * dz)) / tp[136], 1.f);
    }
    // this lane's projections (PersSampler_cuda.cu:155-188)
    float tj[3][P], v[P];  // [column][projection]
#pragma unroll
    for (int p = 0; p < P; p++) {
      const float x0 = gfn_sum4(m0[p].x * xyz[0], m0[p].y * xyz[1], m0[p].z * xyz[2], m0[p].w * 1.f);
      const float x1 = gfn_sum4(m1[p].x * xyz[0], m1[p].y * xyz[1], m1[p].z * xyz[2], m1[p].w * 1.f);
      const float dv0 = 1.f / x1;
      const float dv1 = -x0 / (x1 * x1);
      tj[0][p] = dv0 * m0[p].x + dv1 * m1[p].x;
      tj[1][p] = dv0 * m0[p].y + dv1 * m1[p].y;
      tj[2][p] = dv0 * m0[p].z + dv1 * m1[p].z;
      v[p] = (FILL || rec) ? x0 / x1 : 0.f;
    }
    float pj[3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      float jr[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float q[P];
#pragma unroll
        for (int p = 0; p < P; p++) q[p] = w[rr][p] * tj[c][p];
        jr[c] = G::sum12(q);
      }
      pj[rr] = gfn_sum3(jr[0] * d[0], jr[1] * d[1], jr[2] * d[2]);
    }
    const float pnorm = G::to_all(sqrtf(gfn_sum3(pj[0] * pj[0], pj[1] * pj[1], pj[2] * pj[2])));
    float wp[3] = {0.f, 0.f, 0.f};
    if (FILL || rec) {  // warped point (:155-170): v_i = x0/x1, out_r = sum12(weight[r][i] * v_i); the counting pass records it too
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        float q[P];
#pragma unroll
        for (int p = 0; p < P; p++) q[p] = w[rr][p] * v[p];
        wp[rr] = G::sum12(q);
      }
    }
    // (count pass) the warped point where the record's store wants it: lane 2 (x, y), lane 3 (z)
    const float wp_x2 = G::template to_lane<2>(wp[0]), wp_y2 = G::template to_lane<2>(wp[1]), wp_z3 = G::template to_lane<3>(wp[2]);
    if (active) {
      // the step's noise: sixteen values per ray and visit of memory, NZ consecutive ones in each lane, handed out through the LDS crossbar --
      // a global load per step made every step wait for the memory system (and, loads and stores sharing one counter here, for the stores
      // of the steps before it)
      const int nb = pts_ptr & ~15;
      if (nb != nz_base) {
#pragma unroll
        for (int m = 0; m < NZ; m++) nz[m] = noise[nb + NZ * g + m];
        nz_base = nb;
      }
      float nzv = nz[0];
#pragma unroll
      for (int m = 1; m < NZ; m++) nzv = (pts_ptr & (NZ - 1)) == m ? nz[m] : nzv;
      const float warp_step = sample_l * __shfl(nzv, (pts_ptr & 15) / NZ, L);
      exp_march_step = warp_step / (pnorm + 1e-6f);
      if (scale_by_dis) exp_march_step *= radius_clip;
      cur_march_step = exp_march_step;
      if (FILL && !first && g == 0) {
        const int64_t p = base + pts_ptr;
        if (table_space) {
          const float inv3 = 1.0f / 3.0f;
          wp[0] = (wp[0] + 1.5f) * inv3; wp[1] = (wp[1] + 1.5f) * inv3; wp[2] = (wp[2] + 1.5f) * inv3;
        }
        if (world_pts) { world_pts[3 * p] = xyz[0]; world_pts[3 * p + 1] = xyz[1]; world_pts[3 * p + 2] = xyz[2]; }
        if (dirs) { dirs[3 * p] = d[0]; dirs[3 * p + 1] = d[1]; dirs[3 * p + 2] = d[2]; }
        warped_pts[3 * p] = wp[0]; warped_pts[3 * p + 1] = wp[1]; warped_pts[3 * p + 2] = wp[2];
        ts[p] = cur_t;
        dists[p] = exp_march_step * (pnorm + 1e-6f);
        anchors[3 * p] = trans_idx;
        anchors[3 * p + 1] = cur_oct_idx;
        anchors[3 * p + 2] = block_idx;
        if (ray_id) ray_id[p] = ray;
      }
      if (!FILL && rec && !first) {  // the sample as a 32-byte record {t, dist, leaf, transform, warped x y z, block}: all the sample-parallel fill (or the
        // inference kernel, gfn_fused_render_rec) needs -- ONE 8-byte store from lanes 0 .. 3 of the ray (one store per value from one lane were
        // six memory instructions per step, and on this architecture a wave's next load waits for every store before it: beside a kernel that
        // keeps the L2 busy the march ran at the pace of the L2's write acknowledgements)
        float2 v2;
        v2.x = g == 0 ? cur_t : (g == 1 ? __int_as_float(cur_oct_idx) : (g == 2 ? wp_x2 : wp_z3));
        v2.y = g == 0 ? exp_march_step * (pnorm + 1e-6f) : (g == 1 ? __int_as_float(trans_idx) : (g == 2 ? wp_y2 : __int_as_float(block_idx)));
        int64_t slot = pts_ptr;
        if (pages.table) {
          if ((pts_ptr & (GFN_REC_PAGE - 1)) == 0) {  // the ray's lanes reach this together: its first lane takes the next free page
            int pg = 0;
            if (g == 0) {
              pg = atomicAdd(pages.counter, 1);
              if (pg >= pages.n_pages) {  // pool exhausted: flagged (the caller renders again with a larger pool); the slots stay inside the buffer
                if (overflow) atomicOr(overflow, 16);
                pg = pages.n_pages - 1;
              }
              pages.table[(int64_t)r * (GFN_MAX_SAMPLE_PER_RAY / GFN_REC_PAGE) + (pts_ptr >> GFN_REC_PAGE_LOG2)] = pg;
            }
            pg = __shfl(pg, (int)(threadIdx.x & 63u) & ~(L - 1), 64);
            page_slot0 = pg << GFN_REC_PAGE_LOG2;
          }
          slot = (int64_t)page_slot0 + (pts_ptr & (GFN_REC_PAGE - 1));
        }
        if (g < GFN_MARCH_REC_FLOATS / 2) reinterpret_cast<float2*>(rec + slot * GFN_MARCH_REC_FLOATS)[g] = v2;
      }
      if (!first) pts_ptr += 1;
      while (cur_t + cur_march_step > cur_far) {
        oct_ptr++;
        if (oct_ptr >= n_oct) break;
        cur_oct_idx = hit_idx[oct_ptr];
        nf = hit_nf[oct_ptr];
        cur_near = nf.x;
        cur_far = nf.y;
        const int64_t ex = (int64_t)ceilf(fmaxf((cur_near - cur_t) / exp_march_step, 1.f));
        cur_march_step = exp_march_step * (float)ex;
      }
      cur_t += cur_march_step;
      xyz[0] = o[0] + d[0] * cur_t; xyz[1] = o[1] + d[1] * cur_t; xyz[2] = o[2] + d[2] * cur_t;
      first = false;
      active = pts_ptr < max_n && oct_ptr < n_oct;
    }
  }
  if (!FILL && in_range && g == 0) counts[ray] = pts_ptr;
}

__device__ __forceinline__ float sum12_tree(const float* q) {  // the association of row_sum12 / Eigen's redux
  return ((q[0] + (q[1] + q[2])) + (q[3] + (q[4] + q[5]))) + ((q[6] + (q[7] + q[8])) + (q[9] + (q[10] + q[11])));
}

// Second half of K4 when the count pass has recorded the samples (32 bytes each: t, dist, leaf, transform, warped point, block): with
// the offsets known every sample is written where it belongs, one thread per sample, instead of re-running the serial march.  (Until
// round 2 the record held (t, dist, leaf) and this kernel redid the 12 projections per sample: 0.86 ms on a 138 k-ray render tile, 60
// loads and 12 divisions per sample; until round 4 it fetched the leaf's node for the two indices the march had in registers.)
__global__ __launch_bounds__(256) void fill_from_rec_kernel(
    int n_rays, int table_space, int max_hits, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const int* __restrict__ hit_counts, const float2* __restrict__ hit_nf, const float* __restrict__ rec, const int* __restrict__ pts_offsets,
    float* __restrict__ world_pts, float* __restrict__ warped_pts, float* __restrict__ dirs, float* __restrict__ dists,
    float* __restrict__ ts, int* __restrict__ anchors, int* __restrict__ ray_id, float* __restrict__ first_oct_dis,
    int64_t pts_cap, int* __restrict__ overflow) {
  const int ray = blockIdx.x;
  if (ray >= n_rays) return;
  const int64_t base = pts_offsets[ray];
  int64_t end = pts_offsets[ray + 1];
  if (end > pts_cap) { end = pts_cap; if (overflow && threadIdx.x == 0) atomicOr(overflow, 4); }
  if (threadIdx.x == 0 && first_oct_dis) first_oct_dis[ray] = hit_counts[ray] > 0 ? hit_nf[(int64_t)ray * max_hits].x : 1e9f;
  const int n = (int)(end > base ? end - base : 0);
  const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
  const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
  for (int k = threadIdx.x; k < n; k += 256) {
    const float4* rp = reinterpret_cast<const float4*>(rec + ((int64_t)ray * GFN_MAX_SAMPLE_PER_RAY + k) * GFN_MARCH_REC_FLOATS);
    const float4 r0 = rp[0], r1 = rp[1];
    const float t = r0.x;
    const float xyz[3] = {o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t};
    float wp[3] = {r1.x, r1.y, r1.z};  // as march_group_kernel<L, true> computes it (same lanes, same DPP trees)
    if (table_space) {
      const float inv3 = 1.0f / 3.0f;
      wp[0] = (wp[0] + 1.5f) * inv3; wp[1] = (wp[1] + 1.5f) * inv3; wp[2] = (wp[2] + 1.5f) * inv3;
    }
    const int64_t p = base + k;
    if (world_pts) { world_pts[3 * p] = xyz[0]; world_pts[3 * p + 1] = xyz[1]; world_pts[3 * p + 2] = xyz[2]; }
    if (dirs) { dirs[3 * p] = d[0]; dirs[3 * p + 1] = d[1]; dirs[3 * p + 2] = d[2]; }
    warped_pts[3 * p] = wp[0]; warped_pts[3 * p + 1] = wp[1]; warped_pts[3 * p + 2] = wp[2];
    ts[p] = t;
    dists[p] = r0.y;
    anchors[3 * p] = __float_as_int(r0.w); anchors[3 * p + 1] = __float_as_int(r0.z); anchors[3 * p + 2] = __float_as_int(r1.w);
    if (ray_id) ray_id[p] = ray;
  }
}

__global__ void normalize_dirs_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
  const float nrm = sqrtf((x * x + y * y) + z * z);
  out[3 * i] = x / nrm; out[3 * i + 1] = y / nrm; out[3 * i + 2] = z / nrm;
}

// packed -> dense [R,1024,.] zero-padded (PersSampler_cuda.cu:437-476).  One
// workgroup per ray; every dense element is written (no separate zero fill).
__global__ __launch_bounds__(256) void to_dense_kernel(
    const int* __restrict__ pts_offsets, const float* __restrict__ world_pts,
    const float* __restrict__ warped_pts, const float* __restrict__ dirs,
    const float* __restrict__ dists, const float* __restrict__ ts, const int* __restrict__ anchors,
    float* __restrict__ d_world, float* __restrict__ d_warped, float* __restrict__ d_dirs,
    float* __restrict__ d_dists, float* __restrict__ d_ts, int64_t* __restrict__ d_anchors,
    int64_t* __restrict__ d_se) {
  const int ray = blockIdx.x;
  const int s = pts_offsets[ray], e = pts_offsets[ray + 1];
  const int n = min(e - s, GFN_MAX_SAMPLE_PER_RAY);
  const int64_t dbase = (int64_t)ray * GFN_MAX_SAMPLE_PER_RAY;
  if (threadIdx.x == 0 && d_se) { d_se[2 * ray] = s; d_se[2 * ray + 1] = e; }  // :216-217,:312
  for (int i = threadIdx.x; i < GFN_MAX_SAMPLE_PER_RAY * 3; i += blockDim.x) {
    const bool v = (i / 3) < n;
    const int64_t src = (int64_t)s * 3 + i;
    if (d_world) d_world[dbase * 3 + i] = (v && world_pts) ? world_pts[src] : 0.f;
    if (d_warped) d_warped[dbase * 3 + i] = v ? warped_pts[src] : 0.f;
    if (d_dirs) d_dirs[dbase * 3 + i] = (v && dirs) ? dirs[src] : 0.f;
    if (d_anchors) d_anchors[dbase * 3 + i] = v ? (int64_t)anchors[src] : 0;
  }
  for (int i = threadIdx.x; i < GFN_MAX_SAMPLE_PER_RAY; i += blockDim.x) {
    const bool v = i < n;
    if (d_dists) d_dists[dbase + i] = v ? dists[s + i] : 0.f;
    if (d_ts) d_ts[dbase + i] = v ? ts[s + i] : 0.f;
  }
}

// ---- UpdateOctNodes --------------------------------------------------------
#define OCC_WEIGHT_BASE 512
#define ABS_WEIGHT_THRES 0.01
#define REL_WEIGHT_THRES 0.1
#define OCC_ALPHA_BASE 32
#define ABS_ALPHA_THRES 0.02
#define REL_ALPHA_THRES 0.1

__global__ void fill_i64_kernel(long long* p, long long v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// K5 MarkVistNodeKernel (:518-574) on packed samples.  The reference walks a ray with one thread: per contiguous run
// of samples in one node it takes max weight / max alpha / the run length and atomicMax'es
//   (max_w_run > w_thres ? OCC_WEIGHT_BASE : -1),  (max_a_run > a_thres ? OCC_ALPHA_BASE : -1),  run length
// into per-node arrays initialised to -1 / 0.  Every one of these is a max, so a wave per ray can let the samples speak
// for themselves -- and the first two are maxima over {-1, ONE constant}: "raise the adder to the base" is a plain store
// of that constant (every writer stores the same value over an initial -1; no atomic unit is involved, and stores to one
// address merge on their way out).  Only the run length needs a real atomicMax, and only the last sample of a run (or
// of the 64-sample sweep) posts it: each sample's 1-based position inside its run bounds the run length from below.
// A run's samples above the threshold are reduced to its last lane first (segmented OR by ballot), so a run costs at most
// two stores and one atomic however long it is (an early version issued two contended atomics per sample: 0.32 ms per
// 4096 x 128 batch where this takes a few tens of microseconds).
__global__ __launch_bounds__(256) void mark_visit_kernel(
    int n_rays, const int* __restrict__ pts_offsets, const int* __restrict__ anchors,
    const float* __restrict__ weights, const float* __restrict__ alphas, int n_nodes, long long* w_adder,
    long long* a_adder, long long* mark, long long* visit_cnt) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int s = pts_offsets[ray], e = pts_offsets[ray + 1];
  if (s >= e) return;
  float max_w = 0.f, max_a = 0.f;
  for (int i = s + lane; i < e; i += 64) { max_w = fmaxf(max_w, weights[i]); max_a = fmaxf(max_a, alphas[i]); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    max_w = fmaxf(max_w, __shfl_xor(max_w, off, 64));
    max_a = fmaxf(max_a, __shfl_xor(max_a, off, 64));
  }
  const float w_thres = fminf((float)(max_w * REL_WEIGHT_THRES), (float)ABS_WEIGHT_THRES);  // :544 (double math)
  const float a_thres = fminf((float)(max_a * REL_ALPHA_THRES), (float)ABS_ALPHA_THRES);
  int carry_oi = -1;          // node of the previous sweep's last sample
  long long carry_cnt = 0;    // its position inside its run
  for (int base = s; base < e; base += 64) {
    const int i = base + lane;
    const bool v = i < e;
    const int oi = v ? anchors[3 * i + 1] : -2;
    int prev = __shfl_up(oi, 1, 64);
    if (lane == 0) prev = carry_oi;
    const bool head = oi != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));  // heads at lanes <= this one
    const int first = upto ? 63 - __builtin_clzll(upto) : 0;                                 // first lane of this lane's run in the sweep
    const long long cnt = upto ? (long long)(lane - first + 1) : (long long)(lane + 1) + carry_cnt;
    const int next = __shfl_down(oi, 1, 64);
    const bool last = lane == 63 || next != oi;
    // samples of this run (within the sweep) above the thresholds: bits [first, lane] of the ballots
    const unsigned long long w_hit = __ballot(v && weights[v ? i : s] > w_thres);
    const unsigned long long a_hit = __ballot(v && alphas[v ? i : s] > a_thres);
    const unsigned long long run_mask = (lane == 63 ? ~0ull : ((2ull << lane) - 1ull)) & ~((1ull << first) - 1ull);
    if (v && last && oi >= 0 && oi < n_nodes) {  // samples marched through an older (larger) tree must not write past the arrays
      mark[oi] = 1;
      if (w_hit & run_mask) w_adder[oi] = (long long)OCC_WEIGHT_BASE;
      if (a_hit & run_mask) a_adder[oi] = (long long)OCC_ALPHA_BASE;
      atomicMax(visit_cnt + oi, cnt);
    }
    carry_oi = __shfl(oi, 63, 64);
    carry_cnt = __shfl((int)cnt, 63, 64);
  }
}

// host-side torch ops of :641-655 + K6 MarkInvalidNodes (:576-582)
__global__ void node_stats_kernel(int64_t n_nodes, const long long* __restrict__ w_adder,
                                  const long long* __restrict__ a_adder,
                                  const long long* __restrict__ mark, long long* weight_stats,
                                  long long* alpha_stats, uint8_t* nodes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  auto upd = [](long long s, long long adder, long long mk) {
    const long long occ = adder > 0;
    const long long m = occ * adder;
    s = s > m ? s : m;
    s += mk * (1 - occ) * adder;
    return s < -100 ? -100 : (s > (1 << 20) ? (long long)(1 << 20) : s);
  };
  const long long ws = upd(weight_stats[i], w_adder[i], mark[i]);
  const long long as = upd(alpha_stats[i], a_adder[i], mark[i]);
  weight_stats[i] = ws;
  alpha_stats[i] = as;
  if (ws < 0 || as < 0) *reinterpret_cast<long long*>(nodes + i * GFN_TREE_NODE_BYTES + 96) = -1;
}

// K7 MarkInvisibleNodes (PersSampler_cuda.cu:680-742): a node no training camera can see loses its warp
// (trans_idx = -1).  One wave per node, lanes stride over the cameras, the wave stops at the first camera that sees the
// node (the reference loops one thread per node over all cameras).  Arithmetic in the reference's order: Eigen forms
// w2c * c.homogeneous() as leftCols(3) * c + col(3) with the 3-term sum a + (b + c); radius = side_len * 0.707 in double.
__device__ __forceinline__ bool camera_sees(const float* __restrict__ K, const float* __restrict__ W, const float* __restrict__ bd,
                                            float cx_, float cy_, float cz_, float radius) {
  float p[3];
#pragma unroll
  for (int r = 0; r < 3; r++) p[r] = (W[4 * r] * cx_ + (W[4 * r + 1] * cy_ + W[4 * r + 2] * cz_)) + W[4 * r + 3];
  if (-p[2] < bd[0] - radius || -p[2] > bd[1] + radius) return false;
  const float nrm = sqrtf(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));
  if (nrm < radius) return true;
  const float cx = K[2], cy = K[5], fx = K[0], fy = K[4];
  const float bias_x = radius / -p[2] * fx, bias_y = radius / -p[2] * fy;
  const float img_x = p[0] / -p[2] * fx, img_y = p[1] / -p[2] * fy;
  return !(img_x + bias_x < -cx || img_x > cx + bias_x || img_y + bias_y < -cy || img_y > cy + bias_y);
}

__global__ __launch_bounds__(256) void mark_invisible_kernel(int64_t n_nodes, int64_t n_cams, uint8_t* __restrict__ nodes,
                                                            const float* __restrict__ intri, const float* __restrict__ w2c,
                                                            const float* __restrict__ bounds) {
  const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_nodes) return;  // wave-uniform
  const int lane = threadIdx.x & 63;
  gfn_tree_node* nd = reinterpret_cast<gfn_tree_node*>(nodes + node * GFN_TREE_NODE_BYTES);
  const float cx = nd->center[0], cy = nd->center[1], cz = nd->center[2];
  const float radius = (float)((double)nd->side_len * 0.707);
  bool seen = false;
  for (int64_t c0 = 0; c0 < n_cams && !seen; c0 += 64) {
    const int64_t cam = c0 + lane;
    const bool v = cam < n_cams && camera_sees(intri + 9 * cam, w2c + 12 * cam, bounds + 2 * cam, cx, cy, cz, radius);
    seen = __any(v);
  }
  if (!seen && lane == 0) nd->trans_idx = -1;
}

// K12 QueryTreeNodeCenterKernel (PersSampler_cuda.cu:984-1005): centre of the node each index names; indices outside
// [0, n_nodes) leave their (zero-initialised by the caller, :1013) row untouched.
__global__ void query_centers_kernel(int64_t n_pts, int64_t n_nodes, const uint8_t* __restrict__ nodes,
                                     const long long* __restrict__ anchors, float* __restrict__ centers) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  const long long a = anchors[i];
  if (a < 0 || a >= n_nodes) return;
  const float* c = reinterpret_cast<const float*>(nodes + a * GFN_TREE_NODE_BYTES);
  centers[3 * i] = c[0]; centers[3 * i + 1] = c[1]; centers[3 * i + 2] = c[2];
}

}  // namespace

namespace {
// GFN_MARCH_LANES = 4 | 16 in the environment: the march's lanes per ray for calls that leave gfn_launch_opts.march_lanes at 0 (read once, never written)
const int kEnvMarchLanes = [] { const char* e = getenv("GFN_MARCH_LANES"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 16) ? v : 0; }();
}  // namespace

extern "C" {


int gfn_normalize_dirs(const float* rays_d_raw, float* rays_d, int64_t n_rays, void* stream) {
  if (n_rays == 0) return GFN_OK;
  GFN_CHECK_ARG(rays_d_raw && rays_d && n_rays > 0, "gfn_normalize_dirs: bad arguments");
  normalize_dirs_kernel<<<dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, gfn_stream(stream)>>>(rays_d_raw, rays_d, n_rays);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_trace(const float* rays_o, const float* rays_d, int64_t n_rays, const void* nodes,
              int64_t n_nodes, float near_bound, float far_bound, int max_hits,
              int32_t* oct_offsets, int32_t* hit_idx, float* hit_near_far, int64_t hit_cap,
              int32_t* overflow_dev, void* stream) {
  GFN_CHECK_ARG(oct_offsets, "gfn_trace: null oct_offsets");
  if (n_rays == 0) {
    GFN_CHECK_HIP(gfn_zero_async(oct_offsets, sizeof(int32_t), gfn_stream(stream)));
    return GFN_OK;
  }
  GFN_CHECK_ARG(rays_o && rays_d && nodes && hit_idx && hit_near_far, "gfn_trace: null pointer");
  GFN_CHECK_ARG(n_rays >= 0 && n_rays < (1ll << 30) && n_nodes > 0 && n_nodes < (1ll << 31),
                "gfn_trace: n_rays=%lld n_nodes=%lld out of range", (long long)n_rays, (long long)n_nodes);
  GFN_CHECK_ARG(max_hits > 0 && max_hits <= GFN_MAX_OCT_PER_RAY, "gfn_trace: max_hits=%d not in (0,1024]", max_hits);
  GFN_CHECK_ARG(hit_cap > 0 && hit_cap < (1ll << 31), "gfn_trace: hit_cap=%lld", (long long)hit_cap);
  hipStream_t st = gfn_stream(stream);
  if (n_rays == 0) {
    GFN_CHECK_HIP(gfn_zero_async(oct_offsets, sizeof(int32_t), st));
    return GFN_OK;
  }
  const unsigned grid = (unsigned)((n_rays + kRayBlock - 1) / kRayBlock);
  NodeView nv{reinterpret_cast<const uint8_t*>(nodes)};
  // per-ray counts are staged in hit_idx (unused until the fill pass); the scan consumes them first
  GFN_CHECK_ARG(hit_cap >= n_rays, "gfn_trace: hit_cap (%lld) must be >= n_rays (%lld) (used as count scratch)",
                (long long)hit_cap, (long long)n_rays);
  int* counts = hit_idx;
  trace_kernel<false><<<grid, kRayBlock, 0, st>>>((int)n_rays, nv, rays_o, rays_d, near_bound, far_bound, max_hits,
                                                  counts, nullptr, nullptr, nullptr, hit_cap, overflow_dev);
  GFN_CHECK_LAUNCH();
  launch_scan(counts, (int)n_rays, oct_offsets, reinterpret_cast<int*>(hit_near_far), st);  // hit_near_far (2 x hit_cap words, written by the next launch) is the scratch
  GFN_CHECK_LAUNCH();
  trace_kernel<true><<<grid, kRayBlock, 0, st>>>((int)n_rays, nv, rays_o, rays_d, near_bound, far_bound, max_hits,
                                                 nullptr, oct_offsets, hit_idx,
                                                 reinterpret_cast<float2*>(hit_near_far), hit_cap, overflow_dev);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_march(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
              float sample_l, int scale_by_dis, int table_space, const int32_t* oct_offsets,
              const int32_t* hit_idx, const float* hit_near_far, const void* nodes, const void* transes,
              int32_t* pts_offsets, float* world_pts, float* warped_pts, float* dirs, float* dists,
              float* ts, int32_t* anchors, int32_t* ray_id, float* first_oct_dis, int64_t pts_cap,
              int32_t* overflow_dev, void* stream) {
  GFN_CHECK_ARG(pts_offsets, "gfn_march: null pts_offsets");
  if (n_rays == 0) {
    GFN_CHECK_HIP(gfn_zero_async(pts_offsets, sizeof(int32_t), gfn_stream(stream)));
    return GFN_OK;
  }
  GFN_CHECK_ARG(rays_o && rays_d && noise && oct_offsets && hit_idx && hit_near_far && nodes && transes,
                "gfn_march: null input pointer");
  GFN_CHECK_ARG(pts_offsets && warped_pts && dists && ts && anchors, "gfn_march: null output pointer");
  GFN_CHECK_ARG(n_rays >= 0 && n_rays < (1ll << 30), "gfn_march: n_rays=%lld", (long long)n_rays);
  GFN_CHECK_ARG(pts_cap >= n_rays && pts_cap * 3 < (1ll << 31), "gfn_march: pts_cap=%lld must be in [n_rays, 2^31/3)",
                (long long)pts_cap);
  hipStream_t st = gfn_stream(stream);
  if (n_rays == 0) {
    GFN_CHECK_HIP(gfn_zero_async(pts_offsets, sizeof(int32_t), st));
    return GFN_OK;
  }
  const unsigned grid = (unsigned)((n_rays + kRayBlock - 1) / kRayBlock);
  NodeView nv{reinterpret_cast<const uint8_t*>(nodes)};
  int* counts = anchors;  // scratch until the fill pass (pts_cap >= n_rays checked above)
  march_kernel<false><<<grid, kRayBlock, 0, st>>>(
      (int)n_rays, sample_l, scale_by_dis, table_space, rays_o, rays_d, noise, oct_offsets, hit_idx,
      reinterpret_cast<const float2*>(hit_near_far), nv, reinterpret_cast<const float*>(transes), counts, nullptr,
      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pts_cap, overflow_dev);
  GFN_CHECK_LAUNCH();
  launch_scan(counts, (int)n_rays, pts_offsets, counts + n_rays, st);  // anchors holds 3 x pts_cap >= 3 x n_rays words: counts, then the scan's scratch
  GFN_CHECK_LAUNCH();
  march_kernel<true><<<grid, kRayBlock, 0, st>>>(
      (int)n_rays, sample_l, scale_by_dis, table_space, rays_o, rays_d, noise, oct_offsets, hit_idx,
      reinterpret_cast<const float2*>(hit_near_far), nv, reinterpret_cast<const float*>(transes), nullptr,
      pts_offsets, world_pts, warped_pts, dirs, dists, ts, anchors, ray_id, first_oct_dis, pts_cap, overflow_dev);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_tree_compact(const void* nodes, int64_t n_nodes, void* compact_out, void* stream) {
  if (n_nodes == 0) return GFN_OK;
  GFN_CHECK_ARG(nodes && compact_out && n_nodes > 0 && n_nodes < (1ll << 31), "gfn_tree_compact: bad arguments");
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(compact_out) & 63) == 0, "gfn_tree_compact: output must be 64-byte aligned");
  NodeView nv{reinterpret_cast<const uint8_t*>(nodes)};
  compact_nodes_kernel<<<(unsigned)((n_nodes + 255) / 256), 256, 0, gfn_stream(stream)>>>(nv, n_nodes,
                                                                                        reinterpret_cast<CompactNode*>(compact_out));
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"

// gfn_sample_rays (record_counts == NULL) and gfn_sample_rays_record (the traversal and the counting march only: the per-ray counts go to
// record_counts, the samples stay in march_rec)
static int sample_rays_core(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                            const void* compact_nodes, const void* transes, float near_bound, float far_bound,
                            int max_hits, float sample_l, int scale_by_dis, int table_space,
                            int32_t* hit_counts, int32_t* hit_idx, float* hit_near_far,
                            int32_t* pts_offsets, float* world_pts, float* warped_pts, float* dirs, float* dists,
                            float* ts, int32_t* anchors, int32_t* ray_id, float* first_oct_dis, int64_t pts_cap,
                            int32_t* overflow_dev, float* march_rec, int tree_depth, int32_t* record_counts, RecPages pages,
                            const gfn_launch_opts* opts, void* stream) {
  const bool record_only = record_counts != nullptr;
  GFN_CHECK_ARG(record_only || pts_offsets, "gfn_sample_rays: null pts_offsets");
  GFN_CHECK_ARG(tree_depth <= kMaxDepth, "gfn_sample_rays: tree_depth=%d (at most %d levels)", tree_depth, kMaxDepth);
  hipStream_t st = gfn_stream(stream);
  if (n_rays == 0) {
    if (!record_only) GFN_CHECK_HIP(gfn_zero_async(pts_offsets, sizeof(int32_t), st));
    return GFN_OK;
  }
  GFN_CHECK_ARG(rays_o && rays_d && noise && compact_nodes && transes && hit_counts && hit_idx && hit_near_far,
                "gfn_sample_rays: null input / workspace pointer");
  GFN_CHECK_ARG(record_only ? march_rec != nullptr : (warped_pts && dists && ts && anchors), "gfn_sample_rays: null output pointer");
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(march_rec) & 15) == 0, "gfn_sample_rays: march_rec must be 16-byte aligned");
  GFN_CHECK_ARG(n_rays > 0 && n_rays < (1ll << 30), "gfn_sample_rays: n_rays=%lld", (long long)n_rays);
  GFN_CHECK_ARG(max_hits > 0 && max_hits <= GFN_MAX_OCT_PER_RAY && n_rays * max_hits < (1ll << 31),
                "gfn_sample_rays: max_hits=%d", max_hits);
  GFN_CHECK_ARG(record_only || (pts_cap >= n_rays && pts_cap * 3 < (1ll << 31)), "gfn_sample_rays: pts_cap=%lld must be in [n_rays, 2^31/3)",
                (long long)pts_cap);
  const CompactNode* cn = reinterpret_cast<const CompactNode*>(compact_nodes);
  const int stack = 7 * (tree_depth > 0 ? tree_depth : kMaxDepth) + 2;  // every expansion nets at most +7 entries per level
  trace_group_kernel<<<(unsigned)((n_rays + kTraceRays - 1) / kTraceRays), 256, (size_t)kTraceRays * stack * 12, st>>>(
      (int)n_rays, cn, rays_o, rays_d, near_bound, far_bound, max_hits, hit_counts, hit_idx,
      reinterpret_cast<float2*>(hit_near_far), overflow_dev, stack);
  GFN_CHECK_LAUNCH();
  int* counts = record_only ? record_counts : anchors;  // (anchors: scratch until the fill pass)
  // lanes per ray (MarchGroup): a row of 16 while the batch leaves the machine empty (the shortest chain per step), a quad from 32 Ki rays on
  // (a quarter of the vector work per ray); opts->march_lanes = 4 | 16 (or GFN_MARCH_LANES in the environment) overrides: tests run both on the same rays
  GFN_CHECK_ARG(!opts || opts->march_lanes == 0 || opts->march_lanes == 4 || opts->march_lanes == 16, "gfn_sample_rays: march_lanes=%d (0, 4 or 16)", opts->march_lanes);
  const int forced_lanes = (opts && opts->march_lanes) ? opts->march_lanes : kEnvMarchLanes;
  const bool quad = forced_lanes == 4 || (forced_lanes != 16 && n_rays >= 32768);
  auto march = [&](auto lanes, auto fill, int* counts_out, const int* offsets, float* rec) {
    constexpr int L = decltype(lanes)::value;
    constexpr bool FILL = decltype(fill)::value;
    const int rays_per_block = MarchGroup<L>::THREADS / L;
    const unsigned grid = (unsigned)((n_rays + rays_per_block - 1) / rays_per_block);
    march_group_kernel<L, FILL><<<grid, MarchGroup<L>::THREADS, 0, st>>>(
        (int)n_rays, sample_l, scale_by_dis, table_space, max_hits, rays_o, rays_d, noise, hit_counts, hit_idx,
        reinterpret_cast<const float2*>(hit_near_far), cn, reinterpret_cast<const float*>(transes), counts_out, offsets,
        FILL ? world_pts : nullptr, FILL ? warped_pts : nullptr, FILL ? dirs : nullptr, FILL ? dists : nullptr, FILL ? ts : nullptr,
        FILL ? anchors : nullptr, FILL ? ray_id : nullptr, (FILL || record_only) ? first_oct_dis : nullptr, pts_cap, overflow_dev, rec,
        (record_only && rec) ? pages : RecPages{nullptr, nullptr, 0});
  };
  using Row = std::integral_constant<int, 16>;
  using Quad = std::integral_constant<int, 4>;
  if (quad) march(Quad{}, std::false_type{}, counts, nullptr, march_rec);
  else march(Row{}, std::false_type{}, counts, nullptr, march_rec);
  GFN_CHECK_LAUNCH();
  if (record_only) return GFN_OK;
  launch_scan(counts, (int)n_rays, pts_offsets, counts + n_rays, st);  // anchors holds 3 x pts_cap >= 3 x n_rays words: counts, then the scan's scratch
  GFN_CHECK_LAUNCH();
  if (march_rec)
    fill_from_rec_kernel<<<(unsigned)n_rays, 256, 0, st>>>(
        (int)n_rays, table_space, max_hits, rays_o, rays_d, hit_counts, reinterpret_cast<const float2*>(hit_near_far), march_rec, pts_offsets,
        world_pts, warped_pts, dirs, dists, ts, anchors, ray_id, first_oct_dis, pts_cap, overflow_dev);
  else if (quad) march(Quad{}, std::true_type{}, nullptr, pts_offsets, nullptr);
  else march(Row{}, std::true_type{}, nullptr, pts_offsets, nullptr);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}


extern "C" {

int gfn_sample_rays(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                    const void* compact_nodes, const void* transes, float near_bound, float far_bound,
                    int max_hits, float sample_l, int scale_by_dis, int table_space,
                    int32_t* hit_counts, int32_t* hit_idx, float* hit_near_far,
                    int32_t* pts_offsets, float* world_pts, float* warped_pts, float* dirs, float* dists,
                    float* ts, int32_t* anchors, int32_t* ray_id, float* first_oct_dis, int64_t pts_cap,
                    int32_t* overflow_dev, float* march_rec, int tree_depth, const gfn_launch_opts* opts, void* stream) {
  return sample_rays_core(rays_o, rays_d, noise, n_rays, compact_nodes, transes, near_bound, far_bound, max_hits, sample_l, scale_by_dis,
                          table_space, hit_counts, hit_idx, hit_near_far, pts_offsets, world_pts, warped_pts, dirs, dists, ts, anchors, ray_id,
                          first_oct_dis, pts_cap, overflow_dev, march_rec, tree_depth, nullptr, RecPages{nullptr, nullptr, 0}, opts, stream);
}

int gfn_sample_rays_record(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                           const void* compact_nodes, const void* transes, float near_bound, float far_bound,
                           int max_hits, float sample_l, int scale_by_dis, int32_t* hit_counts, int32_t* hit_idx,
                           float* hit_near_far, int32_t* counts, float* first_oct_dis, int32_t* overflow_dev,
                           float* march_rec, int32_t* page_table, int32_t* page_counter, int64_t n_pages, int tree_depth,
                           const gfn_launch_opts* opts, void* stream) {
  if (n_rays == 0) return GFN_OK;  // (an empty batch has no outputs to point at)
  GFN_CHECK_ARG(counts && first_oct_dis && march_rec, "gfn_sample_rays_record: null output pointer");
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(march_rec) & 15) == 0, "gfn_sample_rays_record: march_rec must be 16-byte aligned");
  GFN_CHECK_ARG(!page_table || (page_counter && overflow_dev && n_pages >= 1 && n_pages < (1ll << 31) / GFN_REC_PAGE),
                "gfn_sample_rays_record: paged records need the page counter, the overflow word and 1 <= n_pages < 2^31 / %d", GFN_REC_PAGE);
  if (page_table) GFN_CHECK_HIP(gfn_zero_async(page_counter, sizeof(int32_t), gfn_stream(stream)));
  return sample_rays_core(rays_o, rays_d, noise, n_rays, compact_nodes, transes, near_bound, far_bound, max_hits, sample_l, scale_by_dis, 1,
                          hit_counts, hit_idx, hit_near_far, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          first_oct_dis, 0, overflow_dev, march_rec, tree_depth, counts, RecPages{page_table, page_counter, (int)n_pages}, opts, stream);
}

int gfn_samples_to_dense(const int32_t* pts_offsets, int64_t n_rays, const float* world_pts,
                         const float* warped_pts, const float* dirs, const float* dists,
                         const float* ts, const int32_t* anchors, float* d_world, float* d_warped,
                         float* d_dirs, float* d_dists, float* d_ts, int64_t* d_anchors,
                         int64_t* d_pts_idx_start_end, void* stream) {
  GFN_CHECK_ARG(pts_offsets && n_rays >= 0, "gfn_samples_to_dense: bad arguments");
  GFN_CHECK_ARG((!d_warped || warped_pts) && (!d_dists || dists) && (!d_ts || ts) && (!d_anchors || anchors),
                "gfn_samples_to_dense: dense output requested without its packed source");
  if (n_rays == 0) return GFN_OK;
  to_dense_kernel<<<(unsigned)n_rays, 256, 0, gfn_stream(stream)>>>(pts_offsets, world_pts, warped_pts, dirs, dists, ts,
                                                                   anchors, d_world, d_warped, d_dirs, d_dists, d_ts,
                                                                   d_anchors, d_pts_idx_start_end);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_update_nodes_mark(const int32_t* pts_offsets, int64_t n_rays, const int32_t* anchors,
                          const float* weights, const float* alphas, int64_t n_nodes, int64_t* visit_cnt,
                          int64_t* scratch, void* stream) {
  GFN_CHECK_ARG(pts_offsets && anchors && weights && alphas && visit_cnt && scratch, "gfn_update_nodes_mark: null pointer");
  GFN_CHECK_ARG(n_rays >= 0 && n_nodes > 0 && n_nodes < (1ll << 31), "gfn_update_nodes_mark: bad sizes");
  hipStream_t st = gfn_stream(stream);
  long long* w_adder = reinterpret_cast<long long*>(scratch);
  long long* a_adder = w_adder + n_nodes;
  long long* mark = a_adder + n_nodes;
  fill_i64_kernel<<<(unsigned)((2 * n_nodes + 255) / 256), 256, 0, st>>>(w_adder, -1, 2 * n_nodes);  // :617-618
  GFN_CHECK_LAUNCH();
  GFN_CHECK_HIP(gfn_zero_async(mark, sizeof(long long) * n_nodes, st));
  if (n_rays > 0) {
    mark_visit_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, st>>>(
        (int)n_rays, pts_offsets, anchors, weights, alphas, (int)n_nodes, w_adder, a_adder, mark,
        reinterpret_cast<long long*>(visit_cnt));
    GFN_CHECK_LAUNCH();
  }
  return GFN_OK;
}

int gfn_update_nodes_apply(int64_t n_nodes, const int64_t* scratch, int64_t* weight_stats, int64_t* alpha_stats,
                           void* nodes, void* stream) {
  GFN_CHECK_ARG(scratch && weight_stats && alpha_stats && nodes, "gfn_update_nodes_apply: null pointer");
  GFN_CHECK_ARG(n_nodes > 0 && n_nodes < (1ll << 31), "gfn_update_nodes_apply: bad sizes");
  const long long* w_adder = reinterpret_cast<const long long*>(scratch);
  node_stats_kernel<<<(unsigned)((n_nodes + 255) / 256), 256, 0, gfn_stream(stream)>>>(
      n_nodes, w_adder, w_adder + n_nodes, w_adder + 2 * n_nodes, reinterpret_cast<long long*>(weight_stats),
      reinterpret_cast<long long*>(alpha_stats), reinterpret_cast<uint8_t*>(nodes));
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_update_nodes(const int32_t* pts_offsets, int64_t n_rays, const int32_t* anchors,
                     const float* weights, const float* alphas, int64_t n_nodes,
                     int64_t* weight_stats, int64_t* alpha_stats, int64_t* visit_cnt, void* nodes,
                     int64_t* scratch, void* stream) {
  const int rc = gfn_update_nodes_mark(pts_offsets, n_rays, anchors, weights, alphas, n_nodes, visit_cnt, scratch, stream);
  if (rc != GFN_OK) return rc;
  return gfn_update_nodes_apply(n_nodes, scratch, weight_stats, alpha_stats, nodes, stream);
}

int gfn_query_tree_nodes_centers(const void* nodes, int64_t n_nodes, const int64_t* anchors, int64_t n_pts, float* centers,
                                 void* stream) {
  if (n_pts == 0) return GFN_OK;
  GFN_CHECK_ARG(nodes && anchors && centers && n_nodes > 0 && n_pts > 0, "gfn_query_tree_nodes_centers: bad arguments");
  query_centers_kernel<<<(unsigned)((n_pts + 255) / 256), 256, 0, gfn_stream(stream)>>>(
      n_pts, n_nodes, reinterpret_cast<const uint8_t*>(nodes), reinterpret_cast<const long long*>(anchors), centers);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

int gfn_mark_invisible_nodes(void* nodes, int64_t n_nodes, const float* intri, const float* w2c, const float* bounds,
                             int64_t n_cams, void* stream) {
  GFN_CHECK_ARG(nodes && intri && w2c && bounds, "gfn_mark_invisible_nodes: null pointer");
  GFN_CHECK_ARG(n_nodes > 0 && n_cams > 0 && n_nodes < (1ll << 31), "gfn_mark_invisible_nodes: bad sizes");
  mark_invisible_kernel<<<(unsigned)((n_nodes + 3) / 4), 256, 0, gfn_stream(stream)>>>(n_nodes, n_cams, reinterpret_cast<uint8_t*>(nodes),
                                                                                      intri, w2c, bounds);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
