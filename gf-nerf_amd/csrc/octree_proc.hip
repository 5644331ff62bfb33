// octree_proc.hip -- PersOctree::ProcOctree (gfnerf/bindings/PtsSampler/PersSampler.cpp:154-417) ON THE DEVICE.
//
// The reference copies the node blob to the host, prunes / path-compresses / renumbers / subdivides a std::vector<TreeNode>
// there and uploads the result (PersSampler_cuda.cu:667-676: every compact_freq iterations and at the subdivision
// milestones).  csrc/octree_proc.cpp is that computation in closed form on the host; this file is the same closed form on
// the GPU, byte for byte (tests/test_gpu_octree_proc.py), so the tree never leaves HBM:
//   alive(u)   = u's subtree holds a leaf with trans_idx >= 0       -> every live leaf marks its ancestors
//   spliced(v) = alive, not the root, internal, exactly one alive child
//   kept(u)    = root, or alive and not spliced; new index = rank among kept nodes (prefix scan in index order)
//   child slot = first kept node down that slot (follow spliced chains); parent = first kept ancestor
//   subdivide  = pre-order renumbering of the compacted tree in which a leaf with visit_cnt > 4 (every leaf: brute force)
//                is followed immediately by its 8 octants: subtree sizes bottom-up, pre-order ranks top-down.
// The tree is small (10^3 .. 10^5 nodes) and this runs once per ~1000 training steps: ONE workgroup walks the phases with
// workgroup barriers in between -- no grid-wide synchronisation, no host round trip; what matters is that the training
// stream never drains.  The new node count lands in n_out_dev; status_dev != 0 reports a blob that is not a tree.
#include "gfn_common.h"

namespace {

constexpr int kT = 1024;
constexpr long long kInitNodeStat = 1000;  // PersSampler.h:14 INIT_NODE_STAT

struct Work {  // int32 scratch arrays of n_in (compaction) resp. n_mid (subdivision) entries
  int* alive;
  int* kids;
  int* new_idx;
  int* only;   // the single alive child of a spliced node
  int* sub;    // subdivision: leaf is split
  int* size;   // subtree size in the output numbering
  int* pre;    // pre-order rank in the output numbering
};

__device__ __forceinline__ const gfn_tree_node* node_at(const uint8_t* blob, long long i) {
  return reinterpret_cast<const gfn_tree_node*>(blob + i * GFN_TREE_NODE_BYTES);
}
__device__ __forceinline__ gfn_tree_node* node_at(uint8_t* blob, long long i) {
  return reinterpret_cast<gfn_tree_node*>(blob + i * GFN_TREE_NODE_BYTES);
}
__device__ __forceinline__ void copy_node(uint8_t* dst, const uint8_t* src) {  // all 128 bytes: padding travels with the node
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int k = 0; k < GFN_TREE_NODE_BYTES / 16; k++) d[k] = s[k];
}

// exclusive prefix sum of flag[0..n) by one workgroup -> out, returns the total
__device__ int block_scan(const int* flag, int* out, int n, int* sh /*[kT + 1]*/) {
  const int tid = threadIdx.x;
  const int per = (n + kT - 1) / kT;
  const int lo = tid * per, hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; i++) s += flag[i];
  sh[tid] = s;
  __syncthreads();
  for (int off = 1; off < kT; off <<= 1) {
    const int v = tid >= off ? sh[tid - off] : 0;
    __syncthreads();
    sh[tid] += v;
    __syncthreads();
  }
  const int total = sh[kT - 1];
  int run = sh[tid] - s;
  __syncthreads();
  for (int i = lo; i < hi; i++) { out[i] = run; run += flag[i]; }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(kT) void proc_octree_kernel(const uint8_t* __restrict__ in, long long n_in_cap, const long long* __restrict__ n_in_dev,
                                                         const long long* __restrict__ w_in, const long long* __restrict__ a_in,
                                                         const long long* __restrict__ visit_in, int subdivide, int brute_force,
                                                         uint8_t* __restrict__ mid, long long* __restrict__ w_mid, long long* __restrict__ a_mid,
                                                         int* __restrict__ origin_mid, uint8_t* __restrict__ out, long long* __restrict__ w_out,
                                                         long long* __restrict__ a_out, long long out_cap, long long* __restrict__ n_out_dev,
                                                         int* __restrict__ status_dev, Work wk) {
  __shared__ int sh[kT + 1];
  __shared__ int changed, bad;
  const int tid = threadIdx.x;
  const int n = (int)(n_in_dev ? min(*n_in_dev, n_in_cap) : n_in_cap);
  if (tid == 0) { bad = 0; changed = 0; }
  __syncthreads();
  // ---- the blob must be a tree rooted at node 0: child -> parent links agree, and every non-root node is some slot of its parent
  for (int u = tid; u < n; u += kT) {
    const gfn_tree_node* t = node_at(in, u);
    if (u == 0 && t->parent != -1) bad = 1;
    for (int st = 0; st < 8; st++) {
      const long long c = t->childs[st];
      if (c >= n || (c >= 0 && (c == 0 || node_at(in, c)->parent != u))) bad = 1;
      if (c >= 0 && t->is_leaf_node) bad = 1;  // a leaf has no children (the build and every ProcOctree keep it so)
    }
    if (u != 0) {
      const long long p = t->parent;
      bool listed = false;
      if (p >= 0 && p < n)
        for (int st = 0; st < 8; st++) listed |= node_at(in, p)->childs[st] == u;
      if (!listed) bad = 1;
    }
    wk.alive[u] = 0;
  }
  __syncthreads();
  if (n < 1 || bad) {
    if (tid == 0) { *status_dev = 1; *n_out_dev = 0; }
    return;
  }
  // ---- alive: every leaf with a warp marks itself and its ancestors (stop at the first one somebody else marked already)
  for (int u = tid; u < n; u += kT) {
    const gfn_tree_node* t = node_at(in, u);
    if (t->is_leaf_node && t->trans_idx >= 0) {
      long long v = u;
      while (v >= 0) {
        if (atomicExch(&wk.alive[v], 1) == 1) break;
        v = node_at(in, v)->parent;
      }
    }
  }
  __syncthreads();
  // an internal node is alive only through its children: a marked node always has an alive leaf below it, and an internal node
  // flagged is_leaf_node == 0 never marks itself -- exactly the host's post-order rule
  for (int u = tid; u < n; u += kT) {
    const gfn_tree_node* t = node_at(in, u);
    int cnt = 0, only = -1;
    if (!t->is_leaf_node)
      for (int st = 0; st < 8; st++) {
        const long long c = t->childs[st];
        if (c >= 0 && wk.alive[c]) { cnt++; only = (int)c; }
      }
    wk.kids[u] = cnt;
    wk.only[u] = only;  // the LAST alive slot, like the host's loop (it is the only one when the node is spliced)
  }
  __syncthreads();
  auto spliced = [&](long long v) { return v != 0 && wk.alive[v] && !node_at(in, v)->is_leaf_node && wk.kids[v] == 1; };
  for (int u = tid; u < n; u += kT) wk.sub[u] = (u == 0 || (wk.alive[u] && !spliced(u))) ? 1 : 0;  // kept flags (sub reused)
  __syncthreads();
  const int m = block_scan(wk.sub, wk.new_idx, n, sh);
  uint8_t* cdst = subdivide ? mid : out;
  long long* cw = subdivide ? w_mid : w_out;
  long long* ca = subdivide ? a_mid : a_out;
  if (!subdivide && m > out_cap) {
    if (tid == 0) { *status_dev = 2; *n_out_dev = 0; }
    return;
  }
  for (int u = tid; u < n; u += kT) {
    if (!wk.sub[u]) continue;
    const int k = wk.new_idx[u];
    uint8_t* dstb = cdst + (long long)k * GFN_TREE_NODE_BYTES;
    copy_node(dstb, in + (long long)u * GFN_TREE_NODE_BYTES);
    gfn_tree_node* nd = reinterpret_cast<gfn_tree_node*>(dstb);
    const gfn_tree_node* t = node_at(in, u);
    for (int st = 0; st < 8; st++) {
      long long c = t->childs[st];
      if (c >= 0 && !wk.alive[c]) c = -1;
      while (c >= 0 && spliced(c)) c = wk.only[c];  // follow the chain to the first kept node
      nd->childs[st] = c >= 0 ? wk.new_idx[c] : -1;
    }
    if (u != 0) {
      long long p = t->parent;
      while (spliced(p)) p = node_at(in, p)->parent;
      nd->parent = wk.new_idx[p];
    }
    cw[k] = w_in[u]; ca[k] = a_in[u];
    if (subdivide) origin_mid[k] = u;
  }
  __syncthreads();
  if (!subdivide) {
    if (tid == 0) { *n_out_dev = m; *status_dev = 0; }
    return;
  }
  // ---- subdivision of the compacted tree `mid` (m nodes): sizes bottom-up, pre-order ranks top-down, then emit
  for (int u = tid; u < m; u += kT) {
    const gfn_tree_node* t = node_at(mid, u);
    const int split = t->is_leaf_node && (brute_force || visit_in[origin_mid[u]] > 4) ? 1 : 0;
    wk.sub[u] = split;
    wk.size[u] = t->is_leaf_node ? 1 + 8 * split : 0;  // 0 = not known yet
    wk.pre[u] = -1;
  }
  __syncthreads();
  for (;;) {  // an internal node learns its size once all of its children know theirs: <= depth rounds
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int u = tid; u < m; u += kT) {
      if (wk.size[u]) continue;
      const gfn_tree_node* t = node_at(mid, u);
      int s = 1;
      bool ready = true;
      for (int st = 0; st < 8; st++) {
        const long long c = t->childs[st];
        if (c < 0) continue;
        const int sc = wk.size[c];
        if (!sc) { ready = false; break; }
        s += sc;
      }
      if (ready) { wk.size[u] = s; changed = 1; }
    }
    __syncthreads();
    const int again = changed;
    __syncthreads();
    if (!again) break;
  }
  const long long m_out = wk.size[0];
  if (m_out > out_cap || m_out <= 0) {
    if (tid == 0) { *status_dev = 2; *n_out_dev = 0; }
    return;
  }
  if (tid == 0) wk.pre[0] = 0;
  __syncthreads();
  for (int u = tid; u < m; u += kT) wk.alive[u] = 0;  // reused: "children ranked"
  __syncthreads();
  for (;;) {  // a node with a rank hands ranks to its children, slots ascending (the host's pre-order)
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int u = tid; u < m; u += kT) {
      if (wk.pre[u] < 0 || wk.alive[u]) continue;
      const gfn_tree_node* t = node_at(mid, u);
      int run = wk.pre[u] + 1;
      if (!t->is_leaf_node)
        for (int st = 0; st < 8; st++) {
          const long long c = t->childs[st];
          if (c < 0) continue;
          wk.pre[c] = run;
          run += wk.size[c];
        }
      wk.alive[u] = 1;
      changed = 1;
    }
    __syncthreads();
    const int again = changed;
    __syncthreads();
    if (!again) break;
  }
  for (int u = tid; u < m; u += kT) {
    const int nu = wk.pre[u];
    if (nu < 0) continue;  // unreachable from the root: cannot happen in a well-formed tree
    uint8_t* dstb = out + (long long)nu * GFN_TREE_NODE_BYTES;
    copy_node(dstb, mid + (long long)u * GFN_TREE_NODE_BYTES);
    gfn_tree_node* nd = reinterpret_cast<gfn_tree_node*>(dstb);
    const gfn_tree_node* t = node_at(mid, u);
    nd->parent = u == 0 ? -1 : wk.pre[t->parent];
    w_out[nu] = w_mid[u]; a_out[nu] = a_mid[u];
    if (!t->is_leaf_node) {
      for (int st = 0; st < 8; st++) {
        const long long c = t->childs[st];
        if (c >= 0) nd->childs[st] = wk.pre[c];
      }
    } else if (wk.sub[u]) {
      const float half = t->side_len * .5f;
      for (int st = 0; st < 8; st++) {
        const int kid_idx = nu + 1 + st;
        uint8_t* kb = out + (long long)kid_idx * GFN_TREE_NODE_BYTES;
        uint4* kz = reinterpret_cast<uint4*>(kb);
#pragma unroll
        for (int q = 0; q < GFN_TREE_NODE_BYTES / 16; q++) kz[q] = make_uint4(0u, 0u, 0u, 0u);  // value-initialised: block_idx = 0
        gfn_tree_node* kid = reinterpret_cast<gfn_tree_node*>(kb);
        const float off[3] = {(float)((st >> 2) & 1) - .5f, (float)((st >> 1) & 1) - .5f, (float)(st & 1) - .5f};
        for (int k = 0; k < 3; k++) kid->center[k] = t->center[k] + half * off[k];
        kid->side_len = half;
        kid->parent = nu;
        for (int k = 0; k < 8; k++) kid->childs[k] = -1;
        kid->is_leaf_node = 1;
        kid->trans_idx = t->trans_idx;
        nd->childs[st] = kid_idx;
        w_out[kid_idx] = w_mid[u]; a_out[kid_idx] = a_mid[u];
      }
      nd->is_leaf_node = 0;
      nd->trans_idx = -1;
      w_out[nu] = kInitNodeStat; a_out[nu] = kInitNodeStat;
    }
  }
  __syncthreads();
  if (tid == 0) { *n_out_dev = m_out; *status_dev = 0; }
}

}  // namespace

extern "C" {

int64_t gfn_octree_proc_workspace(int64_t n_nodes, int subdivide) {
  // 7 int32 arrays + (subdivide) the compacted copy: node blob, two i64 stat arrays, origin
  int64_t b = 7 * 4 * n_nodes;
  if (subdivide) b += n_nodes * (GFN_TREE_NODE_BYTES + 8 + 8 + 4);
  return (b + 255) / 256 * 256 + 256;
}

int gfn_octree_proc(const void* nodes, int64_t n_nodes, const int64_t* n_nodes_dev, const int64_t* weight_stats, const int64_t* alpha_stats,
                    const int64_t* visit_cnt, int subdivide, int brute_force, void* nodes_out, int64_t* weight_stats_out,
                    int64_t* alpha_stats_out, int64_t out_cap, int64_t* n_out_dev, int32_t* status_dev, void* workspace, void* stream) {
  GFN_CHECK_ARG(nodes && weight_stats && alpha_stats && visit_cnt && nodes_out && weight_stats_out && alpha_stats_out && n_out_dev && status_dev && workspace,
                "gfn_octree_proc: null pointer");
  GFN_CHECK_ARG(n_nodes >= 1 && n_nodes < (1ll << 28) && out_cap >= 1 && out_cap < (1ll << 30), "gfn_octree_proc: n_nodes=%lld out_cap=%lld",
                (long long)n_nodes, (long long)out_cap);
  GFN_CHECK_ARG((reinterpret_cast<uintptr_t>(nodes) & 15) == 0 && (reinterpret_cast<uintptr_t>(nodes_out) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
                "gfn_octree_proc: blobs must be 16-byte aligned, the workspace 256-byte aligned");
  uint8_t* p = static_cast<uint8_t*>(workspace);
  Work wk;
  int* ints = reinterpret_cast<int*>(p);
  wk.alive = ints; wk.kids = ints + n_nodes; wk.new_idx = ints + 2 * n_nodes; wk.only = ints + 3 * n_nodes; wk.sub = ints + 4 * n_nodes;
  wk.size = ints + 5 * n_nodes; wk.pre = ints + 6 * n_nodes;
  p += (7 * 4 * n_nodes + 255) / 256 * 256;
  uint8_t* mid = nullptr;
  long long *w_mid = nullptr, *a_mid = nullptr;
  int* origin = nullptr;
  if (subdivide) {
    mid = p; p += n_nodes * GFN_TREE_NODE_BYTES;
    w_mid = reinterpret_cast<long long*>(p); p += 8 * n_nodes;
    a_mid = reinterpret_cast<long long*>(p); p += 8 * n_nodes;
    origin = reinterpret_cast<int*>(p);
  }
  proc_octree_kernel<<<1, kT, 0, gfn_stream(stream)>>>(
      static_cast<const uint8_t*>(nodes), n_nodes, reinterpret_cast<const long long*>(n_nodes_dev), reinterpret_cast<const long long*>(weight_stats),
      reinterpret_cast<const long long*>(alpha_stats), reinterpret_cast<const long long*>(visit_cnt), subdivide, brute_force, mid, w_mid, a_mid, origin,
      static_cast<uint8_t*>(nodes_out), reinterpret_cast<long long*>(weight_stats_out), reinterpret_cast<long long*>(alpha_stats_out), out_cap,
      reinterpret_cast<long long*>(n_out_dev), status_dev, wk);
  GFN_CHECK_LAUNCH();
  return GFN_OK;
}

}  // extern "C"
