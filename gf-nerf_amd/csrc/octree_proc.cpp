// octree_proc.cpp -- octree maintenance between training steps: prune, path-compress, renumber, subdivide.
//
// Replaces PersOctree::ProcOctree (gfnerf/bindings/PtsSampler/PersSampler.cpp:154-417), which the reference also
// runs on the host (blob D2H -> std::vector<TreeNode> -> H2D) every compact_freq iterations and at the subdivision
// milestones (PersSampler_cuda.cu:667-676).  The reference iterates to a fixpoint and splices single-child chains
// node by node; the same result is computed here in closed form, one pass each:
//   alive(u)   = u's subtree holds a leaf with trans_idx >= 0           (post-order)
//   spliced(v) = alive(v), v is not the root, exactly one alive child   (the reference's "compress path")
//   kept(u)    = root, or alive and not spliced; a kept node's child slot points at the first kept node down that slot
//   new index  = rank among kept nodes in index order; statistics travel with their node
//   subdivide  = pre-order renumbering in which a leaf with visit_cnt > 4 (or every leaf: brute force) becomes an
//                internal node followed immediately by its 8 octants (which inherit its warp and statistics).
// Byte layouts are the checkpoint ABI (gfn_tree_node, 128 B).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/gfnerf_amd.h"

void gfn_set_error(const char* fmt, ...);

namespace {

constexpr int64_t kInitNodeStat = 1000;  // PersSampler.h:14 INIT_NODE_STAT

struct Tree {
  std::vector<gfn_tree_node> nodes;
  std::vector<int64_t> w, a, origin;  // origin: index in the input tree (-1 for new octants)
};

bool well_formed(const gfn_tree_node* t, int64_t n) {
  if (n < 1 || t[0].parent != -1) return false;
  for (int64_t u = 0; u < n; u++)
    for (int st = 0; st < 8; st++) {
      const int64_t c = t[u].childs[st];
      if (c >= n || (c >= 0 && (c == 0 || t[c].parent != u))) return false;
    }
  return true;
}

// post-order over the child pointers (no assumption on index order)
void compute_alive(const gfn_tree_node* t, int64_t n, std::vector<uint8_t>& alive, std::vector<int8_t>& n_alive_kids) {
  alive.assign(n, 0);
  n_alive_kids.assign(n, 0);
  std::vector<std::pair<int64_t, int>> stack;  // (node, next slot)
  stack.emplace_back(0, 0);
  while (!stack.empty()) {
    auto& [u, st] = stack.back();
    if (t[u].is_leaf_node) {
      alive[u] = t[u].trans_idx >= 0;
      stack.pop_back();
      continue;
    }
    if (st < 8) {
      const int64_t c = t[u].childs[st++];
      if (c >= 0) stack.emplace_back(c, 0);
      continue;
    }
    int cnt = 0;
    for (int k = 0; k < 8; k++)
      if (t[u].childs[k] >= 0 && alive[t[u].childs[k]]) cnt++;
    n_alive_kids[u] = (int8_t)cnt;
    alive[u] = cnt > 0;
    stack.pop_back();
  }
}

Tree compact_tree(const gfn_tree_node* t, int64_t n, const int64_t* w, const int64_t* a) {
  std::vector<uint8_t> alive;
  std::vector<int8_t> kids;
  compute_alive(t, n, alive, kids);
  auto spliced = [&](int64_t v) { return v != 0 && alive[v] && !t[v].is_leaf_node && kids[v] == 1; };
  std::vector<int64_t> new_idx(n, -1);
  int64_t m = 0;
  for (int64_t u = 0; u < n; u++)
    if (u == 0 || (alive[u] && !spliced(u))) new_idx[u] = m++;
  Tree out;
  out.nodes.resize(m); out.w.resize(m); out.a.resize(m); out.origin.resize(m);
  for (int64_t u = 0; u < n; u++) {
    if (new_idx[u] < 0) continue;
    gfn_tree_node nd;
    memcpy(&nd, &t[u], sizeof(nd));  // padding bytes are part of the checkpoint blob: they travel with the node
    for (int st = 0; st < 8; st++) {
      int64_t c = nd.childs[st];
      if (c >= 0 && !alive[c]) c = -1;
      while (c >= 0 && spliced(c)) {  // follow the chain to the first kept node
        int64_t next = -1;
        for (int k = 0; k < 8; k++)
          if (t[c].childs[k] >= 0 && alive[t[c].childs[k]]) next = t[c].childs[k];
        c = next;
      }
      nd.childs[st] = c >= 0 ? new_idx[c] : -1;
    }
    if (u != 0) {
      int64_t p = t[u].parent;
      while (spliced(p)) p = t[p].parent;
      nd.parent = new_idx[p];
      // a node the pruning left without children is a leaf (the reference re-flags every childless non-root node, :193-207)
    }
    const int64_t k = new_idx[u];
    memcpy(&out.nodes[k], &nd, sizeof(nd)); out.w[k] = w[u]; out.a[k] = a[u]; out.origin[k] = u;
  }
  return out;
}

Tree subdivide_tree(const Tree& in, const int64_t* visit_cnt, bool brute_force) {
  Tree out;
  const int64_t n = (int64_t)in.nodes.size();
  out.nodes.reserve(n * 2); out.w.reserve(n * 2); out.a.reserve(n * 2);
  struct Item { int64_t u, parent; int slot; };
  std::vector<Item> stack;
  stack.push_back({0, -1, -1});
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const int64_t nu = (int64_t)out.nodes.size();
    out.nodes.emplace_back(); memcpy(&out.nodes[nu], &in.nodes[it.u], sizeof(gfn_tree_node));
    out.w.push_back(in.w[it.u]); out.a.push_back(in.a[it.u]);
    out.nodes[nu].parent = it.parent;
    if (it.parent >= 0) out.nodes[it.parent].childs[it.slot] = nu;
    const gfn_tree_node& src = in.nodes[it.u];
    if (src.is_leaf_node) {
      if (!brute_force && visit_cnt[in.origin[it.u]] <= 4) continue;
      const float half = src.side_len * .5f;
      for (int st = 0; st < 8; st++) {
        gfn_tree_node kid;
        memset(&kid, 0, sizeof(kid));  // value-initialised like the reference's emplace_back(): block_idx = 0
        const float off[3] = {(float)((st >> 2) & 1) - .5f, (float)((st >> 1) & 1) - .5f, (float)(st & 1) - .5f};
        for (int k = 0; k < 3; k++) kid.center[k] = src.center[k] + half * off[k];
        kid.side_len = half;
        kid.parent = nu;
        for (int k = 0; k < 8; k++) kid.childs[k] = -1;
        kid.is_leaf_node = 1;
        kid.trans_idx = src.trans_idx;
        out.nodes[nu].childs[st] = (int64_t)out.nodes.size();
        out.nodes.emplace_back(); memcpy(&out.nodes.back(), &kid, sizeof(kid));
        out.w.push_back(in.w[it.u]); out.a.push_back(in.a[it.u]);
      }
      out.nodes[nu].is_leaf_node = 0;
      out.nodes[nu].trans_idx = -1;
      out.w[nu] = kInitNodeStat; out.a[nu] = kInitNodeStat;
    } else {
      for (int st = 7; st >= 0; st--)  // reversed: the stack pops slot 0 first (pre-order, slots ascending)
        if (src.childs[st] >= 0) stack.push_back({src.childs[st], nu, st});
    }
  }
  return out;
}

}  // namespace

extern "C" {

int gfn_octree_proc_host(const void* nodes_host, int64_t n_nodes, const int64_t* weight_stats_host,
                         const int64_t* alpha_stats_host, const int64_t* visit_cnt_host, int compact, int subdivide,
                         int brute_force, void** nodes_out_host, int64_t* n_nodes_out, int64_t** weight_stats_out_host,
                         int64_t** alpha_stats_out_host) {
  if (!nodes_host || !weight_stats_host || !alpha_stats_host || !visit_cnt_host || !nodes_out_host || !n_nodes_out ||
      !weight_stats_out_host || !alpha_stats_out_host || n_nodes < 1) {
    gfn_set_error("gfn_octree_proc_host: bad arguments");
    return GFN_ERR_ARG;
  }
  const gfn_tree_node* t = static_cast<const gfn_tree_node*>(nodes_host);
  if (!well_formed(t, n_nodes)) {
    gfn_set_error("gfn_octree_proc_host: node blob is not a tree rooted at node 0 (parent / child links disagree)");
    return GFN_ERR_ARG;
  }
  if (!compact) {
    // every call site of the reference passes compact = true (PersSampler_cuda.cu:668-675); without it the reference's
    // renumbering drops invalid leaves while their parents still point at them and trips its own CHECK_GE (:300)
    gfn_set_error("gfn_octree_proc_host: compact = 0 is not a state the reference can be in");
    return GFN_ERR_ARG;
  }
  Tree tr = compact_tree(t, n_nodes, weight_stats_host, alpha_stats_host);
  if (subdivide) tr = subdivide_tree(tr, visit_cnt_host, brute_force != 0);
  const size_t m = tr.nodes.size();
  void* nb = malloc(m * sizeof(gfn_tree_node));
  int64_t* wb = static_cast<int64_t*>(malloc(m * sizeof(int64_t)));
  int64_t* ab = static_cast<int64_t*>(malloc(m * sizeof(int64_t)));
  if (!nb || !wb || !ab) {
    free(nb); free(wb); free(ab);
    gfn_set_error("gfn_octree_proc_host: out of host memory");
    return GFN_ERR_ARG;
  }
  memcpy(nb, tr.nodes.data(), m * sizeof(gfn_tree_node));
  memcpy(wb, tr.w.data(), m * sizeof(int64_t));
  memcpy(ab, tr.a.data(), m * sizeof(int64_t));
  *nodes_out_host = nb; *n_nodes_out = (int64_t)m; *weight_stats_out_host = wb; *alpha_stats_out_host = ab;
  return GFN_OK;
}

}  // extern "C"
