/*
 * gfnerf_amd.h -- C ABI of the MI355X-native GF-NeRF per-ray hot path.
 *
 * This is the drop-in boundary for the reference's native operator interface,
 * the TorchScript custom classes registered in
 *   gfnerf/bindings/hashanchored/bindings.cpp:343-399  (my_classes.Hash3DAnchored,
 *   my_classes.PersSampler)
 * plus the PyTorch ops the reference runs between them on the hot path
 * (gfnerf/mlp.py, nerfstudio/cameras/rays.py:178-200, model_components/renderers.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the callee never allocates outputs, never synchronises and never throws:
 *     it enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     0, or a negative gfn_status with the text in gfn_last_error();
 *   - sample arrays are PACKED (CSR): ray r owns samples
 *     [pts_offsets[r], pts_offsets[r+1]); the total N lives on the device in
 *     pts_offsets[n_rays].  Kernels that consume packed samples take `n_max`
 *     (host-known capacity, used for grid sizing) and `n_dev` (device pointer to
 *     the live count, may be NULL when n_max is exact);
 *   - dense [R,1024,.] views of the reference (PersSampler_cuda.cu:437-444) are
 *     produced on request by gfn_samples_to_dense.
 */
#ifndef GFNERF_AMD_H
#define GFNERF_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFN_N_LEVELS 16          /* Hash3DAnchored.h:18 */
#define GFN_N_CHANNELS 2         /* Hash3DAnchored.h:17 */
#define GFN_N_PROS 12            /* PersSampler.h:15 */
#define GFN_MAX_SAMPLE_PER_RAY 1024   /* PersSampler_cuda.cu:9 */
#define GFN_MAX_OCT_PER_RAY 1024      /* PersSampler_cuda.cu:8 */
#define GFN_TREE_NODE_BYTES 128  /* sizeof(TreeNode),  PersSampler.h:39-48 */
#define GFN_TRANS_INFO_BYTES 576 /* sizeof(TransInfo), PersSampler.h:31-37 */

typedef enum {
  GFN_OK = 0,
  GFN_ERR_ARG = -1,      /* bad argument (null pointer, size, range) */
  GFN_ERR_HIP = -2,      /* a HIP runtime call failed */
  GFN_ERR_NO_DEVICE = -3 /* no gfx950 device visible */
} gfn_status;

/* ABI revision of this header: bumped whenever an exported signature changes, is added or removed.  gfn_version() returns
 * the revision the loaded library was built from; a binding written for another revision must refuse to call it (the
 * ctypes loader gf-nerf_amd/_lib.py does: arguments of a changed signature would otherwise shift silently).
 *   100: rounds 1-2.   101: round 3 (gfn_adam_step_ex / gfn_fused_forward / gfn_fused_render lost an int, gfn_hash_fwd_lm /
 *   gfn_field_fwd_x3 / gfn_field_bwd_scatter_x3 gained pointers, gfn_hash_fwd_pair / gfn_table_pair_f16 removed).
 *   102: round 4.   103: round 4 (gfn_sample_rays_record, gfn_fused_render_rec added; GFN_MARCH_REC_FLOATS 6 -> 8: a caller's march_rec
 *   buffer grows by a third).   104, 105: round 5 (gfn_set_plane_dither, gfn_set_march_lanes added; one-plane forms of the x3 plane
 *   arguments; gfn_sample_rays_record / gfn_fused_render_rec gained the paged record form's arguments).   106: round 5
 *   (gfn_sample_rays_continue and the two-phase arguments of gfn_sample_rays_record / gfn_fused_render_rec; gfn_set_record_merge_auto).
 *   107: round 6 -- the five process-wide setters (gfn_set_decoder_cus, gfn_set_plane_dither, gfn_set_record_merge,
 *   gfn_set_record_merge_auto, gfn_set_march_lanes) are GONE: launch policy is a `const gfn_launch_opts*` argument of the entry
 *   points that read it (gfn_hash_fwd_lm, gfn_sample_rays, gfn_sample_rays_record, gfn_field_fwd_x3, gfn_field_bwd_dx_x3,
 *   gfn_field_bwd_x3, gfn_field_bwd_scatter_x3, gfn_fused_forward, gfn_fused_render, gfn_fused_render_rec); the two-phase inference
 *   march is gone too (gfn_sample_rays_continue and those arguments: a 6 % gain on a random-init scene, a 3x loss on a trained one);
 *   gfn_train_losses, gfn_loss_grad_combine added; gfn_fused_render(_rec) gained cam_uniform.
 *   108: round 6 -- gfn_launch_opts.reserved became colour_x2 (same size and offsets: a 107 caller's zeroed word is the default). */
#define GFN_ABI_VERSION 108

const char* gfn_last_error(void);
int gfn_version(void);

/* Launch policy of ONE call.  The library keeps no copy and no state of its own: two engines, or a render thread beside the training
 * thread, each pass their own (until ABI 106 these were process-wide setters that callers set, launched under and restored -- the
 * stash-on-`this` hazard of Hash3DAnchored.cpp:104-106 rebuilt; ctypes releases the GIL around every call).  NULL, or a zeroed
 * struct, is the default policy.
 *   decoder_cus   cap on the workgroups (= CUs) of the persistent kernels -- the fused x3 decoder kernels (one 512-thread workgroup
 *                 per CU with the whole LDS), gfn_fused_forward / gfn_fused_render(_rec), gfn_hash_fwd_lm's sweep -- so that they
 *                 leave the other CUs to a kernel of another stream (the next batch's sampler); 0 or >= 256: every CU.  Only the
 *                 grid depends on it, never a value.
 *   plain_rounding  != 0: the one-plane (planes_lo == NULL) producers round WITHOUT their per-sample dither (csrc/x3_device.h:
 *                 sample q's activations are stored as rne(a (1 + u_q)), its dz as rne(dz (1 - u_q)), u_q a fixed hash of q in
 *                 +-2^-11, so that neighbouring samples' rounding errors are independent); default: dithered.
 *   record_merge  gfn_field_bwd_scatter_x3 / gfn_field_bwd_dx_x3 in record form without planes (the focal stage): GFN_MERGE_OFF one
 *                 record per sample and cell; GFN_MERGE_ON contributions of consecutive samples of a ray that fall into one cell are
 *                 summed first (the global stage always does; a merged run is rounded to f16 once more: the choice changes bits);
 *                 GFN_MERGE_DEVICE both forms are launched and the DEVICE runs the one that fits this step's own counts -- the
 *                 merging form when *merge_front >= merge_frac * *merge_n (device i32 words: the samples in front of the
 *                 termination points, e.g. gfn_phase_list's front count, and the batch's samples) -- so that a replayed step and an
 *                 eagerly launched one decide alike.
 *   march_lanes   lanes per ray of the samplers' march: 16 (a row per ray: batches that leave the machine empty), 4 (a quad: a
 *                 quarter of the vector work per ray), 0 = by batch size (a quad from 32 Ki rays on; GFN_MARCH_LANES in the
 *                 environment, read once, overrides that).  Both give the same bits.
 *   colour_x2     gfn_fused_render_rec in the tile renderer's form (two tables with their own primes, no bias pool, the embedding folded:
 *                 cam_uniform or none): != 0: the colour head's layer 2 (128 -> 128, half of the decoder's MFMAs; behind head layer 1, two
 *                 layers in front of the sigmoid) takes its ACTIVATION operand as one f16 number, rounded to nearest, instead of the
 *                 split pair hi + lo' -- two MFMAs per product sum instead of three and no lo' half to make (csrc/x3_device.h X2_H1).
 *                 Everything the DENSITY depends on (hash features, base network, head layer 1's geo input) keeps the split-f16 x3
 *                 form: sigma, accumulation, depth and every ray's termination point are the same bits either way; rgb moves by
 *                 that layer's f16 rounding (measured: DESIGN.md section 4; 5.5e-5 ... 1.3e-4 at the worst pixel of full-HD frames of trained scenes).  Other
 *                 forms of the launch ignore it.  GFN_RENDER_X2=<layer mask> in the environment (read per call; a measurement,
 *                 tools/exp_render_x2.py) overrides it; 0 there forces the x3 form. */
#define GFN_MERGE_OFF 0
#define GFN_MERGE_ON 1
#define GFN_MERGE_DEVICE 2
typedef struct gfn_launch_opts {
  int32_t decoder_cus;
  int32_t plain_rounding;
  int32_t record_merge;
  int32_t march_lanes;
  const int32_t* merge_front;
  const int32_t* merge_n;
  float merge_frac;
  int32_t colour_x2;
} gfn_launch_opts; /* 40 B */
/* number of visible HIP devices whose arch is gfx950 (0 on a CPU-only host). */
int gfn_device_count(void);

/* ---------------------------------------------------------------------- *
 * Checkpoint ABI structs (byte layouts of the blobs PersSampler::States()
 * returns, PersSampler.cpp:969-979).
 * ---------------------------------------------------------------------- */
typedef struct {
  float center[3];
  float side_len;
  int64_t parent;
  int64_t childs[8];
  uint8_t is_leaf_node;
  uint8_t pad0[7];
  int64_t trans_idx;
  int64_t block_idx;
  uint8_t pad1[16];
} gfn_tree_node; /* 128 B */

typedef struct {
  float w2xz[GFN_N_PROS][2][4];
  float weight[3][GFN_N_PROS];
  float center[3];
  float side_len;
  float dis_summary;
  uint8_t pad[28];
} gfn_trans_info; /* 576 B */

/* ====================================================================== *
 * Hash3DAnchored  (replaces my_classes.Hash3DAnchored.AnchoredQuery and its
 * autograd backward: bindings.cpp:317-321; Hash3DAnchored_cuda.cu:160-239)
 * ====================================================================== */
/* scales[l] = 2^(7 l/15 + 3) (Hash3DAnchored_cuda.cu:28), host helper. */
void gfn_hash_level_scales_host(float out_host[GFN_N_LEVELS]);

/* feat_pool f32 [16*T*2] master table (States()[0]); prim_pool i32 [16,V,3];
 * bias_pool f32 [16*V,3] or NULL for the all-zero pool; pts f32 [n,3] already in table space
 * ((warped+1.5)/3, nerfacto_field.py:431); anchors i32 or i64, element i at
 * anchors[i*anchor_stride] (stride 1 = the reference's [n] tensor, 3 = column 0 of
 * the packed sampler's [n,3] anchors).  out f32 [n,32].  anchors_i64 != 0 selects
 * the reference's int64 anchors.  accumulate != 0: out += features (the focal
 * stage's residual table, nerfacto_field.py:485-486). */
int gfn_hash_fwd(const float* feat_pool, const int32_t* prim_pool, const float* bias_pool,
                 const float* pts, const void* anchors, int anchors_i64, int anchor_stride,
                 int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                 int accumulate, float* out, void* stream);

/* Same lookup from f16 shadow tables (u16 [16*T*2], element i = f16(master[i]) RNE, kept
 * current by gfn_adam_step_ex): identical results to gfn_hash_fwd, half the gather bytes.
 * table2_f16 != NULL fuses the focal stage's residual lookup: out = feat(table) + feat(table2);
 * prim_pool2 = the second table's own primes (every Hash3DAnchored draws its own,
 * Hash3DAnchored.cpp:43-50), or NULL when it shares prim_pool. */
int gfn_hash_fwd_f16(const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool,
                     const int32_t* prim_pool2, const float* bias_pool, const float* pts, const void* anchors, int anchors_i64,
                     int anchor_stride, int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                     float* out, void* stream);
/* The same lookup as gfn_hash_fwd_f16 in LEVEL-MAJOR order (csrc/lm_lookup.hip): every workgroup owns a run of
 * samples and all workgroups sweep level by level, table by table, and -- for levels >= halves_from -- one
 * address part (1 / 2^split_log2 of the level's slots; split_log2 = 0: halves, 4 MB each at T = 2^21)
 * at a time, so that the slots being gathered fit the XCDs' L2 instead of missing to the
 * fabric.  Bit-identical features.  anchors i32, element i at anchors[i*anchor_stride].  planar != 0: out is
 * level-major, float2 [16][n_max] (coalesced stores; what gfn_field_fwd_x3 reads with feat_stride = n_max).
 * accumulate != 0: out += features.  One launch per table.
 * list / list_n (device i32 [*list_n] ascending sample indices + its length, both or neither): only those samples are looked
 * up -- every workgroup takes an equal run of the LIST --, the others' feature slots are left untouched, and the launch does
 * nothing when *list_n == 0: the phased (termination-aware) training forward, see gfn_composite_fwd_phased. */
int gfn_hash_fwd_lm(const uint16_t* table_f16, const uint16_t* table2_f16, const int32_t* prim_pool,
                    const int32_t* prim_pool2, const float* bias_pool, const float* pts, const int32_t* anchors,
                    int anchor_stride, int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                    int halves_from, int split_log2, int planar, int accumulate, float* out,
                    const int32_t* list, const int32_t* list_n, const gfn_launch_opts* opts, void* stream);
/* f32 master -> f16 shadow (the reference's per-call pool.to(f16), Hash3DAnchored_cuda.cu:185). */
int gfn_table_to_f16(const float* feat_pool, uint16_t* table_f16, int64_t n_elems, void* stream);

/* grad_out f32 [n,32] -> grad_pool [16*T*2], ACCUMULATED (caller zeroes).  Every
 * contribution is quantised like Hash3DAnchored_cuda.cu:147-151,219 (x128 -> f16,
 * product -> f16); contributions of consecutive samples that fall in the same cell are
 * summed in registers first.  grad_f16 == 0: f32 pool, atomics in f32, the /128 (:238)
 * already applied.  grad_f16 != 0: u16(f16) pool accumulated with packed f16 atomics and
 * still carrying the x128 scale -- the reference's own accumulation (:151,:222); the
 * consumer divides by 128 (gfn_adam_step_ex grad_scale). */
int gfn_hash_bwd(const int32_t* prim_pool, const float* bias_pool, const float* pts,
                 const void* anchors, int anchors_i64, int anchor_stride, const float* grad_out,
                 int64_t n_max, const int32_t* n_dev, int64_t n_volumes, int log2_T,
                 void* grad_pool, int grad_f16, void* stream);

/* ====================================================================== *
 * PersSampler  (replaces my_classes.PersSampler.GetSamples / UpdateOctNodes:
 * bindings.cpp:37-63; PersSampler_cuda.cu:321-477, 584-665)
 * ====================================================================== */
/* rays_d_raw f32 [R,3] -> unit directions (PersSampler_cuda.cu:323). */
int gfn_normalize_dirs(const float* rays_d_raw, float* rays_d, int64_t n_rays, void* stream);

/* K3 (PersSampler_cuda.cu:53-152), both passes + device-side scan, no host sync.
 * nodes: blob of gfn_tree_node[n_nodes].  oct_offsets i32 [R+1] (exclusive scan,
 * ray order -> deterministic, unlike the reference's atomicAdd packing);
 * hit_idx i32 [hit_cap], hit_near_far f32 [hit_cap,2].  hit_cap >= R*max_hits
 * can never overflow; a smaller cap truncates and sets *overflow_dev != 0. */
int gfn_trace(const float* rays_o, const float* rays_d, int64_t n_rays, const void* nodes,
              int64_t n_nodes, float near_bound, float far_bound, int max_hits,
              int32_t* oct_offsets, int32_t* hit_idx, float* hit_near_far, int64_t hit_cap,
              int32_t* overflow_dev, void* stream);

/* K4 (PersSampler_cuda.cu:190-318), count + scan + fill.  noise f32 [1024+R+10]
 * (rays_noise, :380-389, already multiplied by the fineness).  Packed outputs of
 * capacity pts_cap: world_pts/warped_pts/dirs f32 [.,3] (world_pts, dirs may be
 * NULL), dists, ts f32 [.], anchors i32 [.,3] (trans_idx, oct_idx, block_idx),
 * ray_id i32 [.] (may be NULL); pts_offsets i32 [R+1]; first_oct_dis f32 [R].
 * table_space != 0 stores (warped + 1.5) / 3 (nerfacto_field.py:431) in warped_pts. */
int gfn_march(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
              float sample_l, int scale_by_dis, int table_space, const int32_t* oct_offsets, const int32_t* hit_idx,
              const float* hit_near_far, const void* nodes, const void* transes,
              int32_t* pts_offsets, float* world_pts, float* warped_pts, float* dirs, float* dists,
              float* ts, int32_t* anchors, int32_t* ray_id, float* first_oct_dis, int64_t pts_cap,
              int32_t* overflow_dev, void* stream);

/* Fast path of GetSamples() for the training / render loop: K3 once into per-ray dense
 * slots + K4 with a group of lanes per ray (a 16-lane row below 32 Ki rays, a quad from there on); bit-identical samples to gfn_trace + gfn_march.
 * compact_nodes: output of gfn_tree_compact (64 B per node, 64-byte aligned), to be refreshed
 * whenever the node blob changes (gfn_update_nodes).  Workspace: hit_counts i32 [R],
 * hit_idx i32 [R*max_hits], hit_near_far f32 [R*max_hits,2], and march_rec f32 [R*1024*GFN_MARCH_REC_FLOATS], 16-byte aligned (may be
 * NULL): with it the counting march records every sample (32 bytes: t, dist, leaf, transform, warped point, block) and the outputs are
 * rebuilt one thread per sample instead of marching every ray a second time -- same bits, ~1/3 less time.  tree_depth: depth of
 * the deepest node (root = 0) or an upper bound of it, 0 = the reference's 24: the traversal's per-ray stack in LDS holds
 * 7 * tree_depth + 2 entries, and a shallower tree lets a CU traverse more rays at once. */
int gfn_tree_compact(const void* nodes, int64_t n_nodes, void* compact_out, void* stream);
int gfn_sample_rays(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                    const void* compact_nodes, const void* transes, float near_bound, float far_bound,
                    int max_hits, float sample_l, int scale_by_dis, int table_space,
                    int32_t* hit_counts, int32_t* hit_idx, float* hit_near_far,
                    int32_t* pts_offsets, float* world_pts, float* warped_pts, float* dirs, float* dists,
                    float* ts, int32_t* anchors, int32_t* ray_id, float* first_oct_dis, int64_t pts_cap,
                    int32_t* overflow_dev, float* march_rec, int tree_depth, const gfn_launch_opts* opts, void* stream);
/* The first half of the same only: traversal + counting march.  The samples stay where the march recorded them -- slot k of ray r at
 * march_rec[(r * 1024 + k) * GFN_MARCH_REC_FLOATS], counts[r] of them (table_space does not apply: the record holds the warped point) --
 * for a consumer that walks rays, not packed samples: gfn_fused_render_rec.  No scan, no fill, no second copy of the samples (a
 * 138 k-ray render tile: 1.4 GB of the 2.8 GB its samples cost the memory system).  first_oct_dis f32 [R] as gfn_sample_rays.
 * page_table != NULL (round 5): the PAGED form -- march_rec is a pool of n_pages pages of GFN_REC_PAGE slots, handed to the rays as they
 * march (one atomic counter, page_counter i32 [1], zeroed here); page p of ray r is page_table[r * (1024 / GFN_REC_PAGE) + p] (i32
 * [R * 32]; entries beyond a ray's count are not written), slot k of ray r at march_rec[(page_table[r * 32 + k / 32] * 32 + k % 32) *
 * GFN_MARCH_REC_FLOATS].  The pool is sized by the samples (~7 KB per ray at 180 samples per ray) instead of the cap (32 KB per ray); when it
 * runs out, bit 16 of *overflow_dev is set (the slots written from then on alias the pool's last page): the caller samples again
 * into a larger pool.  Bit 8 (gfn_sample_rays too): some ray's traversal stopped at max_hits leaves with nodes still to visit -- the
 * reference's truncation at its max_oct_intersect_per_ray = 1024; a caller that passes a smaller max_hits (hit lists of n_rays *
 * max_hits entries) finds out here that it must ask again with a larger one. */
int gfn_sample_rays_record(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                           const void* compact_nodes, const void* transes, float near_bound, float far_bound,
                           int max_hits, float sample_l, int scale_by_dis, int32_t* hit_counts, int32_t* hit_idx,
                           float* hit_near_far, int32_t* counts, float* first_oct_dis, int32_t* overflow_dev,
                           float* march_rec, int32_t* page_table, int32_t* page_counter, int64_t n_pages, int tree_depth,
                           const gfn_launch_opts* opts, void* stream);

/* Packed -> the reference's dense, zero-padded GetSamples() tensors
 * (PersSampler_cuda.cu:437-476): [R,1024,3] f32 x3, [R,1024] f32 x2,
 * anchors i64 [R,1024,3], pts_idx_start_end i64 [R,2].  Any output may be NULL. */
int gfn_samples_to_dense(const int32_t* pts_offsets, int64_t n_rays, const float* world_pts,
                         const float* warped_pts, const float* dirs, const float* dists,
                         const float* ts, const int32_t* anchors, float* d_world, float* d_warped,
                         float* d_dirs, float* d_dists, float* d_ts, int64_t* d_anchors,
                         int64_t* d_pts_idx_start_end, void* stream);

/* K5 + stat update + K6 (PersSampler_cuda.cu:518-665) on packed samples.
 * weight_stats/alpha_stats/visit_cnt i64 [n_nodes] and the node blob are updated
 * in place; scratch i64 [3*n_nodes]. */
int gfn_update_nodes(const int32_t* pts_offsets, int64_t n_rays, const int32_t* anchors,
                     const float* weights, const float* alphas, int64_t n_nodes,
                     int64_t* weight_stats, int64_t* alpha_stats, int64_t* visit_cnt, void* nodes,
                     int64_t* scratch, void* stream);

/* The same in two phases, for ray-data-parallel training: _mark is K5 (:518-574) into scratch
 * ([w_adder | a_adder | mark] i64 [3*n_nodes], initialised here) and visit_cnt; every quantity is a
 * maximum over rays, so ranks combine scratch and visit_cnt with an elementwise MAX all-reduce and
 * _apply (the statistics update :641-655 + K6 :576-582) then gives every replica the tree one
 * process would have computed from all the rays.  Sample node indices >= n_nodes are skipped. */
int gfn_update_nodes_mark(const int32_t* pts_offsets, int64_t n_rays, const int32_t* anchors,
                          const float* weights, const float* alphas, int64_t n_nodes, int64_t* visit_cnt,
                          int64_t* scratch, void* stream);
int gfn_update_nodes_apply(int64_t n_nodes, const int64_t* scratch, int64_t* weight_stats, int64_t* alpha_stats,
                           void* nodes, void* stream);

/* K12 QueryTreeNodeCenterKernel / PersSampler::QueryTreeNodeCenters (PersSampler_cuda.cu:984-1029;
 * `qurey_tree_nodes_centers`, bindings.cpp:396): centers[i] = tree_nodes[anchors[i]].center for
 * anchors i64 [n_pts] in [0, n_nodes); other rows are left as the caller initialised them (zeros). */
int gfn_query_tree_nodes_centers(const void* nodes, int64_t n_nodes, const int64_t* anchors, int64_t n_pts, float* centers,
                                 void* stream);

/* K7 MarkInvisibleNodes (PersSampler_cuda.cu:680-742): trans_idx = -1 for every node that no camera
 * sees.  intri f32 [n_cams,3,3], w2c f32 [n_cams,3,4], bounds f32 [n_cams,2], row-major (the tensors
 * InitSampler receives, bindings.cpp:24-35). */
int gfn_mark_invisible_nodes(void* nodes, int64_t n_nodes, const float* intri, const float* w2c,
                             const float* bounds, int64_t n_cams, void* stream);

/* PersOctree::ProcOctree (PersSampler.cpp:154-417), host side like the reference: compact != 0 prunes
 * subtrees without a valid leaf, splices single-child chains and renumbers; subdivide != 0 splits every
 * leaf with visit_cnt > 4 (every leaf when brute_force) into 8 octants that inherit its warp and
 * statistics.  compact must be non-zero (every call site of the reference passes true).  Inputs are HOST arrays (nodes blob, i64 stats [n_nodes]); outputs are malloc'd host
 * arrays (gfn_free_host).  The caller uploads them and zeroes visit_cnt (:415). */
int gfn_octree_proc_host(const void* nodes_host, int64_t n_nodes, const int64_t* weight_stats_host,
                         const int64_t* alpha_stats_host, const int64_t* visit_cnt_host, int compact,
                         int subdivide, int brute_force, void** nodes_out_host, int64_t* n_nodes_out,
                         int64_t** weight_stats_out_host, int64_t** alpha_stats_out_host);

/* ====================================================================== *
 * Field pieces (replace gfnerf/mlp.py MLPNetwork, the tcnn SH(4) encoding and
 * the glue of GFNeRFField.get_density/get_outputs, nerfacto_field.py:412-591)
 * ====================================================================== */
/* y[n,out_dim] = act(x[n,in_dim(ld ldx)] W[out_dim,in_dim]^T + b); act 0 none,
 * 1 ReLU, 2 Sigmoid.  f32 in / f32 accumulate on v_mfma_f32_32x32x2_f32. */
int gfn_linear_fwd(const float* x, int ldx, const float* W, const float* b, int64_t n_max,
                   const int32_t* n_dev, int in_dim, int out_dim, int act, float* y, int ldy,
                   void* stream);
/* dz = dy * act'(y);  dx[n,in_dim] = dz W (NULL to skip);  dW += dz^T x, db += sum dz
 * (ACCUMULATED, f32 atomics across workgroups; caller zeroes). */
int gfn_linear_bwd(const float* x, int ldx, const float* W, const float* y, int ldy,
                   const float* dy, int lddy, int64_t n_max, const int32_t* n_dev, int in_dim,
                   int out_dim, int act, float* dx, int lddx, float* dW, float* db, void* stream);

/* The whole decoder in one launch (nerfacto_field.py:455-555): feat f32 [n,32] (hash features)
 * -> base_network -> sigma = trunc_exp(h0+1); [SH | geo | appearance] -> mlp_head -> rgb.
 * weights: 10 device pointers {bW0[128,32], bb0, bW1[16,128], bb1, hW0[128,63], hb0,
 * hW1[128,128], hb1, hW2[3,128], hb2} (hidden_dim 128, config.py:124-125).  sh f32 [R,16]
 * (gfn_sh4), emb f32 [n_cams,32] or NULL (zeros), cam_idx i32 [R], ray_id i32 [n].
 * h0 f32 [n] and masks u64 [n,3,2] (ReLU bit masks of the three 128-wide layers) are written
 * when non-NULL and are all the backward needs. */
int gfn_field_fwd(const float* feat, const int32_t* ray_id, const float* sh, const float* emb,
                  const int32_t* cam_idx, const float* const* weights, int64_t n_max,
                  const int32_t* n_dev, float* sigma, float* rgb, float* h0, uint64_t* masks, void* stream);
/* (d_sigma [n], d_rgb [n,3]) -> dfeat f32 [n,32] with the decoder frozen (focal stage). */
int gfn_field_bwd_dx(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0,
                     const uint64_t* masks, const float* const* weights, int64_t n_max,
                     const int32_t* n_dev, float* dfeat, void* stream);

/* The same two kernels on the f16 matrix pipe at f32 accuracy (every operand split into two f16
 * halves x ~= hi + lo' * 2^-11, three MFMAs per product sum, ~1e-6 relative; see csrc/field_x3.hip).
 * The weight images are built by gfn_field_pack_x3 whenever the weights change (either image may be
 * NULL) and are read-only to the kernels; buffers of GFN_FIELD_X3_{FWD,BWD}_BYTES, 16-byte aligned.
 *
 * Global stage (decoder trains): gfn_field_fwd_x3 with planes_{hi,lo} != NULL also keeps every layer
 * input as split-f16 planes.  planes_hi != NULL with planes_lo == NULL (the same in gfn_field_bwd_x3, gfn_field_bwd_scatter_x3,
 * gfn_fused_forward; dz_lo / a_lo of gfn_wgrad_x3_layers): ONE plane per operand, its values rounded to f16 once, to nearest
 * even, under a per-sample dither (gfn_launch_opts.plain_rounding switches it off) -- for the weight gradient only, a sum over every sample of the batch
 * in which zero-mean, sample-to-sample independent rounding averages out; half the plane traffic (csrc/x3_device.h, PL_RNE).
 * The backward's one-plane buffer has GFN_FIELD_X3_BWD_PLANE_LD_RNE columns: dza1 -- whose layer contracts against the hash
 * features, exact in f16 -- keeps its lo' plane there (C=408 W=128) and is not dithered, nor is feat.  A plane buffer holds plane_cap * LD u16 (plane_cap >= n_max, multiple of
 * 8) and is PLANE-MAJOR: the plane with first column C and width W is the contiguous array
 * [plane_cap][W] at element offset plane_cap * C.  Forward (LD = GFN_FIELD_X3_FWD_PLANE_LD): feat
 * C=0 W=32 | a1 C=32 W=128 | xp C=160 W=64 (base output rows 0..15, SH 16, appearance 32) | c1 C=224
 * W=128 | c2 C=352 W=128.  gfn_field_bwd_x3 writes dz of every layer (LD = GFN_FIELD_X3_BWD_PLANE_LD):
 * dz3 C=0 W=8 | dh C=8 W=16 | dz2 C=24 W=128 | dz1 C=152 W=128 | dza1 C=280 W=128, all multiplied by the power-of-two scale
 * gfn_grad_scale_host(*gmax_dev) (gmax_dev: device f32 = largest |gradient| entering the decoder,
 * from gfn_grad_max into a zeroed word), and accumulates the appearance-embedding gradient into
 * demb f32 [n_cams,32] (may be NULL).  gfn_wgrad_x3 turns pairs of planes into weight gradients.
 * feat_stride: 0 = feat is f32 [n,32]; else feat is level-major float2 [16][feat_stride] (gfn_hash_fwd_lm, planar).
 * gfn_field_fwd_x3 with list / list_n (as in gfn_hash_fwd_lm) decodes only the listed samples, 32 consecutive ENTRIES per tile;
 * outputs and plane rows of the other samples are not touched; nothing is done when *list_n == 0. */
#define GFN_FIELD_X3_FWD_BYTES 149248
#define GFN_FIELD_X3_BWD_BYTES 131072
#define GFN_FIELD_X3_FWD_PLANE_LD 480
#define GFN_FIELD_X3_BWD_PLANE_LD 408
#define GFN_FIELD_X3_BWD_PLANE_LD_RNE 536 /* one-plane form (planes_lo == NULL): the backward's buffer also holds dza1's lo' plane, C=408 W=128 */
int gfn_field_pack_x3(const float* const* weights, void* fwd_image, void* bwd_image, void* stream);
int gfn_field_fwd_x3(const float* feat, const int32_t* ray_id, const float* sh, const float* emb,
                     const int32_t* cam_idx, const void* fwd_image, int64_t n_max, const int32_t* n_dev,
                     float* sigma, float* rgb, float* h0, uint64_t* masks, uint16_t* planes_hi,
                     uint16_t* planes_lo, int64_t plane_cap, int64_t feat_stride, const int32_t* list,
                     const int32_t* list_n, const gfn_launch_opts* opts, void* stream);
int gfn_field_bwd_dx_x3(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0,
                        const uint64_t* masks, const void* bwd_image, int64_t n_max, const int32_t* n_dev,
                        float* dfeat, const gfn_launch_opts* opts, void* stream);
int gfn_field_bwd_x3(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0,
                     const uint64_t* masks, const void* bwd_image, int64_t n_max, const int32_t* n_dev,
                     float* dfeat, const float* gmax_dev, uint16_t* planes_hi, uint16_t* planes_lo,
                     int64_t plane_cap, const int32_t* ray_id, const int32_t* cam_idx, float* demb,
                     const gfn_launch_opts* opts, void* stream);
/* Decoder backward + hash scatter in one kernel: d feat never leaves the chip -- each lane of the d-feat
 * tile hashes 8 levels of its sample and adds into grad_pool (same quantisation / accumulation rules as
 * gfn_hash_bwd; grad_f16 as there; pts / anchors [n,3] i32 / prim_pool of the table being trained).
 * planes_* / gmax_dev / ray_id / cam_idx / demb as in gfn_field_bwd_x3 (all NULL in the focal stage). */
int gfn_field_bwd_scatter_x3(const float* d_sigma, const float* d_rgb, const float* rgb, const float* h0,
                             const uint64_t* masks, const void* bwd_image, int64_t n_max, const int32_t* n_dev,
                             const float* pts, const int32_t* anchors, const int32_t* prim_pool,
                             const float* bias_pool, int64_t n_volumes, int log2_T, void* grad_pool, int grad_f16,
                             const float* gmax_dev, uint16_t* planes_hi, uint16_t* planes_lo, int64_t plane_cap,
                             const int32_t* ray_id, const int32_t* cam_idx, float* demb, void* records,
                             uint32_t* record_counts, int64_t record_capacity, const int32_t* list,
                             const int32_t* list_n, const gfn_launch_opts* opts, void* stream);
/* records != NULL (f16 pool only) DEFERS the atomics of gfn_field_bwd_scatter_x3: the pool's live slots (f16
 * pairs [0, 17T/2)) are cut into ranges of 2^GFN_GRAD_RANGE_LOG2 slots, and every add becomes a record (slot u32,
 * packed f16x2 value) in the stream of (range, workgroup): records[(range * 256 + workgroup) * cap + i] with
 * cap = record_capacity / (n_ranges * 256) 8-byte records (16-byte aligned buffer), counts in record_counts u32
 * [n_ranges * 256] (gfn_grad_record_ranges(log2_T) ranges); a full stream falls back to the atomic.
 * gfn_grad_records_reduce then lets the workgroup that owns a range add its 256 streams into a 128 KiB LDS image and
 * add the image to grad_pool_f16 with plain stores (csrc/grad_part.hip) -- the L2 executes only ~21 G scattered
 * atomics/s.  The image holds EXACT fixed-point sums (one ds_add_u64 of both halves in units of 2^-24, the f16 grid;
 * rounded to f16 once) for every range whose sum of |component| stays below 128, i.e. whose 32-bit fields cannot
 * have overflowed -- LDS integer atomics run 6x faster than floating-point ones (tools/lds_atom_bench.hip) -- and
 * the reference's own f16 accumulation (ds_pk_add_f16, Hash3DAnchored_cuda.cu:151) for the others, or for all of
 * them with f16_accumulate != 0.  log2_T <= 21.
 * record_counts must be all zero before the FIRST backward in record form; gfn_grad_records_reduce CONSUMES
 * them (leaves them zero for the next backward; counts_seen, when not NULL, receives a copy of what it
 * consumed), so a training loop never fills them.  range_begin / range_end: only the ranges [begin, end) (0, 0 = all
 * of them) -- a caller may step the slots of finished ranges (range r = f16 elements [r << 15, (r + 1) << 15) of
 * the pool) while later ranges still accumulate. */
#define GFN_GRAD_RANGE_LOG2 14
#define GFN_MARCH_REC_FLOATS 8 /* per sample slot of gfn_sample_rays' march_rec: t, dist, leaf (i32), transform (i32), warped x y z, block (i32) */
#define GFN_REC_PAGE_LOG2 5    /* gfn_sample_rays_record's paged form: pages of 32 sample slots (1 KB) */
#define GFN_REC_PAGE (1 << GFN_REC_PAGE_LOG2)
#define GFN_GRAD_MAX_RANGES 1088
int gfn_grad_record_ranges(int log2_T);
int gfn_grad_records_reduce(const void* records, uint32_t* record_counts, int64_t record_capacity, int log2_T,
                            void* grad_pool_f16, uint32_t* counts_seen, int f16_accumulate, int range_begin,
                            int range_end, void* stream);
int gfn_grad_max(const float* d_sigma, const float* d_rgb, const float* h0, int64_t n_max,
                 const int32_t* n_dev, float* gmax_dev, void* stream);
float gfn_grad_scale_host(float gmax);

/* Weight gradient of one decoder layer from split-f16 planes (global stage):
 *   dW[m*ldw + col(k)] += (1/scale) sum_s dz[s][m] a[s][k],   db[m] += (1/scale) sum_s dz[s][m]
 * for m < m_valid, k < k_valid; col(k) = col_map[k] (skipped when negative) or k.  A plane is two
 * u16(f16) arrays, hi and lo' with x ~= hi + lo' * 2^-11; feat_* = stored columns (multiple
 * of 8, <= 128; tile shapes 128x128, 32x128, 128x64, 128x32 after rounding up to 32).  tiled == 0: the
 * arrays are rows [n][ld]; tiled != 0: the layout the fused decoder kernels write,
 * [n / 32][feat / 8][32 samples][8 features] (ld == feat, n a multiple of 32 in the allocation).  gmax_dev
 * (device f32, may be NULL = no scale) is the word gfn_field_bwd_x3 took its scale from: the dz plane
 * carries the factor gfn_grad_scale_host(*gmax_dev), divided out here.
 * ACCUMULATES with f32 atomics (caller zeroes).  Replaces autograd's dW = dz^T a of the nn.Linear
 * layers in gfnerf/mlp.py:35-57. */
int gfn_wgrad_x3(const uint16_t* dz_hi, const uint16_t* dz_lo, int ld_dz, int feat_dz, const uint16_t* a_hi,
                 const uint16_t* a_lo, int ld_a, int feat_a, int64_t n_max, const int32_t* n_dev,
                 const float* gmax_dev, int m_valid, int k_valid, const int32_t* col_map, float* dW, int ldw,
                 float* db, int tiled, void* stream);

/* All layers of the decoder in ONE launch over tile-major planes (the global-stage step: five layers), and without the
 * final atomics: with parts != NULL workgroup b STORES its partial sums of every valid dW / db entry at
 * parts + b * part_stride + (dW - grad_base) -- the partial vectors mirror the flat gradient vector
 * [grad_base, grad_base + grad_len) that all dW / db point into; parts holds gfn_wgrad_x3_parts() such vectors and must
 * have been zeroed ONCE (entries no layer writes stay zero) -- and gfn_wgrad_reduce adds their sum to the gradient
 * (out += sum_b parts[b * part_stride + i], i < n).  parts == NULL: f32 atomics into dW / db like gfn_wgrad_x3. */
typedef struct {
  const uint16_t *dz_hi, *dz_lo, *a_hi, *a_lo; /* whole tile-major planes [n/32][feat/8][32][8] */
  int32_t feat_dz, feat_a;                      /* plane widths: 128x128, 8x128, 128x64, 16x128 or 128x32 */
  int32_t m_valid, k_valid;
  const int32_t* col_map;                       /* plane column -> dW column (< 0: skipped), or NULL */
  float* dW; int32_t ldw;
  float* db;                                    /* or NULL */
} gfn_wgrad_layer;
int gfn_wgrad_x3_parts(void);
int gfn_wgrad_x3_layers(const gfn_wgrad_layer* layers, int n_layers, int64_t n_max, const int32_t* n_dev,
                        const float* gmax_dev, const float* grad_base, int64_t grad_len, float* parts,
                        int64_t part_stride, void* stream);
int gfn_wgrad_reduce(const float* parts, int n_parts, int64_t part_stride, int64_t n, float* out, void* stream);

/* Everything after the sampler in ONE persistent kernel (csrc/fused_fwd.hip): f16-shadow hash lookup
 * (+ residual table2 with its own primes, or NULL) -> x3 decoder -> front-to-back compositing with early
 * termination.  A wave walks one ray in 32-sample tiles and leaves it once the transmittance is below
 * early_T (early_T <= 0: every sample, the reference's semantics); samples behind that point are never
 * looked up and get sigma = rgb = weight = alpha = 0.  Inputs: the packed sampler outputs (pts in table
 * space, anchors i32 [n,3], dt, ts), sh f32 [R,16], emb / cam_idx, a forward weight image.  Per-sample
 * outputs sigma [n], rgb [n,3] (always), h0 + masks (training), weights / alphas / feat [n,32] (optional),
 * activation planes (global stage); per-ray outputs out_rgb [R,3], out_acc, out_depth (un-clipped),
 * t_minmax [2] (for gfn_depth_finalize).  work_counter: 8 device i32 of scratch (one ray queue per
 * eighth of the batch, drained first by the workgroups of one XCD: neighbouring rays share an L2). */
int gfn_fused_forward(const int32_t* pts_offsets, int64_t n_rays, const float* pts, const int32_t* anchors,
                      const float* dt, const float* ts, const uint16_t* table_f16, const uint16_t* table2_f16,
                      const int32_t* prim_pool, const int32_t* prim_pool2, const float* bias_pool,
                      int64_t n_volumes, int log2_T, const void* fwd_image, const float* sh, const float* emb,
                      const int32_t* cam_idx, float early_T, float* sigma, float* rgb, float* h0, uint64_t* masks,
                      float* weights, float* alphas, float* feat, uint16_t* planes_hi, uint16_t* planes_lo,
                      int64_t plane_cap, float* out_rgb, float* out_acc, float* out_depth, float* t_minmax,
                      int32_t* work_counter, const gfn_launch_opts* opts, void* stream);

/* Inference form of gfn_fused_forward (csrc/fused_fwd.hip, fused_render_patch_kernel): the same lookup + decoder +
 * compositing with early termination, but a wave owns a PATCH of 32 neighbouring rays (tile_width > 0: the rays are a
 * tile_width-wide block of pixels, row major, and a patch is patch_width x 32/patch_width pixels; tile_width == 0: 32
 * consecutive rays) and walks them in lockstep, sample index by sample index -- neighbouring pixels' rays cross the same
 * table cells, and lanes of one gather that want the same address cost what one lane costs.  Only the per-ray outputs
 * are written (out_acc / out_depth / t_minmax may be NULL).  n_rays must be a multiple of 32 (and of whole patch rows).
 * cam_uniform != 0: the caller promises that every ray of the launch carries the camera cam_idx[0] (the rays of ONE image: perssampler.py:431-433,
 * nerfacto_field.py:526-538) -- the appearance part of head layer 1, W_h0[:, 31:63] emb[cam], is then added to the layer's bias once per workgroup instead
 * of multiplied out again for every sample (24 of 216 MFMAs per 32 samples); read from the device, so a captured launch serves any camera.
 * Replaces GFNeRFModel.get_outputs under get_outputs_for_camera_ray_bundle (nerfacto.py:531-566, base_model.py:163-186). */
int gfn_fused_render(const int32_t* pts_offsets, int64_t n_rays, const float* pts, const int32_t* anchors,
                     const float* dt, const float* ts, const uint16_t* table_f16, const uint16_t* table2_f16,
                     const int32_t* prim_pool, const int32_t* prim_pool2, const float* bias_pool,
                     int64_t n_volumes, int log2_T, const void* fwd_image, const float* sh, const float* emb,
                     const int32_t* cam_idx, int cam_uniform, float early_T, int tile_width, int patch_width, float* out_rgb,
                     float* out_acc, float* out_depth, float* t_minmax, int32_t* work_counter, const gfn_launch_opts* opts,
                     void* stream);
/* The same on the samples of gfn_sample_rays_record: counts i32 [R] and march_rec (16-byte aligned) instead of the packed arrays.  Slot k
 * of ray r is read where the counting march left it; the table-space point is made from the recorded warped point as the fill pass
 * makes it ((w + 1.5) * (1/3)), so the images are the bits of gfn_sample_rays(table_space = 1) + gfn_fused_render.
 * page_table: NULL, or the page table of gfn_sample_rays_record's paged form (march_rec is then the page pool). */
int gfn_fused_render_rec(const int32_t* counts, int64_t n_rays, const float* march_rec, const int32_t* page_table, const uint16_t* table_f16,
                         const uint16_t* table2_f16, const int32_t* prim_pool, const int32_t* prim_pool2, const float* bias_pool,
                         int64_t n_volumes, int log2_T, const void* fwd_image, const float* sh, const float* emb,
                         const int32_t* cam_idx, int cam_uniform, float early_T, int tile_width, int patch_width, float* out_rgb,
                         float* out_acc, float* out_depth, float* t_minmax, int32_t* work_counter, const gfn_launch_opts* opts,
                         void* stream);

/* SH degree 4 of per-ray unit directions, tiny-cuda-nn convention on
 * (d+1)/2 (nerfacto_field.py:64-70,152-158,519-521), rounded to f16 like tcnn's
 * default output.  dirs f32 [R,3] -> out f32 [R,16]. */
int gfn_sh4(const float* dirs, int64_t n_rays, float* out, void* stream);

/* Head input assembly + density activation (nerfacto_field.py:491-503,540-547):
 *   sigma[i] = exp(h[i,0] + 1);  xh[i, 0:16] = sh[ray_id[i]];  xh[i,16:31] = h[i,1:16];
 *   xh[i,31:63] = emb[cam_idx[ray_id[i]]] (zeros if emb == NULL);  xh[i,63] = 0.
 * h f32 [n,16], xh f32 [n,64]. */
int gfn_head_pack(const float* h, const float* sh, const float* emb, const int32_t* cam_idx,
                  const int32_t* ray_id, int64_t n_max, const int32_t* n_dev, float* sigma,
                  float* xh, void* stream);
/* backward of gfn_head_pack: dh[i,0] = dsigma[i]*exp(clamp(h0+1,-15,15)) (trunc_exp,
 * activations.py:24-42); dh[i,1:16] = dxh[i,16:31]; demb[cam[ray]] += sum over the ray's
 * samples of dxh[i,31:63] (ACCUMULATED; demb may be NULL). */
int gfn_head_unpack_bwd(const float* h, const float* dsigma, const float* dxh,
                        const int32_t* cam_idx, const int32_t* pts_offsets, int64_t n_rays,
                        int64_t n_max, const int32_t* n_dev, float* dh, float* demb, void* stream);

/* ====================================================================== *
 * Ray generation (Cameras.generate_rays, nerfstudio/cameras/cameras.py:560-734,
 * perspective cameras without distortion; GF's lookat_directions :704)
 * ====================================================================== */
/* c2w f32 [n_cams,3,4]; intr f32 [n_cams,4] = (fx, fy, cx, cy).  Rays are either explicit pairs
 * (cam_idx i32 [n], coords f32 [n,2] = (y, x) with the 0.5 pixel offset already added) or, with
 * coords == NULL, the raster run of pixels first_pixel .. first_pixel+n-1 of an image `width`
 * pixels wide seen by camera cam0 (cam_idx == NULL) -- what render.py renders tile by tile.
 * Outputs (origins, lookat, pixel_area, dir_norm may be NULL): origins / directions (unit) /
 * lookat f32 [n,3], pixel_area / dir_norm f32 [n]. */
int gfn_generate_rays(const float* c2w, const float* intr, const int32_t* cam_idx, int32_t cam0,
                      const float* coords, int64_t first_pixel, int32_t width, int64_t n, float* origins,
                      float* directions, float* lookat, float* pixel_area, float* dir_norm, void* stream);

/* ====================================================================== *
 * Compositing  (replaces RaySamples.get_weights_f2nerf, rays.py:178-200, and
 * RGB / Accumulation / Depth("expected") renderers, renderers.py:69-138,195-283)
 * ====================================================================== */
/* early_T > 0 stops a ray once its transmittance drops below early_T.
 * weights/alphas f32 [n] may be NULL.  out_depth is the un-clipped expectation;
 * t_minmax f32 [2] (may be NULL) receives the batch-global {min,max} of t incl.
 * the reference's zero padding, for gfn_depth_finalize. */
int gfn_composite_fwd(const int32_t* pts_offsets, int64_t n_rays, const float* sigma,
                      const float* dt, const float* t, const float* rgb, float early_T,
                      float* weights, float* alphas, float* out_rgb, float* out_acc,
                      float* out_depth, float* t_minmax, void* stream);
/* The termination-aware ("phased") training forward.  The reference evaluates every sample of every ray (1024 padded ones,
 * PersSampler_cuda.cu:437-444, nerfacto_field.py:443); with early_T > 0 the samples behind a ray's termination point
 * (transmittance < early_T) cannot change the image by more than early_T, and here they are not even looked up:
 *   gfn_phase_list (need_more = NULL)   list_a = the first min(K, n_r) samples of every ray, K = *k_split, compacted by a prefix
 *                           sum over the rays (coff i32 [n_rays + 1] workspace); stats[0] = entries, stats[1] = rays.  K >= 1024
 *                           (one phase): no list is written and stats[0] = -1, which the two kernels below read as "every
 *                           live sample"
 *   gfn_hash_fwd_lm / gfn_field_fwd_x3 (list = list_a, list_n = stats)                       -- phase A
 *                           (sigma and rgb of every other sample must be zero: gfn_phase_list clears them when given
 *                           zero_sigma f32 [n] / zero_rgb f32 [n,3], phase A only)
 *   gfn_composite_fwd_phased(k_split = K)   composites all rays (samples behind K count as empty) and marks the rays that are
 *                           longer than K and not yet terminated: need_more u8 [n_rays]; term i32 [n_rays] = samples in front
 *                           of each ray's termination point
 *   gfn_phase_list (need_more)          list_b = samples K.. of those rays; stats[2] = entries, stats[3] = rays
 *   the same two kernels with list = list_b, list_n = stats + 2                              -- phase B (nothing, as a rule)
 *   gfn_composite_fwd_phased(ray_gate = need_more, gate = stats + 2)   composites those rays again, all samples now known
 *   gfn_phase_list (term)               the samples in front of every ray's termination point (the first term[r]); stats[4] =
 *                           entries, stats[5] = rays: the list gfn_field_bwd_scatter_x3 walks (every other sample's incoming
 *                           gradient is zero)
 *   gfn_two_phase_plan      picks the next step's K from this batch's (ray length, termination point) pairs: the K that
 *                           minimises samples(phase A) + [phase B needed] * (fixed_cost_samples + samples(phase B)); when that
 *                           K covers every ray: the latest termination point + the width of the distribution's tail; 1024 when
 *                           the split would save too little (a new split has to save the fraction save_enter of the samples, a running one save_stay); stats_last = (samples of phase A, rays of phase B, samples of phase B, the K just used).
 *                           enabled == 0: K = 1024 (one phase: every sample, the reference's semantics with early_T = 0).
 *                           n_rays < 2^20.
 * All of it is decided on the device: no host round trip, capturable in a hipGraph.  The backward needs no change: samples
 * that were never evaluated hold zero sigma / rgb and receive zero gradients. */
int gfn_phase_list(const int32_t* pts_offsets, int64_t n_rays, const int32_t* k_split, const uint8_t* need_more,
                   const int32_t* term, int32_t* coff, int32_t* list, int32_t* stats, float* zero_sigma, float* zero_rgb,
                   void* stream);
int gfn_composite_fwd_phased(const int32_t* pts_offsets, int64_t n_rays, const float* sigma, const float* dt,
                             const float* t, const float* rgb, float early_T, float* weights, float* alphas,
                             float* out_rgb, float* out_acc, float* out_depth, float* t_minmax,
                             const uint8_t* ray_gate, const int32_t* gate, const int32_t* k_split,
                             uint8_t* need_more, int32_t* term, void* stream);
int gfn_two_phase_plan(const int32_t* pts_offsets, int64_t n_rays, const int32_t* term, int32_t* k_split,
                       int32_t* stats, int32_t* stats_last, float fixed_cost_samples, float save_enter, float save_stay,
                       int enabled, void* stream);
int gfn_depth_finalize(float* depth, int64_t n_rays, const float* t_minmax, float scale_factor,
                       void* stream);
int gfn_composite_bwd(const int32_t* pts_offsets, int64_t n_rays, const float* sigma,
                      const float* dt, const float* rgb, const float* g_rgb, const float* g_acc,
                      float early_T, float* d_sigma, float* d_rgb, void* stream);

/* ====================================================================== *
 * Batch assembly (upstream of ray generation): PixelSampler /
 * ErrorPixelSampler (nerfstudio/data/pixel_samplers.py:47-117,636-723) and the error-map upkeep
 * (gfnerf/gf_pipeline.py:180-185 -> nerfstudio/data/utils/dataloaders.py:140-142).  Images, error map
 * and its two-level CDF stay in HBM; random numbers are INPUTS, so every call is a pure function.
 * A pixel is addressed by lin = (c * H + y) * W + x over the cached image stack [n_images,H,W].
 * ====================================================================== */
#define GFN_ERRMAP_BLOCK 4096      /* pixels per f64 partial sum of the error-map CDF */
#define GFN_PIXEL_CHUNK (1 << 24)  /* weighted_choice_multinomial's chunk (pixel_samplers.py:639) */
#define GFN_PIXEL_ATTEMPTS 64      /* redraws of a duplicate before it is kept (and counted in status[1]) */

/* PixelSampler.sample_method without a mask (:63-67): lin[i] of floor(u[i] * (n_images, H, W)), u f32 [n,3]
 * in [0,1) (f32 product like torch's; clamped where the reference would index out of range). */
int gfn_pixel_uniform(const float* u, int64_t n, int32_t n_images, int32_t height, int32_t width, int64_t* lin,
                      void* stream);
/* collate_image_dataset_batch (:73-103) + RayGenerator's coordinate lookup (ray_generators.py:47-52) for the
 * pixels lin i64 [n].  images: f32 [n_images,H,W,3], or u8 (images_u8 != 0; converted as u8 / 255.0f like the
 * dataset does); image_idx / rel_cam_idx i32 [n_images] (NULL = identity).  Outputs, each may be NULL:
 * indices i32 [n,3] = (image_idx[c], y, x); slot i32 [n] = c; rel_cam i32 [n] = rel_cam_idx[c];
 * gt f32 [n,3]; coords f32 [n,2] = (y + 0.5, x + 0.5), the form gfn_generate_rays takes. */
int gfn_pixel_collate(const int64_t* lin, int64_t n, int32_t n_images, int32_t height, int32_t width,
                      const void* images, int32_t images_u8, const int32_t* image_idx,
                      const int32_t* rel_cam_idx, int32_t* indices, int32_t* slot, int32_t* rel_cam, float* gt,
                      float* coords, void* stream);
/* block_sums f64 [ceil(n_pixels / GFN_ERRMAP_BLOCK)] of error_map f32 [n_pixels], from scratch. */
int gfn_errmap_block_sums(const float* error_map, int64_t n_pixels, double* block_sums, void* stream);
/* error_map[c,y,x] = |gt - pred| summed over rgb for indices i32 [n,3] = (c, y, x) (out-of-range rows are
 * skipped); block_sums (may be NULL) follows by the exact f64 difference. */
int gfn_errmap_update(const int32_t* indices, int64_t n, const float* pred, const float* gt, int32_t n_images,
                      int32_t height, int32_t width, float* error_map, double* block_sums, void* stream);
/* ErrorPixelSampler.sample_method (:676-723): draws [0, n_weighted) follow weighted_choice_multinomial
 * (probability proportional to error_map, without replacement, per-2^24-pixel-chunk quota and left-over rule
 * of :636-674), draws [n_weighted, n) are uniform without replacement; the two groups are independent.
 * bits i64 [n]: 64 random bits per draw (attempt a of draw d uses splitmix64(bits[d] + (a+1) * golden)).
 * Workspaces: prefix_ws f64 [blocks + 1], cand_ws i64 [n], pending_ws i32 [n], table_ws 16 B x
 * gfn_errmap_table_slots(larger group) (may be NULL when that is <= 128 KiB: the set then lives in LDS).
 * Out: lin i64 [n]; status i32 [2] = {weighted draws whose range had zero total weight (torch.multinomial
 * raises; drawn uniformly here), duplicates left after GFN_PIXEL_ATTEMPTS redraws (a map with nearly all of
 * its weight on fewer pixels than draws)}. */
int64_t gfn_errmap_table_slots(int64_t n);
int gfn_errmap_sample(const float* error_map, int64_t n_pixels, const double* block_sums, const int64_t* bits,
                      int64_t n, int64_t n_weighted, double* prefix_ws, int64_t* cand_ws, int32_t* pending_ws,
                      void* table_ws, int64_t* lin, int32_t* status, void* stream);

/* ====================================================================== *
 * Loss + optimiser (nerfstudio losses.py:73-84,
 * torch.optim.Adam as configured in gfnerf/config.py:130)
 * ====================================================================== */
/* loss_dev[0] += sum(sqrt((pred-gt)^2+eps^2))/rows; g_pred = d loss / d pred. */
int gfn_charbonnier(const float* pred, const float* gt, int64_t rows, float eps, float* loss_dev,
                    float* g_pred, void* stream);
/* The launches between the compositing and the backward of a training iteration through the plugin surface, in one (csrc/render.hip): out3[0] =
 * the Charbonnier loss of gfn_charbonnier (WRITTEN, nothing to zero first; sums in a fixed order: the same bits every run), out3[1] = 0 (for gfn_s3im
 * to add its loss into), out3[2] = PSNR = -10 log10(mean((pred - gt)^2)) (get_metrics_dict, gfnerf/nerfacto.py:300,624); g_char f32 [rows,3] = d out3[0] /
 * d pred (may be NULL); g_zero f32 [rows,3] zeroed (may be NULL: the buffer gfn_s3im scatters its gradient into); and, with error_map != NULL, the block
 * stage's error-map write-back (gf_pipeline.py:180-185) exactly as gfn_errmap_update does it, the image of row i being slot[i] when slot != NULL (the
 * position in the cached stack) and indices[3 i] otherwise.  workspace: NULL (one workgroup does it all), or GFN_TRAIN_LOSSES_WS f32 words that are zero
 * before the first call and belong to ONE caller at a time (an engine's own: the kernel leaves them zero again) -- up to 64 workgroups, the last one to
 * finish adds the others' partial sums in workgroup order.  rows <= 2^22. */
#define GFN_TRAIN_LOSSES_WS 132
int gfn_train_losses(const float* pred, const float* gt, int64_t rows, float eps, float* out3, float* g_char, float* g_zero,
                     const int32_t* indices, const int32_t* slot, int32_t n_images, int32_t height, int32_t width, float* error_map,
                     double* block_sums, float* workspace, void* stream);
/* out[i] = g_a[i] * w_a[0] + g_b[i] * w_b[0] (w_* device f32 words; a NULL weight or gradient drops its term): the gradient of
 * `loss = rgb_loss + s3im_loss` (engine/trainer.py:391) with respect to the prediction from the two losses' own gradients and the gradients autograd
 * hands their backward, in one launch. */
int gfn_loss_grad_combine(const float* g_a, const float* w_a, const float* g_b, const float* w_b, int64_t n, float* out, void* stream);
/* S3IM (losses.py:713-794 as configured by gfnerf/config.py:96-100: 4x4 gaussian window, stride 4, zero
 * padding 1): loss_dev[0] += mult * (1 - mean SSIM) over the virtual image [3, patch_h, n_idx/patch_h]
 * whose pixel k is ray index[k] (arange + randperms, :780-789); g_pred [R,3] += mult * d/d pred (may be
 * NULL).  gt, pred f32 [R,3]. */
int gfn_s3im(const float* gt, const float* pred, const int32_t* index, int64_t n_idx, int patch_h, float mult,
             float* loss_dev, float* g_pred, void* stream);
/* torch.optim.Adam (gfnerf/config.py:130), one step of one parameter group.  The two step-dependent factors
 * (lr / (1 - beta1^step), sqrt(1 - beta2^step)) are computed here from `step` -- or, when hyper_dev (device f32[2]) is not
 * NULL, read from it by the kernel: a step captured in a hipGraph takes them from memory the host refreshes before every
 * replay (gfn_adam_hyper_host gives the same two floats).  */
void gfn_adam_hyper_host(int step, float lr, float beta1, float beta2, float* out2);
int gfn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr,
                  float beta1, float beta2, float eps, const float* hyper_dev, void* stream);
/* Adam for a hash table: g is f32 (g_f16 == 0) or f16 (g_f16 != 0) and is multiplied by
 * g_scale first (1/128 for gfn_hash_bwd's f16 pool); shadow_f16 (may be NULL) receives
 * f16(p) of every updated element so the f16 lookups stay current. */
int gfn_adam_step_ex(float* p, const void* g, int g_f16, float g_scale, float* m, float* v, int64_t n,
                     int step, float lr, float beta1, float beta2, float eps, uint16_t* shadow_f16,
                     const float* hyper_dev, void* stream);
int gfn_fill_zero(void* p, int64_t bytes, void* stream);
/* The random draws of one training step, on the device, in one launch (replaces torch.rand's Philox stream, which a captured step
 * can only use at the price of two extra fills in front of every replay): noise[i] = (u + add) * mul with u uniform in [0, 1) -- the
 * march noise of GetSamples, PersSampler_cuda.cu:380-389: (rand + 0.5) * fineness -- and keys[j] = 63 random bits, whose argsort
 * per row is a uniform random permutation (S3IM's torch.randperm, losses.py:780-789).  Philox4x32-10 keyed by `seed`, counter =
 * (element, state[0]); state = device u64[2] {calls so far, scratch}, zero-initialised by the caller and advanced by the launch
 * itself, so a hipGraph replay of this launch draws fresh numbers.  The reference's random streams cannot be reproduced (they are
 * the CUDA generator's); the distribution is what the tests hold. */
int gfn_step_draws(uint64_t seed, uint64_t* state, float* noise, int64_t n_noise, float add, float mul, int64_t* keys, int64_t n_keys,
                   void* stream);
/* n <= 8 small device-to-device copies (src[i] -> dst[i], bytes[i] a multiple of 4, 4-byte aligned) in ONE launch: the batch
 * hand-over in front of a captured training step (gfnerf_amd.stepgraph).  src / dst / bytes are HOST arrays. */
int gfn_copy_list(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream);

/* Local step of the direct (all-to-all + all-gather) gradient all-reduce of the ray-data-parallel global stage
 * (replaces DDP's bucketed all-reduce, gf_pipeline.py:136-138, which never sees the hash table): recv holds
 * n_shards copies [n_shards][shard_elems] of this rank's shard (f16 when is_f16, else f32), one per peer;
 * out[i] = (sum over copies, accumulated in f32 in rank order) * scale.  shard_elems % 8 == 0. */
int gfn_reduce_shards(const void* recv, int n_shards, int64_t shard_elems, int is_f16, float scale, void* out, void* stream);

/* ProcOctree on the DEVICE (csrc/octree_proc.hip): the same result as gfn_octree_proc_host, byte for byte, without the
 * blob ever leaving HBM.  nodes / stats / visit_cnt: device arrays of n_nodes entries (n_nodes_dev, may be NULL: a device
 * count <= n_nodes that overrides it); outputs: device arrays of out_cap entries (out_cap >= n_nodes, >= 9 * n_nodes when
 * subdivide); n_out_dev i64[1] receives the new node count, status_dev i32[1] 0 / 1 (blob is not a tree) / 2 (out_cap too
 * small); workspace: gfn_octree_proc_workspace(n_nodes, subdivide) bytes, 256-byte aligned.  One launch, stream ordered. */
int64_t gfn_octree_proc_workspace(int64_t n_nodes, int subdivide);
int gfn_octree_proc(const void* nodes, int64_t n_nodes, const int64_t* n_nodes_dev, const int64_t* weight_stats, const int64_t* alpha_stats,
                    const int64_t* visit_cnt, int subdivide, int brute_force, void* nodes_out, int64_t* weight_stats_out,
                    int64_t* alpha_stats_out, int64_t out_cap, int64_t* n_out_dev, int32_t* status_dev, void* workspace, void* stream);

/* ====================================================================== *
 * Host-side octree construction (replaces PersOctree::PersOctree,
 * PersSampler.cpp:92-152,516-831).  Pure host C++; returns malloc'd blobs the
 * caller frees with gfn_free_host.
 * ====================================================================== */
int gfn_octree_build_host(const float* c2w_host /*[n_cams,3,4]*/, const float* intri_host /*[n_cams,3,3]*/,
                          const float* bounds_host /*[n_cams,2]*/, int64_t n_cams, int max_depth,
                          float bbox_side_len, float split_dist_thres, uint64_t seed, int threads,
                          void** nodes_out_host, int64_t* n_nodes_out, void** trans_out_host,
                          int64_t* n_trans_out);
void gfn_free_host(void* p);

/* The pieces gfn_octree_build_host is made of, each with the reference function it replaces; tests hold them against
 * vectors captured from the reference's own PyTorch statement of the same functions (gfnerf/persoctree.py).  Host only.
 *   distance_summary  DistanceSummary (PersSampler.cpp:12-26; persoctree.py:234-246): exp of the mean log distance below
 *                     the 25 % quantile (linear interpolation), 1e8 for an empty set.
 *   pca12             PCA (PersSampler.cpp:593-611; persoctree.py:183-194) of vals [n_pts,12]: evals f32 [12] descending,
 *                     evecs f32 [12,12] row-major with column k the k-th eigenvector.
 *   visible_cams      GetVisiCams (PersSampler.cpp:45-88): visible_out[c] = 1 iff one of camera c's 128 x res_h pixel rays
 *                     crosses the box (center, side) inside the camera's [near, far] (persoctree.py:248-290 samples every
 *                     100th pixel instead, so only cameras away from the visibility border must agree).
 *   construct_trans   PersOctree::ConstructTrans (PersSampler.cpp:613-831) for the cameras c2w_visible [n,3,4] (intri f32
 *                     [9] of camera 0, :755): first_cam >= 0 injects the farthest-point selection's random first pick
 *                     (:641), pts [n_pts,3] the probe points (:540; NULL: 32^3 drawn from `seed`); trans_out receives one
 *                     gfn_trans_info, good_cams_out (may be NULL) the six selected camera indices.
 *   fit_weight        its third step alone (:770-815) for given frames w2xz f32 [12,2,4]: weight_out f32 [3,12]. */
float gfn_octree_distance_summary_host(const float* dis, int64_t n);
int gfn_octree_pca12_host(const float* vals, int64_t n_pts, float* evals_out, float* evecs_out);
int gfn_octree_visible_cams_host(const float* c2w, const float* intri, const float* bounds, int64_t n_cams, const float* center,
                                 float side, uint8_t* visible_out);
int gfn_octree_construct_trans_host(const float* c2w_visible, const float* intri, int64_t n_visible, const float* center, float side,
                                    int64_t first_cam, const float* pts, int64_t n_pts, uint64_t seed, void* trans_out,
                                    int64_t* good_cams_out);
int gfn_octree_fit_weight_host(const float* w2xz, const float* pts, int64_t n_pts, float* weight_out);

#ifdef __cplusplus
}
#endif
#endif /* GFNERF_AMD_H */
