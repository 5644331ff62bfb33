/*
 * gfn_oracle.h -- CPU restatement ("oracle") of the GF-NeRF per-ray hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product path (gf-nerf_amd/) never links or calls
 * this library and fails loudly when its HIP extension is missing.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - PINNED by vectors generated from the importable reference Python
 *     (tests/golden/make_golden.py): MLPNetwork fwd/bwd, get_weights_f2nerf +
 *     RGB/Accumulation/Depth renderers, trunc_exp, SH magnitudes (nerfstudio
 *     convention), Charbonnier.
 *   - PARITY UNPINNED: the native CUDA path (Hash3DAnchored K1/K2, PersSampler
 *     K3/K4/K5/K6) has no fixtures, tests or buildable sources in the reference
 *     (needs nvcc + un-vendored tiny-cuda-nn + Eigen); these functions are a
 *     line-by-line restatement of the cited .cu/.cpp files only.  tcnn's SH sign
 *     convention is likewise unpinned (third-party, absent).
 *
 * Floating point: built with -ffp-contract=off; every multi-term sum spells out
 * its association (Eigen 3.4 redux_novec_unroller halving tree for the fixed
 * size products the reference evaluates through Eigen).  One thing this cannot
 * settle: the reference's device code is built with nvcc's default -fmad=true
 * (gfnerf/bindings/setup.py:103-121 passes no -fmad), which lets the compiler
 * fuse a * b + c of the .cu files into FMAs -- which ones it fuses is a property
 * of the reference's binary that nothing here can observe.  The restatement (and
 * the HIP product, bit for bit with it) rounds every product.  For K1's 8-term
 * blend (Hash3DAnchored_cuda.cu:73-77) a fused chain differs only where the f32
 * sum lies within an ulp of a rounding boundary of the f16 the result is cast to;
 * K4's warp arithmetic is pinned by the reference's PyTorch statement
 * (gfnerf/persoctree.py), which fuses nothing.  Part of "parity unpinned" above.
 */
#ifndef GFN_ORACLE_H
#define GFN_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFN_N_LEVELS 16
#define GFN_N_CHANNELS 2
#define GFN_N_PROS 12
#define GFN_MAX_STACK 48
#define GFN_MAX_SAMPLE_PER_RAY 1024

/* PersSampler.h:39-48 (alignas(32)) -- byte layout is checkpoint ABI. */
typedef struct {
  float center[3];       /* @0   */
  float side_len;        /* @12  */
  int64_t parent;        /* @16  */
  int64_t childs[8];     /* @24  */
  uint8_t is_leaf_node;  /* @88  */
  uint8_t pad0[7];
  int64_t trans_idx;     /* @96  */
  int64_t block_idx;     /* @104 */
  uint8_t pad1[16];      /* -> 128 */
} gfn_o_tree_node;

/* PersSampler.h:31-37 (alignas(32)). */
typedef struct {
  float w2xz[GFN_N_PROS][2][4]; /* @0   row-major 2x4 each */
  float weight[3][GFN_N_PROS];  /* @384 row-major 3x12 */
  float center[3];              /* @528 */
  float side_len;               /* @540 */
  float dis_summary;            /* @544 */
  uint8_t pad[28];              /* -> 576 */
} gfn_o_trans_info;

/* ---- helpers ---------------------------------------------------------- */
uint16_t gfn_o_f32_to_f16(float f);
float gfn_o_f16_to_f32(uint16_t h);
/* per-level scale 2^(7*l/15+3), Hash3DAnchored_cuda.cu:28 */
void gfn_o_level_scales(float out[GFN_N_LEVELS]);
/* PersSampler.cpp:138-151: child visit order LUT, 8x8 */
void gfn_o_search_order(uint8_t out[64]);
/* PersSampler_cuda.cu:323: d / ||d|| */
void gfn_o_normalize_dirs(const float* d_in, float* d_out, int64_t n);

/* ---- a6: Hash3DAnchored forward (Hash3DAnchored_cuda.cu:11-79,160-196) -- */
/* feat_pool: f32 master [L*T*2]; prim_pool i32 [L,V,3]; bias_pool f32 [L*V,3];
 * pts f32 [N,3]; anchors i64 [N]; out f32 [N,32]. */
void gfn_o_hash_fwd(const float* feat_pool, const int32_t* prim_pool, const float* bias_pool,
                    const float* pts, const int64_t* anchors, int64_t n_pts, int64_t n_volumes,
                    int log2_T, float* out, int threads);
/* a7: backward (Hash3DAnchored_cuda.cu:81-155,198-239).  grad_pool f32 [L*T*2]
 * (zeroed here).  Contributions are quantised exactly like the reference
 * (g*128 -> f16, g16*w -> f16) but SUMMED in double, then /128. */
void gfn_o_hash_bwd(const int32_t* prim_pool, const float* bias_pool, const float* pts,
                    const int64_t* anchors, const float* grad_out, int64_t n_pts, int64_t n_volumes,
                    int log2_T, float* grad_pool, int threads);

/* ---- a2: K3 ray/octree traversal (PersSampler_cuda.cu:21-152) ---------- */
/* Two-call protocol like the reference's two passes: call with hit_idx==NULL to
 * get counts -> offsets[R+1] (exclusive scan in ray order), then again to fill. */
void gfn_o_trace(const float* rays_o, const float* rays_d, int64_t n_rays,
                 const gfn_o_tree_node* nodes, float near_bound, float far_bound,
                 int64_t max_hits, int64_t* offsets, int64_t* hit_idx, float* hit_near_far,
                 int threads);

/* ---- a3: K4 ray march (PersSampler_cuda.cu:155-318) -------------------- */
/* Same protocol: counts first (sample arrays NULL) -> pts_offsets[R+1]; then fill
 * PACKED arrays of size pts_offsets[R].  first_oct_dis f32 [R]. */
void gfn_o_march(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                 float sample_l, int scale_by_dis,
                 const int64_t* oct_offsets, const int64_t* hit_idx, const float* hit_near_far,
                 const gfn_o_tree_node* nodes, const gfn_o_trans_info* transes,
                 int64_t* pts_offsets,
                 float* world_pts, float* warped_pts, float* dirs, float* dists, float* ts,
                 int64_t* anchors /* [N,3] trans,oct,block */, float* first_oct_dis, int threads);

/* QueryFrameTransform / Jac exposed for unit tests (PersSampler_cuda.cu:155-188) */
void gfn_o_warp_point(const gfn_o_trans_info* tr, const float xyz[3], float out[3]);
void gfn_o_warp_jac(const gfn_o_trans_info* tr, const float xyz[3], float jac[9]);

/* ---- a9: SH degree 4 ---------------------------------------------------- */
/* convention 0: tiny-cuda-nn (input d01 in [0,1], mapped to 2*d01-1, odd-m terms
 * negated, optionally rounded to f16 like tcnn's default output precision);
 * convention 1: nerfstudio/utils/math.py:27-87 on raw directions. */
void gfn_o_sh4(const float* dirs, int64_t n, int convention, int round_f16, float* out);

/* ---- a8: MLPNetwork (gfnerf/mlp.py:35-57) ------------------------------- */
/* y[n,o] = act(b[o] + sum_k x[n,k] W[o,k]); act 0 none, 1 relu, 2 sigmoid */
void gfn_o_linear_fwd(const float* x, const float* W, const float* b, int64_t n, int in_dim,
                      int out_dim, int act, float* y, int threads);
/* given y (post-activation) and dy -> dx (may be NULL), dW, db (accumulated in double) */
void gfn_o_linear_bwd(const float* x, const float* W, const float* y, const float* dy, int64_t n,
                      int in_dim, int out_dim, int act, float* dx, float* dW, float* db, int threads);

/* ---- a11 + a5: density activation  sigma = trunc_exp(h0 + 1) ------------- */
void gfn_o_density_fwd(const float* h, int64_t n, int stride, float* sigma);
void gfn_o_density_bwd(const float* h, const float* dsigma, int64_t n, int stride, float* dh0);

/* ---- a12/a13: compositing over PACKED samples --------------------------- */
/* offsets i64 [R+1]; sigma, dt, t f32 [N]; rgb f32 [N,3].
 * early_T <= 0: reference semantics (all samples).  early_T > 0: stop once the
 * transmittance falls below early_T (the HIP path's early termination).
 * Outputs: weights/alphas/trans [N] (may be NULL), out_rgb [R,3], out_acc [R],
 * out_depth [R] (NOT yet clipped; see gfn_o_depth_finalize). */
void gfn_o_composite_fwd(const int64_t* offsets, int64_t n_rays, const float* sigma,
                         const float* dt, const float* t, const float* rgb, float early_T,
                         float* weights, float* alphas, float* trans, float* out_rgb,
                         float* out_acc, float* out_depth, int threads);
/* renderers.py:279-281: clip to batch-global [t.min(), t.max()] over the DENSE
 * [R,1024] tensor (zero padding counts whenever a ray has < 1024 samples), then
 * / scale_factor (nerfacto.py:546). */
void gfn_o_depth_finalize(float* depth, const int64_t* offsets, int64_t n_rays, const float* t,
                          float scale_factor);
/* backward of rgb/acc w.r.t. sigma and rgb. g_rgb [R,3], g_acc [R] (may be NULL) */
void gfn_o_composite_bwd(const int64_t* offsets, int64_t n_rays, const float* sigma,
                         const float* dt, const float* rgb, const float* g_rgb,
                         const float* g_acc, float* d_sigma, float* d_rgb, int threads);

/* ---- a15: UpdateOctNodes K5/K6 (PersSampler_cuda.cu:518-665) ------------ */
/* packed weights/alphas/oct_idx; stats i64 [n_nodes] updated in place; nodes'
 * trans_idx set to -1 where a stat < 0. visit_cnt i64 [n_nodes] atomicMax'd. */
void gfn_o_update_nodes(const int64_t* pts_offsets, int64_t n_rays, const int64_t* oct_idx,
                        const float* weights, const float* alphas, int64_t n_nodes,
                        int64_t* weight_stats, int64_t* alpha_stats, int64_t* visit_cnt,
                        gfn_o_tree_node* nodes);

/* ---- a15: ProcOctree (PersSampler.cpp:154-417) -------------------------- */
/* compact (prune invalid leaves to a fixpoint, compress single-child paths, renumber) and / or
 * subdivide (leaves with visit_cnt > 4, every leaf when brute_force).  Inputs are untouched; the new
 * tree is malloc'd: *nodes_out [*n_out], *weight_out / *alpha_out [*n_out] (free with gfn_o_free).
 * The reference zeroes visit_cnt afterwards (:415). */
void gfn_o_proc_octree(const gfn_o_tree_node* nodes, int64_t n_nodes, const int64_t* weight_stats,
                       const int64_t* alpha_stats, const int64_t* visit_cnt, int compact, int subdivide,
                       int brute_force, gfn_o_tree_node** nodes_out, int64_t* n_out, int64_t** weight_out,
                       int64_t** alpha_out);
void gfn_o_free(void* p);
/* K7 MarkInvisibleNodes (PersSampler_cuda.cu:680-742): trans_idx = -1 for nodes no camera sees.
 * intri f32 [n_cams,3,3], w2c f32 [n_cams,3,4], bounds f32 [n_cams,2], all row-major. */
void gfn_o_mark_invisible(gfn_o_tree_node* nodes, int64_t n_nodes, const float* intri, const float* w2c,
                          const float* bounds, int64_t n_cams);
/* ---- Cameras.generate_rays, perspective cameras without distortion
 * (nerfstudio/cameras/cameras.py:560-734; GF's lookat_directions :704).  c2w f32 [n_cams,3,4], intr f32
 * [n_cams,4] = (fx, fy, cx, cy), cam_idx i32 [n], coords f32 [n,2] = (y, x) incl. the 0.5 pixel offset.
 * Outputs [n,3] x3, [n] x2 (any may be NULL): origins, unit directions, c2w[:, :, 2], pixel area, |d|. */
void gfn_o_generate_rays(const float* c2w, const float* intr, const int32_t* cam_idx, const float* coords, int64_t n,
                         float* origins, float* directions, float* lookat, float* pixel_area, float* dir_norm);
/* ---- losses / optimiser (the rows SURVEY 8(f) marks next) ---------------- */
/* losses.py:73-84 CharbonnierLoss: mean(sqrt((p-g)^2 + eps^2)) */
double gfn_o_charbonnier(const float* pred, const float* gt, int64_t n, float eps, float* d_pred);
/* losses.py:713-794 S3IM: 1 - mean SSIM (4x4 gaussian window sigma 1.5, stride 4, zero padding 1, C1 = .01^2,
 * C2 = .03^2) of the virtual patches [3, patch_h, n_idx/patch_h] built from img[index[k]]; a, b f32 [R,3], index i32
 * [n_idx] (the concatenated arange + 9 randperms, :780-789).  d_b (may be NULL) = d loss / d b. */
double gfn_o_s3im(const float* a, const float* b, const int32_t* index, int64_t n_idx, int patch_h, float* d_b, int64_t n_rays);
/* torch.optim.Adam single step (no amsgrad, no weight decay) */
void gfn_o_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr,
                     float beta1, float beta2, float eps);

#ifdef __cplusplus
}
#endif
#endif
