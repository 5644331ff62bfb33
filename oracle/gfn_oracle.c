/*
 * gfn_oracle.c -- CPU restatement of the GF-NeRF per-ray hot path.
 * TEST INFRASTRUCTURE ONLY (see gfn_oracle.h).  Each function cites the
 * reference file:line it follows (paths relative to /root/reference).
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "gfn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

_Static_assert(sizeof(gfn_o_tree_node) == 128, "TreeNode ABI");
_Static_assert(sizeof(gfn_o_trans_info) == 576, "TransInfo ABI");
_Static_assert(__builtin_offsetof(gfn_o_tree_node, trans_idx) == 96, "TreeNode.trans_idx");
_Static_assert(__builtin_offsetof(gfn_o_tree_node, is_leaf_node) == 88, "TreeNode.is_leaf");
_Static_assert(__builtin_offsetof(gfn_o_trans_info, weight) == 384, "TransInfo.weight");
_Static_assert(__builtin_offsetof(gfn_o_trans_info, center) == 528, "TransInfo.center");
_Static_assert(__builtin_offsetof(gfn_o_trans_info, dis_summary) == 544, "TransInfo.dis_summary");

static int use_threads(int threads) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  return threads;
#else
  (void)threads;
  return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* fp16 <-> fp32, round-to-nearest-even (what __float2half / v_cvt_f16_f32 do) */
uint16_t gfn_o_f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x0200u : 0));
  }
  if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (absx < 0x38800000u) { /* subnormal half or zero: |f| < 2^-14 */
    if (absx < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
    int e = (int)(absx >> 23);                      /* biased exp */
    uint32_t m = (absx & 0x7fffffu) | 0x800000u;    /* 24-bit mantissa */
    int shift = 126 - e;                            /* 14..24: m * 2^(e-150) / 2^-24 = m >> shift */
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | q);
  }
  uint32_t e = (absx >> 23) - 112u; /* rebias 127 -> 15 */
  uint32_t m = absx & 0x7fffffu;
  uint32_t q = (e << 10) | (m >> 13);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
  return (uint16_t)(sign | q);
}

float gfn_o_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {
      int s = 0;
      while (!(m & 0x400u)) { m <<= 1; s++; }
      m &= 0x3ffu;
      x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

/* Hash3DAnchored_cuda.cu:28  mul = exp2f((10-3)*l/15 + 3).  CUDA's exp2f is a
 * <=2ulp approximation we cannot reproduce; both this oracle and the HIP path use
 * this host-computed table (exact exp2 in double of the f32 argument, rounded). */
void gfn_o_level_scales(float out[GFN_N_LEVELS]) {
  for (int l = 0; l < GFN_N_LEVELS; l++) {
    float arg = (10.f - 3.f) * (float)l / (float)(GFN_N_LEVELS - 1) + 3.f;
    out[l] = (float)exp2((double)arg);
  }
}

/* PersSampler.cpp:138-151.  std::sort with cmp(a,b) = (a & bt) ^ (st & bt),
 * bt = lowest set bit of a^b.  cmp is a strict total order on 0..7, so any
 * correct sort yields the same permutation; insertion sort here. */
void gfn_o_search_order(uint8_t out[64]) {
  for (int st = 0; st < 8; st++) {
    int v[8];
    for (int i = 0; i < 8; i++) v[i] = i;
    for (int i = 1; i < 8; i++) {
      int x = v[i], j = i - 1;
      while (j >= 0) {
        int a = x, b = v[j];
        int bt = (a ^ b) & -(a ^ b);
        int less = ((a & bt) ^ (st & bt)) != 0; /* x < v[j] ? */
        if (!less) break;
        v[j + 1] = v[j];
        j--;
      }
      v[j + 1] = x;
    }
    for (int i = 0; i < 8; i++) out[st * 8 + i] = (uint8_t)v[i];
  }
}

/* PersSampler_cuda.cu:323  rays_d / linalg_norm(rays_d, 2, -1, keepdim) */
void gfn_o_normalize_dirs(const float* d_in, float* d_out, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    float x = d_in[3 * i], y = d_in[3 * i + 1], z = d_in[3 * i + 2];
    float nrm = sqrtf((x * x + y * y) + z * z);
    d_out[3 * i] = x / nrm;
    d_out[3 * i + 1] = y / nrm;
    d_out[3 * i + 2] = z / nrm;
  }
}

/* ------------------------------------------------------------------------ */
/* Hash addressing shared by fwd/bwd: Hash3DAnchored_cuda.cu:27-69 / :98-139 */
typedef struct {
  uint32_t pos[8]; /* order 000,001,010,011,100,101,110,111 (x,y,z bits) */
  float w[8];
} hash_corner_t;

static void hash_corners(const float* pt3, float scale, const int32_t* prim, const float* bias,
                         uint32_t local_size, hash_corner_t* c) {
  float px = pt3[0] * scale, py = pt3[1] * scale, pz = pt3[2] * scale; /* pt *= mul  :29 */
  px = px + bias[0]; py = py + bias[1]; pz = pz + bias[2];             /* :41 */
  float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  /* static_cast<unsigned>(floorf(.)) :44-46.  Negative / huge inputs are UB in
   * C; CUDA saturates.  The hot path only produces p >= 0 (x in [0,1]); the
   * oracle and the HIP kernel both define the conversion as saturating. */
  uint32_t ux = fx <= 0.f ? 0u : (fx >= 4294967040.f ? 0xffffffffu : (uint32_t)fx);
  uint32_t uy = fy <= 0.f ? 0u : (fy >= 4294967040.f ? 0xffffffffu : (uint32_t)fy);
  uint32_t uz = fz <= 0.f ? 0u : (fz >= 4294967040.f ? 0xffffffffu : (uint32_t)fz);
  uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  uint32_t x0 = ux * pa, x1 = (ux + 1u) * pa;
  uint32_t y0 = uy * pb, y1 = (uy + 1u) * pb;
  uint32_t z0 = uz * pc, z1 = (uz + 1u) * pc;
  c->pos[0] = (x0 ^ y0 ^ z0) % local_size; /* :48-55 */
  c->pos[1] = (x0 ^ y0 ^ z1) % local_size;
  c->pos[2] = (x0 ^ y1 ^ z0) % local_size;
  c->pos[3] = (x0 ^ y1 ^ z1) % local_size;
  c->pos[4] = (x1 ^ y0 ^ z0) % local_size;
  c->pos[5] = (x1 ^ y0 ^ z1) % local_size;
  c->pos[6] = (x1 ^ y1 ^ z0) % local_size;
  c->pos[7] = (x1 ^ y1 ^ z1) % local_size;
  float a = px - fx, b = py - fy, cc = pz - fz; /* :58-60 */
  float na = 1.f - a, nb = 1.f - b, nc = 1.f - cc;
  c->w[0] = (na * nb) * nc; /* :62-69, left-to-right products */
  c->w[1] = (na * nb) * cc;
  c->w[2] = (na * b) * nc;
  c->w[3] = (na * b) * cc;
  c->w[4] = (a * nb) * nc;
  c->w[5] = (a * nb) * cc;
  c->w[6] = (a * b) * nc;
  c->w[7] = (a * b) * cc;
}

void gfn_o_hash_fwd(const float* feat_pool, const int32_t* prim_pool, const float* bias_pool,
                    const float* pts, const int64_t* anchors, int64_t n_pts, int64_t n_volumes,
                    int log2_T, float* out, int threads) {
  float scales[GFN_N_LEVELS];
  gfn_o_level_scales(scales);
  const uint32_t T = 1u << log2_T;
  /* Hash3DAnchored.cpp:66-67: local_size = (pool/L >> 4) << 4 == T for log2_T>=4 */
  const uint32_t local_size = (T >> 4) << 4;
  threads = use_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t i = 0; i < n_pts; i++) {
    int64_t vol = anchors[i];
    for (int l = 0; l < GFN_N_LEVELS; l++) {
      hash_corner_t c;
      int64_t tr = (int64_t)l * n_volumes + vol; /* :32,:40 */
      hash_corners(pts + 3 * i, scales[l], prim_pool + 3 * tr, bias_pool + 3 * tr, local_size, &c);
      /* QUIRK kept for checkpoint parity (:37 + :74): level base is l*T HALVES,
       * slot address pos*2+k, so flat element index = l*T + pos*2 + k. */
      const float* base = feat_pool + (int64_t)l * T;
      for (int k = 0; k < GFN_N_CHANNELS; k++) {
        float acc = 0.f;
        for (int d = 0; d < 8; d++) {
          /* :185 pool.to(fp16) then float(half) */
          float f = gfn_o_f16_to_f32(gfn_o_f32_to_f16(base[(int64_t)c.pos[d] * 2 + k]));
          float term = c.w[d] * f;
          acc = (d == 0) ? term : acc + term; /* :73-77 left-to-right sum */
        }
        /* (T)(...) -> half, then out.to(fp32) :73,:195 */
        out[i * 32 + l * 2 + k] = gfn_o_f16_to_f32(gfn_o_f32_to_f16(acc));
      }
    }
  }
}

void gfn_o_hash_bwd(const int32_t* prim_pool, const float* bias_pool, const float* pts,
                    const int64_t* anchors, const float* grad_out, int64_t n_pts, int64_t n_volumes,
                    int log2_T, float* grad_pool, int threads) {
  float scales[GFN_N_LEVELS];
  gfn_o_level_scales(scales);
  const uint32_t T = 1u << log2_T;
  const uint32_t local_size = (T >> 4) << 4;
  const int64_t pool_elems = (int64_t)GFN_N_LEVELS * T * 2;
  double* acc = (double*)calloc((size_t)pool_elems, sizeof(double));
  threads = use_threads(threads);
  /* Every addend is an f16 value (a multiple of 2^-24 below 2^16): their sum in double is EXACT whatever the order, so the
   * points may be scattered by any number of threads (atomic adds) and the pool is the same bits as the serial loop's. */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t i = 0; i < n_pts; i++) {
    int64_t vol = anchors[i];
    for (int l = 0; l < GFN_N_LEVELS; l++) {
      hash_corner_t c;
      int64_t tr = (int64_t)l * n_volumes + vol;
      hash_corners(pts + 3 * i, scales[l], prim_pool + 3 * tr, bias_pool + 3 * tr, local_size, &c);
      /* :219 grad_in = (grad_out * 128).half() */
      float g0 = gfn_o_f16_to_f32(gfn_o_f32_to_f16(grad_out[i * 32 + l * 2 + 0] * 128.f));
      float g1 = gfn_o_f16_to_f32(gfn_o_f32_to_f16(grad_out[i * 32 + l * 2 + 1] * 128.f));
      if (g0 != 0.f || g1 != 0.f) { /* :149 */
        for (int d = 0; d < 8; d++) {
          /* :150 {(half)(w0*ws[d]), (half)(w1*ws[d])} */
          float c0 = gfn_o_f16_to_f32(gfn_o_f32_to_f16(g0 * c.w[d]));
          float c1 = gfn_o_f16_to_f32(gfn_o_f32_to_f16(g1 * c.w[d]));
          int64_t e = (int64_t)l * T + (int64_t)c.pos[d] * 2; /* :109 + :151 */
#pragma omp atomic
          acc[e] += (double)c0;
#pragma omp atomic
          acc[e + 1] += (double)c1;
        }
      }
    }
  }
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t e = 0; e < pool_elems; e++) grad_pool[e] = (float)(acc[e] / 128.0); /* :238 */
  free(acc);
}

/* ------------------------------------------------------------------------ */
/* PersSampler_cuda.cu:21-51 GetIntersection */
static void get_intersection(const float o[3], const float d[3], const float c[3], float side_len,
                             float* near, float* far) {
  float tmp[3][2];
  float hf = side_len * .5f;
  for (int i = 0; i < 3; i++) {
    if (d[i] < 1e-6f && d[i] > -1e-6f) {
      if (o[i] > c[i] - hf && o[i] < c[i] + hf) {
        tmp[i][0] = -1e6f; tmp[i][1] = 1e6f;
      } else {
        tmp[i][0] = 1e6f; tmp[i][1] = -1e6f;
      }
    } else if (d[i] > 0) {
      tmp[i][0] = ((c[i] - hf) - o[i]) / d[i];
      tmp[i][1] = ((c[i] + hf) - o[i]) / d[i];
    } else {
      tmp[i][0] = ((c[i] + hf) - o[i]) / d[i];
      tmp[i][1] = ((c[i] - hf) - o[i]) / d[i];
    }
  }
  *near = fmaxf(*near, fmaxf(tmp[0][0], fmaxf(tmp[1][0], tmp[2][0])));
  *far = fminf(*far, fminf(tmp[0][1], fminf(tmp[1][1], tmp[2][1])));
}

/* PersSampler_cuda.cu:53-152, one ray.  Returns the hit count; fills when
 * out_idx != NULL. */
static int64_t trace_one(const float o[3], const float d[3], const gfn_o_tree_node* nodes,
                         const uint8_t* search_order_all, float overall_near, float overall_far,
                         int64_t max_cnt, int64_t* out_idx, float* out_nf) {
  int64_t stack_info[GFN_MAX_STACK];
  int64_t stack_ptr = 0, cnt = 0;
  stack_info[0] = 0;
  stack_info[1] = -1;
  int64_t ray_st = ((int64_t)(d[0] > 0.f) << 2) | ((int64_t)(d[1] > 0.f) << 1) | ((int64_t)(d[2] > 0.f));
  const uint8_t* so = search_order_all + ray_st * 8;
  while (stack_ptr >= 0 && cnt < max_cnt) {
    int64_t u = stack_info[stack_ptr * 2];
    const gfn_o_tree_node* node = nodes + u;
    if (stack_info[stack_ptr * 2 + 1] == -1) {
      float cur_near = overall_near, cur_far = overall_far;
      get_intersection(o, d, node->center, node->side_len, &cur_near, &cur_far);
      if (cur_near < cur_far) {
        int64_t child_ptr = 0;
        while (child_ptr < 8 && node->childs[so[child_ptr]] < 0) child_ptr++;
        if (child_ptr < 8) {
          stack_info[stack_ptr * 2 + 1] = child_ptr;
          stack_ptr++;
          stack_info[stack_ptr * 2] = node->childs[so[child_ptr]];
          stack_info[stack_ptr * 2 + 1] = -1;
        } else {
          if (node->trans_idx >= 0) {
            if (out_idx) {
              out_idx[cnt] = u;
              out_nf[cnt * 2] = cur_near;
              out_nf[cnt * 2 + 1] = cur_far;
            }
            cnt++;
          }
          stack_ptr--;
        }
      } else {
        stack_ptr--;
      }
    } else {
      int64_t child_ptr = stack_info[stack_ptr * 2 + 1] + 1;
      while (child_ptr < 8 && node->childs[so[child_ptr]] < 0) child_ptr++;
      if (child_ptr < 8) {
        stack_info[stack_ptr * 2 + 1] = child_ptr;
        stack_ptr++;
        stack_info[stack_ptr * 2] = node->childs[so[child_ptr]];
        stack_info[stack_ptr * 2 + 1] = -1;
      } else {
        stack_ptr--;
      }
    }
  }
  return cnt;
}

void gfn_o_trace(const float* rays_o, const float* rays_d, int64_t n_rays,
                 const gfn_o_tree_node* nodes, float near_bound, float far_bound,
                 int64_t max_hits, int64_t* offsets, int64_t* hit_idx, float* hit_near_far,
                 int threads) {
  uint8_t so[64];
  gfn_o_search_order(so);
  threads = use_threads(threads);
  if (!hit_idx) {
    int64_t* cnt = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_rays > 0 ? n_rays : 1));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rays; r++)
      cnt[r] = trace_one(rays_o + 3 * r, rays_d + 3 * r, nodes, so, near_bound, far_bound, max_hits,
                         NULL, NULL);
    offsets[0] = 0;
    for (int64_t r = 0; r < n_rays; r++) offsets[r + 1] = offsets[r] + cnt[r];
    free(cnt);
  } else {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rays; r++) {
      /* FILL pass: max_intersect_cnt = end - start  (:76) */
      trace_one(rays_o + 3 * r, rays_d + 3 * r, nodes, so, near_bound, far_bound,
                offsets[r + 1] - offsets[r], hit_idx + offsets[r], hit_near_far + 2 * offsets[r]);
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Eigen 3.4 fixed-size reductions (redux_novec_unroller halves the range). */
static inline float sum3(float a, float b, float c) { return a + (b + c); }
static inline float sum4(float a, float b, float c, float d) { return (a + b) + (c + d); }
static inline float sum12(const float* p) {
  return (sum3(p[0], p[1], p[2]) + sum3(p[3], p[4], p[5])) +
         (sum3(p[6], p[7], p[8]) + sum3(p[9], p[10], p[11]));
}

/* PersSampler_cuda.cu:155-170 QueryFrameTransform */
void gfn_o_warp_point(const gfn_o_trans_info* tr, const float xyz[3], float out[3]) {
  float v[GFN_N_PROS];
  for (int i = 0; i < GFN_N_PROS; i++) {
    const float(*m)[4] = tr->w2xz[i];
    float x0 = sum4(m[0][0] * xyz[0], m[0][1] * xyz[1], m[0][2] * xyz[2], m[0][3] * 1.f);
    float x1 = sum4(m[1][0] * xyz[0], m[1][1] * xyz[1], m[1][2] * xyz[2], m[1][3] * 1.f);
    v[i] = x0 / x1; /* :164 */
  }
  for (int r = 0; r < 3; r++) {
    float p[GFN_N_PROS];
    for (int i = 0; i < GFN_N_PROS; i++) p[i] = tr->weight[r][i] * v[i];
    out[r] = sum12(p); /* :167 */
  }
}

/* PersSampler_cuda.cu:172-188 QueryFrameTransformJac (row-major 3x3) */
void gfn_o_warp_jac(const gfn_o_trans_info* tr, const float xyz[3], float jac[9]) {
  float tj[GFN_N_PROS][3];
  for (int i = 0; i < GFN_N_PROS; i++) {
    const float(*m)[4] = tr->w2xz[i];
    float x0 = sum4(m[0][0] * xyz[0], m[0][1] * xyz[1], m[0][2] * xyz[2], m[0][3] * 1.f);
    float x1 = sum4(m[1][0] * xyz[0], m[1][1] * xyz[1], m[1][2] * xyz[2], m[1][3] * 1.f);
    float dv0 = 1.f / x1;          /* :182 */
    float dv1 = -x0 / (x1 * x1);
    for (int c = 0; c < 3; c++) tj[i][c] = dv0 * m[0][c] + dv1 * m[1][c]; /* :183 1x2 . 2x3 */
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      float p[GFN_N_PROS];
      for (int i = 0; i < GFN_N_PROS; i++) p[i] = tr->weight[r][i] * tj[i][c];
      jac[r * 3 + c] = sum12(p); /* :186 */
    }
}

/* PersSampler_cuda.cu:190-318, one ray.  fill != 0 writes samples. */
static int64_t march_one(const float o[3], const float d[3], const float* noise /* +ray_idx */,
                         float sample_l, int scale_by_dis, int64_t n_oct, const int64_t* oct_idx,
                         const float* oct_nf, const gfn_o_tree_node* nodes,
                         const gfn_o_trans_info* transes, int64_t max_n_samples, int fill,
                         float* world_pts, float* warped_pts, float* dirs, float* dists, float* ts,
                         int64_t* anchors) {
  if (max_n_samples <= 0) return 0; /* :238 */
  int64_t oct_ptr = 0, pts_ptr = 0;
  /* the reference reads oct_intersect_idx[0] even when n_oct == 0 (:244); that
   * value is never used because the loop below does not execute. */
  if (n_oct <= 0) return 0;
  int64_t cur_oct_idx = oct_idx[0];
  float cur_march_step = 0.f, exp_march_step = 0.f;
  float cur_t = oct_nf[0];
  float cur_far = oct_nf[1];
  float cur_near = oct_nf[0];
  float cur_xyz[3] = {o[0] + d[0] * cur_t, o[1] + d[1] * cur_t, o[2] + d[2] * cur_t};
  int the_first_pts = 1;
  while (pts_ptr < max_n_samples && oct_ptr < n_oct) {
    const gfn_o_tree_node* cur_node = nodes + cur_oct_idx;
    const gfn_o_trans_info* tr = transes + cur_node->trans_idx;
    float jac[9];
    float dx = o[0] - tr->center[0], dy = o[1] - tr->center[1], dz = o[2] - tr->center[2];
    float cur_radius = sqrtf(sum3(dx * dx, dy * dy, dz * dz)) / tr->dis_summary; /* :266 */
    float cur_radius_clip = fmaxf(cur_radius, 1.f);
    gfn_o_warp_jac(tr, cur_xyz, jac);
    float pj[3];
    for (int r = 0; r < 3; r++) pj[r] = sum3(jac[r * 3] * d[0], jac[r * 3 + 1] * d[1], jac[r * 3 + 2] * d[2]);
    float pnorm = sqrtf(sum3(pj[0] * pj[0], pj[1] * pj[1], pj[2] * pj[2]));
    float exp_march_step_warp = sample_l * noise[pts_ptr]; /* :270 */
    exp_march_step = exp_march_step_warp / (pnorm + 1e-6f);
    if (scale_by_dis) exp_march_step *= cur_radius_clip;
    cur_march_step = exp_march_step;
    if (fill && !the_first_pts) { /* :278-290 */
      world_pts[pts_ptr * 3] = cur_xyz[0]; world_pts[pts_ptr * 3 + 1] = cur_xyz[1]; world_pts[pts_ptr * 3 + 2] = cur_xyz[2];
      ts[pts_ptr] = cur_t;
      dirs[pts_ptr * 3] = d[0]; dirs[pts_ptr * 3 + 1] = d[1]; dirs[pts_ptr * 3 + 2] = d[2];
      gfn_o_warp_point(tr, cur_xyz, warped_pts + pts_ptr * 3);
      dists[pts_ptr] = exp_march_step * (pnorm + 1e-6f);
      anchors[pts_ptr * 3] = cur_node->trans_idx;
      anchors[pts_ptr * 3 + 1] = cur_oct_idx;
      anchors[pts_ptr * 3 + 2] = cur_node->block_idx;
    }
    if (!the_first_pts) pts_ptr += 1;
    while (cur_t + cur_march_step > cur_far) { /* :295-305 */
      oct_ptr++;
      if (oct_ptr >= n_oct) break;
      cur_oct_idx = oct_idx[oct_ptr];
      cur_near = oct_nf[oct_ptr * 2];
      cur_far = oct_nf[oct_ptr * 2 + 1];
      int64_t ex_march_steps = (int64_t)ceilf(fmaxf((cur_near - cur_t) / exp_march_step, 1.f));
      cur_march_step = exp_march_step * (float)ex_march_steps;
    }
    cur_t += cur_march_step;
    cur_xyz[0] = o[0] + d[0] * cur_t; cur_xyz[1] = o[1] + d[1] * cur_t; cur_xyz[2] = o[2] + d[2] * cur_t;
    the_first_pts = 0;
  }
  return pts_ptr;
}

void gfn_o_march(const float* rays_o, const float* rays_d, const float* noise, int64_t n_rays,
                 float sample_l, int scale_by_dis, const int64_t* oct_offsets,
                 const int64_t* hit_idx, const float* hit_near_far, const gfn_o_tree_node* nodes,
                 const gfn_o_trans_info* transes, int64_t* pts_offsets, float* world_pts,
                 float* warped_pts, float* dirs, float* dists, float* ts, int64_t* anchors,
                 float* first_oct_dis, int threads) {
  threads = use_threads(threads);
  if (!world_pts) {
    int64_t* cnt = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_rays > 0 ? n_rays : 1));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (int64_t r = 0; r < n_rays; r++) {
      int64_t s = oct_offsets[r], e = oct_offsets[r + 1];
      cnt[r] = march_one(rays_o + 3 * r, rays_d + 3 * r, noise + r, sample_l, scale_by_dis, e - s,
                         hit_idx + s, hit_near_far + 2 * s, nodes, transes, GFN_MAX_SAMPLE_PER_RAY, 0,
                         NULL, NULL, NULL, NULL, NULL, NULL);
    }
    pts_offsets[0] = 0; /* :407 cumsum */
    for (int64_t r = 0; r < n_rays; r++) pts_offsets[r + 1] = pts_offsets[r] + cnt[r];
    free(cnt);
  } else {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (int64_t r = 0; r < n_rays; r++) {
      int64_t s = oct_offsets[r], e = oct_offsets[r + 1];
      int64_t p = pts_offsets[r];
      int64_t cap = pts_offsets[r + 1] - p;
      if (cap > GFN_MAX_SAMPLE_PER_RAY) cap = GFN_MAX_SAMPLE_PER_RAY; /* :237 */
      if (first_oct_dis) first_oct_dis[r] = (s < e) ? hit_near_far[2 * s] : 1e9f; /* :229-234 */
      march_one(rays_o + 3 * r, rays_d + 3 * r, noise + r, sample_l, scale_by_dis, e - s, hit_idx + s,
                hit_near_far + 2 * s, nodes, transes, cap, 1, world_pts + 3 * p, warped_pts + 3 * p,
                dirs + 3 * p, dists + p, ts + p, anchors + 3 * p);
    }
  }
}

/* ------------------------------------------------------------------------ */
/* a9: SH degree 4.  convention 0 = tiny-cuda-nn (unpinned, third-party absent):
 * input (d+1)/2 from nerfacto_field.py:64-70,519-521 is mapped back to 2x-1.
 * convention 1 = nerfstudio/utils/math.py:27-87 (pinned by golden vector). */
void gfn_o_sh4(const float* dirs, int64_t n, int convention, int round_f16, float* out) {
  for (int64_t i = 0; i < n; i++) {
    float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
    float sgn = 1.f;
    if (convention == 0) {
      x = x * 2.f - 1.f; y = y * 2.f - 1.f; z = z * 2.f - 1.f;
      sgn = -1.f;
    }
    float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float* o = out + 16 * i;
    o[0] = 0.28209479177387814f;
    o[1] = sgn * 0.4886025119029199f * y;
    o[2] = 0.4886025119029199f * z;
    o[3] = sgn * 0.4886025119029199f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = sgn * 1.0925484305920792f * yz;
    o[6] = 0.9461746957575601f * z2 - 0.31539156525251999f;
    o[7] = sgn * 1.0925484305920792f * xz;
    o[8] = 0.5462742152960396f * x2 - 0.5462742152960396f * y2;
    o[9] = sgn * 0.5900435899266435f * y * (3.f * x2 - y2);
    o[10] = 2.890611442640554f * xy * z;
    o[11] = sgn * 0.4570457994644658f * y * (5.f * z2 - 1.f);
    o[12] = 0.3731763325901154f * z * (5.f * z2 - 3.f);
    o[13] = sgn * 0.4570457994644658f * x * (5.f * z2 - 1.f);
    o[14] = 1.445305721320277f * z * (x2 - y2);
    o[15] = sgn * 0.5900435899266435f * x * (x2 - 3.f * y2);
    if (round_f16)
      for (int k = 0; k < 16; k++) o[k] = gfn_o_f16_to_f32(gfn_o_f32_to_f16(o[k]));
  }
}

/* ------------------------------------------------------------------------ */
/* a8: gfnerf/mlp.py:35-57 -- nn.Linear (+ReLU / Sigmoid) */
void gfn_o_linear_fwd(const float* x, const float* W, const float* b, int64_t n, int in_dim,
                      int out_dim, int act, float* y, int threads) {
  threads = use_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t i = 0; i < n; i++) {
    const float* xi = x + i * in_dim;
    for (int o = 0; o < out_dim; o++) {
      const float* w = W + (int64_t)o * in_dim;
      float acc = b ? b[o] : 0.f;
      for (int k = 0; k < in_dim; k++) acc += xi[k] * w[k];
      if (act == 1) acc = acc > 0.f ? acc : 0.f;
      else if (act == 2) acc = 1.f / (1.f + expf(-acc));
      y[i * out_dim + o] = acc;
    }
  }
}

void gfn_o_linear_bwd(const float* x, const float* W, const float* y, const float* dy, int64_t n,
                      int in_dim, int out_dim, int act, float* dx, float* dW, float* db, int threads) {
  /* threads == 1: one pass in sample order (the form the parity tests were pinned with).  More: every thread owns a contiguous
   * run of samples and its own double accumulators, added up in thread order at the end (deterministic for a thread count). */
  threads = use_threads(threads);
  if (threads > n) threads = n > 0 ? (int)n : 1;
  const size_t nW = (size_t)in_dim * out_dim;
  double* aW = (double*)calloc(nW * (size_t)threads, sizeof(double));
  double* ab = (double*)calloc((size_t)out_dim * (size_t)threads, sizeof(double));
#pragma omp parallel num_threads(threads)
  {
    const int tid = omp_get_thread_num();
    const int64_t lo = n * tid / threads, hi = n * (tid + 1) / threads;
    double* tW = aW + nW * (size_t)tid;
    double* tb = ab + (size_t)out_dim * (size_t)tid;
    float* dz = (float*)malloc(sizeof(float) * (size_t)out_dim);
    for (int64_t i = lo; i < hi; i++) {
      for (int o = 0; o < out_dim; o++) {
        float g = dy[i * out_dim + o];
        float yo = y[i * out_dim + o];
        if (act == 1) g = yo > 0.f ? g : 0.f;
        else if (act == 2) g = g * yo * (1.f - yo);
        dz[o] = g;
        tb[o] += g;
        for (int k = 0; k < in_dim; k++) tW[(int64_t)o * in_dim + k] += (double)g * x[i * in_dim + k];
      }
      if (dx) {
        for (int k = 0; k < in_dim; k++) {
          float acc = 0.f;
          for (int o = 0; o < out_dim; o++) acc += dz[o] * W[(int64_t)o * in_dim + k];
          dx[i * in_dim + k] = acc;
        }
      }
    }
    free(dz);
  }
  for (int t = 1; t < threads; t++) {
    for (size_t e = 0; e < nW; e++) aW[e] += aW[nW * (size_t)t + e];
    for (int o = 0; o < out_dim; o++) ab[o] += ab[(size_t)out_dim * (size_t)t + o];
  }
  if (dW) for (int64_t e = 0; e < (int64_t)nW; e++) dW[e] = (float)aW[e];
  if (db) for (int o = 0; o < out_dim; o++) db[o] = (float)ab[o];
  free(aW); free(ab);
}

/* a11 activations.py:24-42 + nerfacto_field.py:499  sigma = trunc_exp(h0 + 1) */
void gfn_o_density_fwd(const float* h, int64_t n, int stride, float* sigma) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) sigma[i] = expf(h[i * stride] + 1.0f);
}
void gfn_o_density_bwd(const float* h, const float* dsigma, int64_t n, int stride, float* dh0) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    float x = h[i * stride] + 1.0f;
    x = x < -15.f ? -15.f : (x > 15.f ? 15.f : x);
    dh0[i] = dsigma[i] * expf(x);
  }
}

/* ------------------------------------------------------------------------ */
/* torch.nan_to_num defaults: nan->0, +inf->FLT_MAX, -inf->-FLT_MAX */
static inline float nan_to_num(float v) {
  if (isnan(v)) return 0.f;
  if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

/* a12 rays.py:178-200 + a13 renderers.py:97-98,216,279 over packed samples */
void gfn_o_composite_fwd(const int64_t* offsets, int64_t n_rays, const float* sigma,
                         const float* dt, const float* t, const float* rgb, float early_T,
                         float* weights, float* alphas, float* trans, float* out_rgb,
                         float* out_acc, float* out_depth, int threads) {
  threads = use_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
  for (int64_t r = 0; r < n_rays; r++) {
    float csum = 0.f; /* exclusive cumsum of delta*sigma  (:191-195) */
    float cr = 0.f, cg = 0.f, cb = 0.f, acc = 0.f, dep = 0.f;
    int stopped = 0;
    for (int64_t i = offsets[r]; i < offsets[r + 1]; i++) {
      float ds = dt[i] * sigma[i];
      float alpha = 1.f - expf(-ds);
      float T = expf(-csum);
      float w = nan_to_num(alpha * T);
      if (early_T > 0.f && T < early_T) stopped = 1;
      if (stopped) { w = 0.f; }
      if (weights) weights[i] = w;
      if (alphas) alphas[i] = alpha;
      if (trans) trans[i] = T;
      if (!stopped) {
        cr += w * rgb[3 * i]; cg += w * rgb[3 * i + 1]; cb += w * rgb[3 * i + 2];
        acc += w;
        dep += w * t[i];
      }
      csum += ds;
    }
    out_rgb[3 * r] = cr; out_rgb[3 * r + 1] = cg; out_rgb[3 * r + 2] = cb;
    if (out_acc) out_acc[r] = acc;
    if (out_depth) out_depth[r] = dep / (acc + 1e-10f); /* renderers.py:279 */
  }
}

void gfn_o_depth_finalize(float* depth, const int64_t* offsets, int64_t n_rays, const float* t,
                          float scale_factor) {
  float tmin = INFINITY, tmax = -INFINITY;
  int any_pad = 0;
  for (int64_t r = 0; r < n_rays; r++) {
    if (offsets[r + 1] - offsets[r] < GFN_MAX_SAMPLE_PER_RAY) any_pad = 1;
    for (int64_t i = offsets[r]; i < offsets[r + 1]; i++) {
      tmin = fminf(tmin, t[i]);
      tmax = fmaxf(tmax, t[i]);
    }
  }
  if (any_pad) { tmin = fminf(tmin, 0.f); tmax = fmaxf(tmax, 0.f); } /* dense zero padding */
  for (int64_t r = 0; r < n_rays; r++) {
    float d = depth[r];
    d = d < tmin ? tmin : (d > tmax ? tmax : d); /* renderers.py:281 */
    /* nerfacto.py:546 `depth / self.scale_factor`: on the device the reference runs on torch
     * evaluates tensor / python-scalar as a * (1/b) in f32 (ATen BinaryDivTrueKernel.cu) */
    depth[r] = d * (1.0f / scale_factor);
  }
}

/* backward of (rgb, acc) wrt sigma, rgb through w_i = alpha_i T_i:
 *   dw_i/dsigma_i = dt_i T_{i+1};  dw_k/dsigma_i = -dt_i w_k (k > i). */
void gfn_o_composite_bwd(const int64_t* offsets, int64_t n_rays, const float* sigma,
                         const float* dt, const float* rgb, const float* g_rgb,
                         const float* g_acc, float* d_sigma, float* d_rgb, int threads) {
  threads = use_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16) /* rays are independent: the same bits for any thread count */
  for (int64_t r = 0; r < n_rays; r++) {
    int64_t s = offsets[r], e = offsets[r + 1];
    int64_t n = e - s;
    if (n <= 0) continue;
    double* w = (double*)malloc(sizeof(double) * (size_t)n);
    double* Tn = (double*)malloc(sizeof(double) * (size_t)n); /* T_{i+1} */
    double csum = 0.0;
    for (int64_t i = 0; i < n; i++) {
      double ds = (double)dt[s + i] * sigma[s + i];
      double T = exp(-csum);
      w[i] = (1.0 - exp(-ds)) * T;
      csum += ds;
      Tn[i] = exp(-csum);
    }
    double suffix = 0.0; /* sum_{k>i} G_k w_k */
    for (int64_t i = n - 1; i >= 0; i--) {
      double G = (double)g_rgb[3 * r] * rgb[3 * (s + i)] + (double)g_rgb[3 * r + 1] * rgb[3 * (s + i) + 1] +
                 (double)g_rgb[3 * r + 2] * rgb[3 * (s + i) + 2] + (g_acc ? (double)g_acc[r] : 0.0);
      d_sigma[s + i] = (float)((double)dt[s + i] * (G * Tn[i] - suffix));
      suffix += G * w[i];
      if (d_rgb) {
        d_rgb[3 * (s + i)] = (float)(w[i] * g_rgb[3 * r]);
        d_rgb[3 * (s + i) + 1] = (float)(w[i] * g_rgb[3 * r + 1]);
        d_rgb[3 * (s + i) + 2] = (float)(w[i] * g_rgb[3 * r + 2]);
      }
    }
    free(w); free(Tn);
  }
}

/* ------------------------------------------------------------------------ */
/* a15: PersSampler_cuda.cu:518-665 */
#define OCC_WEIGHT_BASE 512
#define ABS_WEIGHT_THRES 0.01
#define REL_WEIGHT_THRES 0.1
#define OCC_ALPHA_BASE 32
#define ABS_ALPHA_THRES 0.02
#define REL_ALPHA_THRES 0.1

static inline void amax(int64_t* p, int64_t v) { if (v > *p) *p = v; }

void gfn_o_update_nodes(const int64_t* pts_offsets, int64_t n_rays, const int64_t* oct_idx,
                        const float* weights, const float* alphas, int64_t n_nodes,
                        int64_t* weight_stats, int64_t* alpha_stats, int64_t* visit_cnt,
                        gfn_o_tree_node* nodes) {
  int64_t* w_adder = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_nodes);
  int64_t* a_adder = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_nodes);
  int64_t* mark = (int64_t*)calloc((size_t)n_nodes, sizeof(int64_t));
  for (int64_t i = 0; i < n_nodes; i++) { w_adder[i] = -1; a_adder[i] = -1; } /* :617-618 */
  for (int64_t r = 0; r < n_rays; r++) { /* MarkVistNodeKernel :518-574 */
    int64_t s = pts_offsets[r], e = pts_offsets[r + 1];
    if (s >= e) continue;
    float max_w = 0.f, max_a = 0.f;
    for (int64_t i = s; i < e; i++) { max_w = fmaxf(max_w, weights[i]); max_a = fmaxf(max_a, alphas[i]); }
    /* float * double-literal -> evaluated in double, then narrowed (:544-545) */
    const float w_thres = fminf((float)(max_w * REL_WEIGHT_THRES), (float)ABS_WEIGHT_THRES);
    const float a_thres = fminf((float)(max_a * REL_ALPHA_THRES), (float)ABS_ALPHA_THRES);
    float cw = 0.f, ca = 0.f;
    int64_t cur = -1, cnt = 0;
    for (int64_t i = s; i < e; i++) {
      if (cur != oct_idx[i]) {
        if (cur >= 0) {
          amax(w_adder + cur, cw > w_thres ? OCC_WEIGHT_BASE : -1);
          amax(a_adder + cur, ca > a_thres ? OCC_ALPHA_BASE : -1);
          amax(visit_cnt + cur, cnt);
          mark[cur] = 1;
        }
        cur = oct_idx[i]; cw = 0.f; ca = 0.f; cnt = 0;
      }
      cw = fmaxf(cw, weights[i]);
      ca = fmaxf(ca, alphas[i]);
      cnt += 1;
    }
    if (cur >= 0) {
      amax(w_adder + cur, cw > w_thres ? OCC_WEIGHT_BASE : -1);
      amax(a_adder + cur, ca > a_thres ? OCC_ALPHA_BASE : -1);
      amax(visit_cnt + cur, cnt);
      mark[cur] = 1;
    }
  }
  for (int64_t i = 0; i < n_nodes; i++) { /* :641-655 */
    int64_t occ = w_adder[i] > 0;
    int64_t s = weight_stats[i];
    int64_t m = occ * w_adder[i];
    s = s > m ? s : m;
    s += mark[i] * (1 - occ) * w_adder[i];
    s = s < -100 ? -100 : (s > (1 << 20) ? (1 << 20) : s);
    weight_stats[i] = s;
    occ = a_adder[i] > 0;
    s = alpha_stats[i];
    m = occ * a_adder[i];
    s = s > m ? s : m;
    s += mark[i] * (1 - occ) * a_adder[i];
    s = s < -100 ? -100 : (s > (1 << 20) ? (1 << 20) : s);
    alpha_stats[i] = s;
    if (weight_stats[i] < 0 || alpha_stats[i] < 0) nodes[i].trans_idx = -1; /* :576-582 */
  }
  free(w_adder); free(a_adder); free(mark);
}

/* ------------------------------------------------------------------------ */
/* losses.py:73-84: sum(sqrt((x-y)^2 + eps^2)) / batch  (out_norm 'b');
 * n = number of scalars, rows = batch dimension */
double gfn_o_charbonnier(const float* pred, const float* gt, int64_t n, float eps, float* d_pred) {
  /* caller passes n = rows*3; normalisation by rows = n/3 */
  double s = 0.0;
  double rows = (double)(n / 3);
  for (int64_t i = 0; i < n; i++) {
    double d = (double)pred[i] - gt[i];
    double q = sqrt(d * d + (double)eps * eps);
    s += q;
    if (d_pred) d_pred[i] = (float)(d / q / rows);
  }
  return s / rows;
}

/* torch.optim.Adam (single tensor, no amsgrad, wd 0, maximize False) */
void gfn_o_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr,
                     float beta1, float beta2, float eps) {
  double bc1 = 1.0 - pow((double)beta1, step);
  double bc2 = 1.0 - pow((double)beta2, step);
  float step_size = (float)(lr / bc1);
  float bc2_sqrt = (float)sqrt(bc2);
  for (int64_t i = 0; i < n; i++) {
    m[i] = m[i] + (g[i] - m[i]) * (1.f - beta1); /* lerp */
    v[i] = v[i] * beta2 + (1.f - beta2) * g[i] * g[i];
    float denom = sqrtf(v[i]) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (m[i] / denom);
  }
}


/* ======================================================================== *
 * a15: PersOctree::ProcOctree (PersSampler.cpp:154-417), statement by statement.
 * ======================================================================== */
#define GFN_O_INIT_NODE_STAT 1000 /* PersSampler.h INIT_NODE_STAT */

typedef struct {
  gfn_o_tree_node* nodes;
  int64_t* w;
  int64_t* a;
  int64_t n, cap;
} o_vec;

static void o_vec_push(o_vec* v, const gfn_o_tree_node* nd, int64_t w, int64_t a) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->nodes = (gfn_o_tree_node*)realloc(v->nodes, (size_t)v->cap * sizeof(gfn_o_tree_node));
    v->w = (int64_t*)realloc(v->w, (size_t)v->cap * sizeof(int64_t));
    v->a = (int64_t*)realloc(v->a, (size_t)v->cap * sizeof(int64_t));
  }
  /* memcpy, not struct assignment: padding bytes (part of the checkpoint blob) must travel too */
  if (nd) memcpy(&v->nodes[v->n], nd, sizeof(gfn_o_tree_node)); else memset(&v->nodes[v->n], 0, sizeof(gfn_o_tree_node));
  v->w[v->n] = w;
  v->a[v->n] = a;
  v->n++;
}

static int o_single_child(const gfn_o_tree_node* t, int64_t u) { /* :234-252 */
  int64_t cnt = 0, ret = -1;
  for (int64_t i = 0; i < 8; i++)
    if (t[u].childs[i] >= 0) { ret = i; cnt++; }
  return cnt == 1 ? (int)ret : -1;
}

typedef struct {
  const gfn_o_tree_node* wp; const int64_t* w_wp; const int64_t* a_wp;
  const int64_t* visit_cnt; const int64_t* inv_idx; int brute_force; o_vec* out;
} o_subdiv_ctx;

static int64_t o_subdiv(o_subdiv_ctx* c, int64_t u, int64_t pa) { /* :336-395 */
  o_vec* nn = c->out;
  const int64_t new_u = nn->n;
  o_vec_push(nn, &c->wp[u], c->w_wp[u], c->a_wp[u]);
  nn->nodes[new_u].parent = pa;
  if (c->wp[u].is_leaf_node) {
    if (!c->brute_force && c->visit_cnt[c->inv_idx[u]] <= 4) return new_u;
    for (int64_t st = 0; st < 8; st++) {
      const float off[3] = {(float)((st >> 2) & 1) - .5f, (float)((st >> 1) & 1) - .5f, (float)(st & 1) - .5f};
      float sub_center[3];
      for (int k = 0; k < 3; k++) /* center + side_len * .5f * offset, left to right (:353) */
        sub_center[k] = nn->nodes[new_u].center[k] + (nn->nodes[new_u].side_len * .5f) * off[k];
      const int64_t v = nn->n;
      o_vec_push(nn, NULL, nn->w[new_u], nn->a[new_u]);  /* emplace_back(): value-initialised node */
      nn->nodes[new_u].childs[st] = v;
      for (int k = 0; k < 3; k++) nn->nodes[v].center[k] = sub_center[k];
      nn->nodes[v].side_len = nn->nodes[new_u].side_len * .5f;
      nn->nodes[v].parent = new_u;
      for (int k = 0; k < 8; k++) nn->nodes[v].childs[k] = -1;
      nn->nodes[v].is_leaf_node = 1;
      nn->nodes[v].trans_idx = nn->nodes[new_u].trans_idx;
      /* block_idx left at its value-initialised 0 (:366 is commented out in the reference) */
    }
    nn->nodes[new_u].is_leaf_node = 0;
    nn->nodes[new_u].trans_idx = -1;
    nn->w[new_u] = GFN_O_INIT_NODE_STAT;
    nn->a[new_u] = GFN_O_INIT_NODE_STAT;
  } else {
    for (int64_t st = 0; st < 8; st++) {
      if (nn->nodes[new_u].childs[st] >= 0) {
        const int64_t v = o_subdiv(c, nn->nodes[new_u].childs[st], new_u);
        nn->nodes[new_u].childs[st] = v;
      }
    }
  }
  return new_u;
}

void gfn_o_proc_octree(const gfn_o_tree_node* nodes, int64_t n_nodes, const int64_t* weight_stats,
                       const int64_t* alpha_stats, const int64_t* visit_cnt, int compact, int subdivide,
                       int brute_force, gfn_o_tree_node** nodes_out, int64_t* n_out, int64_t** weight_out,
                       int64_t** alpha_out) {
  gfn_o_tree_node* t = (gfn_o_tree_node*)malloc((size_t)n_nodes * sizeof(gfn_o_tree_node));
  memcpy(t, nodes, (size_t)n_nodes * sizeof(gfn_o_tree_node));
  /* First, compact tree nodes (:175-223) */
  while (compact) {
    for (int64_t u = 0; u < n_nodes; u++) {
      if (!t[u].is_leaf_node) continue;
      if (t[u].trans_idx < 0 && t[u].parent >= 0) {
        const int64_t v = t[u].parent;
        for (int64_t st = 0; st < 8; st++)
          if (t[v].childs[st] == u) t[v].childs[st] = -1;
      }
    }
    int update_flag = 0;
    for (int64_t u = 1; u < n_nodes; u++) { /* root can not be leaf node */
      int has_valid_childs = 0;
      for (int64_t st = 0; st < 8; st++)
        if (t[u].childs[st] >= 0) { has_valid_childs = 1; break; }
      if (!has_valid_childs) {
        if (!t[u].is_leaf_node) update_flag = 1;
        t[u].is_leaf_node = 1;
      }
    }
    if (!update_flag) break;
  }
  /* Compress path (:226-276) */
  if (compact) {
    for (int64_t u = 0; u < n_nodes; u++) {
      if (t[u].is_leaf_node && t[u].trans_idx < 0) continue;
      int64_t v = t[u].parent;
      int st = -1;
      while (v >= 0 && t[v].parent >= 0 && (st = o_single_child(t, v)) >= 0) {
        const int64_t vv = t[v].parent;
        for (int64_t i = 0; i < 8; i++)
          if (t[vv].childs[i] == v) t[vv].childs[i] = u;
        t[u].parent = vv;
        t[v].trans_idx = -1;
        t[v].is_leaf_node = 1; /* The flag to remove it */
        v = vv;
      }
      (void)st;
    }
  }
  /* renumber (:278-321) */
  int64_t* new_idx = (int64_t*)malloc((size_t)n_nodes * sizeof(int64_t));
  int64_t* inv_idx = (int64_t*)malloc((size_t)n_nodes * sizeof(int64_t));
  int64_t n_compacted = 0;
  for (int64_t u = 0; u < n_nodes; u++) {
    new_idx[u] = -1;
    if (!t[u].is_leaf_node || t[u].trans_idx >= 0) {
      inv_idx[n_compacted] = u;
      new_idx[u] = n_compacted++;
    }
  }
  o_vec nv = {0};
  for (int64_t u = 0; u < n_nodes; u++) {
    if (new_idx[u] < 0) continue;
    gfn_o_tree_node node;
    memcpy(&node, &t[u], sizeof(node));
    if (node.parent >= 0) node.parent = new_idx[node.parent];
    for (int64_t st = 0; st < 8; st++)
      if (node.childs[st] >= 0) node.childs[st] = new_idx[node.childs[st]];
    o_vec_push(&nv, &node, weight_stats[u], alpha_stats[u]);
  }
  /* Sub-divide (:326-398) */
  if (subdivide) {
    o_vec out = {0};
    o_subdiv_ctx c = {nv.nodes, nv.w, nv.a, visit_cnt, inv_idx, brute_force, &out};
    o_subdiv(&c, 0, -1);
    free(nv.nodes); free(nv.w); free(nv.a);
    nv = out;
  }
  free(t); free(new_idx); free(inv_idx);
  *nodes_out = nv.nodes; *n_out = nv.n; *weight_out = nv.w; *alpha_out = nv.a;
}

void gfn_o_free(void* p) { free(p); }

/* CheckVisible (PersSampler_cuda.cu:680-707).  Eigen 3.4 evaluates Matrix<3,4> * v.homogeneous() WITHOUT forming the
 * 4-vector (Geometry/Homogeneous.h, homogeneous_left_product_impl): dst = lhs.leftCols(3) * v; dst += lhs.col(3),
 * the 3x3 . 3x1 part coefficient-based with the 3-term redux a + (b + c).  (QueryFrameTransform, by contrast,
 * materialises the Wec4f first, :156-157, and gets the 4-term halving tree.)  norm() = sqrt(x*x + (y*y + z*z)). */
static int64_t o_check_visible(const float c[3], float side_len, const float* K, const float* W, const float* bd) {
  float p[3];
  for (int r = 0; r < 3; r++) p[r] = (W[4 * r] * c[0] + (W[4 * r + 1] * c[1] + W[4 * r + 2] * c[2])) + W[4 * r + 3];
  const float radius = (float)((double)side_len * 0.707);  /* float * double literal, rounded back on assignment */
  if (-p[2] < bd[0] - radius || -p[2] > bd[1] + radius) return 0;
  const float nrm = sqrtf(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));
  if (nrm < radius) return 1;
  const float cx = K[2], cy = K[5], fx = K[0], fy = K[4];
  const float bias_x = radius / -p[2] * fx, bias_y = radius / -p[2] * fy;
  const float img_x = p[0] / -p[2] * fx, img_y = p[1] / -p[2] * fy;
  if (img_x + bias_x < -cx || img_x > cx + bias_x || img_y + bias_y < -cy || img_y > cy + bias_y) return 0;
  return 1;
}

void gfn_o_mark_invisible(gfn_o_tree_node* nodes, int64_t n_nodes, const float* intri, const float* w2c,
                          const float* bounds, int64_t n_cams) {
  for (int64_t u = 0; u < n_nodes; u++) {
    int64_t n_vis = 0;
    for (int64_t cam = 0; cam < n_cams; cam++)
      n_vis += o_check_visible(nodes[u].center, nodes[u].side_len, intri + 9 * cam, w2c + 12 * cam, bounds + 2 * cam);
    if (n_vis < 1) nodes[u].trans_idx = -1;
  }
}


/* ======================================================================== *
 * Cameras.generate_rays (cameras.py:560-734), perspective, distortion-free.
 * ======================================================================== */
static void o_cam_dir(const float* R /* c2w row-major 3x4 */, float cxn, float cyn, float out[3], float* norm_out) {
  /* directions_stack = [x, y, -1]; rotated = sum(dir[None, :] * rotation, -1) (:697-699): row i = sum_j d[j] R[i][j];
   * normalize_with_norm: x / max(||x||, eps) (camera_utils.py:240-252) */
  const float d[3] = {cxn, cyn, -1.0f};
  float v[3];
  for (int i = 0; i < 3; i++) v[i] = (d[0] * R[4 * i] + d[1] * R[4 * i + 1]) + d[2] * R[4 * i + 2];
  float nrm = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  const float eps = (float)(2.220446049250313e-16 * 4.0); /* _EPS = np.finfo(float).eps * 4.0 (camera_utils.py:28), .to(x) */
  if (nrm < eps) nrm = eps;
  for (int i = 0; i < 3; i++) out[i] = v[i] / nrm;
  if (norm_out) *norm_out = nrm;
}

void gfn_o_generate_rays(const float* c2w, const float* intr, const int32_t* cam_idx, const float* coords, int64_t n,
                         float* origins, float* directions, float* lookat, float* pixel_area, float* dir_norm) {
  for (int64_t i = 0; i < n; i++) {
    const float* M = c2w + 12 * (int64_t)cam_idx[i];
    const float fx = intr[4 * cam_idx[i]], fy = intr[4 * cam_idx[i] + 1], cx = intr[4 * cam_idx[i] + 2], cy = intr[4 * cam_idx[i] + 3];
    const float y = coords[2 * i], x = coords[2 * i + 1];
    /* :606-608 */
    const float c0x = (x - cx) / fx, c0y = -(y - cy) / fy;
    const float cxx = (x - cx + 1.0f) / fx, cyy = -(y - cy + 1.0f) / fy;
    float d0[3], d1[3], d2[3], nrm;
    o_cam_dir(M, c0x, c0y, d0, &nrm);
    o_cam_dir(M, cxx, c0y, d1, NULL);
    o_cam_dir(M, c0x, cyy, d2, NULL);
    if (origins) for (int k = 0; k < 3; k++) origins[3 * i + k] = M[4 * k + 3];
    if (lookat) for (int k = 0; k < 3; k++) lookat[3 * i + k] = M[4 * k + 2];
    if (directions) for (int k = 0; k < 3; k++) directions[3 * i + k] = d0[k];
    if (dir_norm) dir_norm[i] = nrm;
    if (pixel_area) { /* :712-716 */
      float sx = 0.f, sy = 0.f;
      for (int k = 0; k < 3; k++) { const float a = d0[k] - d1[k], b = d0[k] - d2[k]; sx += a * a; sy += b * b; }
      pixel_area[i] = sqrtf(sx) * sqrtf(sy);
    }
  }
}


/* ======================================================================== *
 * S3IM (nerfstudio/model_components/losses.py:713-794), kernel 4, stride 4.
 * ======================================================================== */
double gfn_o_s3im(const float* a, const float* b, const int32_t* index, int64_t n_idx, int patch_h, float* d_b, int64_t n_rays) {
  const int K = 4, stride = 4, pad = (K - 1) / 2; /* :741 padding = (kernel_size-1)//2 */
  const int64_t H = patch_h, W = n_idx / patch_h; /* reshape(1, 3, patch_h, -1), :790-791 */
  const int64_t OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
  double g[4], gs = 0; /* gaussian(4, 1.5), :726-728 */
  for (int x = 0; x < K; x++) { g[x] = exp(-(double)((x - K / 2) * (x - K / 2)) / (2.0 * 1.5 * 1.5)); gs += g[x]; }
  float w[4][4];
  { float g32[4]; for (int x = 0; x < K; x++) g32[x] = (float)g[x] / (float)gs; /* torch.Tensor (f32) / sum */
    for (int i = 0; i < K; i++) for (int j = 0; j < K; j++) w[i][j] = g32[i] * g32[j]; }
  const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
  const double count = (double)(3 * OH * OW);
  if (d_b) memset(d_b, 0, (size_t)n_rays * 3 * sizeof(float));
  double total = 0;
  for (int c = 0; c < 3; c++)
    for (int64_t oy = 0; oy < OH; oy++)
      for (int64_t ox = 0; ox < OW; ox++) {
        double mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
        for (int i = 0; i < K; i++)
          for (int j = 0; j < K; j++) {
            const int64_t y = oy * stride - pad + i, x = ox * stride - pad + j;
            if (y < 0 || y >= H || x < 0 || x >= W) continue; /* zero padding */
            const int64_t r = index[y * W + x];
            const double x1 = a[3 * r + c], x2 = b[3 * r + c], ww = w[i][j];
            mu1 += ww * x1; mu2 += ww * x2; e11 += ww * x1 * x1; e22 += ww * x2 * x2; e12 += ww * x1 * x2;
          }
        const double s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const double A1 = 2 * mu1 * mu2 + C1, A2 = 2 * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        total += (A1 * A2) / (B1 * B2);
        if (!d_b) continue;
        for (int i = 0; i < K; i++)
          for (int j = 0; j < K; j++) {
            const int64_t y = oy * stride - pad + i, x = ox * stride - pad + j;
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
            const int64_t r = index[y * W + x];
            const double x1 = a[3 * r + c], x2 = b[3 * r + c], ww = w[i][j];
            const double dA1 = 2 * mu1 * ww, dA2 = 2 * ww * (x1 - mu1), dB1 = 2 * mu2 * ww, dB2 = 2 * ww * (x2 - mu2);
            const double ds = (dA1 * A2 + A1 * dA2) / (B1 * B2) - (A1 * A2) * (dB1 * B2 + B1 * dB2) / ((B1 * B2) * (B1 * B2));
            d_b[3 * r + c] += (float)(-ds / count); /* loss = 1 - mean(ssim_map) */
          }
      }
  return 1.0 - total / count;
}
