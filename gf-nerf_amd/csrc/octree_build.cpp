// octree_build.cpp -- host-side construction of the perspective-warped octree.
//
// Replaces PersOctree::PersOctree / ConstructTreeNode / ConstructTrans / PCA and the
// anonymous helpers DistanceSummary / GetVisiCams
// (gfnerf/bindings/PtsSampler/PersSampler.cpp:12-88, 92-152, 516-831), which the
// reference runs as one libtorch call chain per node with CUDA round trips.
// Here it is plain C++17 + OpenMP: no torch, no Eigen.  The outputs are the exact
// checkpoint byte layouts (TreeNode 128 B, TransInfo 576 B).
//
// Not bit-comparable with the reference: it draws torch::rand / randint streams
// (PersSampler.cpp:540,641) that cannot be reproduced; this builder uses its own
// seeded generator.  Visibility is computed with the same 128 x res_h pixel-ray
// slab test as GetVisiCams, restricted (exactly) to the pixels whose rays can
// reach the node's bounding sphere.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/gfnerf_amd.h"

void gfn_set_error(const char* fmt, ...);

namespace {

struct Rng {  // splitmix64 -> uniform floats
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  float uniform() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }  // [0,1)
  int64_t randint(int64_t n) { return (int64_t)(next() % (uint64_t)n); }
};

struct Vec3 {
  float x, y, z;
};
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm(Vec3 a) { return sqrtf(dot(a, a)); }

// general 3x3 inverse (row-major)
bool inv3(const float m[9], float o[9]) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  if (det == 0.0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  o[0] = (float)(A * id); o[1] = (float)(-(b * i - c * h) * id); o[2] = (float)((b * f - c * e) * id);
  o[3] = (float)(B * id); o[4] = (float)((a * i - c * g) * id);  o[5] = (float)(-(a * f - c * d) * id);
  o[6] = (float)(C * id); o[7] = (float)(-(a * h - b * g) * id); o[8] = (float)((a * e - b * d) * id);
  return true;
}

// PersSampler.cpp:12-26.  torch.quantile default = linear interpolation.
float distance_summary(const std::vector<float>& dis) {
  if (dis.empty()) return 1e8f;
  std::vector<float> ld(dis.size());
  for (size_t i = 0; i < dis.size(); i++) ld[i] = logf(dis[i]);
  std::vector<float> sorted = ld;
  std::sort(sorted.begin(), sorted.end());
  const float pos = 0.25f * (float)(sorted.size() - 1);
  const size_t lo = (size_t)floorf(pos);
  const size_t hi = std::min(lo + 1, sorted.size() - 1);
  const float thres = sorted[lo] + (pos - (float)lo) * (sorted[hi] - sorted[lo]);
  double msum = 0.0, wsum = 0.0, all = 0.0;
  for (float v : ld) {
    all += v;
    if (v < thres) { msum += 1.0; wsum += v; }
  }
  if (msum < 1e-3) return expf((float)(all / (double)ld.size()));
  return expf((float)(wsum / msum));
}

struct Builder {
  int64_t n_cams;
  const float* c2w;    // [n,3,4]
  const float* intri;  // [n,3,3]
  const float* bounds; // [n,2]
  int max_depth;
  float split_dist_thres;
  int threads;
  std::vector<gfn_tree_node> nodes;
  std::vector<gfn_trans_info> trans;
  std::vector<float> w2c_rot;  // inverse of c2w[:, :3,:3], [n,9]
  struct LeafJob { int64_t node; std::vector<int64_t> visi; };
  std::vector<LeafJob> leaf_jobs;
  // pixel grid of GetVisiCams (PersSampler.cpp:51-65), intrinsics of camera 0 for all
  float half_w, half_h, cx, cy, fx, fy;
  int res_w, res_h;
  bool failed = false;


  Vec3 cam_pos(int64_t c) const { return {c2w[c * 12 + 3], c2w[c * 12 + 7], c2w[c * 12 + 11]}; }

  void init() {
    half_w = intri[2]; half_h = intri[5]; cx = intri[2]; cy = intri[5]; fx = intri[0]; fy = intri[4];
    res_w = 128;
    res_h = (int)lroundf((float)res_w / half_w * half_h);
    w2c_rot.resize((size_t)n_cams * 9);
    for (int64_t c = 0; c < n_cams; c++) {
      float r[9];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 3 + j] = c2w[c * 12 + i * 4 + j];
      if (!inv3(r, &w2c_rot[c * 9])) failed = true;
    }
  }

  // one pixel ray of camera c against the node box: PersSampler.cpp:65-78
  bool pixel_hits(int64_t c, int pi, int pj, Vec3 lo, Vec3 hi) const {
    const float i = res_h > 1 ? .5f + (half_h * 2.f - 1.f) * (float)pi / (float)(res_h - 1) : .5f;
    const float j = res_w > 1 ? .5f + (half_w * 2.f - 1.f) * (float)pj / (float)(res_w - 1) : .5f;
    const float ccx = (j - cx) / fx, ccy = -(i - cy) / fy, ccz = -1.f;
    const float* m = c2w + c * 12;
    const float d[3] = {m[0] * ccx + m[1] * ccy + m[2] * ccz, m[4] * ccx + m[5] * ccy + m[6] * ccz,
                        m[8] * ccx + m[9] * ccy + m[10] * ccz};
    const float o[3] = {m[3], m[7], m[11]};
    const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    float far = INFINITY, near = -INFINITY;
    for (int k = 0; k < 3; k++) {
      float a = (l[k] - o[k]) / d[k], b = (h[k] - o[k]) / d[k];
      // nan_to_num(x, 0, 1e6, -1e6)
      a = std::isnan(a) ? 0.f : (std::isinf(a) ? (a > 0 ? 1e6f : -1e6f) : a);
      b = std::isnan(b) ? 0.f : (std::isinf(b) ? (b > 0 ? 1e6f : -1e6f) : b);
      far = fminf(far, fmaxf(a, b));
      near = fmaxf(near, fminf(a, b));
    }
    far = fminf(far, bounds[c * 2 + 1]);
    near = fmaxf(near, bounds[c * 2 + 0]);
    return far > near;
  }

  // GetVisiCams (PersSampler.cpp:45-88): camera c sees the node iff ANY of its
  // res_h x 128 pixel rays crosses the box inside [near, far].  Only the pixels whose
  // rays can reach the node's bounding sphere are tested (a ray that misses the
  // sphere misses the box), which is exact and makes the test O(1) for far nodes.
  bool cam_sees(int64_t c, Vec3 center, float side) const {
    const Vec3 lo = {center.x - side * .5f, center.y - side * .5f, center.z - side * .5f};
    const Vec3 hi = {center.x + side * .5f, center.y + side * .5f, center.z + side * .5f};
    const float rad = side * 0.8660255f * 1.0001f;
    const Vec3 rel = center - cam_pos(c);
    const float* wr = &w2c_rot[c * 9];
    const float px = wr[0] * rel.x + wr[1] * rel.y + wr[2] * rel.z;
    const float py = wr[3] * rel.x + wr[4] * rel.y + wr[5] * rel.z;
    const float pz = wr[6] * rel.x + wr[7] * rel.y + wr[8] * rel.z;
    const float z = -pz;  // depth along the viewing direction
    // A ray's parameter t is the camera-space depth of the point it reaches (the pixel direction has depth 1), so a hit
    // needs a point of the node's bounding sphere at a depth inside [near, far] of this camera: clip the sphere's depth
    // range with the bounds first, then bound the pixels whose rays can reach that slab of the sphere.
    const float zn = fmaxf(z - rad, bounds[c * 2 + 0]), zf = fminf(z + rad, bounds[c * 2 + 1]);
    if (zf < zn) return false;
    int j0 = 0, j1 = res_w - 1, i0 = 0, i1 = res_h - 1;
    if (zn > 1e-6f) {
      const float ulo = fminf((px - rad) / zn, (px - rad) / zf), uhi = fmaxf((px + rad) / zn, (px + rad) / zf);
      const float vlo = fminf((py - rad) / zn, (py - rad) / zf), vhi = fmaxf((py + rad) / zn, (py + rad) / zf);
      // pixel coordinate j = u*fx + cx ; i = cy - v*fy ; grid index = (coord - .5) * (res-1)/(2*half-1)
      const float sj = (float)(res_w - 1) / (half_w * 2.f - 1.f), si = (float)(res_h - 1) / (half_h * 2.f - 1.f);
      const float jlo = (ulo * fx + cx - .5f) * sj, jhi = (uhi * fx + cx - .5f) * sj;
      const float ilo = (cy - vhi * fy - .5f) * si, ihi = (cy - vlo * fy - .5f) * si;
      if (jhi < -1.f || jlo > (float)res_w || ihi < -1.f || ilo > (float)res_h) return false;
      j0 = std::max(0, (int)floorf(fmaxf(jlo, -1.f)) - 1); j1 = std::min(res_w - 1, (int)ceilf(fminf(jhi, (float)res_w)) + 1);
      i0 = std::max(0, (int)floorf(fmaxf(ilo, -1.f)) - 1); i1 = std::min(res_h - 1, (int)ceilf(fminf(ihi, (float)res_h)) + 1);
    }
    for (int pi = i0; pi <= i1; pi++)
      for (int pj = j0; pj <= j1; pj++)
        if (pixel_hits(c, pi, pj, lo, hi)) return true;
    return false;
  }

  std::vector<int64_t> visible_cams(Vec3 center, float side) const {
    std::vector<uint8_t> vis((size_t)n_cams, 0);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (int64_t c = 0; c < n_cams; c++) vis[c] = cam_sees(c, center, side) ? 1 : 0;
    std::vector<int64_t> ret;
    for (int64_t c = 0; c < n_cams; c++)
      if (vis[c]) ret.push_back(c);
    return ret;
  }

  // symmetric 12x12 eigen-decomposition, cyclic Jacobi (replaces linalg_eigh, :602)
  static void eigh12(double A[12][12], double evals[12], double V[12][12]) {
    const int n = 12;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 64; sweep++) {
      double off = 0;
      for (int p = 0; p < n; p++)
        for (int q = p + 1; q < n; q++) off += A[p][q] * A[p][q];
      if (off < 1e-30) break;
      for (int p = 0; p < n; p++)
        for (int q = p + 1; q < n; q++) {
          if (fabs(A[p][q]) < 1e-300) continue;
          const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
          for (int k = 0; k < n; k++) {
            const double akp = A[k][p], akq = A[k][q];
            A[k][p] = c * akp - s * akq;
            A[k][q] = s * akp + c * akq;
          }
          for (int k = 0; k < n; k++) {
            const double apk = A[p][k], aqk = A[q][k];
            A[p][k] = c * apk - s * aqk;
            A[q][k] = s * apk + c * aqk;
          }
          for (int k = 0; k < n; k++) {
            const double vkp = V[k][p], vkq = V[k][q];
            V[k][p] = c * vkp - s * vkq;
            V[k][q] = s * vkp + c * vkq;
          }
        }
    }
    for (int i = 0; i < n; i++) evals[i] = A[i][i];
  }

  // PCA (PersSampler.cpp:593-611): eigen-decomposition of the 12x12 covariance of the projected values, eigenvalues in
  // descending order, V[:, k] the k-th eigenvector (f32 like the reference's .to(kFloat32))
  static void pca12(const float* tv, int n_pts, double evals_sorted[12], float V[12][12]) {
    double mean[GFN_N_PROS] = {0};
    for (int p = 0; p < n_pts; p++)
      for (int i = 0; i < GFN_N_PROS; i++) mean[i] += tv[(size_t)p * GFN_N_PROS + i];
    for (int i = 0; i < GFN_N_PROS; i++) mean[i] /= n_pts;
    double cov[12][12] = {{0}};
    for (int p = 0; p < n_pts; p++) {
      double m[GFN_N_PROS];
      for (int i = 0; i < GFN_N_PROS; i++) m[i] = tv[(size_t)p * GFN_N_PROS + i] - mean[i];
      for (int i = 0; i < GFN_N_PROS; i++)
        for (int j = i; j < GFN_N_PROS; j++) cov[i][j] += m[i] * m[j];
    }
    for (int i = 0; i < GFN_N_PROS; i++)
      for (int j = i; j < GFN_N_PROS; j++) { cov[i][j] /= n_pts; cov[j][i] = cov[i][j]; }
    double evals[12], evecs[12][12];
    eigh12(cov, evals, evecs);
    int order[12];
    for (int i = 0; i < 12; i++) order[i] = i;
    std::sort(order, order + 12, [&](int a, int b) { return evals[a] > evals[b]; });
    for (int k = 0; k < 12; k++) {
      evals_sorted[k] = evals[order[k]];
      for (int i = 0; i < 12; i++) V[i][k] = (float)evecs[i][order[k]];
    }
  }

  // PersSampler.cpp:613-831
  // first: the camera (index into visi) the farthest-point selection starts from, -1 = drawn (torch::randint in the reference,
  // :641); pts_in: probe points [n_pts_in,3], NULL = 32^3 drawn uniformly in the node (torch::rand, :540); good_out: the six
  // selected cameras (indices into visi)
  bool construct_trans(const std::vector<int64_t>& visi, Vec3 center, float side, gfn_trans_info* out, Rng& rng,
                       int par, int64_t first = -1, const float* pts_in = nullptr, int n_pts_in = 0,
                       int64_t* good_out = nullptr) const {
    float frame[GFN_N_PROS][2][4];
    float dis_summary;
    if (!select_frames(visi, center, first, frame, &dis_summary, good_out, rng)) return false;
    std::vector<float> own;
    int n_pts = n_pts_in;
    if (!pts_in) {  // probe points in the node (:539-540)
      n_pts = 32 * 32 * 32;
      own.resize((size_t)n_pts * 3);
      for (int i = 0; i < n_pts * 3; i += 3) {
        own[i] = (rng.uniform() - .5f) * side + center.x;
        own[i + 1] = (rng.uniform() - .5f) * side + center.y;
        own[i + 2] = (rng.uniform() - .5f) * side + center.z;
      }
      pts_in = own.data();
    }
    memset(out, 0, sizeof(*out));
    if (!fit_weight(frame, pts_in, n_pts, out->weight, par)) return false;
    memcpy(out->w2xz, frame, sizeof(frame));
    out->center[0] = center.x; out->center[1] = center.y; out->center[2] = center.z;
    out->side_len = side;
    out->dis_summary = dis_summary;
    return true;
  }

  // first and second step of ConstructTrans (:619-768): align distances, pick six cameras, re-aim them at the node centre
  bool select_frames(const std::vector<int64_t>& visi, Vec3 center, int64_t first, float frame[GFN_N_PROS][2][4],
                     float* dis_summary_out, int64_t* good_out, Rng& rng) const {
    const int n_virt = GFN_N_PROS / 2;
    const int64_t n_cur = (int64_t)visi.size();
    std::vector<Vec3> cpos(n_cur);
    std::vector<float> dis(n_cur);
    for (int64_t i = 0; i < n_cur; i++) {
      cpos[i] = cam_pos(visi[i]);
      dis[i] = norm(cpos[i] - center);
    }
    const float dis_summary = distance_summary(dis);  // :627
    std::vector<Vec3> normed(n_cur);
    for (int64_t i = 0; i < n_cur; i++) normed[i] = (cpos[i] - center) * (1.f / dis[i]);
    // farthest-point selection of 6 cameras (:638-673)
    std::vector<int64_t> good;
    std::vector<uint8_t> mark(n_cur, 0);
    good.push_back(first >= 0 && first < n_cur ? first : rng.randint(n_cur));
    mark[good[0]] = 1;
    std::vector<float> mind(n_cur, 1e8f);
    for (int cnt = 1; cnt < n_virt && cnt < n_cur; cnt++) {
      const int64_t last = good.back();
      int64_t candi = -1;
      float max_dis = -1.f;
      for (int64_t i = 0; i < n_cur; i++) {
        if (mark[i]) continue;
        mind[i] = fminf(mind[i], norm(normed[i] - normed[last]));
        if (mind[i] > max_dis) { max_dis = mind[i]; candi = i; }
      }
      if (candi < 0) return false;
      mark[candi] = 1;
      good.push_back(candi);
    }
    for (size_t i = 0; good.size() < (size_t)n_virt; i++) good.push_back(good[i]);
    if (good_out)
      for (int g = 0; g < n_virt; g++) good_out[g] = good[g];
    *dis_summary_out = dis_summary;

    const float focal = intri[0] / intri[2];  // :755
    for (int g = 0; g < n_virt; g++) {
      const int64_t ci = good[g];
      const float cam_scale = fminf(fmaxf(dis[ci] / dis_summary, 1.f), 1e9f);                 // :684
      const Vec3 rel = normed[ci] * fminf(fmaxf(dis[ci], dis_summary), 1e9f);                  // :685
      const Vec3 gpos = rel + center;
      const float* ax = &w2c_rot[visi[ci] * 9];  // rows = camera axes in world (linalg_inv, :623)
      const Vec3 expect_z = rel * (1.f / norm(rel));
      const Vec3 from_z = {ax[6], ax[7], ax[8]};
      Vec3 cr = cross(from_z, expect_z);
      float cos_v = dot(from_z, expect_z), sin_v = norm(cr);
      sin_v = std::max(-0.999999f, std::min(sin_v, 0.999999f));
      cos_v = std::max(-0.999999f, std::min(cos_v, 0.999999f));
      float angle = asinf(sin_v);
      if (cos_v < 0.f) angle = (float)M_PI - angle;
      const float crn = norm(cr);
      Vec3 k = crn > 0.f ? cr * (1.f / crn) : Vec3{1.f, 0.f, 0.f};
      // Rodrigues (Eigen::AngleAxisf, :741)
      const float c = cosf(angle), s = sinf(angle), t = 1.f - c;
      const float R[9] = {t * k.x * k.x + c,       t * k.x * k.y - s * k.z, t * k.x * k.z + s * k.y,
                          t * k.x * k.y + s * k.z, t * k.y * k.y + c,       t * k.y * k.z - s * k.x,
                          t * k.x * k.z - s * k.y, t * k.y * k.z + s * k.x, t * k.z * k.z + c};
      Vec3 axes[3];
      for (int r = 0; r < 3; r++) {  // good_cam_axis @ rots^T (:746): every axis rotated by R
        const Vec3 a = {ax[r * 3], ax[r * 3 + 1], ax[r * 3 + 2]};
        axes[r] = {R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z,
                   R[6] * a.x + R[7] * a.y + R[8] * a.z};
      }
      const Vec3 xa = axes[0] * (focal * cam_scale), ya = axes[1] * (focal * cam_scale), za = axes[2];
      const Vec3 rows0[2] = {xa, ya};
      for (int which = 0; which < 2; which++) {  // i<6: x-axis rows, i>=6: y-axis rows (:760-768)
        float(*f)[4] = frame[g + which * n_virt];
        f[0][0] = rows0[which].x; f[0][1] = rows0[which].y; f[0][2] = rows0[which].z;
        f[0][3] = -dot(rows0[which], gpos);
        f[1][0] = za.x; f[1][1] = za.y; f[1][2] = za.z;
        f[1][3] = -dot(za, gpos);
      }
    }

    return true;
  }

  // third step of ConstructTrans (:770-815): project the probe points through the 12 frames, PCA of the projected values,
  // rows normalised by the mean expected step
  // par: threads of the per-point loops (1 when the caller already runs one leaf per thread)
  bool fit_weight(const float frame[GFN_N_PROS][2][4], const float* pts, int n_pts, float weight[3][GFN_N_PROS],
                  int par) const {
    std::vector<float> tv((size_t)n_pts * GFN_N_PROS);          // transed value a/b
    std::vector<float> dvd((size_t)n_pts * GFN_N_PROS * 3);     // dv/dxyz
    bool bad = false;
#pragma omp parallel for num_threads(par) schedule(static) reduction(|| : bad)
    for (int p = 0; p < n_pts; p++) {
      const float x = pts[p * 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
      for (int i = 0; i < GFN_N_PROS; i++) {
        const float a = frame[i][0][0] * x + frame[i][0][1] * y + frame[i][0][2] * z + frame[i][0][3];
        const float b = frame[i][1][0] * x + frame[i][1][1] * y + frame[i][1][2] * z + frame[i][1][3];
        if (!(b < 0.f)) bad = true;  // CHECK(max < 0), :790
        const float dv_da = 1.f / b, dv_db = a / -(b * b);
        tv[(size_t)p * GFN_N_PROS + i] = a / b;
        for (int c = 0; c < 3; c++)
          dvd[((size_t)p * GFN_N_PROS + i) * 3 + c] = dv_da * frame[i][0][c] + dv_db * frame[i][1][c];
      }
    }
    if (bad) return false;

    // PCA (:593-611, :803-804)
    double evals[12];
    float Vall[12][12];
    pca12(tv.data(), n_pts, evals, Vall);
    float V[3][GFN_N_PROS];
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < GFN_N_PROS; i++) V[r][i] = Vall[i][r];

    // step normalisation (:807-815)
    double mean_step[3] = {0, 0, 0};
    bool singular = false;
#pragma omp parallel num_threads(par)
    {
      double local[3] = {0, 0, 0};
      bool lsing = false;
#pragma omp for schedule(static) nowait
      for (int p = 0; p < n_pts; p++) {
        float jac[9];
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) {
            float acc = 0.f;
            for (int i = 0; i < GFN_N_PROS; i++) acc += V[r][i] * dvd[((size_t)p * GFN_N_PROS + i) * 3 + c];
            jac[r * 3 + c] = acc;
          }
        float inv[9];
        if (!inv3(jac, inv)) { lsing = true; continue; }
        float jmax[3] = {0, 0, 0};
        for (int i = 0; i < GFN_N_PROS; i++)
          for (int c = 0; c < 3; c++) {
            const float* dv = &dvd[((size_t)p * GFN_N_PROS + i) * 3];
            const float v = fabsf(dv[0] * inv[c] + dv[1] * inv[3 + c] + dv[2] * inv[6 + c]);
            jmax[c] = fmaxf(jmax[c], v);
          }
        for (int c = 0; c < 3; c++) local[c] += 1.0 / jmax[c];
      }
#pragma omp critical
      {
        for (int c = 0; c < 3; c++) mean_step[c] += local[c];
        singular = singular || lsing;
      }
    }
    if (singular) return false;
    for (int r = 0; r < 3; r++) {
      const float ms = (float)(mean_step[r] / n_pts);
      for (int i = 0; i < GFN_N_PROS; i++) {
        const float v = V[r][i] / ms;
        if (!std::isfinite(v)) return false;
        weight[r][i] = v;
      }
    }
    return true;
  }

  // PersSampler.cpp:516-591
  void construct_node(int64_t u, int depth, Vec3 center, float side) {
    {
      gfn_tree_node& nd = nodes[u];
      nd.center[0] = center.x; nd.center[1] = center.y; nd.center[2] = center.z;
      nd.side_len = side;
      nd.is_leaf_node = 0;
      nd.trans_idx = -1;
      nd.block_idx = -1;
      for (int i = 0; i < 8; i++) nd.childs[i] = -1;
    }
    if (depth > max_depth) {
      nodes[u].is_leaf_node = 1;
      return;
    }
    const std::vector<int64_t> visi = visible_cams(center, side);
    std::vector<float> visi_dis(visi.size());
    for (size_t i = 0; i < visi.size(); i++) visi_dis[i] = norm(cam_pos(visi[i]) - center);
    const float summary = distance_summary(visi_dis);
    const bool unaddressed = (visi.size() >= GFN_N_PROS / 2) && (summary < side * split_dist_thres);  // :555
    if (unaddressed) {
      for (int st = 0; st < 8; st++) {
        const int64_t v = (int64_t)nodes.size();
        nodes.emplace_back();
        memset(&nodes.back(), 0, sizeof(gfn_tree_node));
        const Vec3 off = {(float)((st >> 2) & 1) - .5f, (float)((st >> 1) & 1) - .5f, (float)(st & 1) - .5f};
        const Vec3 sub = center + off * (side * .5f);
        nodes[u].childs[st] = v;
        nodes[v].parent = u;
        construct_node(v, depth + 1, sub, side * .5f);
      }
    } else if (visi.size() < GFN_N_PROS / 2) {
      nodes[u].is_leaf_node = 1;  // leaf without enough cameras: invalid (:573-577)
    } else {
      nodes[u].is_leaf_node = 1;
      leaf_jobs.push_back({u, visi});  // fitted by fit_leaves(), one leaf per thread
    }
  }

  // ConstructTrans of every leaf that has enough cameras (:579-588).  The leaves are independent, so they run one per
  // thread (a leaf's own loops are only 32^3 points long: splitting THOSE over the host's cores is fork/join-bound), each
  // with its own random stream derived from (seed, position of the leaf in depth-first order), which makes the result
  // independent of the thread count.  Transform indices are handed out in depth-first order afterwards, as the reference's
  // recursion does.
  void fit_leaves(uint64_t seed, int leaf_threads) {
    const int64_t n = (int64_t)leaf_jobs.size();
    std::vector<gfn_trans_info> fitted((size_t)n);
    std::vector<uint8_t> ok((size_t)n, 0);
#pragma omp parallel for num_threads(leaf_threads) schedule(dynamic, 1)
    for (int64_t j = 0; j < n; j++) {
      const gfn_tree_node& nd = nodes[leaf_jobs[j].node];
      Rng leaf_rng(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(j + 1)));
      ok[j] = construct_trans(leaf_jobs[j].visi, {nd.center[0], nd.center[1], nd.center[2]}, nd.side_len, &fitted[j],
                              leaf_rng, 1)
                  ? 1 : 0;
    }
    for (int64_t j = 0; j < n; j++)
      if (ok[j]) {  // else: the reference would abort on its CHECKs (:790,:799); keep the leaf invalid instead
        nodes[leaf_jobs[j].node].trans_idx = (int64_t)trans.size();
        trans.push_back(fitted[j]);
      }
    leaf_jobs.clear();
  }
};

}  // namespace

extern "C" {

int gfn_octree_build_host(const float* c2w_host, const float* intri_host, const float* bounds_host,
                          int64_t n_cams, int max_depth, float bbox_side_len, float split_dist_thres,
                          uint64_t seed, int threads, void** nodes_out_host, int64_t* n_nodes_out,
                          void** trans_out_host, int64_t* n_trans_out) {
  if (!c2w_host || !intri_host || !bounds_host || !nodes_out_host || !n_nodes_out || !trans_out_host ||
      !n_trans_out || n_cams <= 0 || max_depth < 0 || max_depth > 22 || !(bbox_side_len > 0.f)) {
    gfn_set_error("gfn_octree_build_host: bad arguments");
    return GFN_ERR_ARG;
  }
  int leaf_threads = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  leaf_threads = threads;            // one leaf per thread: scales with the host
  if (threads > 16) threads = 16;    // the per-node camera loop is short; more threads only add fork/join cost
#else
  threads = 1;
#endif
  Builder b;
  b.n_cams = n_cams; b.c2w = c2w_host; b.intri = intri_host; b.bounds = bounds_host;
  b.max_depth = max_depth; b.split_dist_thres = split_dist_thres; b.threads = threads;
  b.init();
  if (b.failed) {
    gfn_set_error("gfn_octree_build_host: singular camera rotation");
    return GFN_ERR_ARG;
  }
  b.nodes.emplace_back();
  memset(&b.nodes[0], 0, sizeof(gfn_tree_node));
  b.nodes[0].parent = -1;
  b.construct_node(0, 0, {0.f, 0.f, 0.f}, bbox_side_len);
  b.fit_leaves(seed, leaf_threads);
  const size_t nb = b.nodes.size() * sizeof(gfn_tree_node);
  const size_t tb = std::max<size_t>(b.trans.size(), 1) * sizeof(gfn_trans_info);
  void* np = malloc(nb);
  void* tp = calloc(1, tb);
  if (!np || !tp) {
    free(np); free(tp);
    gfn_set_error("gfn_octree_build_host: out of memory");
    return GFN_ERR_ARG;
  }
  memcpy(np, b.nodes.data(), nb);
  if (!b.trans.empty()) memcpy(tp, b.trans.data(), b.trans.size() * sizeof(gfn_trans_info));
  *nodes_out_host = np; *n_nodes_out = (int64_t)b.nodes.size();
  *trans_out_host = tp; *n_trans_out = (int64_t)b.trans.size();
  return GFN_OK;
}

/* ---- pieces of the build, exposed so that tests can hold them against the reference's own Python statement of them
 * (gfnerf/persoctree.py: distance_summary :234-246, PCA :183-194, get_visi_cams :248-290, construct_trans :86-181) ---- */

float gfn_octree_distance_summary_host(const float* dis, int64_t n) {
  return distance_summary(std::vector<float>(dis, dis + (n > 0 ? n : 0)));
}

int gfn_octree_pca12_host(const float* vals, int64_t n_pts, float* evals_out, float* evecs_out) {
  if (!vals || !evals_out || !evecs_out || n_pts <= 0 || n_pts > (1 << 24)) {
    gfn_set_error("gfn_octree_pca12_host: bad arguments");
    return GFN_ERR_ARG;
  }
  double ev[12];
  float V[12][12];
  Builder::pca12(vals, (int)n_pts, ev, V);
  for (int i = 0; i < 12; i++) evals_out[i] = (float)ev[i];
  memcpy(evecs_out, V, sizeof(V));
  return GFN_OK;
}

static int make_builder(Builder& b, const float* c2w, const float* intri, const float* bounds, int64_t n_cams, const char* who) {
  if (!c2w || !intri || !bounds || n_cams <= 0) {
    gfn_set_error("%s: bad arguments", who);
    return GFN_ERR_ARG;
  }
  b.n_cams = n_cams; b.c2w = c2w; b.intri = intri; b.bounds = bounds;
  b.max_depth = 0; b.split_dist_thres = 1.5f; b.threads = 1;
  b.init();
  if (b.failed) {
    gfn_set_error("%s: singular camera rotation", who);
    return GFN_ERR_ARG;
  }
  return GFN_OK;
}

int gfn_octree_visible_cams_host(const float* c2w, const float* intri, const float* bounds, int64_t n_cams, const float* center,
                                 float side, uint8_t* visible_out) {
  Builder b;
  if (!center || !visible_out) {
    gfn_set_error("gfn_octree_visible_cams_host: bad arguments");
    return GFN_ERR_ARG;
  }
  const int rc = make_builder(b, c2w, intri, bounds, n_cams, "gfn_octree_visible_cams_host");
  if (rc != GFN_OK) return rc;
  for (int64_t c = 0; c < n_cams; c++) visible_out[c] = b.cam_sees(c, {center[0], center[1], center[2]}, side) ? 1 : 0;
  return GFN_OK;
}

int gfn_octree_construct_trans_host(const float* c2w_visible, const float* intri, int64_t n_visible, const float* center, float side,
                                    int64_t first_cam, const float* pts, int64_t n_pts, uint64_t seed, void* trans_out,
                                    int64_t* good_cams_out) {
  Builder b;
  if (!center || !trans_out || (pts && n_pts <= 0) || n_pts > (1 << 24)) {
    gfn_set_error("gfn_octree_construct_trans_host: bad arguments");
    return GFN_ERR_ARG;
  }
  std::vector<float> bounds((size_t)std::max<int64_t>(n_visible, 1) * 2, 0.f);
  const int rc = make_builder(b, c2w_visible, intri, bounds.data(), n_visible, "gfn_octree_construct_trans_host");
  if (rc != GFN_OK) return rc;
  std::vector<int64_t> visi((size_t)n_visible);
  for (int64_t i = 0; i < n_visible; i++) visi[i] = i;
  Rng rng(seed);
  if (!b.construct_trans(visi, {center[0], center[1], center[2]}, side, reinterpret_cast<gfn_trans_info*>(trans_out), rng, 1,
                         first_cam, pts, (int)n_pts, good_cams_out)) {
    gfn_set_error("gfn_octree_construct_trans_host: degenerate frames (a probe point behind a virtual camera, or a singular Jacobian)");
    return GFN_ERR_ARG;
  }
  return GFN_OK;
}

int gfn_octree_fit_weight_host(const float* w2xz, const float* pts, int64_t n_pts, float* weight_out) {
  if (!w2xz || !pts || !weight_out || n_pts <= 0 || n_pts > (1 << 24)) {
    gfn_set_error("gfn_octree_fit_weight_host: bad arguments");
    return GFN_ERR_ARG;
  }
  Builder b;
  b.threads = 1;
  float frame[GFN_N_PROS][2][4];
  memcpy(frame, w2xz, sizeof(frame));
  float weight[3][GFN_N_PROS];
  if (!b.fit_weight(frame, pts, (int)n_pts, weight, 1)) {
    gfn_set_error("gfn_octree_fit_weight_host: degenerate frames");
    return GFN_ERR_ARG;
  }
  memcpy(weight_out, weight, sizeof(weight));
  return GFN_OK;
}

void gfn_free_host(void* p) { free(p); }

}  // extern "C"
