/*
 * pft_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see pft_oracle.h).
 *
 * CPU restatement of the PCL 1.8.0 particle-filter tracking path used by
 * /root/reference/src/auto_tracking.cpp:201-254 (configuration) and :691-693 (setInputCloud/compute).
 * Every function names the upstream PCL 1.8.0 file it follows; PCL is not available in the build
 * container, so those are restated from the published sources.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off (no FMA contraction: PCL/x86-64 release builds have none).
 * Floating-point types (float vs double) and operation order follow PCL exactly where stated.
 */
#define _GNU_SOURCE
#include "pft_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static double omp_get_wtime(void) { return 0.0; }
#endif

/* ------------------------------------------------------------------------------------------- */
/* configuration: /root/reference/src/auto_tracking.cpp:187-253                                 */
/* ------------------------------------------------------------------------------------------- */
void orc_config_default(orc_config_t* c) {
  memset(c, 0, sizeof(*c));
  c->particle_num = 400;  /* :231 */
  c->iteration_num = 2;   /* :229 */
  for (int k = 0; k < 6; k++) {
    c->step_cov[k] = 0.015 * 0.015; /* :187 */
    c->init_cov[k] = 0.00001;       /* :192 */
    c->init_mean[k] = 0.0;          /* :193 */
  }
  c->step_cov[3] *= 40.0; /* :188-190 */
  c->step_cov[4] *= 40.0;
  c->step_cov[5] *= 40.0;
  c->alpha = 15.0;
  c->max_distance = 0.1;       /* :253 */
  c->octree_resolution = 0.01; /* :251 */
  c->distance_weight = 1.0;
  c->hsv_weight = 0.1; /* :246 */
  c->h_weight = 1.0;
  c->s_weight = 1.0;
  c->v_weight = 0.0;
  c->hsv_pcl180_argorder = 1;
  c->threads = 16; /* :845 */
  c->emulate_pcl_alloc = 1;
  c->seed = 1;
  c->kld_adaptive = 0;
  c->kld_max_particles = 500; /* :209 */
  c->kld_delta = 0.99;        /* :210 */
  c->kld_epsilon = 0.2;       /* :211 */
  for (int k = 0; k < 6; k++) c->kld_bin_size[k] = 0.1; /* :212-219 */
  c->motion_ratio = 0.25;
  c->exact_nearest = 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A1  pcl::getTransformation<float>  (PCL 1.8.0 common/include/pcl/common/impl/eigen.hpp)       */
/* ------------------------------------------------------------------------------------------- */
void orc_get_transformation(float x, float y, float z, float roll, float pitch, float yaw, float m[16]) {
  float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch);
  float E = cosf(roll), F = sinf(roll), DE = D * E, DF = D * F;
  m[0] = A * C;  m[1] = A * DF - B * E;  m[2] = B * F + A * DE;  m[3] = x;
  m[4] = B * C;  m[5] = A * E + B * DF;  m[6] = B * DE - A * F;  m[7] = y;
  m[8] = -D;     m[9] = C * F;           m[10] = C * E;          m[11] = z;
  m[12] = 0;     m[13] = 0;              m[14] = 0;              m[15] = 1;
}

/* TEST-ONLY variant of A1 (orc_tracker_set_trig_mode(t, 1)): sin / cos evaluated in double and rounded to float,
 * which is how the product's device code forms the matrix (DESIGN.md "numerics").  PCL itself calls cosf / sinf
 * (above, the default); the two agree to <= 1 ulp(float) per trig value.  With this variant the oracle and the
 * device see identical matrices, so whole tracking runs can be compared bit for bit over many frames
 * (tests/test_gpu_longrun.py); the default-mode comparison stays in place beside it. */
static void get_transformation_double_trig(float x, float y, float z, float roll, float pitch, float yaw, float m[16]) {
  float A = (float)cos((double)yaw), B = (float)sin((double)yaw), C = (float)cos((double)pitch),
        D = (float)sin((double)pitch);
  float E = (float)cos((double)roll), F = (float)sin((double)roll), DE = D * E, DF = D * F;
  m[0] = A * C;  m[1] = A * DF - B * E;  m[2] = B * F + A * DE;  m[3] = x;
  m[4] = B * C;  m[5] = A * E + B * DF;  m[6] = B * DE - A * F;  m[7] = y;
  m[8] = -D;     m[9] = C * F;           m[10] = C * E;          m[11] = z;
  m[12] = 0;     m[13] = 0;              m[14] = 0;              m[15] = 1;
}

/* A0  ParticleXYZRPY::toState -> pcl::getTranslationAndEulerAngles (same file) */
void orc_to_state(const float m[16], orc_particle_t* out) {
  memset(out, 0, sizeof(*out));
  out->x = m[3];
  out->y = m[7];
  out->z = m[11];
  out->w = 1.0f;
  out->roll = atan2f(m[9], m[10]);
  out->pitch = asinf(-m[8]);
  out->yaw = atan2f(m[4], m[0]);
}

/* A2  pcl::transformPointCloud, is_dense branch (PCL 1.8.0 common/impl/transforms.hpp):
 * every field copied, then x' = T00*x + T01*y + T02*z + T03 evaluated left to right in float. */
void orc_transform_cloud(const orc_point_t* in, size_t n, const float m[16], orc_point_t* out) {
  for (size_t i = 0; i < n; i++) {
    orc_point_t p = in[i];
    float px = in[i].x, py = in[i].y, pz = in[i].z;
    p.x = m[0] * px + m[1] * py + m[2] * pz + m[3];
    p.y = m[4] * px + m[5] * py + m[6] * pz + m[7];
    p.z = m[8] * px + m[9] * py + m[10] * pz + m[11];
    out[i] = p;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* A7b RGB2HSV + HSVColorCoherence (PCL 1.8.0 tracking/impl/hsv_color_coherence.hpp)             */
/* ------------------------------------------------------------------------------------------- */
int orc_div_table(int i) {
  /* upstream literal table == round((255 << 12) / i), div_table[0] = 0 */
  if (i <= 0) return 0;
  return (int)lrint(1044480.0 / (double)i);
}

void orc_rgb2hsv_int(int r, int g, int b, int* ph, int* ps, int* pv) {
  const int hsv_shift = 12;
  int hr = 180, hscale = 15;
  int h, s, v = b;
  int vmin = b, diff;
  int vr, vg;
  v = v > g ? v : g;
  v = v > r ? v : r;
  vmin = vmin < g ? vmin : g;
  vmin = vmin < r ? vmin : r;
  diff = v - vmin;
  vr = v == r ? -1 : 0;
  vg = v == g ? -1 : 0;
  s = diff * orc_div_table(v) >> hsv_shift;
  h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * orc_div_table(diff) * hscale + (1 << (hsv_shift + 6))) >> (7 + hsv_shift);
  h += h < 0 ? hr : 0;
  *ph = h;
  *ps = s;
  *pv = v;
}

void orc_rgb2hsv(int r, int g, int b, float* fh, float* fs, float* fv) {
  int h, s, v;
  orc_rgb2hsv_int(r, g, b, &h, &s, &v);
  *fh = (float)h / 180.0f;
  *fs = (float)s / 255.0f;
  *fv = (float)v / 255.0f;
}

static inline void hsv_of_rgba(const orc_config_t* c, uint32_t rgba, float* h, float* s, float* v) {
  int Blue = (int)(rgba & 0xff), Green = (int)((rgba >> 8) & 0xff), Red = (int)((rgba >> 16) & 0xff);
  if (c->hsv_pcl180_argorder)
    orc_rgb2hsv(Red, Blue, Green, h, s, v); /* RGB2HSV (rgb.Red, rgb.Blue, rgb.Green, ...) as upstream */
  else
    orc_rgb2hsv(Red, Green, Blue, h, s, v);
}

double orc_hsv_coherence(const orc_config_t* c, uint32_t src_rgba, uint32_t tgt_rgba) {
  float source_h, source_s, source_v, target_h, target_s, target_v;
  hsv_of_rgba(c, src_rgba, &source_h, &source_s, &source_v);
  hsv_of_rgba(c, tgt_rgba, &target_h, &target_s, &target_v);
  const float _h_diff = fabsf(source_h - target_h);
  float _h_diff2;
  if (source_h < target_h)
    _h_diff2 = fabsf(1.0f + source_h - target_h);
  else
    _h_diff2 = fabsf(1.0f + target_h - source_h);
  float h_diff;
  if (_h_diff < _h_diff2)
    h_diff = (float)c->h_weight * _h_diff * _h_diff;
  else
    h_diff = (float)c->h_weight * _h_diff2 * _h_diff2;
  const float s_diff = (float)c->s_weight * (source_s - target_s) * (source_s - target_s);
  const float v_diff = (float)c->v_weight * (source_v - target_v) * (source_v - target_v);
  const float diff2 = h_diff + s_diff + v_diff;
  return 1.0 / (1.0 + c->hsv_weight * diff2);
}

/* A7a DistanceCoherence (PCL 1.8.0 tracking/impl/distance_coherence.hpp):
 * d = (p - p').norm() on Vector4f maps (4th components both 1 -> 0), float; the SSE3 packet
 * reduction adds (dx2+dy2)+(dz2+0). Result 1/(1+d*d*w) in double. */
double orc_distance_coherence(const orc_config_t* c, const orc_point_t* s, const orc_point_t* t) {
  float dx = s->x - t->x, dy = s->y - t->y, dz = s->z - t->z, dw = s->w - t->w;
  float n2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
  double d = (double)sqrtf(n2);
  return 1.0 / (1.0 + d * d * c->distance_weight);
}

/* ------------------------------------------------------------------------------------------- */
/* A5  OctreePointCloud::addPointsFromInputCloud (PCL 1.8.0 octree/impl/octree_pointcloud.hpp)  */
/* ------------------------------------------------------------------------------------------- */
typedef struct oc_branch {
  void* child[8];
} oc_branch;
typedef struct oc_leaf {
  int* idx;
  int n, cap;
} oc_leaf;

#define ORC_MAX_GROW 64
struct orc_octree {
  const orc_point_t* pts; /* copied cloud */
  orc_point_t* own;
  size_t n;
  double res;
  double min_x, min_y, min_z, max_x, max_y, max_z;
  int bbox_defined;
  unsigned depth;
  unsigned depth_mask;
  oc_branch* root;
  size_t leaf_count, branch_count;
  int emulate_alloc;
  /* bookkeeping for tests only: insertion-time keys and the growth history */
  uint32_t* ins_key; /* 3 per point */
  uint32_t* ins_epoch;
  int n_grow;
  unsigned char grow_shift[ORC_MAX_GROW][3];
  unsigned grow_old_depth[ORC_MAX_GROW];
  uint64_t stat_queries, stat_scanned;
};

static oc_branch* new_branch(void) { return (oc_branch*)calloc(1, sizeof(oc_branch)); }

static void free_rec(void* node, unsigned levels_below) {
  if (!node) return;
  if (levels_below == 0) {
    oc_leaf* l = (oc_leaf*)node;
    free(l->idx);
    free(l);
    return;
  }
  oc_branch* b = (oc_branch*)node;
  for (int c = 0; c < 8; c++) free_rec(b->child[c], levels_below - 1);
  free(b);
}

static void set_tree_depth(orc_octree_t* t, unsigned d) {
  t->depth = d;
  t->depth_mask = 1u << (d - 1);
}

/* OctreePointCloud::getKeyBitSize, reached only for the first point (leaf_count_ == 0) */
static void get_key_bit_size(orc_octree_t* t) {
  const float minValue = FLT_EPSILON;
  unsigned max_key_x = (unsigned)((t->max_x - t->min_x) / t->res);
  unsigned max_key_y = (unsigned)((t->max_y - t->min_y) / t->res);
  unsigned max_key_z = (unsigned)((t->max_z - t->min_z) / t->res);
  unsigned max_voxels = max_key_x > max_key_y ? max_key_x : max_key_y;
  max_voxels = max_voxels > max_key_z ? max_voxels : max_key_z;
  max_voxels = max_voxels > 2u ? max_voxels : 2u;
  double l2 = log((double)max_voxels) / log(2.0);
  unsigned d = (unsigned)ceil(l2 - minValue);
  if (d > 32u) d = 32u;
  double octree_side_len = (double)(1u << d) * t->res - minValue;
  if (t->leaf_count == 0) {
    double ox = (octree_side_len - (t->max_x - t->min_x)) / 2.0;
    double oy = (octree_side_len - (t->max_y - t->min_y)) / 2.0;
    double oz = (octree_side_len - (t->max_z - t->min_z)) / 2.0;
    t->min_x -= ox; t->min_y -= oy; t->min_z -= oz;
    t->max_x += ox; t->max_y += oy; t->max_z += oz;
  } else {
    t->max_x = t->min_x + octree_side_len;
    t->max_y = t->min_y + octree_side_len;
    t->max_z = t->min_z + octree_side_len;
  }
  set_tree_depth(t, d);
}

/* OctreePointCloud::adoptBoundingBoxToPoint */
static void adopt_bbox(orc_octree_t* t, const orc_point_t* p) {
  const float minValue = FLT_EPSILON;
  for (;;) {
    int lx = (p->x < t->min_x), ly = (p->y < t->min_y), lz = (p->z < t->min_z);
    int ux = (p->x >= t->max_x), uy = (p->y >= t->max_y), uz = (p->z >= t->max_z);
    if (lx || ly || lz || ux || uy || uz || !t->bbox_defined) {
      if (t->bbox_defined) {
        unsigned char child_idx = (unsigned char)(((!ux) << 2) | ((!uy) << 1) | (!uz));
        oc_branch* nr = new_branch();
        t->branch_count++;
        nr->child[child_idx] = t->root;
        t->root = nr;
        double side = (double)(1u << t->depth) * t->res;
        if (t->n_grow < ORC_MAX_GROW) {
          t->grow_shift[t->n_grow][0] = (unsigned char)!ux;
          t->grow_shift[t->n_grow][1] = (unsigned char)!uy;
          t->grow_shift[t->n_grow][2] = (unsigned char)!uz;
          t->grow_old_depth[t->n_grow] = t->depth;
        }
        t->n_grow++;
        if (!ux) t->min_x -= side;
        if (!uy) t->min_y -= side;
        if (!uz) t->min_z -= side;
        set_tree_depth(t, t->depth + 1);
        side = (double)(1u << t->depth) * t->res - minValue;
        t->max_x = t->min_x + side;
        t->max_y = t->min_y + side;
        t->max_z = t->min_z + side;
      } else {
        t->min_x = p->x - t->res / 2; t->min_y = p->y - t->res / 2; t->min_z = p->z - t->res / 2;
        t->max_x = p->x + t->res / 2; t->max_y = p->y + t->res / 2; t->max_z = p->z + t->res / 2;
        get_key_bit_size(t);
        t->bbox_defined = 1;
      }
    } else
      break;
  }
}

static void add_point_idx(orc_octree_t* t, int i) {
  const orc_point_t* p = &t->pts[i];
  adopt_bbox(t, p);
  /* genOctreeKeyforPoint: double arithmetic, truncation */
  unsigned kx = (unsigned)((p->x - t->min_x) / t->res);
  unsigned ky = (unsigned)((p->y - t->min_y) / t->res);
  unsigned kz = (unsigned)((p->z - t->min_z) / t->res);
  t->ins_key[3 * i + 0] = kx; t->ins_key[3 * i + 1] = ky; t->ins_key[3 * i + 2] = kz;
  t->ins_epoch[i] = (uint32_t)t->n_grow;
  /* createLeafRecursive */
  oc_branch* b = t->root;
  unsigned mask = t->depth_mask;
  for (;;) {
    int c = ((!!(kx & mask)) << 2) | ((!!(ky & mask)) << 1) | (!!(kz & mask));
    if (mask > 1) {
      if (!b->child[c]) {
        b->child[c] = new_branch();
        t->branch_count++;
      }
      b = (oc_branch*)b->child[c];
      mask >>= 1;
    } else {
      oc_leaf* l = (oc_leaf*)b->child[c];
      if (!l) {
        l = (oc_leaf*)calloc(1, sizeof(oc_leaf));
        b->child[c] = l;
        t->leaf_count++;
      }
      if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 1;
        l->idx = (int*)realloc(l->idx, sizeof(int) * (size_t)l->cap);
      }
      l->idx[l->n++] = i; /* OctreeContainerPointIndices::addPointIndex -> push_back */
      break;
    }
  }
}

/* pcl::search::Octree::setInputCloud: deleteTree(); setInputCloud(); addPointsFromInputCloud() */
orc_octree_t* orc_octree_build(const orc_point_t* pts, size_t n, double resolution, int emulate_pcl_alloc) {
  orc_octree_t* t = (orc_octree_t*)calloc(1, sizeof(*t));
  t->own = (orc_point_t*)malloc(sizeof(orc_point_t) * (n ? n : 1));
  if (n) memcpy(t->own, pts, sizeof(orc_point_t) * n);
  t->pts = t->own;
  t->n = n;
  t->res = resolution;
  t->root = new_branch();
  t->branch_count = 1;
  t->depth = 0;
  t->emulate_alloc = emulate_pcl_alloc;
  t->ins_key = (uint32_t*)calloc(3 * (n ? n : 1), sizeof(uint32_t));
  t->ins_epoch = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
  for (size_t i = 0; i < n; i++) {
    /* addPointsFromInputCloud skips non-finite points; PassThrough already removed them */
    if (isfinite(pts[i].x) && isfinite(pts[i].y) && isfinite(pts[i].z)) add_point_idx(t, (int)i);
  }
  return t;
}

void orc_octree_free(orc_octree_t* t) {
  if (!t) return;
  free_rec(t->root, t->depth);
  free(t->own);
  free(t->ins_key);
  free(t->ins_epoch);
  free(t);
}

void orc_octree_info(const orc_octree_t* t, int* depth, double b[6], size_t* leaf_count, size_t* branch_count) {
  if (depth) *depth = (int)t->depth;
  if (b) {
    b[0] = t->min_x; b[1] = t->min_y; b[2] = t->min_z;
    b[3] = t->max_x; b[4] = t->max_y; b[5] = t->max_z;
  }
  if (leaf_count) *leaf_count = t->leaf_count;
  if (branch_count) *branch_count = t->branch_count;
}

void orc_octree_point_key(const orc_octree_t* t, size_t i, uint32_t key[3]) {
  key[0] = t->ins_key[3 * i]; key[1] = t->ins_key[3 * i + 1]; key[2] = t->ins_key[3 * i + 2];
  int ng = t->n_grow < ORC_MAX_GROW ? t->n_grow : ORC_MAX_GROW;
  for (int s = (int)t->ins_epoch[i]; s < ng; s++)
    for (int a = 0; a < 3; a++)
      if (t->grow_shift[s][a]) key[a] += 1u << t->grow_old_depth[s];
}

/* ------------------------------------------------------------------------------------------- */
/* A6  OctreePointCloudSearch::approxNearestSearch (PCL 1.8.0 octree/impl/octree_search.hpp)    */
/* ------------------------------------------------------------------------------------------- */
/* pointSquaredDist: (a.getVector3fMap() - b.getVector3fMap()).squaredNorm(), float.
 * Eigen's fixed-size-3 unrolled reduction adds x2 + (y2 + z2). */
static inline float point_sq_dist(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + (dy * dy + dz * dz);
}

int orc_octree_approx_nearest(const orc_octree_t* t, const orc_point_t* q, int* result_index, float* sqr_distance) {
  if (t->leaf_count == 0) return 0;
  const oc_branch* node = t->root;
  unsigned kx = 0, ky = 0, kz = 0;
  for (unsigned tree_depth = 1;; tree_depth++) {
    double min_voxel_center_distance = DBL_MAX;
    int min_child_idx = 0xFF;
    unsigned mkx = 0, mky = 0, mkz = 0;
    double vs = t->res * (double)(1u << (t->depth - tree_depth));
    for (int c = 0; c < 8; c++) {
      if (!node->child[c]) continue;
      unsigned nkx = (kx << 1) + (unsigned)(!!(c & 4));
      unsigned nky = (ky << 1) + (unsigned)(!!(c & 2));
      unsigned nkz = (kz << 1) + (unsigned)(!!(c & 1));
      /* genVoxelCenterFromOctreeKey */
      float cx = (float)(((double)nkx + 0.5f) * vs + t->min_x);
      float cy = (float)(((double)nky + 0.5f) * vs + t->min_y);
      float cz = (float)(((double)nkz + 0.5f) * vs + t->min_z);
      double voxelPointDist = point_sq_dist(cx, cy, cz, q->x, q->y, q->z);
      if (voxelPointDist >= min_voxel_center_distance) continue;
      min_voxel_center_distance = voxelPointDist;
      min_child_idx = c;
      mkx = nkx; mky = nky; mkz = nkz;
    }
    const void* child = node->child[min_child_idx];
    if (tree_depth < t->depth) {
      node = (const oc_branch*)child;
      /* U10 (switch point): the key handed to the next level is the MIN child's key
       * (`minChildKey`, as PCL 1.8.0's octree_search.hpp is recalled).  Upstream history also holds a version that
       * passed `new_key` -- the key of the LAST existing child iterated, i.e. of the highest set bit of the child mask --
       * whose voxel centres at the next level then belong to another cell.  That variant would read
       * `kx = nk_last_x; ...` here (and the device: pft_likelihood.hip, the jx / jy / jz update of the generic level).
       * Not decidable offline; min-child is implemented on both sides. */
      kx = mkx; ky = mky; kz = mkz;
      continue;
    }
    const oc_leaf* leaf = (const oc_leaf*)child;
    double smallest = DBL_MAX;
    const int* idx = leaf->idx;
    int* tmp = NULL;
    if (t->emulate_alloc) {
      /* getPointIndices(std::vector<int>&) copies the leaf's indices into a fresh vector per query */
      tmp = (int*)malloc(sizeof(int) * (size_t)leaf->n);
      memcpy(tmp, leaf->idx, sizeof(int) * (size_t)leaf->n);
      idx = tmp;
    }
    for (int i = 0; i < leaf->n; i++) {
      const orc_point_t* cand = &t->pts[idx[i]];
      double sd = point_sq_dist(cand->x, cand->y, cand->z, q->x, q->y, q->z);
      if (sd >= smallest) continue;
      *result_index = idx[i];
      smallest = sd;
      *sqr_distance = (float)sd;
    }
    if (tmp) free(tmp);
    return leaf->n > 0 ? leaf->n : 1;
  }
}

void orc_octree_scan_stats(const orc_octree_t* t, uint64_t* queries, uint64_t* scanned) {
  *queries = t->stat_queries;
  *scanned = t->stat_scanned;
}

/* ------------------------------------------------------------------------------------------- */
/* A8  ParticleFilterTracker::normalizeWeight (PCL 1.8.0 tracking/impl/particle_filter.hpp)      */
/* ------------------------------------------------------------------------------------------- */
void orc_normalize_weights(float* w, size_t n, double alpha, double* fit_ratio) {
  double w_min = DBL_MAX, w_max = -DBL_MAX;
  for (size_t i = 0; i < n; i++) {
    double weight = w[i];
    if (w_min > weight) w_min = weight;
    if (weight != 0.0 && w_max < weight) w_max = weight;
  }
  if (fit_ratio) *fit_ratio = w_min;
  if (w_max != w_min) {
    for (size_t i = 0; i < n; i++)
      if (w[i] != 0.0) w[i] = (float)exp(1.0 - alpha * (w[i] - w_min) / (w_max - w_min));
  } else {
    for (size_t i = 0; i < n; i++) w[i] = 1.0f / (float)n;
  }
  double sum = 0.0;
  for (size_t i = 0; i < n; i++) sum += w[i];
  if (sum != 0.0) {
    for (size_t i = 0; i < n; i++) w[i] = w[i] / (float)sum;
  } else {
    for (size_t i = 0; i < n; i++) w[i] = 1.0f / (float)n;
  }
}

/* ---- TEST-ONLY summation order (orc_tracker_set_sum_mode(t, 1)) ----
 * PCL adds the weights (double) and the weighted poses (float) one after the other in index order; a parallel device
 * cannot, and the product specifies its order instead: the ADJACENT-PAIR TREE over the index range padded with +0.0
 * to a power of two -- T0[i] = x[i], T(k+1)[i] = Tk[2i] + Tk[2i+1], result = the root -- in double, which any
 * power-of-two decomposition into threads / waves / workgroups / GPUs reproduces bit for bit
 * (pcl_tracking_amd/csrc/pft_population.hip).  With this mode, and trig mode 1, the oracle follows the device's
 * arithmetic exactly, so whole tracking runs can be compared bit for bit (tests/test_gpu_longrun.py).  The
 * difference from PCL's own order is <= 1 ulp(float) of the weight sum and ~1e-7 of the mean pose. */
static double tree_sum(double* v, size_t n) { /* destroys v; v has room for the next power of two >= n */
  size_t m = 1;
  while (m < n) m <<= 1;
  for (size_t i = n; i < m; i++) v[i] = 0.0;
  for (; m > 1; m >>= 1)
    for (size_t i = 0; i < m / 2; i++) v[i] = v[2 * i] + v[2 * i + 1];
  return n ? v[0] : 0.0;
}
static size_t pow2_at_least(size_t n) {
  size_t m = 1;
  while (m < n) m <<= 1;
  return m;
}

void orc_normalize_weights_tree(float* w, size_t n, double alpha, double* fit_ratio) {
  double w_min = DBL_MAX, w_max = -DBL_MAX;
  for (size_t i = 0; i < n; i++) {
    double weight = w[i];
    if (w_min > weight) w_min = weight;
    if (weight != 0.0 && w_max < weight) w_max = weight;
  }
  if (fit_ratio) *fit_ratio = w_min;
  if (w_max != w_min) {
    for (size_t i = 0; i < n; i++)
      if (w[i] != 0.0) w[i] = (float)exp(1.0 - alpha * (w[i] - w_min) / (w_max - w_min));
  } else {
    for (size_t i = 0; i < n; i++) w[i] = 1.0f / (float)n;
  }
  double* v = (double*)malloc(sizeof(double) * pow2_at_least(n ? n : 1));
  for (size_t i = 0; i < n; i++) v[i] = (double)w[i];
  const double sum = tree_sum(v, n);
  free(v);
  if (sum != 0.0) {
    for (size_t i = 0; i < n; i++) w[i] = w[i] / (float)sum;
  } else {
    for (size_t i = 0; i < n; i++) w[i] = 1.0f / (float)n;
  }
}

void orc_weighted_mean_tree(const orc_particle_t* p, size_t n, orc_particle_t* rep) {
  orc_particle_t r;
  memset(&r, 0, sizeof(r));
  r.w = 1.0f;
  double* v = (double*)malloc(sizeof(double) * pow2_at_least(n ? n : 1));
  float* out[6] = {&r.x, &r.y, &r.z, &r.roll, &r.pitch, &r.yaw};
  for (int k = 0; k < 6; k++) {
    for (size_t i = 0; i < n; i++) {
      const float c = k == 0 ? p[i].x : k == 1 ? p[i].y : k == 2 ? p[i].z : k == 3 ? p[i].roll : k == 4 ? p[i].pitch : p[i].yaw;
      v[i] = (double)c * (double)p[i].weight;
    }
    *out[k] = (float)tree_sum(v, n);
  }
  free(v);
  r.weight = 1.0f / (float)n;
  *rep = r;
}

/* A9  genAliasTable (same file): Walker alias, H stack from the front, L stack from the back */
void orc_gen_alias_table(const float* w, size_t num, int32_t* a, double* q) {
  if (num == 0) return;
  int* HL = (int*)malloc(sizeof(int) * num);
  ptrdiff_t H = 0;                   /* next free slot of the H stack */
  ptrdiff_t L = (ptrdiff_t)num - 1;  /* next free slot of the L stack */
  for (size_t i = 0; i < num; i++) q[i] = w[i] * (float)num; /* float product widened to double */
  for (size_t i = 0; i < num; i++) a[i] = (int32_t)i;
  for (size_t i = 0; i < num; i++) {
    if (q[i] >= 1.0)
      HL[H++] = (int)i;
    else
      HL[L--] = (int)i;
  }
  while (H != 0 && L != (ptrdiff_t)num - 1) {
    int j = HL[L + 1];
    int k = HL[H - 1];
    a[j] = k;
    q[k] += q[j] - 1;
    L++;
    if (q[k] < 1.0) {
      HL[L--] = k;
      --H;
    }
  }
  free(HL);
}

/* A10 ParticleFilterTracker::update (same file): rep = rep + p * p.weight, sequential;
 * operator*(ParticleXYZRPY, double) multiplies in double and casts each component to float,
 * operator+ adds floats. */
void orc_weighted_mean(const orc_particle_t* p, size_t n, orc_particle_t* rep) {
  orc_particle_t r;
  memset(&r, 0, sizeof(r));
  r.w = 1.0f;
  for (size_t i = 0; i < n; i++) {
    double v = (double)p[i].weight;
    r.x = r.x + (float)(p[i].x * v);
    r.y = r.y + (float)(p[i].y * v);
    r.z = r.z + (float)(p[i].z * v);
    r.roll = r.roll + (float)(p[i].roll * v);
    r.pitch = r.pitch + (float)(p[i].pitch * v);
    r.yaw = r.yaw + (float)(p[i].yaw * v);
  }
  r.weight = 1.0f / (float)n;
  *rep = r;
}

/* ------------------------------------------------------------------------------------------- */
/* RNG specification (replaces PCL's two function-local time(0)-seeded boost::mt19937 engines,  */
/* tracking/src/tracking.cpp sampleNormal + particle_filter.hpp sampleWithReplacement).          */
/* Philox4x32-10, key = seed, counter = (global particle id, slot, epoch, purpose).              */
/* ------------------------------------------------------------------------------------------- */
void orc_philox4x32(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; r++) {
    if (r > 0) {
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double orc_u53(uint32_t a, uint32_t b) {
  uint64_t m = ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
  return (double)m * (1.0 / 9007199254740992.0);
}

static void rng_words(uint64_t seed, uint32_t pid, uint32_t slot, uint32_t epoch, uint32_t purpose, uint32_t o[4]) {
  uint32_t ctr[4] = {pid, slot, epoch, purpose};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  orc_philox4x32(ctr, key, o);
}

double orc_rng_uniform(uint64_t seed, uint32_t pid, uint32_t slot, uint32_t epoch, uint32_t purpose) {
  uint32_t o[4];
  rng_words(seed, pid, slot, epoch, purpose, o);
  return orc_u53(o[0], o[1]);
}

void orc_rng_normal_pair(uint64_t seed, uint32_t pid, uint32_t slot, uint32_t epoch, uint32_t purpose, double* z0,
                         double* z1) {
  uint32_t o[4];
  rng_words(seed, pid, slot, epoch, purpose, o);
  double u1 = 1.0 - orc_u53(o[0], o[1]); /* (0,1] */
  double u2 = orc_u53(o[2], o[3]);
  double r = sqrt(-2.0 * log(u1));
  double th = 6.283185307179586 * u2;
  *z0 = r * cos(th);
  *z1 = r * sin(th);
}

/* ParticleXYZRPY::sample(mean, cov): x += (float) sampleNormal(mean[0], cov[0]) ... in the order
 * x,y,z,roll,pitch,yaw; sampleNormal(mean, cov) draws N(mean, sigma = sqrt(cov)) in double. */
static void particle_sample(orc_particle_t* p, const double mean[6], const double cov[6], uint64_t seed,
                            uint32_t pid, uint32_t epoch, uint32_t purpose) {
  double z[6];
  orc_rng_normal_pair(seed, pid, 1, epoch, purpose, &z[0], &z[1]);
  orc_rng_normal_pair(seed, pid, 2, epoch, purpose, &z[2], &z[3]);
  orc_rng_normal_pair(seed, pid, 3, epoch, purpose, &z[4], &z[5]);
  p->x += (float)(z[0] * sqrt(cov[0]) + mean[0]);
  p->y += (float)(z[1] * sqrt(cov[1]) + mean[1]);
  p->z += (float)(z[2] * sqrt(cov[2]) + mean[2]);
  p->roll += (float)(z[3] * sqrt(cov[3]) + mean[3]);
  p->pitch += (float)(z[4] * sqrt(cov[4]) + mean[4]);
  p->yaw += (float)(z[5] * sqrt(cov[5]) + mean[5]);
}

/* A0 ParticleFilterTracker::initParticles(true) (particle_filter.hpp) */
void orc_init_particles(const orc_config_t* c, const orc_particle_t* rep, uint32_t id_offset, size_t n_local,
                        orc_particle_t* out) {
  for (size_t i = 0; i < n_local; i++) {
    orc_particle_t p;
    memset(&p, 0, sizeof(p));
    p.w = 1.0f;
    particle_sample(&p, c->init_mean, c->init_cov, c->seed, id_offset + (uint32_t)i, 0, 0);
    p.x = p.x + rep->x; p.y = p.y + rep->y; p.z = p.z + rep->z;
    p.roll = p.roll + rep->roll; p.pitch = p.pitch + rep->pitch; p.yaw = p.yaw + rep->yaw;
    p.weight = 1.0f / (float)c->particle_num;
    out[i] = p;
  }
}

/* A11 resampleWithReplacement + sampleWithReplacement, INTENDED semantics:
 * true snapshot of the old particles, u scaled by P, exactly P particles, motion_num == 0. */
void orc_resample(const orc_config_t* c, const orc_particle_t* old, size_t n_total, const int32_t* a,
                  const double* q, const orc_particle_t* rep, uint32_t epoch, uint32_t id_offset, size_t n_local,
                  orc_particle_t* out) {
  static const double zero_mean[6] = {0, 0, 0, 0, 0, 0};
  for (size_t li = 0; li < n_local; li++) {
    uint32_t g = id_offset + (uint32_t)li;
    if (g == 0) {
      out[li] = *rep;
      continue;
    }
    double rU = orc_rng_uniform(c->seed, g, 0, epoch, 1) * (double)n_total;
    int k = (int)rU;
    rU -= k;
    int target = (rU < q[k]) ? k : a[k];
    orc_particle_t p = old[target];
    particle_sample(&p, zero_mean, c->step_cov, c->seed, g, epoch, 1);
    out[li] = p;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* KLD-adaptive resampling (PCL 1.8.0 tracking/kld_adaptive_particle_filter.h, impl/...hpp)      */
/* ------------------------------------------------------------------------------------------- */
double orc_kld_normal_quantile(double u) {
  /* CACM Algorithm 209 "Gauss" coefficients, as upstream (the last-but-one b coefficient is quoted from the
   * algorithm, 5.35310849e-4; an upstream typo there would move z in the 10th digit only) */
  static const double a[9] = {1.24818987e-4, -1.075204047e-3, 5.198775019e-3, -0.019198292004, 0.059054035642,
                              -0.151968751364, 0.319152932694, -0.5319230073, 0.797884560593};
  static const double b[15] = {-4.5255659e-5, 1.5252929e-4, -1.9538132e-5, -6.76904986e-4, 1.390604284e-3,
                               -7.9462082e-4, -2.034254874e-3, 6.549791214e-3, -0.010557625006, 0.011630447319,
                               -9.279453341e-3, 5.353579108e-3, -2.141268741e-3, 5.35310849e-4, 9.99936657524e-1};
  double w, y, z;
  if (u == 0.) return 0.5;
  y = u / 2.0;
  if (y < -6.) return 0.0;
  if (y > 6.) return 1.0;
  if (y < 0.) y = -y;
  if (y < 1.) {
    w = y * y;
    z = a[0];
    for (int i = 1; i < 9; i++) z = z * w + a[i];
    z *= (y * 2.0);
  } else {
    y -= 2.0;
    z = b[0];
    for (int i = 1; i < 15; i++) z = z * y + b[i];
  }
  if (u < 0.0) return (1. - z) / 2.0;
  return (1. + z) / 2.0;
}

double orc_kld_bound(int k, double delta, double epsilon) {
  double z = orc_kld_normal_quantile(delta);
  double chi = 1.0 - 2.0 / (9.0 * (k - 1)) + sqrt(2.0 / (9.0 * (k - 1))) * z;
  return ((k - 1.0) / 2.0 / epsilon) * chi * chi * chi;
}

/* do { j = sampleWithReplacement(a, q); x = particles[j]; x.sample(0, step_cov);
 *      if (rand()/RAND_MAX < motion_ratio) x = x + motion;  S.push_back(x);
 *      bin[i] = (int)(x[i] / bin_size[i]);  if (insertIntoBins(bin, B)) ++k;  ++n;
 * } while (n < max && (k < 2 || n < calcKLBound(k)));
 * RNG: sample n draws from the Philox stream of "particle id" n with purpose 2: slot 0 the alias draw, slots
 * 1-3 the step noise, slot 4 the motion coin (upstream: boost mt19937 + C rand(), unseedable). */
size_t orc_kld_resample(const orc_config_t* c, const orc_particle_t* old, size_t n_old, const int32_t* a,
                        const double* q, const orc_particle_t* motion, uint32_t epoch, orc_particle_t* out,
                        int32_t* bins_out, int32_t* k_out) {
  static const double zero_mean[6] = {0, 0, 0, 0, 0, 0};
  const unsigned maxn = (unsigned)c->kld_max_particles;
  int32_t* B = (int32_t*)malloc(sizeof(int32_t) * 6 * (maxn ? maxn : 1));
  unsigned k = 0, n = 0;
  float bs[6];
  for (int i = 0; i < 6; i++) bs[i] = (float)c->kld_bin_size[i];
  do {
    double rU = orc_rng_uniform(c->seed, n, 0, epoch, 2) * (double)n_old;
    int kk = (int)rU;
    rU -= kk;
    int j_n = (rU < q[kk]) ? kk : a[kk];
    orc_particle_t x = old[j_n];
    particle_sample(&x, zero_mean, c->step_cov, c->seed, n, epoch, 2);
    if (orc_rng_uniform(c->seed, n, 4, epoch, 2) < c->motion_ratio) { /* StateT operator+: the six pose floats */
      x.x = x.x + motion->x; x.y = x.y + motion->y; x.z = x.z + motion->z;
      x.roll = x.roll + motion->roll; x.pitch = x.pitch + motion->pitch; x.yaw = x.yaw + motion->yaw;
    }
    out[n] = x;
    const float v[6] = {x.x, x.y, x.z, x.roll, x.pitch, x.yaw};
    int32_t bin[6];
    for (int i = 0; i < 6; i++) bin[i] = (int32_t)(v[i] / bs[i]);
    if (bins_out) memcpy(bins_out + 6 * n, bin, sizeof(bin));
    int found = 0;
    for (unsigned m = 0; m < k && !found; m++) found = memcmp(B + 6 * m, bin, sizeof(bin)) == 0;
    if (!found) {
      memcpy(B + 6 * k, bin, sizeof(bin));
      ++k;
    }
    ++n;
  } while (n < maxn && (k < 2 || (double)n < orc_kld_bound((int)k, c->kld_delta, c->kld_epsilon)));
  free(B);
  if (k_out) *k_out = (int32_t)k;
  return n;
}

/* ------------------------------------------------------------------------------------------- */
/* The tracker: ParticleFilterOMPTracker (PCL 1.8.0 tracking/impl/particle_filter_omp.hpp,       */
/* particle_filter.hpp, tracker.hpp)                                                             */
/* ------------------------------------------------------------------------------------------- */
struct orc_tracker {
  orc_config_t cfg;
  orc_point_t* ref;
  size_t M;
  const orc_point_t* input;
  size_t N;
  float trans[16];
  orc_particle_t* particles;
  size_t P; /* 0 until initParticles */
  orc_particle_t rep;
  orc_particle_t motion;
  int changed;
  uint32_t resample_epoch;
  double fit_ratio;
  orc_point_t* transed; /* transed_reference_vector_: P clouds of M points */
  size_t transed_cap;
  double stage[7];
  const float* mat_override; /* tests: P row-major 4x4 matrices used instead of toEigenMatrix(p) */
  const double* bbox_override; /* tests: x_min,x_max,y_min,y_max,z_min,z_max used instead of calcBoundingBox */
  int bbox_only;               /* tests: stop after calcBoundingBox */
  int trig_mode;               /* tests: 0 = cosf / sinf as PCL (default), 1 = double sin / cos rounded to float */
  int sum_mode;                /* tests: 0 = PCL's sequential sums (default), 1 = the device's adjacent-pair tree sums */
};

orc_tracker_t* orc_tracker_create(const orc_config_t* c) {
  orc_tracker_t* t = (orc_tracker_t*)calloc(1, sizeof(*t));
  t->cfg = *c;
  for (int i = 0; i < 16; i++) t->trans[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return t;
}

void orc_tracker_set_trig_mode(orc_tracker_t* t, int mode) { t->trig_mode = mode; }
void orc_tracker_set_sum_mode(orc_tracker_t* t, int mode) { t->sum_mode = mode; }

void orc_tracker_destroy(orc_tracker_t* t) {
  if (!t) return;
  free(t->ref);
  free(t->particles);
  free(t->transed);
  free(t);
}

int orc_tracker_set_reference(orc_tracker_t* t, const orc_point_t* pts, size_t n) {
  free(t->ref);
  t->ref = (orc_point_t*)malloc(sizeof(orc_point_t) * (n ? n : 1));
  if (n) memcpy(t->ref, pts, sizeof(orc_point_t) * n);
  t->M = n;
  return 0;
}

int orc_tracker_set_trans(orc_tracker_t* t, const float m[16]) {
  memcpy(t->trans, m, sizeof(float) * 16);
  return 0;
}

int orc_tracker_set_input(orc_tracker_t* t, const orc_point_t* pts, size_t n) {
  t->input = pts;
  t->N = n;
  return 0;
}

void orc_tracker_get_result(const orc_tracker_t* t, orc_particle_t* out) { *out = t->rep; }

size_t orc_tracker_get_particles(const orc_tracker_t* t, orc_particle_t* out, size_t cap) {
  size_t n = t->P < cap ? t->P : cap;
  if (out && n) memcpy(out, t->particles, n * sizeof(orc_particle_t));
  return t->P;
}

int orc_tracker_set_particles(orc_tracker_t* t, const orc_particle_t* p, size_t n) {
  free(t->particles);
  t->particles = (orc_particle_t*)malloc(sizeof(orc_particle_t) * (n ? n : 1));
  memcpy(t->particles, p, sizeof(orc_particle_t) * n);
  t->P = n;
  return 0;
}

double orc_tracker_fit_ratio(const orc_tracker_t* t) { return t->fit_ratio; }
void orc_tracker_set_matrix_override(orc_tracker_t* t, const float* m16) { t->mat_override = m16; }
void orc_tracker_set_bbox_override(orc_tracker_t* t, const double* bbox6) { t->bbox_override = bbox6; }
void orc_tracker_set_bbox_only(orc_tracker_t* t, int on) { t->bbox_only = on; }
void orc_tracker_stage_times(const orc_tracker_t* t, double s[7]) { memcpy(s, t->stage, sizeof(double) * 7); }

static int n_threads(const orc_tracker_t* t) {
#ifdef _OPENMP
  return t->cfg.threads > 0 ? t->cfg.threads : omp_get_max_threads();
#else
  (void)t;
  return 1;
#endif
}

/* pcl::PassThrough::applyFilterIndices on one field (PCL 1.8.0 filters/impl/passthrough.hpp):
 * drop non-finite points, keep  min <= v <= max  (limits are floats), order preserved. */
static size_t pass_through(const orc_point_t* in, const int32_t* in_idx, size_t n, int field, float lo, float hi,
                           orc_point_t* out, int32_t* out_idx) {
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    const orc_point_t* p = &in[i];
    if (!isfinite(p->x) || !isfinite(p->y) || !isfinite(p->z)) continue;
    float v = field == 0 ? p->x : (field == 1 ? p->y : p->z);
    if (!isfinite(v)) continue;
    if (v < lo || v > hi) continue;
    out[m] = *p;
    out_idx[m] = in_idx ? in_idx[i] : (int32_t)i;
    m++;
  }
  return m;
}

/* ParticleFilterOMPTracker::weight() without normalizeWeight(): A2, A3, A4, A5, A7 */
size_t orc_tracker_eval_weights(orc_tracker_t* t, const orc_particle_t* particles, size_t P, float* raw_w,
                                int32_t* nn_idx, float* nn_d2, int32_t* crop_idx, size_t crop_cap, double bbox[6],
                                int* octree_depth, double octree_bounds[6], uint64_t* scan_queries,
                                uint64_t* scan_points) {
  const size_t M = t->M;
  const int nt = n_threads(t);
  (void)nt;
  double t0 = omp_get_wtime();
  if (t->transed_cap < P * M) {
    free(t->transed);
    t->transed = (orc_point_t*)malloc(sizeof(orc_point_t) * ((P * M) > 0 ? P * M : 1));
    t->transed_cap = P * M;
  }
  /* A2: computeTransformedPointCloudWithoutNormal for every particle (OMP loop 1) */
#pragma omp parallel for num_threads(nt) schedule(static)
  for (long i = 0; i < (long)P; i++) {
    float T[16];
    const orc_particle_t* p = &particles[i];
    if (t->mat_override)
      memcpy(T, t->mat_override + 16 * (size_t)i, sizeof(T));
    else if (t->trig_mode == 1)
      get_transformation_double_trig(p->x, p->y, p->z, p->roll, p->pitch, p->yaw, T);
    else
      orc_get_transformation(p->x, p->y, p->z, p->roll, p->pitch, p->yaw, T);
    orc_transform_cloud(t->ref, M, T, t->transed + (size_t)i * M);
  }
  double t1 = omp_get_wtime();
  /* A3: calcBoundingBox: pcl::getMinMax3D per particle cloud (float), folded into doubles */
  double x_min = DBL_MAX, y_min = DBL_MAX, z_min = DBL_MAX;
  double x_max = -DBL_MAX, y_max = -DBL_MAX, z_max = -DBL_MAX;
  for (size_t i = 0; i < P; i++) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const orc_point_t* c = t->transed + i * M;
    for (size_t j = 0; j < M; j++) {
      mn[0] = c[j].x < mn[0] ? c[j].x : mn[0]; mx[0] = c[j].x > mx[0] ? c[j].x : mx[0];
      mn[1] = c[j].y < mn[1] ? c[j].y : mn[1]; mx[1] = c[j].y > mx[1] ? c[j].y : mx[1];
      mn[2] = c[j].z < mn[2] ? c[j].z : mn[2]; mx[2] = c[j].z > mx[2] ? c[j].z : mx[2];
    }
    if (x_min > mn[0]) x_min = mn[0];
    if (x_max < mx[0]) x_max = mx[0];
    if (y_min > mn[1]) y_min = mn[1];
    if (y_max < mx[1]) y_max = mx[1];
    if (z_min > mn[2]) z_min = mn[2];
    if (z_max < mx[2]) z_max = mx[2];
  }
  if (bbox) {
    bbox[0] = x_min; bbox[1] = x_max; bbox[2] = y_min; bbox[3] = y_max; bbox[4] = z_min; bbox[5] = z_max;
  }
  if (t->bbox_only) return 0;
  if (t->bbox_override) { /* tests of the sharded path: the box over ALL ranks' particles */
    x_min = t->bbox_override[0]; x_max = t->bbox_override[1]; y_min = t->bbox_override[2];
    y_max = t->bbox_override[3]; z_min = t->bbox_override[4]; z_max = t->bbox_override[5];
  }
  /* A4: cropInputPointCloud: PassThrough x, then y, then z */
  size_t N = t->N;
  orc_point_t* cx = (orc_point_t*)malloc(sizeof(orc_point_t) * (N ? N : 1));
  orc_point_t* cy = (orc_point_t*)malloc(sizeof(orc_point_t) * (N ? N : 1));
  orc_point_t* cz = (orc_point_t*)malloc(sizeof(orc_point_t) * (N ? N : 1));
  int32_t* ix = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
  int32_t* iy = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
  int32_t* iz = (int32_t*)malloc(sizeof(int32_t) * (N ? N : 1));
  size_t nx = pass_through(t->input, NULL, N, 0, (float)x_min, (float)x_max, cx, ix);
  size_t ny = pass_through(cx, ix, nx, 1, (float)y_min, (float)y_max, cy, iy);
  size_t nc = pass_through(cy, iy, ny, 2, (float)z_min, (float)z_max, cz, iz);
  if (crop_idx)
    for (size_t i = 0; i < nc && i < crop_cap; i++) crop_idx[i] = iz[i];
  double t2 = omp_get_wtime();
  /* A5: coherence_->setTargetCloud(cropped); initCompute() -> search::Octree(0.01).setInputCloud */
  /* (NearestPairPointCloudCoherence, cfg.exact_nearest: the search structure is irrelevant to the result; the true
   * nearest neighbour is found by exhaustive search below) */
  const int exact = t->cfg.exact_nearest;
  orc_octree_t* tree = orc_octree_build(cz, exact ? 0 : nc, t->cfg.octree_resolution, t->cfg.emulate_pcl_alloc);
  if (octree_depth || octree_bounds) orc_octree_info(tree, octree_depth, octree_bounds, NULL, NULL);
  double t3 = omp_get_wtime();
  /* A7: ApproxNearestPairPointCloudCoherence::computeCoherence per particle (OMP loop 2) */
  const double maxd2 = t->cfg.max_distance * t->cfg.max_distance;
  uint64_t tot_q = 0, tot_s = 0;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : tot_q, tot_s)
  for (long i = 0; i < (long)P; i++) {
    double val = 0.0;
    const orc_point_t* cloud = t->transed + (size_t)i * M;
    for (size_t j = 0; j < M; j++) {
      int k_index = 0;
      float k_distance = 0.0f;
      orc_point_t input_point = cloud[j];
      int scanned;
      if (exact) {
        /* NearestPairPointCloudCoherence::computeCoherence (tracking/impl/nearest_pair_point_cloud_coherence.hpp):
         * search_->nearestKSearch(input_point, 1, ...): the nearest target point, squared distance in float
         * (pointSquaredDist); equal distances: the lowest index (upstream leaves the order of ties to std::sort) */
        scanned = (int)nc;
        k_distance = INFINITY;
        for (size_t c = 0; c < nc; c++) {
          const float dd = point_sq_dist(cz[c].x, cz[c].y, cz[c].z, input_point.x, input_point.y, input_point.z);
          if (dd < k_distance) {
            k_distance = dd;
            k_index = (int)c;
          }
        }
      } else {
        scanned = orc_octree_approx_nearest(tree, &input_point, &k_index, &k_distance);
      }
      if (!scanned) { /* empty target: PCL asserts (UB in release); defined here as "no correspondence" */
        if (nn_idx) nn_idx[(size_t)i * M + j] = -1;
        if (nn_d2) nn_d2[(size_t)i * M + j] = INFINITY;
        continue;
      }
      tot_q += 1;
      tot_s += (uint64_t)scanned;
      if (nn_idx) nn_idx[(size_t)i * M + j] = k_index;
      if (nn_d2) nn_d2[(size_t)i * M + j] = k_distance;
      if (k_distance < maxd2) {
        orc_point_t target_point = cz[k_index];
        double coherence_val = 1.0;
        coherence_val *= orc_distance_coherence(&t->cfg, &input_point, &target_point);
        coherence_val *= orc_hsv_coherence(&t->cfg, input_point.rgba, target_point.rgba);
        val += coherence_val;
      }
    }
    if (raw_w) raw_w[i] = -(float)val;
  }
  double t4 = omp_get_wtime();
  if (scan_queries) *scan_queries = tot_q;
  if (scan_points) *scan_points = tot_s;
  orc_octree_free(tree);
  free(cx); free(cy); free(cz); free(ix); free(iy); free(iz);
  t->stage[0] += t1 - t0;
  t->stage[1] += t2 - t1;
  t->stage[2] += t3 - t2;
  t->stage[3] += t4 - t3;
  return nc;
}

static void tracker_weight(orc_tracker_t* t) {
  size_t P = t->P;
  float* w = (float*)malloc(sizeof(float) * P);
  orc_tracker_eval_weights(t, t->particles, P, w, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL);
  /* use_change_detector_ == false: changed_ = true after every weight() */
  t->changed = 1;
  double t0 = omp_get_wtime();
  if (t->sum_mode == 1)
    orc_normalize_weights_tree(w, P, t->cfg.alpha, &t->fit_ratio);
  else
    orc_normalize_weights(w, P, t->cfg.alpha, &t->fit_ratio);
  for (size_t i = 0; i < P; i++) t->particles[i].weight = w[i];
  t->stage[4] += omp_get_wtime() - t0;
  free(w);
}

static void tracker_resample(orc_tracker_t* t) {
  double t0 = omp_get_wtime();
  size_t P = t->P;
  int32_t* a = (int32_t*)malloc(sizeof(int32_t) * P);
  double* q = (double*)malloc(sizeof(double) * P);
  float* w = (float*)malloc(sizeof(float) * P);
  for (size_t i = 0; i < P; i++) w[i] = t->particles[i].weight;
  orc_gen_alias_table(w, P, a, q);
  const size_t cap = t->cfg.kld_adaptive ? (size_t)t->cfg.kld_max_particles : P;
  orc_particle_t* np = (orc_particle_t*)malloc(sizeof(orc_particle_t) * (cap ? cap : 1));
  if (t->cfg.kld_adaptive) /* particles_ = S; particle_num_ = S.size() */
    t->P = orc_kld_resample(&t->cfg, t->particles, P, a, q, &t->motion, t->resample_epoch, np, NULL, NULL);
  else
    orc_resample(&t->cfg, t->particles, P, a, q, &t->rep, t->resample_epoch, 0, P, np);
  t->resample_epoch++;
  free(t->particles);
  t->particles = np;
  free(a); free(q); free(w);
  t->stage[5] += omp_get_wtime() - t0;
}

static void tracker_update(orc_tracker_t* t) {
  double t0 = omp_get_wtime();
  orc_particle_t orig = t->rep, r;
  if (t->sum_mode == 1)
    orc_weighted_mean_tree(t->particles, t->P, &r);
  else
    orc_weighted_mean(t->particles, t->P, &r);
  t->rep = r;
  t->motion.x = r.x - orig.x; t->motion.y = r.y - orig.y; t->motion.z = r.z - orig.z;
  t->motion.roll = r.roll - orig.roll; t->motion.pitch = r.pitch - orig.pitch; t->motion.yaw = r.yaw - orig.yaw;
  t->stage[6] += omp_get_wtime() - t0;
}

/* Tracker::compute -> ParticleFilterTracker::initCompute + computeTracking (A12 schedule) */
int orc_tracker_compute(orc_tracker_t* t) {
  memset(t->stage, 0, sizeof(t->stage));
  if (!t->input || t->N == 0) return 1; /* PCL_ERROR + early return, no exception */
  if (t->P == 0) {
    /* initParticles(true) */
    orc_particle_t rep;
    orc_to_state(t->trans, &rep);
    rep.weight = 1.0f / (float)t->cfg.particle_num;
    t->rep = rep;
    t->P = (size_t)t->cfg.particle_num;
    t->particles = (orc_particle_t*)malloc(sizeof(orc_particle_t) * t->P);
    orc_init_particles(&t->cfg, &rep, 0, t->P, t->particles);
  }
  for (int it = 0; it < t->cfg.iteration_num; it++) {
    if (t->changed) tracker_resample(t);
    tracker_weight(t);
    if (t->changed) tracker_update(t);
  }
  return 0;
}
