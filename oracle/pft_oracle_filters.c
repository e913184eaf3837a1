/*
 * pft_oracle_filters.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pft_oracle.h).
 *
 * CPU restatement of the per-frame input filters the reference runs in front of the tracker
 * (SURVEY.md section 8f row 1):
 *   filterPassThrough   /root/reference/src/auto_tracking.cpp:536-547  (z in [0, 10], keep_organized = false)
 *   gridSampleApprox    /root/reference/src/auto_tracking.cpp:563-575  (ApproximateVoxelGrid, leaf 0.01)
 *   gridSample          /root/reference/src/auto_tracking.cpp:549-561  (VoxelGrid, leaf 0.01; model + wait frames)
 *
 * PARITY UNPINNED: PCL 1.8.0 is absent (pft_oracle.h); these functions restate the published
 * PCL 1.8.0 sources named at each function and are pinned by hand-derived known answers only.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pft_oracle.h"

/* static_cast<int>(floor(v)) as an x86-64 PCL build evaluates it: cvttss2si/cvttsd2si return INT_MIN
 * ("integer indefinite") for NaN and out-of-range values, which C leaves undefined. */
static int floor_to_int(float v) {
  const float f = floorf(v);
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT_MIN;
  return (int)f;
}

/* pcl::PassThrough<PointT>::applyFilterIndices with a field name set, keep_organized = false
 * (PCL 1.8.0 filters/include/pcl/filters/impl/passthrough.hpp): points with a non-finite x, y or z are
 * always removed; then the field value must be finite and, unless `negative`, inside [lo, hi] inclusive
 * (with `negative`: strictly outside).  Stable.  field: 0 = x, 1 = y, 2 = z.  Returns the number kept. */
size_t orc_pass_through(const orc_point_t* pts, size_t n, int field, float lo, float hi, int negative,
                        int32_t* out_idx) {
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    const orc_point_t* p = &pts[i];
    if (!isfinite(p->x) || !isfinite(p->y) || !isfinite(p->z)) continue;
    const float v = field == 0 ? p->x : field == 1 ? p->y : p->z;
    if (!isfinite(v)) continue;
    if (!negative && (v < lo || v > hi)) continue;
    if (negative && v >= lo && v <= hi) continue;
    out_idx[o++] = (int32_t)i;
  }
  return o;
}

/* pcl::ApproximateVoxelGrid<PointXYZRGBA>::applyFilter + flush
 * (PCL 1.8.0 filters/include/pcl/filters/impl/approximate_voxel_grid.hpp), downsample_all_data_ = true:
 * a `hist_size`-entry history table indexed by (ix*7171 + iy*3079 + iz*4231) & (hist_size-1) holds one open
 * voxel per entry; a point of another voxel hashing to an occupied entry flushes that entry's centroid to the
 * output first; entries still open at the end are flushed in table order.  The centroid vector is
 * [x, y, z, (float) rgba, r, g, b] summed in float in arrival order and divided by (float) count; the output
 * colour is (int) r << 16 | (int) g << 8 | (int) b (alpha byte 0), data[3] stays 1.0f. */
typedef struct {
  int ix, iy, iz, count;
  float c[7];
} ahe_t;

static void avg_flush(orc_point_t* out, ahe_t* h) {
  const float cnt = (float)h->count;
  float c[7];
  for (int k = 0; k < 7; k++) c[k] = h->c[k] / cnt;
  memset(out, 0, sizeof(*out));
  out->x = c[0];
  out->y = c[1];
  out->z = c[2];
  out->w = 1.0f;
  /* the generic field copy writes (uint32_t) c[3] into rgba; the RGB special case then overwrites all 4 bytes */
  const int rgb = ((int)c[4]) << 16 | ((int)c[5]) << 8 | ((int)c[6]);
  memcpy(&out->rgba, &rgb, 4);
}

size_t orc_approx_voxel_grid(const orc_point_t* pts, size_t n, const float leaf[3], uint32_t hist_size,
                             orc_point_t* out) {
  if (hist_size == 0 || (hist_size & (hist_size - 1))) return 0;
  /* setLeafSize: inverse_leaf_size_ = Eigen::Array3f::Ones() / leaf_size_.array() */
  const float inv[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};
  ahe_t* hist = (ahe_t*)calloc(hist_size, sizeof(ahe_t));
  size_t op = 0;
  for (size_t cp = 0; cp < n; cp++) {
    const orc_point_t* p = &pts[cp];
    const int ix = floor_to_int(p->x * inv[0]);
    const int iy = floor_to_int(p->y * inv[1]);
    const int iz = floor_to_int(p->z * inv[2]);
    const uint32_t hash =
        ((uint32_t)ix * 7171u + (uint32_t)iy * 3079u + (uint32_t)iz * 4231u) & (hist_size - 1);
    ahe_t* h = &hist[hash];
    if (h->count && (ix != h->ix || iy != h->iy || iz != h->iz)) {
      avg_flush(&out[op++], h);
      h->count = 0;
      for (int k = 0; k < 7; k++) h->c[k] = 0.0f;
    }
    h->ix = ix;
    h->iy = iy;
    h->iz = iz;
    h->count++;
    const float s[7] = {p->x, p->y, p->z, (float)p->rgba, (float)((p->rgba >> 16) & 255u),
                        (float)((p->rgba >> 8) & 255u), (float)(p->rgba & 255u)};
    for (int k = 0; k < 7; k++) h->c[k] += s[k];
  }
  for (uint32_t i = 0; i < hist_size; i++)
    if (hist[i].count) avg_flush(&out[op++], &hist[i]);
  free(hist);
  return op;
}

/* pcl::VoxelGrid<PointXYZRGBA>::applyFilter (PCL 1.8.0 filters/include/pcl/filters/impl/voxel_grid.hpp),
 * no filter field, downsample_all_data_ = true, min_points_per_voxel_ = 0: bounds of the finite points
 * (getMinMax3D), voxel index idx = ijk0 + ijk1*div0 + ijk2*div0*div1, points sorted by idx, one
 * CentroidPoint per voxel (xyz: float sum / (float) n; rgba: float sums of a, r, g, b, each / n, truncated).
 * PCL sorts with std::sort (not stable): the order of the points INSIDE a voxel -- hence the last bits of
 * the float sums -- is whatever libstdc++'s introsort leaves; this restatement sums in input order.
 * Returns the number of output points, or (size_t)-1 when PCL would give up ("leaf size too small"). */
typedef struct {
  uint32_t idx, pt;
} vg_pair_t;

static int vg_cmp(const void* a, const void* b) {
  const vg_pair_t* x = (const vg_pair_t*)a;
  const vg_pair_t* y = (const vg_pair_t*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt);
}

size_t orc_voxel_grid(const orc_point_t* pts, size_t n, const float leaf[3], orc_point_t* out) {
  const float inv[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  size_t nf = 0;
  for (size_t i = 0; i < n; i++) {
    const float v[3] = {pts[i].x, pts[i].y, pts[i].z};
    if (!isfinite(v[0]) || !isfinite(v[1]) || !isfinite(v[2])) continue;
    nf++;
    for (int k = 0; k < 3; k++) {
      if (v[k] < mn[k]) mn[k] = v[k];
      if (v[k] > mx[k]) mx[k] = v[k];
    }
  }
  if (!nf) return 0;
  int64_t d[3];
  for (int k = 0; k < 3; k++) d[k] = (int64_t)((mx[k] - mn[k]) * inv[k]) + 1;
  if (d[0] * d[1] * d[2] > (int64_t)INT32_MAX) return (size_t)-1;
  int minb[3], maxb[3], div[3];
  for (int k = 0; k < 3; k++) {
    minb[k] = floor_to_int(mn[k] * inv[k]);
    maxb[k] = floor_to_int(mx[k] * inv[k]);
    div[k] = maxb[k] - minb[k] + 1;
  }
  const int mul[3] = {1, div[0], div[0] * div[1]};
  vg_pair_t* iv = (vg_pair_t*)malloc(sizeof(vg_pair_t) * nf);
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    const float v[3] = {pts[i].x, pts[i].y, pts[i].z};
    if (!isfinite(v[0]) || !isfinite(v[1]) || !isfinite(v[2])) continue;
    int ijk[3];
    for (int k = 0; k < 3; k++) ijk[k] = (int)(floorf(v[k] * inv[k]) - (float)minb[k]);
    iv[m].idx = (uint32_t)(ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2]);
    iv[m].pt = (uint32_t)i;
    m++;
  }
  qsort(iv, m, sizeof(vg_pair_t), vg_cmp);
  size_t o = 0, a = 0;
  while (a < m) {
    size_t b = a + 1;
    while (b < m && iv[b].idx == iv[a].idx) b++;
    float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0, sa = 0;
    for (size_t j = a; j < b; j++) {
      const orc_point_t* p = &pts[iv[j].pt];
      sx += p->x;
      sy += p->y;
      sz += p->z;
      sr += (float)((p->rgba >> 16) & 255u);
      sg += (float)((p->rgba >> 8) & 255u);
      sb += (float)(p->rgba & 255u);
      sa += (float)(p->rgba >> 24);
    }
    const float cnt = (float)(b - a);
    orc_point_t* q = &out[o++];
    memset(q, 0, sizeof(*q));
    q->x = sx / cnt;
    q->y = sy / cnt;
    q->z = sz / cnt;
    q->w = 1.0f;
    q->rgba = (uint32_t)(sa / cnt) << 24 | (uint32_t)(sr / cnt) << 16 | (uint32_t)(sg / cnt) << 8 |
              (uint32_t)(sb / cnt);
    a = b;
  }
  free(iv);
  return o;
}
