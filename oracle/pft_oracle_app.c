/*
 * pft_oracle_app.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pft_oracle.h).
 *
 * CPU restatement of the host-side steps /root/reference/src/auto_tracking.cpp performs around its trackers
 * (SURVEY.md 8f row 3): the model preparation of the "set object to track" step (:646-677) and the result consumer
 * (drawResult :300-326, viz_cb :432-466).  The reference's own code is removeZeroPoints (:577-595); the rest are calls
 * into PCL 1.8.0 common (compute3DCentroid, transformPointCloud) and Eigen, restated here from the published sources.
 * PARITY UNPINNED like the rest of the oracle: checked against hand-derived known answers
 * (tests/test_oracle_app.py); the product's C++ host code (pcl_tracking_amd/include/pft/common.hpp,
 * examples/tracking_app.hpp) is compared with these functions bit for bit.
 */
#include <math.h>
#include <string.h>

#include "pft_oracle.h"

/* auto_tracking.cpp:577-595: drops NaN points and points within 1 cm of the sensor origin on all three axes
 * (fabs() promotes the float to double; 0.01 is a double literal) */
size_t orc_remove_zero_points(const orc_point_t* in, size_t n, orc_point_t* out) {
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    const orc_point_t p = in[i];
    if (!(fabs((double)p.x) < 0.01 && fabs((double)p.y) < 0.01 && fabs((double)p.z) < 0.01) && !isnan(p.x) && !isnan(p.y) &&
        !isnan(p.z))
      out[m++] = p;
  }
  return m;
}

/* pcl::compute3DCentroid<PointT, float> (PCL 1.8.0 common/impl/centroid.hpp): a float accumulator per axis, points added
 * in index order; dense clouds take every point, others skip non-finite ones; centroid /= (float) count; centroid[3] = 1 */
size_t orc_compute_3d_centroid(const orc_point_t* pts, size_t n, int is_dense, float centroid[4]) {
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  size_t cp = 0;
  for (size_t i = 0; i < n; i++) {
    if (!is_dense && !(isfinite(pts[i].x) && isfinite(pts[i].y) && isfinite(pts[i].z))) continue;
    c0 += pts[i].x;
    c1 += pts[i].y;
    c2 += pts[i].z;
    cp++;
  }
  if (!cp) return 0;
  centroid[0] = c0 / (float)cp;
  centroid[1] = c1 / (float)cp;
  centroid[2] = c2 / (float)cp;
  centroid[3] = 1.0f;
  return cp;
}

/* :663-668: trans = Identity with translation = centroid; transformPointCloud(nonzero_ref, transed_ref, trans.inverse()).
 * Eigen's inverse of an affine transform whose linear part is the identity is the identity with the translation negated
 * (-(I * c) is exact), and the transform then evaluates 1*x + 0*y + 0*z + (-c) in float = x - c.  trans16 receives trans. */
void orc_recentre_model(const orc_point_t* in, size_t n, const float centroid[4], orc_point_t* out, float trans16[16]) {
  float inv[16];
  for (int i = 0; i < 16; i++) {
    trans16[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    inv[i] = trans16[i];
  }
  trans16[3] = centroid[0];
  trans16[7] = centroid[1];
  trans16[11] = centroid[2];
  inv[3] = -centroid[0];
  inv[7] = -centroid[1];
  inv[11] = -centroid[2];
  orc_transform_cloud(in, n, inv, out);
}

/* drawResult (:309-316) + viz_cb (:432-433): transformation = toEigenMatrix(result) (pcl::getTransformation, cosf / sinf);
 * translation += (0, 0, -0.005f); tracked cloud = transformPointCloud(reference_dict[obj], transformation); the published
 * position is compute3DCentroid of the tracked cloud.  moved may be NULL. */
void orc_object_position(const orc_point_t* reference_full, size_t n, const orc_particle_t* result, orc_point_t* moved,
                         float centroid[4]) {
  float T[16];
  orc_get_transformation(result->x, result->y, result->z, result->roll, result->pitch, result->yaw, T);
  T[11] += -0.005f;
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  for (size_t i = 0; i < n; i++) {
    orc_point_t q;
    orc_transform_cloud(&reference_full[i], 1, T, &q);
    if (moved) moved[i] = q;
    c0 += q.x;
    c1 += q.y;
    c2 += q.z;
  }
  centroid[0] = centroid[1] = centroid[2] = 0.0f;
  centroid[3] = 1.0f;
  if (n) {
    centroid[0] = c0 / (float)n;
    centroid[1] = c1 / (float)n;
    centroid[2] = c2 / (float)n;
  }
}
