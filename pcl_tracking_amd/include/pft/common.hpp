// common.hpp -- host-side mirrors of the two pcl::common functions the reference's "set object to track" and result
// consumer steps call around the tracker, in PCL 1.8.0's operation order and precision:
//
//   pcl::compute3DCentroid<PointT, float>   common/impl/centroid.hpp   /root/reference/src/auto_tracking.cpp:433, :663
//   pcl::transformPointCloud<PointT>        common/impl/transforms.hpp /root/reference/src/auto_tracking.cpp:316, :668
//
// Plain loops over a few thousand model points, once per object (re-centring) and once per frame and object (the
// moved model whose centroid the node publishes): host code in the reference, host code here.  The CPU checker under
// oracle/ restates both, and tests/test_cpp_host.py compares the C++ driver with it bit for bit.
#pragma once
#include <cmath>

#include "pft/particle_filter_tracker.hpp"

namespace pft {

// centroid[0..2] = sum of the finite points' coordinates, accumulated in float in index order, divided by their
// number; centroid[3] = 1.  Returns the number of points used (0: centroid untouched), as upstream.
template <typename PointT>
inline unsigned int compute3DCentroid(const PointCloud<PointT>& cloud, float centroid[4]) {
  if (cloud.points.empty()) return 0;
  float c[3] = {0.0f, 0.0f, 0.0f};
  unsigned int cp = 0;
  for (size_t i = 0; i < cloud.points.size(); i++) {
    const PointT& p = cloud.points[i];
    if (!cloud.is_dense && !(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
    c[0] += p.x;
    c[1] += p.y;
    c[2] += p.z;
    cp++;
  }
  if (!cp) return 0;
  centroid[0] = c[0] / static_cast<float>(cp);
  centroid[1] = c[1] / static_cast<float>(cp);
  centroid[2] = c[2] / static_cast<float>(cp);
  centroid[3] = 1.0f;
  return cp;
}

// every field copied, then x' = ((T00 x + T01 y) + T02 z) + T03 in float (same for y', z'); points with a non-finite
// coordinate are left as they are when the cloud is not dense
template <typename PointT>
inline void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Affine3f& T) {
  if (&in != &out) {
    out.points = in.points;
    out.width = in.width;
    out.height = in.height;
    out.is_dense = in.is_dense;
  }
  for (size_t i = 0; i < out.points.size(); i++) {
    const float x = in.points[i].x, y = in.points[i].y, z = in.points[i].z;
    if (!in.is_dense && !(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) continue;
    out.points[i].x = T(0, 0) * x + T(0, 1) * y + T(0, 2) * z + T(0, 3);
    out.points[i].y = T(1, 0) * x + T(1, 1) * y + T(1, 2) * z + T(1, 3);
    out.points[i].z = T(2, 0) * x + T(2, 1) * y + T(2, 2) * z + T(2, 3);
  }
}

// Eigen::Affine3f::inverse() of a pure translation, the only transform the reference inverts (:668): linear part the
// identity, translation negated
inline Affine3f inverseOfTranslation(const Affine3f& t) {
  Affine3f r = Affine3f::Identity();
  r(0, 3) = -t(0, 3);
  r(1, 3) = -t(1, 3);
  r(2, 3) = -t(2, 3);
  return r;
}

}  // namespace pft
