// tracking_app.hpp -- what /root/reference/src/auto_tracking.cpp's OpenNISegmentTracking does around its trackers, without
// ROS / VTK, for any number of objects (shared by auto_tracking_amd.cpp and dist_tracking_amd.cpp):
//
//   buildTrackers()      one tracker per object, configured as initialize_trackers() does        :181-259
//   setObjectsToTrack()  the frame-#2 step: removeZeroPoints, centroid, re-centre, gridSample,
//                        setReferenceCloud / setTrans / setMinIndices                            :577-595, :646-677
//   objectPosition()     drawResult + viz_cb: the full-resolution model moved by the result pose
//                        (5 mm towards the camera "for better visualization") and its centroid,
//                        which the node publishes as the object's position                       :300-326, :432-466
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "pft/common.hpp"
#include "pft/filters.hpp"
#include "pft/pcd_io.hpp"
#include "pft/particle_filter_tracker.hpp"

namespace app {

using namespace pft;
using namespace pft::tracking;

typedef PointXYZRGBA RefPointType;
typedef ParticleXYZRPY ParticleT;
typedef PointCloud<RefPointType> Cloud;
typedef ParticleFilterTracker<RefPointType, ParticleT> ParticleFilter;

struct Options {
  int particles = 400;                  // :231
  uint64_t seed = 1;                    // PCL's engines are time(0)-seeded; object k uses seed + k
  bool use_fixed = true;                // the reference's main() passes false (:821): the KLD-adaptive tracker
  double downsampling_grid_size = 0.01; // :824; 0 = the model is used as given
  unsigned threads = 16;                // :845, meaningless on the GPU
};

// *.pcd = PCD v0.7 with fields x y z rgba (create_model.cpp:219-222 writes them, :741 once loaded them);
// anything else = a raw array of 32-byte pcl::PointXYZRGBA records
inline Cloud::Ptr loadCloud(const char* path) {
  Cloud::Ptr c(new Cloud());
  const size_t len = std::strlen(path);
  if (len > 4 && !std::strcmp(path + len - 4, ".pcd")) {
    if (pft::io::loadPCDFile(path, *c) == -1) {
      std::fprintf(stderr, "pcd file not found or not readable: %s\n", path);
      c->points.clear();
    }
    return c;
  }
  FILE* f = std::fopen(path, "rb");
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    return c;
  }
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  c->points.resize((size_t)sz / sizeof(RefPointType));
  if (std::fread(c->points.data(), sizeof(RefPointType), c->points.size(), f) != c->points.size()) c->points.clear();
  std::fclose(f);
  c->width = (uint32_t)c->points.size();
  return c;
}

// keeps the points that are neither NaN nor within 1 cm of the sensor origin on all three axes (:577-595)
inline void removeZeroPoints(const Cloud& cloud, Cloud& result) {
  result.points.clear();
  for (const RefPointType& p : cloud.points) {
    const bool at_origin = std::fabs(p.x) < 0.01 && std::fabs(p.y) < 0.01 && std::fabs(p.z) < 0.01;
    if (!at_origin && !std::isnan(p.x) && !std::isnan(p.y) && !std::isnan(p.z)) result.points.push_back(p);
  }
  result.width = (uint32_t)result.points.size();
  result.height = 1;
  result.is_dense = true;
}

class TrackingApp {
 public:
  explicit TrackingApp(const Options& o) : opt_(o) {}

  std::map<int, std::shared_ptr<ParticleFilter>> tracker_dict;   // :153
  std::map<int, Cloud::Ptr> ref_cloud_dict;                      // the segmented object clusters, camera frame
  std::map<int, Cloud::Ptr> reference_dict;                      // re-centred, full resolution (:675)
  std::map<int, Cloud::Ptr> tracked_cloud_dict;                  // :325

  // one tracker per object, parameters of auto_tracking.cpp:187-254; `configure` lets a caller add what the reference
  // has no notion of (device, stream, shard) before the handle exists
  template <class F>
  void buildTrackers(int nb_objects, F&& configure) {
    std::vector<double> step_cov(6, 0.015 * 0.015);
    for (int k = 3; k < 6; k++) step_cov[k] *= 40.0;
    const std::vector<double> init_cov(6, 0.00001), init_mean(6, 0.0);
    for (int obj_id = 0; obj_id < nb_objects; obj_id++) {
      std::shared_ptr<ParticleFilter> tr;
      if (opt_.use_fixed) {
        tr.reset(new ParticleFilterOMPTracker<RefPointType, ParticleT>(opt_.threads));
      } else {
        auto* kld = new KLDAdaptiveParticleFilterOMPTracker<RefPointType, ParticleT>(opt_.threads);
        kld->setMaximumParticleNum(500);
        kld->setDelta(0.99);
        kld->setEpsilon(0.2);
        ParticleT bin;
        bin.x = bin.y = bin.z = bin.roll = bin.pitch = bin.yaw = 0.1f;
        kld->setBinSize(bin);
        tr.reset(kld);
      }
      tr->setTrans(Affine3f::Identity());
      tr->setStepNoiseCovariance(step_cov);
      tr->setInitialNoiseCovariance(init_cov);
      tr->setInitialNoiseMean(init_mean);
      tr->setIterationNum(2);
      tr->setParticleNum(opt_.particles);
      tr->setResampleLikelihoodThr(0.00);
      tr->setUseNormal(false);
      tr->setSeed(opt_.seed + (uint64_t)obj_id);
      ApproxNearestPairPointCloudCoherence<RefPointType>::Ptr coherence(new ApproxNearestPairPointCloudCoherence<RefPointType>());
      coherence->addPointCoherence(std::make_shared<DistanceCoherence<RefPointType>>());
      auto color = std::make_shared<HSVColorCoherence<RefPointType>>();
      color->setWeight(0.1);
      coherence->addPointCoherence(color);
      coherence->setSearchMethod(std::make_shared<search::Octree<RefPointType>>(0.01));
      coherence->setMaximumDistance(0.1);
      tr->setCloudCoherence(coherence);
      configure(*tr, obj_id);
      tracker_dict[obj_id] = tr;
    }
  }

  // returns false if an object's model is empty
  bool setObjectsToTrack() {
    for (auto& kv : tracker_dict) {
      const int obj_id = kv.first;
      const Cloud::Ptr ref_cloud = ref_cloud_dict[obj_id];
      Cloud::Ptr nonzero_ref(new Cloud());
      removeZeroPoints(*ref_cloud, *nonzero_ref);
      if (nonzero_ref->empty()) {
        std::fprintf(stderr, "object %d: empty model\n", obj_id);
        return false;
      }
      float c[4] = {0, 0, 0, 1};
      compute3DCentroid(*nonzero_ref, c);  // the object's initial position
      Affine3f trans = Affine3f::Identity();
      trans(0, 3) = c[0];
      trans(1, 3) = c[1];
      trans(2, 3) = c[2];
      Cloud::Ptr transed_ref(new Cloud());
      transformPointCloud(*nonzero_ref, *transed_ref, inverseOfTranslation(trans));
      Cloud::Ptr transed_ref_downsampled(new Cloud());
      if (opt_.downsampling_grid_size > 0) {  // gridSample (:549-561): pcl::VoxelGrid, on the device
        pft::VoxelGrid grid;
        const float leaf = (float)opt_.downsampling_grid_size;
        grid.setLeafSize(leaf, leaf, leaf);
        grid.setInputCloud(transed_ref);
        grid.filter(*transed_ref_downsampled);
      } else {
        *transed_ref_downsampled = *transed_ref;
      }
      std::fprintf(stderr, "object %d ref_cloud: %zu data points, nonzero_ref: %zu, downsampled: %zu\n", obj_id,
                   ref_cloud->points.size(), nonzero_ref->points.size(), transed_ref_downsampled->points.size());
      kv.second->setReferenceCloud(transed_ref_downsampled);
      kv.second->setTrans(trans);
      reference_dict[obj_id] = transed_ref;
      kv.second->setMinIndices((int)ref_cloud->points.size() / 2);
    }
    return true;
  }

  // the tracked cloud of an object (drawResult) and its centroid (viz_cb), from a result pose
  void objectPosition(int obj_id, const ParticleT& result, float centroid[4]) {
    Affine3f transformation = tracker_dict[obj_id]->toEigenMatrix(result);
    transformation(2, 3) += -0.005f;  // "move a little bit for better visualization": the published centroid carries it
    Cloud::Ptr result_cloud(new Cloud());
    transformPointCloud(*reference_dict[obj_id], *result_cloud, transformation);
    tracked_cloud_dict[obj_id] = result_cloud;
    centroid[0] = centroid[1] = centroid[2] = 0.0f;
    centroid[3] = 1.0f;
    compute3DCentroid(*result_cloud, centroid);
  }

  const Options& options() const { return opt_; }

 private:
  Options opt_;
};

// pose and published position of one object for one frame; %.9g round-trips a float, so a test can feed the pose back to the oracle
inline void printObjectLine(size_t frame, int obj_id, const ParticleT& r, const float c[4]) {
  std::printf("frame %zu object %d pose %.9g %.9g %.9g %.9g %.9g %.9g  centroid %.9g %.9g %.9g\n", frame, obj_id, r.x, r.y, r.z,
              r.roll, r.pitch, r.yaw, c[0], c[1], c[2]);
}

}  // namespace app
