// auto_tracking_amd.cpp -- the tracking part of /root/reference/src/auto_tracking.cpp without ROS / VTK:
// initialize_trackers() (:181-259), the "set object to track" step (:643-677: removeZeroPoints, centroid,
// re-centre, setReferenceCloud, setTrans) and the per-frame loop (:688-697: setInputCloud, compute), followed
// by what drawResult() does with the pose (:309-316: toEigenMatrix(getResult())).
//
//   auto_tracking_amd <model.bin> <frame0.bin> [frame1.bin ...] [--particles N] [--seed S] [--raw] [--kld]
//
// *.pcd = PCD v0.7 ascii / binary with fields x y z rgba (what create_model.cpp:219-222 writes);
// *.bin = raw arrays of 32-byte pcl::PointXYZRGBA records.
// The model is the segmented object cluster in the camera frame.  Without --raw the frames are already
// filtered and downsampled; with --raw they are sensor frames and go through cloud_cb's front end first
// (:637 filterPassThrough, :683 gridSampleApprox) on the device, the result staying in HBM for the tracker.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pft/filters.hpp"
#include "pft/pcd_io.hpp"
#include "pft/particle_filter_tracker.hpp"

using namespace pft;
using namespace pft::tracking;

typedef PointXYZRGBA RefPointType;
typedef ParticleXYZRPY ParticleT;
typedef PointCloud<RefPointType> Cloud;
typedef ParticleFilterTracker<RefPointType, ParticleT> ParticleFilter;

static Cloud::Ptr load_bin(const char* path) {
  Cloud::Ptr c(new Cloud());
  const size_t len = std::strlen(path);
  if (len > 4 && !std::strcmp(path + len - 4, ".pcd")) {  // what create_model.cpp:219-222 writes (:741 loadPCDFile)
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
  long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  c->points.resize((size_t)sz / sizeof(RefPointType));
  if (std::fread(c->points.data(), sizeof(RefPointType), c->points.size(), f) != c->points.size()) c->points.clear();
  std::fclose(f);
  c->width = (uint32_t)c->points.size();
  return c;
}

// auto_tracking.cpp:577-595
static void removeZeroPoints(const Cloud& cloud, Cloud& result) {
  for (size_t i = 0; i < cloud.points.size(); i++) {
    const RefPointType& p = cloud.points[i];
    if (!(std::fabs(p.x) < 0.01 && std::fabs(p.y) < 0.01 && std::fabs(p.z) < 0.01) && !std::isnan(p.x) &&
        !std::isnan(p.y) && !std::isnan(p.z))
      result.points.push_back(p);
  }
  result.width = (uint32_t)result.points.size();
  result.height = 1;
  result.is_dense = true;
}

int main(int argc, char** argv) {
  std::vector<const char*> files;
  int particles = 400;
  uint64_t seed = 1;
  double downsampling_grid_size_ = 0.01;  // :824; --model-leaf 0 skips gridSample of the model
  bool raw = false, use_fixed_ = true;  // the reference defaults to use_fixed = false (:821); --kld selects that branch
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "--raw")) raw = true;
    else if (!std::strcmp(argv[i], "--kld")) use_fixed_ = false;
    else if (!std::strcmp(argv[i], "--model-leaf") && i + 1 < argc) downsampling_grid_size_ = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--particles") && i + 1 < argc) particles = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
    else files.push_back(argv[i]);
  }
  if (files.size() < 2) {
    std::fprintf(stderr, "usage: %s <model.bin> <frame.bin>... [--particles N] [--seed S] [--raw] [--kld] [--model-leaf L]\n", argv[0]);
    return 2;
  }

  // ---- initialize_trackers(), auto_tracking.cpp:187-254 ----
  std::vector<double> default_step_covariance(6, 0.015 * 0.015);
  default_step_covariance[3] *= 40.0;
  default_step_covariance[4] *= 40.0;
  default_step_covariance[5] *= 40.0;
  std::vector<double> initial_noise_covariance(6, 0.00001);
  std::vector<double> default_initial_mean(6, 0.0);

  std::shared_ptr<ParticleFilter> tracker_;
  if (use_fixed_) {
    std::shared_ptr<ParticleFilterOMPTracker<RefPointType, ParticleT>> tracker(
        new ParticleFilterOMPTracker<RefPointType, ParticleT>(16));
    tracker_ = tracker;
  } else {  // :207-222
    std::shared_ptr<KLDAdaptiveParticleFilterOMPTracker<RefPointType, ParticleT>> tracker(
        new KLDAdaptiveParticleFilterOMPTracker<RefPointType, ParticleT>(16));
    tracker->setMaximumParticleNum(500);
    tracker->setDelta(0.99);
    tracker->setEpsilon(0.2);
    ParticleT bin_size;
    bin_size.x = 0.1f;
    bin_size.y = 0.1f;
    bin_size.z = 0.1f;
    bin_size.roll = 0.1f;
    bin_size.pitch = 0.1f;
    bin_size.yaw = 0.1f;
    tracker->setBinSize(bin_size);
    tracker_ = tracker;
  }
  tracker_->setTrans(Affine3f::Identity());
  tracker_->setStepNoiseCovariance(default_step_covariance);
  tracker_->setInitialNoiseCovariance(initial_noise_covariance);
  tracker_->setInitialNoiseMean(default_initial_mean);
  tracker_->setIterationNum(2);
  tracker_->setParticleNum(particles);
  tracker_->setResampleLikelihoodThr(0.00);
  tracker_->setUseNormal(false);
  tracker_->setSeed(seed);

  ApproxNearestPairPointCloudCoherence<RefPointType>::Ptr coherence(
      new ApproxNearestPairPointCloudCoherence<RefPointType>());
  std::shared_ptr<DistanceCoherence<RefPointType>> distance_coherence(new DistanceCoherence<RefPointType>());
  coherence->addPointCoherence(distance_coherence);
  std::shared_ptr<HSVColorCoherence<RefPointType>> color_coherence(new HSVColorCoherence<RefPointType>());
  color_coherence->setWeight(0.1);
  coherence->addPointCoherence(color_coherence);
  std::shared_ptr<search::Octree<RefPointType>> search(new search::Octree<RefPointType>(0.01));
  coherence->setSearchMethod(search);
  coherence->setMaximumDistance(0.1);
  tracker_->setCloudCoherence(coherence);

  // ---- "set object to track", auto_tracking.cpp:655-676 ----
  Cloud::Ptr ref_cloud = load_bin(files[0]);
  Cloud::Ptr nonzero_ref(new Cloud());
  removeZeroPoints(*ref_cloud, *nonzero_ref);
  if (nonzero_ref->empty()) {
    std::fprintf(stderr, "empty model\n");
    return 1;
  }
  double cx = 0, cy = 0, cz = 0;  // pcl::compute3DCentroid accumulates in the scalar type of the result (float)
  {
    float sx = 0, sy = 0, sz = 0;
    for (const auto& p : nonzero_ref->points) {
      sx += p.x;
      sy += p.y;
      sz += p.z;
    }
    cx = sx / (float)nonzero_ref->size();
    cy = sy / (float)nonzero_ref->size();
    cz = sz / (float)nonzero_ref->size();
  }
  Affine3f trans = Affine3f::Identity();
  trans(0, 3) = (float)cx;
  trans(1, 3) = (float)cy;
  trans(2, 3) = (float)cz;
  Cloud::Ptr transed_ref(new Cloud(*nonzero_ref));
  for (auto& p : transed_ref->points) {  // transformPointCloud by trans.inverse(): a pure translation
    p.x -= (float)cx;
    p.y -= (float)cy;
    p.z -= (float)cz;
  }
  Cloud::Ptr transed_ref_downsampled(new Cloud());
  if (downsampling_grid_size_ > 0) {  // gridSample (:549-561, :672): pcl::VoxelGrid on the device
    pft::VoxelGrid grid;
    const float leaf = (float)downsampling_grid_size_;
    grid.setLeafSize(leaf, leaf, leaf);
    grid.setInputCloud(transed_ref);
    grid.filter(*transed_ref_downsampled);
  } else {
    *transed_ref_downsampled = *transed_ref;
  }
  std::fprintf(stderr, "ref_cloud: %zu data points, nonzero_ref: %zu, downsampled: %zu\n", ref_cloud->points.size(),
               nonzero_ref->points.size(), transed_ref_downsampled->points.size());
  tracker_->setReferenceCloud(transed_ref_downsampled);
  tracker_->setTrans(trans);
  const Cloud::Ptr reference_ = transed_ref;  // reference_dict[obj_id] (:675): the full-resolution model, for drawResult
  tracker_->setMinIndices((int)ref_cloud->points.size() / 2);

  // ---- "track the object", auto_tracking.cpp:688-697, then drawResult :309-310 ----
  InputFilter front_end;  // filterPassThrough (z in [0, 10]) + gridSampleApprox (0.01), fused on the device
  for (size_t f = 1; f < files.size(); f++) {
    Cloud::Ptr cloud = load_bin(files[f]);
    if (raw) {
      const pft_point_xyzrgba* d_cloud = nullptr;
      size_t n_down = 0;
      front_end.setInputCloud(cloud);
      front_end.filterDevice(&d_cloud, &n_down);
      std::fprintf(stderr, "PointCloud before downsampled: %zu data points.\nPointCloud after downsampled: %zu data points.\n",
                   front_end.passedPoints(), n_down);  // auto_tracking.cpp:682, 684
      tracker_->setInputCloudDevice(d_cloud, n_down);
    } else {
      tracker_->setInputCloud(cloud);
    }
    tracker_->compute();
    ParticleXYZRPY result = tracker_->getResult();
    Affine3f transformation = tracker_->toEigenMatrix(result);
    // drawResult (:309-316) + viz_cb (:432-466): the full-resolution model moved by the result pose and its
    // centroid, which the node publishes as the object position
    float sx = 0, sy = 0, sz = 0;
    for (const auto& p : reference_->points) {
      sx += transformation(0, 0) * p.x + transformation(0, 1) * p.y + transformation(0, 2) * p.z + transformation(0, 3);
      sy += transformation(1, 0) * p.x + transformation(1, 1) * p.y + transformation(1, 2) * p.z + transformation(1, 3);
      sz += transformation(2, 0) * p.x + transformation(2, 1) * p.y + transformation(2, 2) * p.z + transformation(2, 3);
    }
    const float nref = (float)reference_->points.size();
    std::printf("frame %zu pose %.6f %.6f %.6f %.6f %.6f %.6f  t = [%.5f %.5f %.5f]  centroid = [%.5f %.5f %.5f]\n", f,
                result.x, result.y, result.z, result.roll, result.pitch, result.yaw, transformation(0, 3),
                transformation(1, 3), transformation(2, 3), sx / nref, sy / nref, sz / nref);
  }
  return 0;
}
