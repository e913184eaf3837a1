// auto_tracking_amd.cpp -- the tracking part of /root/reference/src/auto_tracking.cpp without ROS / VTK, for the
// nb_objects objects the reference tracks at once: one tracker per object (:199-257), the "set object to track" step
// (:646-677), then per frame the loop over tracker_dict (:688-697: setInputCloud, compute inside try / catch (int)) and what
// drawResult / viz_cb do with each pose (:300-326, :432-466).  The shared steps live in tracking_app.hpp.
//
//   auto_tracking_amd <model0> [<model1> ...] --frames <frame0> [<frame1> ...] [--particles N] [--seed S] [--raw] [--kld]
//                     [--model-leaf L]
//   (one model only: `auto_tracking_amd <model> <frame0> [frame1 ...]` also works)
//
// *.pcd = PCD v0.7 ascii / binary / binary_compressed with fields x y z rgba (what create_model.cpp:219-222 writes);
// anything else = raw arrays of 32-byte pcl::PointXYZRGBA records.  A model is a segmented object cluster in the
// camera frame.  Without --raw the frames are already filtered and downsampled; with --raw they are sensor frames and
// go through cloud_cb's front end first (:637 filterPassThrough, :683 gridSampleApprox) on the device, the result
// staying in HBM for all the trackers.  Every object is an independent handle on its own HIP stream: the loop below
// enqueues all of them before it reads any result, so they overlap on the GPU.
#include <cstdlib>

#include "tracking_app.hpp"

using namespace app;

int main(int argc, char** argv) {
  std::vector<const char*> models, frames;
  Options opt;
  bool raw = false, in_frames = false;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "--raw")) raw = true;
    else if (!std::strcmp(argv[i], "--kld")) opt.use_fixed = false;
    else if (!std::strcmp(argv[i], "--frames")) in_frames = true;
    else if (!std::strcmp(argv[i], "--model-leaf") && i + 1 < argc) opt.downsampling_grid_size = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--particles") && i + 1 < argc) opt.particles = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) opt.seed = std::strtoull(argv[++i], nullptr, 10);
    else (in_frames ? frames : models).push_back(argv[i]);
  }
  if (!in_frames && models.size() >= 2) {  // `<model> <frame>...`
    frames.assign(models.begin() + 1, models.end());
    models.resize(1);
  }
  if (models.empty() || frames.empty()) {
    std::fprintf(stderr, "usage: %s <model>... --frames <frame>... [--particles N] [--seed S] [--raw] [--kld] [--model-leaf L]\n", argv[0]);
    return 2;
  }

  TrackingApp v(opt);
  const int nb_objects = (int)models.size();
  for (int obj_id = 0; obj_id < nb_objects; obj_id++) v.ref_cloud_dict[obj_id] = loadCloud(models[obj_id]);
  v.buildTrackers(nb_objects, [](ParticleFilter& tr, int) { tr.setThrowOnFailure(false); });
  if (!v.setObjectsToTrack()) return 1;

  InputFilter front_end;  // filterPassThrough (z in [0, 10]) + gridSampleApprox (0.01), fused on the device
  for (size_t f = 0; f < frames.size(); f++) {
    Cloud::Ptr cloud = loadCloud(frames[f]);
    const pft_point_xyzrgba* d_cloud = nullptr;
    size_t n_down = 0;
    if (raw) {
      front_end.setInputCloud(cloud);
      front_end.filterDevice(&d_cloud, &n_down);
      std::fprintf(stderr, "PointCloud before downsampled: %zu data points.\nPointCloud after downsampled: %zu data points.\n",
                   front_end.passedPoints(), n_down);  // auto_tracking.cpp:682, 684
    }
    for (auto& kv : v.tracker_dict) {  // :688-697 -- asynchronous: all objects are in flight before a result is read
      if (raw) kv.second->setInputCloudDevice(d_cloud, n_down);
      else kv.second->setInputCloud(cloud);
      try {
        kv.second->compute();
      } catch (int e) {
        std::fprintf(stderr, "Object not recognized (%s)\n", pft_status_string(e));
      }
    }
    for (auto& kv : v.tracker_dict) {
      const ParticleT result = kv.second->getResult();
      float centroid[4];
      v.objectPosition(kv.first, result, centroid);
      printObjectLine(f + 1, kv.first, result, centroid);
    }
  }
  return 0;
}
