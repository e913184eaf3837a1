// filters.hpp -- header-only C++ mirror, over the C ABI of include/pft_filters.h, of the PCL filter classes
// that /root/reference/src/auto_tracking.cpp runs in front of the tracker:
//
//   pcl::PassThrough<PointType>            filterPassThrough   :536-547
//   pcl::VoxelGrid<PointType>              gridSample          :549-561
//   pcl::ApproximateVoxelGrid<PointType>   gridSampleApprox    :563-575
//
// Same member names and argument meaning (setFilterFieldName / setFilterLimits / setKeepOrganized /
// setLeafSize / setInputCloud / filter).  pft::InputFilter is the fused form the MI355X path prefers:
// PassThrough + voxel grid in one device pipeline whose output cloud stays in HBM and is handed to
// ParticleFilterTracker::setInputCloudDevice without touching the host.  All compute happens in the HIP
// library; filter() throws std::runtime_error on a HIP failure, as a PCL filter never silently degrades.
#pragma once
#include <stdexcept>
#include <string>

#include "pft_filters.h"
#include "pft/particle_filter_tracker.hpp"

namespace pft {

class InputFilter {
 public:
  explicit InputFilter(int device_id = 0, void* hip_stream = nullptr) {
    pft_filter_default_config(&cfg_);
    cfg_.device_id = device_id;
    if (hip_stream) {
      cfg_.stream = hip_stream;
      cfg_.stream_is_external = 1;
    }
  }
  virtual ~InputFilter() { close(); }
  InputFilter(const InputFilter&) = delete;
  InputFilter& operator=(const InputFilter&) = delete;

  void setPassThrough(const std::string& field, float lo, float hi, bool negative = false, bool enable = true) {
    cfg_.pass_enable = enable ? 1 : 0;
    cfg_.pass_field = field == "x" ? 0 : (field == "y" ? 1 : 2);
    cfg_.pass_min = lo;
    cfg_.pass_max = hi;
    cfg_.pass_negative = negative ? 1 : 0;
    close();
  }
  void setVoxelMode(int mode) {
    cfg_.voxel_mode = mode;
    close();
  }
  void setLeafSize(float lx, float ly, float lz) {
    cfg_.leaf_size[0] = lx;
    cfg_.leaf_size[1] = ly;
    cfg_.leaf_size[2] = lz;
    close();
  }
  void setHistorySize(unsigned n) {
    cfg_.approx_hist_size = n;
    close();
  }

  void setInputCloud(const PointCloud<PointXYZRGBA>::ConstPtr& cloud) {
    input_ = cloud;
    dev_in_ = nullptr;
  }
  // a cloud already in HBM (n 32-byte points)
  void setInputCloudDevice(const pft_point_xyzrgba* device_points, size_t n) {
    input_.reset();
    dev_in_ = device_points;
    dev_n_ = n;
  }

  // pcl::Filter::filter(PointCloud& output)
  void filter(PointCloud<PointXYZRGBA>& output) {
    apply();
    size_t n_pass = 0, n_out = 0;
    check(pft_filter_counts(h_, &n_pass, &n_out), "pft_filter_counts");
    output.points.resize(n_out);
    size_t got = 0;
    check(pft_filter_get_output(h_, output.points.data(), n_out, &got), "pft_filter_get_output");
    output.width = static_cast<uint32_t>(n_out);
    output.height = 1;          // downsampling breaks the organised structure
    output.is_dense = cfg_.voxel_mode != PFT_VOXEL_APPROX;  // ApproximateVoxelGrid: false, PassThrough / VoxelGrid: true
  }
  // fused path: output stays in HBM (valid until the next filter call on this object)
  void filterDevice(const pft_point_xyzrgba** device_points, size_t* n_out) {
    apply();
    check(pft_filter_output_device(h_, device_points, n_out), "pft_filter_output_device");
  }
  size_t passedPoints() const {
    size_t n_pass = 0, n_out = 0;
    if (h_) pft_filter_counts(h_, &n_pass, &n_out);
    return n_pass;
  }
  double lastMilliseconds() const {
    double ms = 0.0;
    if (h_) pft_filter_last_ms(h_, &ms);
    return ms;
  }

 protected:
  pft_filter_config cfg_;

  void close() {
    if (h_) pft_filter_destroy(h_);
    h_ = nullptr;
  }

 private:
  pft_filter* h_ = nullptr;
  PointCloud<PointXYZRGBA>::ConstPtr input_;
  const pft_point_xyzrgba* dev_in_ = nullptr;
  size_t dev_n_ = 0;

  void check(int st, const char* what) {
    if (st != PFT_OK)
      throw std::runtime_error(std::string(what) + ": " + pft_status_string(st) + " " +
                               (h_ ? pft_filter_last_error_string(h_) : ""));
  }
  void apply() {
    if (!h_) check(pft_filter_create(&cfg_, &h_), "pft_filter_create");
    if (dev_in_)
      check(pft_filter_apply_device(h_, dev_in_, dev_n_), "pft_filter_apply_device");
    else if (input_)
      check(pft_filter_apply(h_, input_->points.data(), input_->points.size()), "pft_filter_apply");
    else
      throw std::runtime_error("filter() without an input cloud");
  }
};

// pcl::PassThrough<PointXYZRGBA> (auto_tracking.cpp:536-547)
class PassThrough : public InputFilter {
 public:
  explicit PassThrough(int device_id = 0, void* hip_stream = nullptr) : InputFilter(device_id, hip_stream) {
    cfg_.voxel_mode = PFT_VOXEL_NONE;
    cfg_.pass_enable = 1;
  }
  void setFilterFieldName(const std::string& name) {
    cfg_.pass_field = name == "x" ? 0 : (name == "y" ? 1 : 2);
    close();
  }
  void setFilterLimits(float lo, float hi) {
    cfg_.pass_min = lo;
    cfg_.pass_max = hi;
    close();
  }
  void setFilterLimitsNegative(bool negative) {
    cfg_.pass_negative = negative ? 1 : 0;
    close();
  }
  void setKeepOrganized(bool keep) {
    if (keep) throw std::invalid_argument("keep_organized = true is not on the reference's path (auto_tracking.cpp:543)");
  }
};

// pcl::ApproximateVoxelGrid<PointXYZRGBA> (auto_tracking.cpp:563-575)
class ApproximateVoxelGrid : public InputFilter {
 public:
  explicit ApproximateVoxelGrid(int device_id = 0, void* hip_stream = nullptr) : InputFilter(device_id, hip_stream) {
    cfg_.voxel_mode = PFT_VOXEL_APPROX;
    cfg_.pass_enable = 0;
  }
};

// pcl::VoxelGrid<PointXYZRGBA> (auto_tracking.cpp:549-561)
class VoxelGrid : public InputFilter {
 public:
  explicit VoxelGrid(int device_id = 0, void* hip_stream = nullptr) : InputFilter(device_id, hip_stream) {
    cfg_.voxel_mode = PFT_VOXEL_EXACT;
    cfg_.pass_enable = 0;
  }
};

}  // namespace pft
