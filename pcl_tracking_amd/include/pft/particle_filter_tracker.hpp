// particle_filter_tracker.hpp -- header-only C++ mirror, over the C ABI of include/pft.h, of the PCL classes
// that /root/reference/src/auto_tracking.cpp instantiates and calls on its hot path:
//
//   pcl::tracking::ParticleFilterOMPTracker<RefPointType, ParticleT>      :203-204
//   pcl::tracking::KLDAdaptiveParticleFilterOMPTracker<...>               :207-222 (the runtime default, :821)
//   pcl::tracking::ParticleFilterTracker<RefPointType, ParticleT>         :153 (base type of tracker_dict)
//   pcl::tracking::ApproxNearestPairPointCloudCoherence<RefPointType>     :235-236
//   pcl::tracking::DistanceCoherence / HSVColorCoherence                  :240-247
//   pcl::search::Octree<RefPointType>                                     :250
//   pcl::tracking::ParticleXYZRPY, pcl::PointXYZRGBA, pcl::PointCloud<T>
//
// Same member names, argument meaning and error behaviour (compute() never throws; a missing input cloud
// is reported on stderr and the call returns, as PCL's PCL_ERROR + early return does).  The types live in
// namespace pft so the header can sit next to a real PCL; on a machine that has PCL, the two lines of
// INTEGRATION.md switch auto_tracking.cpp over.  All compute happens in the HIP library.
#pragma once
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "pft.h"

namespace pft {

struct PointXYZRGBA : pft_point_xyzrgba {  // layout of pcl::PointXYZRGBA
  PointXYZRGBA() {
    x = y = z = 0.0f;
    w = 1.0f;
    rgba = 0;
    pad[0] = pad[1] = pad[2] = 0;
  }
};
static_assert(sizeof(PointXYZRGBA) == 32, "pcl::PointXYZRGBA is 32 bytes");

struct ParticleXYZRPY : pft_particle {  // layout of pcl::tracking::ParticleXYZRPY
  ParticleXYZRPY() {
    x = y = z = roll = pitch = yaw = 0.0f;
    w = 1.0f;
    weight = 0.0f;
  }
  static int stateDimension() { return 6; }
  float operator[](unsigned i) const {
    switch (i) {
      case 0: return x;
      case 1: return y;
      case 2: return z;
      case 3: return roll;
      case 4: return pitch;
      case 5: return yaw;
      default: return 0.0f;
    }
  }
};
static_assert(sizeof(ParticleXYZRPY) == 32, "pcl::tracking::ParticleXYZRPY is 32 bytes");

// the part of Eigen::Affine3f the driver uses: a row-major 4x4
struct Affine3f {
  float m[16];
  Affine3f() { *this = Identity(); }
  static Affine3f Identity() {
    Affine3f a(0);
    for (int i = 0; i < 4; i++) a.m[5 * i] = 1.0f;
    return a;
  }
  float& operator()(int r, int c) { return m[4 * r + c]; }
  float operator()(int r, int c) const { return m[4 * r + c]; }

 private:
  explicit Affine3f(int) { std::memset(m, 0, sizeof(m)); }
};

template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 1;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
};

namespace search {
template <typename PointT>
class Octree {
 public:
  explicit Octree(double resolution) : resolution_(resolution) {}
  double getResolution() const { return resolution_; }

 private:
  double resolution_;
};
}  // namespace search

namespace tracking {

template <typename PointInT>
class PointCoherence {
 public:
  virtual ~PointCoherence() {}
  enum Kind { DISTANCE, HSV };
  virtual Kind kind() const = 0;
};

template <typename PointInT>
class DistanceCoherence : public PointCoherence<PointInT> {
 public:
  DistanceCoherence() : weight_(1.0) {}
  void setWeight(double w) { weight_ = w; }
  double getWeight() const { return weight_; }
  typename PointCoherence<PointInT>::Kind kind() const override { return PointCoherence<PointInT>::DISTANCE; }

 private:
  double weight_;
};

template <typename PointInT>
class HSVColorCoherence : public PointCoherence<PointInT> {
 public:
  HSVColorCoherence() : weight_(1.0), h_weight_(1.0), s_weight_(1.0), v_weight_(0.0) {}
  void setWeight(double w) { weight_ = w; }
  double getWeight() const { return weight_; }
  void setHWeight(double w) { h_weight_ = w; }
  void setSWeight(double w) { s_weight_ = w; }
  void setVWeight(double w) { v_weight_ = w; }
  double getHWeight() const { return h_weight_; }
  double getSWeight() const { return s_weight_; }
  double getVWeight() const { return v_weight_; }
  typename PointCoherence<PointInT>::Kind kind() const override { return PointCoherence<PointInT>::HSV; }

 private:
  double weight_, h_weight_, s_weight_, v_weight_;
};

template <typename PointInT>
class ApproxNearestPairPointCloudCoherence {
 public:
  typedef std::shared_ptr<ApproxNearestPairPointCloudCoherence<PointInT>> Ptr;
  typedef std::shared_ptr<PointCoherence<PointInT>> PointCoherencePtr;
  ApproxNearestPairPointCloudCoherence() : maximum_distance_(1e30), resolution_(0.01) {}
  virtual ~ApproxNearestPairPointCloudCoherence() {}
  virtual bool exactNearest() const { return false; }
  void addPointCoherence(const PointCoherencePtr& c) { point_coherences_.push_back(c); }
  // upstream keeps its own search::Octree(0.01) and ignores the object passed here; the reference passes
  // the same 0.01 (auto_tracking.cpp:250), so the resolution is taken from it
  void setSearchMethod(const std::shared_ptr<search::Octree<PointInT>>& s) {
    if (s) resolution_ = s->getResolution();
  }
  void setMaximumDistance(double d) { maximum_distance_ = d; }
  double getMaximumDistance() const { return maximum_distance_; }
  double getResolution() const { return resolution_; }
  const std::vector<PointCoherencePtr>& getPointCoherences() const { return point_coherences_; }

 private:
  std::vector<PointCoherencePtr> point_coherences_;
  double maximum_distance_, resolution_;
};

// pcl::tracking::NearestPairPointCloudCoherence: the true nearest neighbour instead of the greedy octree descent
// (the alternative auto_tracking.cpp keeps commented out at :237-238, :249); same setters
template <typename PointInT>
class NearestPairPointCloudCoherence : public ApproxNearestPairPointCloudCoherence<PointInT> {
 public:
  bool exactNearest() const override { return true; }
};

template <typename PointInT, typename StateT>
class ParticleFilterTracker {
 public:
  typedef PointCloud<PointInT> PointCloudIn;
  typedef typename PointCloudIn::ConstPtr PointCloudInConstPtr;
  typedef PointCloud<StateT> PointCloudState;
  typedef typename PointCloudState::Ptr PointCloudStatePtr;
  typedef ApproxNearestPairPointCloudCoherence<PointInT> CloudCoherence;
  typedef typename CloudCoherence::Ptr CoherencePtr;

  ParticleFilterTracker() : handle_(nullptr) {
    pft_config_default(&cfg_);
    trans_ = Affine3f::Identity();
  }
  virtual ~ParticleFilterTracker() {
    if (handle_) pft_destroy(handle_);
  }
  ParticleFilterTracker(const ParticleFilterTracker&) = delete;
  ParticleFilterTracker& operator=(const ParticleFilterTracker&) = delete;

  // ---- setters used at auto_tracking.cpp:225-254 ----
  void setTrans(const Affine3f& trans) {
    trans_ = trans;
    if (handle_) pft_set_trans(handle_, trans_.m);
  }
  void setStepNoiseCovariance(const std::vector<double>& cov) { copy6(cov, cfg_.step_noise_cov); }
  void setInitialNoiseCovariance(const std::vector<double>& cov) { copy6(cov, cfg_.initial_noise_cov); }
  void setInitialNoiseMean(const std::vector<double>& mean) { copy6(mean, cfg_.initial_noise_mean); }
  void setIterationNum(int n) { guard(); cfg_.iteration_num = n; }
  void setParticleNum(int n) { guard(); cfg_.particle_num = n; }
  void setResampleLikelihoodThr(double v) { guard(); cfg_.resample_likelihood_thr = v; }
  void setUseNormal(bool b) { guard(); cfg_.use_normal = b ? 1 : 0; }
  void setAlpha(double a) { guard(); cfg_.alpha = a; }
  void setMinIndices(int) {}  // read only when use_normal_ is true
  void setSeed(uint64_t s) { guard(); cfg_.seed = s; }      // PCL's engines are time(0)-seeded
  void setDevice(int id) { guard(); cfg_.device_id = id; }
  // enqueue on the caller's HIP stream (hipStream_t; nullptr = the default stream) instead of a stream of the handle's own
  void setStream(void* hip_stream) { guard(); cfg_.stream = hip_stream; cfg_.stream_is_external = 1; }
  // particle sharding over the GPUs of a node (one process per GPU): this handle owns the global particle ids
  // [rank * N / world, (rank + 1) * N / world); such a handle is driven through the pft_dist_* phases of pft.h with the
  // two collectives in between (examples/dist_tracking_amd.cpp), not through compute()
  void setShard(int rank, int world_size) { guard(); cfg_.rank = rank; cfg_.world_size = world_size; }
  // compute() is asynchronous and PCL's returns void.  The reference wraps it in try / catch (int)
  // (auto_tracking.cpp:692-696): with this switch on, compute() waits for the frame and throws the pft_status (an int)
  // if the device reported a failure for it (octree capacity / depth, crop time-out: pft.h)
  void setThrowOnFailure(bool b) { throw_on_failure_ = b; }
  // the handle exists from the first compute() on; create() makes it now (for callers of the C ABI's phase API)
  bool create() { return ensure(); }
  void setCloudCoherence(const CoherencePtr& c) {
    guard();
    coherence_ = c;
    cfg_.max_distance = c->getMaximumDistance();
    cfg_.octree_resolution = c->getResolution();
    cfg_.exact_nearest = c->exactNearest() ? 1 : 0;
    const auto& pcs = c->getPointCoherences();
    if (pcs.size() != 2 || pcs[0]->kind() != PointCoherence<PointInT>::DISTANCE ||
        pcs[1]->kind() != PointCoherence<PointInT>::HSV)
      throw std::invalid_argument("supported point coherences: DistanceCoherence then HSVColorCoherence");
    auto* d = static_cast<DistanceCoherence<PointInT>*>(pcs[0].get());
    auto* h = static_cast<HSVColorCoherence<PointInT>*>(pcs[1].get());
    cfg_.distance_weight = d->getWeight();
    cfg_.hsv_weight = h->getWeight();
    cfg_.h_weight = h->getHWeight();
    cfg_.s_weight = h->getSWeight();
    cfg_.v_weight = h->getVWeight();
  }

  // ---- data, auto_tracking.cpp:673, 691 ----
  void setReferenceCloud(const PointCloudInConstPtr& ref) {
    ref_ = ref;
    if (handle_ && ref_) check(pft_set_reference(handle_, ref_->points.data(), ref_->points.size()), "setReferenceCloud");
  }
  PointCloudInConstPtr getReferenceCloud() const { return ref_; }
  void setInputCloud(const PointCloudInConstPtr& cloud) {
    input_ = cloud;
    dev_input_ = nullptr;
  }
  // a cloud already in HBM (n 32-byte points), e.g. the output of pft::InputFilter::filterDevice
  void setInputCloudDevice(const pft_point_xyzrgba* device_points, size_t n) {
    input_.reset();
    dev_input_ = device_points;
    dev_n_ = n;
  }

  // ---- auto_tracking.cpp:693 ----
  void compute() {
    if (dev_input_ && dev_n_) {
      if (!ensure()) return;
      if (check(pft_set_input_device(handle_, dev_input_, dev_n_), "setInputCloudDevice") != PFT_OK) return;
      finish(check(pft_compute(handle_), "compute"));
      return;
    }
    if (!input_ || input_->points.empty()) {
      std::fprintf(stderr, "[pft::ParticleFilterTracker::compute] input cloud is empty or not set\n");
      return;  // PCL: PCL_ERROR + early return, no exception
    }
    if (!ensure()) return;
    if (check(pft_set_input(handle_, input_->points.data(), input_->points.size()), "setInputCloud") != PFT_OK) return;
    finish(check(pft_compute(handle_), "compute"));
  }

  // ---- auto_tracking.cpp:309-310, 270 ----
  StateT getResult() const {
    StateT r;
    if (handle_) check(pft_get_result(handle_, &r), "getResult");  // also reports device-side failures of the frame
    return r;
  }
  Affine3f toEigenMatrix(const StateT& particle) const {
    Affine3f a;
    pft_to_matrix(&particle, a.m);
    return a;
  }
  PointCloudStatePtr getParticles() const {
    PointCloudStatePtr out(new PointCloudState());
    if (!handle_) return out;
    size_t n = 0;
    pft_get_particles(handle_, nullptr, 0, &n);
    out->points.resize(n);
    if (n) check(pft_get_particles(handle_, out->points.data(), n, &n), "getParticles");
    out->width = (uint32_t)n;
    return out;
  }
  double getFitRatio() const {
    double v = 0.0;
    if (handle_) check(pft_get_fit_ratio(handle_, &v), "getFitRatio");
    return v;
  }
  int getIterationNum() const { return cfg_.iteration_num; }
  int getParticleNum() const { return cfg_.particle_num; }
  pft_tracker* nativeHandle() { return handle_; }

 protected:
  void guard() const {
    if (handle_) throw std::logic_error("tracker parameters are fixed after the first compute()");
  }
  void copy6(const std::vector<double>& v, double* dst) {
    guard();
    for (size_t i = 0; i < 6 && i < v.size(); i++) dst[i] = v[i];
  }
  int check(int st, const char* what) const {
    if (st != PFT_OK)
      std::fprintf(stderr, "[pft::ParticleFilterTracker::%s] %s: %s\n", what, pft_status_string(st),
                   handle_ ? pft_last_error_string(handle_) : "");
    return st;
  }
  void finish(int st) {
    if (!throw_on_failure_) return;
    if (st == PFT_OK) st = check(pft_synchronize(handle_), "compute");
    if (st != PFT_OK) throw st;
  }
  bool ensure() {
    if (handle_) return true;
    int st = pft_create(&cfg_, &handle_);
    if (st != PFT_OK) {
      check(st, "create");
      handle_ = nullptr;
      return false;
    }
    pft_set_trans(handle_, trans_.m);
    if (ref_) check(pft_set_reference(handle_, ref_->points.data(), ref_->points.size()), "setReferenceCloud");
    return true;
  }

  pft_config cfg_;
  pft_tracker* handle_;
  Affine3f trans_;
  PointCloudInConstPtr ref_, input_;
  const pft_point_xyzrgba* dev_input_ = nullptr;
  size_t dev_n_ = 0;
  bool throw_on_failure_ = false;
  CoherencePtr coherence_;
};

// the class the reference actually news (auto_tracking.cpp:203-204); the thread count is the OpenMP team
// size of PCL's CPU loops and has no meaning here
template <typename PointInT, typename StateT>
class ParticleFilterOMPTracker : public ParticleFilterTracker<PointInT, StateT> {
 public:
  explicit ParticleFilterOMPTracker(unsigned int nr_threads = 0) : threads_(nr_threads) {}
  unsigned int getNumberOfThreads() const { return threads_; }

 private:
  unsigned int threads_;
};

// the class the reference news unless use_fixed is set (auto_tracking.cpp:207-222, default :821): particle_num is
// the initial count, every resample draws until the KL bound is met (at most setMaximumParticleNum)
template <typename PointInT, typename StateT>
class KLDAdaptiveParticleFilterOMPTracker : public ParticleFilterTracker<PointInT, StateT> {
 public:
  explicit KLDAdaptiveParticleFilterOMPTracker(unsigned int nr_threads = 0) : threads_(nr_threads) {
    this->cfg_.kld_adaptive = 1;
  }
  void setMaximumParticleNum(unsigned int nr) { this->guard(); this->cfg_.maximum_particle_num = (int)nr; }
  void setDelta(double delta) { this->guard(); this->cfg_.kld_delta = delta; }
  void setEpsilon(double eps) { this->guard(); this->cfg_.kld_epsilon = eps; }
  void setBinSize(const StateT& bin_size) {
    this->guard();
    for (unsigned i = 0; i < 6; i++) this->cfg_.kld_bin_size[i] = bin_size[i];
  }
  unsigned int getNumberOfThreads() const { return threads_; }

 private:
  unsigned int threads_;
};

}  // namespace tracking
}  // namespace pft
