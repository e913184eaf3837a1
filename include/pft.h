/*
 * pft.h -- C ABI of the MI355X-native particle-filter point-cloud tracker.
 *
 * Drop-in boundary for the ONE hot path of cmaestre/pcl_tracking: the
 *   pcl::tracking::ParticleFilter(OMP)Tracker<pcl::PointXYZRGBA, pcl::tracking::ParticleXYZRPY>
 * object that /root/reference/src/auto_tracking.cpp builds at :201-206, configures at :225-254,
 * feeds at :673-676 (setReferenceCloud / setTrans) and :691 (setInputCloud) and runs at :693
 * (compute()), reading the pose back at :309-310 (getResult / toEigenMatrix) and :270 (getParticles).
 * The reference has no FFI for this path (it calls the PCL C++ classes directly); each entry point
 * below names the PCL member call it replaces.  The header-only C++ mirror of those classes over
 * this ABI is pcl_tracking_amd/include/pft/particle_filter_tracker.hpp; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Plain C: POD structs in PCL's memory layout, raw pointers and sizes, int status codes.
 * Nothing throws across this boundary.  A handle is not re-entrant (the reference serialises its
 * callback under a mutex, auto_tracking.cpp:604).  All compute runs as HIP kernels on gfx950; there
 * is no CPU fallback: without a usable GPU pft_create() fails with PFT_ERR_NO_DEVICE.
 */
#ifndef PFT_H
#define PFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFT_ABI_VERSION 4

/* pcl::PointXYZRGBA (32 B, 16-aligned): x,y,z,1.0f | rgba (bytes b,g,r,a) | 12 B pad */
typedef struct pft_point_xyzrgba {
  float x, y, z, w;
  uint32_t rgba;
  uint32_t pad[3];
} pft_point_xyzrgba;

/* pcl::tracking::ParticleXYZRPY (32 B, 16-aligned): x,y,z,1.0f | roll,pitch,yaw,weight */
typedef struct pft_particle {
  float x, y, z, w;
  float roll, pitch, yaw, weight;
} pft_particle;

typedef enum pft_status {
  PFT_OK = 0,
  PFT_ERR_INVALID_ARG = 1,
  PFT_ERR_NO_INPUT = 2,     /* compute() without an input cloud: PCL prints PCL_ERROR and returns */
  PFT_ERR_NO_REFERENCE = 3, /* compute() before setReferenceCloud */
  PFT_ERR_NO_DEVICE = 4,    /* no gfx950 GPU / HIP runtime unusable: there is no CPU fallback */
  PFT_ERR_HIP = 5,          /* a HIP call failed; see pft_last_error_string */
  PFT_ERR_CAPACITY = 6,     /* a size exceeds what the handle was created for */
  PFT_ERR_STATE = 7         /* call not valid in the current state */
} pft_status;

/* One POD holding every parameter the reference sets (auto_tracking.cpp:187-253) plus PCL's
 * constructor defaults it relies on.  pft_config_default() fills the reference's values. */
typedef struct pft_config {
  uint32_t abi_version;        /* PFT_ABI_VERSION */
  int32_t device_id;           /* HIP device ordinal */
  void* stream;                /* hipStream_t to enqueue on (used when stream_is_external != 0; the null
                                  handle then means HIP's default stream) */
  int32_t stream_is_external;  /* 0: the handle creates its own non-blocking stream */
  int32_t particle_num;        /* setParticleNum            :231   400 */
  int32_t iteration_num;       /* setIterationNum           :229   2 */
  double step_noise_cov[6];    /* setStepNoiseCovariance    :226   0.015^2 (x40 for r,p,y) */
  double initial_noise_cov[6]; /* setInitialNoiseCovariance :227   1e-5 */
  double initial_noise_mean[6];/* setInitialNoiseMean       :228   0 */
  double alpha;                /* ParticleFilterTracker ctor default 15.0 */
  double resample_likelihood_thr; /* setResampleLikelihoodThr :232 (inert upstream; kept for API parity) */
  double max_distance;         /* coherence->setMaximumDistance :253  0.1 */
  double octree_resolution;    /* search::Octree(0.01)      :251 */
  double distance_weight;      /* DistanceCoherence default weight 1.0 */
  double hsv_weight;           /* HSVColorCoherence::setWeight :246  0.1 */
  double h_weight, s_weight, v_weight; /* HSVColorCoherence ctor defaults 1, 1, 0 */
  int32_t hsv_pcl180_argorder; /* 1 = RGB2HSV(Red, Blue, Green) as written in PCL 1.8.0 */
  int32_t use_normal;          /* setUseNormal :233; only 0 is supported */
  uint64_t seed;               /* key of the counter-based RNG (PCL's engines are time(0)-seeded) */
  /* multi-GPU sharding: this handle owns global particle ids [rank*P/world, (rank+1)*P/world) */
  int32_t rank, world_size;
  /* capacities (0 = grow on demand at set_reference / set_input) */
  uint32_t max_reference_points, max_input_points;
  /* KLDAdaptiveParticleFilterOMPTracker, the tracker auto_tracking.cpp runs unless use_fixed is set (:207-222, :821):
   * particle_num is then only the initial count; every resample draws until the KL bound is met */
  int32_t kld_adaptive;          /* 0: ParticleFilterOMPTracker (:203-204)   1: KLD-adaptive (:207-208) */
  int32_t maximum_particle_num;  /* setMaximumParticleNum :209   500 */
  double kld_delta;              /* setDelta              :210   0.99 */
  double kld_epsilon;            /* setEpsilon            :211   0.2 */
  double kld_bin_size[6];        /* setBinSize            :212-219  0.1 each */
  double motion_ratio;           /* ParticleFilterTracker ctor default 0.25 (used by the KLD resample only) */
  /* NearestPairPointCloudCoherence (true nearest neighbour) instead of ApproxNearestPair...: the alternative the
   * reference keeps commented out at auto_tracking.cpp:237-238, :249 */
  int32_t exact_nearest;
} pft_config;

typedef struct pft_tracker pft_tracker;

void pft_config_default(pft_config* cfg);
const char* pft_status_string(int status);

/* new ParticleFilterOMPTracker<...>(threads) + the setters of :225-254 */
int pft_create(const pft_config* cfg, pft_tracker** out);
void pft_destroy(pft_tracker* t);
const char* pft_last_error_string(const pft_tracker* t);

/* tracker_->setReferenceCloud(cloud) :673 -- host pointer, PCL layout, copied */
int pft_set_reference(pft_tracker* t, const pft_point_xyzrgba* pts, size_t n);
/* tracker_->setTrans(Eigen::Affine3f) :225, :674 -- row-major 4x4 */
int pft_set_trans(pft_tracker* t, const float m[16]);
/* tracker_->setInputCloud(cloud) :691 -- host pointer, copied to HBM before the call returns */
int pft_set_input(pft_tracker* t, const pft_point_xyzrgba* pts, size_t n);
/* same, for a cloud already resident in HBM (device pointer, PCL layout; borrowed until the next
 * pft_set_input* call) */
int pft_set_input_device(pft_tracker* t, const void* device_pts, size_t n);
/* tracker_->compute() :693 -- first call runs initParticles; then iteration_num x
 * [resample, weight, update].  Asynchronous on the handle's stream. */
int pft_compute(pft_tracker* t);
/* tracker_->getResult() :309 -- synchronises the stream.  Like every call that synchronises (pft_get_particles,
 * pft_get_fit_ratio, pft_synchronize, pft_eval_weights) it also reports device-side failures of the iterations run
 * since the last such call: PFT_ERR_CAPACITY (octree node capacity, depth or bounding-box growth steps exceeded) or
 * PFT_ERR_HIP (the one-pass crop gave up waiting), pft_last_error_string naming the flag.  The affected iteration ran
 * without a target cloud (all likelihoods zero), so the pose returned with the error is the unweighted particle mean;
 * the flags are per iteration and the next pft_compute starts clean.  (PCL's compute() is void; the reference's caller
 * wraps it in try / catch, auto_tracking.cpp:692-696.) */
int pft_get_result(pft_tracker* t, pft_particle* out);
/* tracker_->getParticles() :270 -- copy-out of all particle_num particles (all ranks' shards) */
int pft_get_particles(pft_tracker* t, pft_particle* out, size_t cap, size_t* n);
/* tracker_->toEigenMatrix(result) :310 == pcl::getTransformation; host-side helper, row-major 4x4 */
void pft_to_matrix(const pft_particle* p, float m[16]);
/* ParticleXYZRPY::toState(Affine3f) */
void pft_to_state(const float m[16], pft_particle* out);
/* fit_ratio_ diagnostic (w_min of the last normalizeWeight) */
int pft_get_fit_ratio(pft_tracker* t, double* out);
int pft_synchronize(pft_tracker* t);

/* ---- multi-GPU phase API (one handle per rank; the collectives between the phases are issued by
 *      the host layer on the same stream, see pcl_tracking_amd/dist.py and DESIGN.md) ----
 * The host layer owns three device buffers and binds them once:
 *   bbox6    6 floats {-xmin,-ymin,-zmin,xmax,ymax,zmax}: ONE max all-reduce gives the global AABB
 *   shard    P/world particles (32 B each) with the raw weight in .weight
 *   gathered P particles: the all-gather of every rank's shard, in rank order
 * iteration = phase_a -> [all-reduce(max) bbox6] -> phase_b -> [all-gather shard -> gathered] -> phase_c */
int pft_dist_bind(pft_tracker* t, void* bbox6_dev, void* shard_dev, void* gathered_dev);
int pft_dist_begin_frame(pft_tracker* t);          /* initParticles on the first frame */
int pft_dist_phase_a(pft_tracker* t, int iteration); /* resample shard, pose->matrix, local AABB */
int pft_dist_phase_b(pft_tracker* t);              /* crop, octree, likelihood, raw weights into shard */
int pft_dist_phase_c(pft_tracker* t);              /* normalise, update, alias table over all particles */

/* ---- test hooks (used by tests/ to compare every stage with the oracle) ---- */
int pft_set_particles(pft_tracker* t, const pft_particle* p, size_t n);
/* deterministic chain A1-A7 on explicit particles: raw_w[P] = -(float)sum; optional per-pair
 * approximate-NN index into the cropped cloud (nn_idx[P*M]) and squared distance (nn_d2[P*M]) */
int pft_eval_weights(pft_tracker* t, const pft_particle* particles, size_t P, float* raw_w, int32_t* nn_idx,
                     float* nn_d2);
int pft_debug_get_bbox(pft_tracker* t, float bbox[6]); /* x_min,x_max,y_min,y_max,z_min,z_max */
int pft_debug_get_crop(pft_tracker* t, int32_t* idx, size_t cap, size_t* n);
int pft_debug_get_octree(pft_tracker* t, int32_t* depth, double min_xyz[3], double max_xyz[3], uint32_t* n_leaves,
                         uint32_t* n_nodes);
int pft_debug_get_point_keys(pft_tracker* t, uint32_t* keys3, size_t cap_points);
int pft_debug_get_scan_stats(pft_tracker* t, uint64_t* queries, uint64_t* scanned_points);
/* limits for the error-path tests: max_words != 0 lowers the octree node capacity (never above the allocation),
 * sorted_npass != 0 fixes the radix passes of the sorted builder (0 = derived from the previous depth) */
int pft_debug_set_limits(pft_tracker* t, uint32_t max_words, int sorted_npass);
/* checkpoint of the filter state between two frames (population with weights, alias table, representative state,
 * motion, KLD particle count, resample epoch); restore is one kernel on the handle's stream.  bench.py replays the
 * same frame with it (stationary workload); tests use it to compare two schedules from the same state. */
int pft_debug_state_save(pft_tracker* t);
int pft_debug_state_restore(pft_tracker* t);
/* OR `bits` into the device-side error flags right after the next crop launch (what a failing stage leaves behind) */
int pft_debug_inject_error(pft_tracker* t, uint32_t bits);
/* the pinned status block: [0] last crop size, [1] last depth, [2] flags of the last failed iteration, [3] unreported flags */
int pft_debug_get_host_stat(pft_tracker* t, uint32_t out4[4]);
/* wall-clock stamps (100 MHz ticks) taken at phase boundaries inside the single-workgroup kernels of the
 * last iteration: [0..15] octree build, [16..31] population.  The kernels take the stamps only in the diagnostic variant
 * library (-DPFT_DIAG); the product library returns zeros */
int pft_debug_get_ticks(pft_tracker* t, uint64_t* ticks32);
/* descent statistics of the last pft_eval_weights call that asked for the NN arrays: [0..10] queries by
 * number of generic levels, [11] queries that used the jump table, [12] wave iterations, [13..15] sums of
 * the per-wave maxima of generic levels / fast levels / leaf size, [16..26] wave iterations by max generic */
int pft_debug_get_descent_stats(pft_tracker* t, uint64_t* dbg32);
/* host-only (no device needed): the positions of the reference points the bounding box of the particles' transformed
 * clouds is taken over (A3: calcBoundingBox of the tracker that /root/reference/src/auto_tracking.cpp:691-693 runs) -- the
 * convex hull's vertices plus the shell the float evaluation can reach; `keep` has room for n indices (ascending),
 * *n_keep receives their number (n itself for a degenerate cloud).  Same points, same order as the library uses for a
 * reference cloud handed to pft_set_reference in this order. */
int pft_debug_aabb_support_subset(const pft_point_xyzrgba* pts, size_t n, uint32_t* keep, size_t* n_keep);
#ifdef PFT_DIAG
/* diagnostic variant library only (tools/build_variant.py diag -DPFT_DIAG; never in libpft_hip.so): skip stages of the
 * likelihood kernel for timing (bit0 generic levels, bit1 leaf scan, bit2 coherence); results are wrong while set */
void pft_debug_set_ablate(int mask);
#endif
/* diagnostic: resident likelihood workgroups per CU according to the HIP occupancy API */
int pft_debug_likelihood_occupancy(void);
int pft_debug_normalize(pft_tracker* t, float* w_inout, size_t n, double* fit_ratio);
int pft_debug_alias(pft_tracker* t, const float* w, size_t n, int32_t* a, double* q);
int pft_debug_weighted_mean(pft_tracker* t, const pft_particle* p, size_t n, pft_particle* out);
int pft_debug_init_particles(pft_tracker* t, const pft_particle* rep, uint32_t id_offset, size_t n_local,
                             pft_particle* out);
int pft_debug_resample(pft_tracker* t, const pft_particle* old, size_t n_total, const int32_t* a, const double* q,
                       const pft_particle* rep, uint32_t epoch, uint32_t id_offset, size_t n_local,
                       pft_particle* out);
int pft_debug_pose_to_matrix(pft_tracker* t, const pft_particle* p, size_t n, float* m12);
/* KLD resample alone: n_old particles + their explicit alias table (a, q) + motion -> the new particle set
 * (capacity maximum_particle_num), their 6-D bins, the new count and the number of distinct bins */
int pft_debug_kld_resample(pft_tracker* t, const pft_particle* old, size_t n_old, const int32_t* a, const double* q,
                           const pft_particle* motion, uint32_t epoch, pft_particle* out, int32_t* bins6,
                           uint32_t* n_out, uint32_t* k_out);
/* host-side KLDAdaptiveParticleFilterTracker::normalQuantile / calcKLBound (what the resample kernel is given) */
double pft_kld_normal_quantile(double u);
double pft_kld_bound(int k, double delta, double epsilon);

/* ---- per-kernel HIP-event timing on the handle's stream ---- */
enum {
  PFT_K_RESAMPLE = 0, PFT_K_AABB = 1, PFT_K_CROP = 2, PFT_K_OCTREE = 3, PFT_K_LIKELIHOOD = 4,
  PFT_K_POPULATION = 5, PFT_K_PACK = 6, PFT_K_COUNT = 7
};
int pft_profile_enable(pft_tracker* t, int on);
int pft_profile_get(pft_tracker* t, int kernel_id, double* total_ms, uint64_t* launches);
int pft_profile_reset(pft_tracker* t);
const char* pft_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif
