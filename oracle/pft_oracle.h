/*
 * pft_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11 + OpenMP) of the particle-filter tracking hot path that
 * cmaestre/pcl_tracking drives through PCL 1.8.0
 *   ParticleFilterOMPTracker<PointXYZRGBA, ParticleXYZRPY> + ApproxNearestPairPointCloudCoherence
 *   (reference call sites: /root/reference/src/auto_tracking.cpp:201-254, 673-676, 691-693).
 *
 * PARITY UNPINNED: the arithmetic lives in PCL 1.8.0 (find_package(PCL 1.8.0 EXACT),
 * /root/reference/CMakeLists.txt:4), which is neither vendored in the reference nor installed
 * in the build container, and the reference ships no tests / golden vectors / recorded frames
 * for this path.  This file restates the published PCL 1.8.0 algorithm
 * (upstream file named at every function) and is checked against hand-derived known answers
 * only.  Golden fixtures under tests/golden/ pin THIS restatement, not PCL.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (pcl_tracking_amd/) never links, imports or falls back to it.
 */
#ifndef PFT_ORACLE_H
#define PFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZRGBA, 32 bytes (PCL 1.8.0 common/include/pcl/impl/point_types.hpp). rgba bytes: b,g,r,a */
typedef struct {
  float x, y, z, w;
  uint32_t rgba;
  uint32_t pad[3];
} orc_point_t;

/* pcl::tracking::ParticleXYZRPY, 32 bytes (PCL 1.8.0 tracking/include/pcl/tracking/impl/tracking.hpp) */
typedef struct {
  float x, y, z, w;
  float roll, pitch, yaw, weight;
} orc_particle_t;

/* Parameters of the path; defaults = /root/reference/src/auto_tracking.cpp:187-253 + PCL ctor defaults */
typedef struct {
  int32_t particle_num;      /* :231  400 */
  int32_t iteration_num;     /* :229  2 */
  double step_cov[6];        /* :187-190, 226 */
  double init_cov[6];        /* :192, 227 */
  double init_mean[6];       /* :193, 228 */
  double alpha;              /* PCL particle_filter.h ctor: 15.0 */
  double max_distance;       /* :253  0.1 */
  double octree_resolution;  /* :251  0.01 (ApproxNearestPair keeps its own search::Octree(0.01)) */
  double distance_weight;    /* DistanceCoherence ctor: 1.0 */
  double hsv_weight;         /* :246  0.1 */
  double h_weight, s_weight, v_weight; /* HSVColorCoherence ctor: 1, 1, 0 */
  int32_t hsv_pcl180_argorder; /* 1: RGB2HSV(Red, Blue, Green) as written upstream */
  int32_t threads;           /* :845 16 OpenMP threads; <=0 -> omp default */
  int32_t emulate_pcl_alloc; /* 1: per-query index-vector malloc as OctreePointCloudSearch does */
  uint64_t seed;             /* counter-based RNG key (PCL itself is time(0)-seeded, unobservable) */
  /* KLDAdaptiveParticleFilterOMPTracker, the reference's runtime default (use_fixed == false, :821; :207-222) */
  int32_t kld_adaptive;      /* 0: ParticleFilterOMPTracker (:203-204)   1: KLD-adaptive variant (:207-208) */
  int32_t kld_max_particles; /* :209  setMaximumParticleNum(500) */
  double kld_delta;          /* :210  0.99 */
  double kld_epsilon;        /* :211  0.2 */
  double kld_bin_size[6];    /* :212-219  0.1 each (stored as the float members of a ParticleXYZRPY upstream) */
  double motion_ratio;       /* PCL particle_filter.h ctor: 0.25 */
  /* NearestPairPointCloudCoherence instead of the Approx... one: the alternative left commented out at
   * /root/reference/src/auto_tracking.cpp:237-238: true nearest neighbour in the cropped cloud */
  int32_t exact_nearest;
} orc_config_t;

void orc_config_default(orc_config_t* c);

/* ---- A1: pcl::getTransformation (common/impl/eigen.hpp); m = row-major 4x4 ---- */
void orc_get_transformation(float x, float y, float z, float roll, float pitch, float yaw, float m[16]);
/* ---- A0: ParticleXYZRPY::toState -> pcl::getTranslationAndEulerAngles ---- */
void orc_to_state(const float m[16], orc_particle_t* out);
/* ---- A2: pcl::transformPointCloud dense branch (common/impl/transforms.hpp) ---- */
void orc_transform_cloud(const orc_point_t* in, size_t n, const float m[16], orc_point_t* out);

/* ---- A7b: RGB2HSV + HSVColorCoherence (tracking/impl/hsv_color_coherence.hpp) ---- */
int orc_div_table(int i);
void orc_rgb2hsv(int r, int g, int b, float* fh, float* fs, float* fv);
/* integer h (0..179), s (0..255), v (0..255) of the same routine, for packing tests */
void orc_rgb2hsv_int(int r, int g, int b, int* h, int* s, int* v);
double orc_hsv_coherence(const orc_config_t* c, uint32_t src_rgba, uint32_t tgt_rgba);
/* ---- A7a: DistanceCoherence (tracking/impl/distance_coherence.hpp) ---- */
double orc_distance_coherence(const orc_config_t* c, const orc_point_t* s, const orc_point_t* t);

/* ---- A5/A6: OctreePointCloudSearch (octree/impl/octree_pointcloud.hpp, octree_search.hpp) ---- */
typedef struct orc_octree orc_octree_t;
orc_octree_t* orc_octree_build(const orc_point_t* pts, size_t n, double resolution, int emulate_pcl_alloc);
void orc_octree_free(orc_octree_t* t);
/* info[0..2]=min xyz, info[3..5]=max xyz */
void orc_octree_info(const orc_octree_t* t, int* depth, double bounds[6], size_t* leaf_count, size_t* branch_count);
/* key of point i as assigned at insertion time, expressed in the FINAL key frame (shifts applied) */
void orc_octree_point_key(const orc_octree_t* t, size_t i, uint32_t key[3]);
/* returns 0 when the tree is empty (PCL asserts), 1 otherwise */
int orc_octree_approx_nearest(const orc_octree_t* t, const orc_point_t* q, int* idx, float* sqr_dist);
/* mean leaf occupancy counter for the roofline's algorithmic bytes */
void orc_octree_scan_stats(const orc_octree_t* t, uint64_t* queries, uint64_t* scanned_points);

/* ---- A8: normalizeWeight; in-place on w[n] ---- */
void orc_normalize_weights(float* w, size_t n, double alpha, double* fit_ratio);
/* ---- A9: genAliasTable ---- */
void orc_gen_alias_table(const float* w, size_t n, int32_t* a, double* q);
/* ---- A10: update (weighted mean); returns representative with weight 1/n ---- */
void orc_weighted_mean(const orc_particle_t* p, size_t n, orc_particle_t* rep);

/* ---- RNG spec shared (by specification, not by code) with the product: Philox4x32-10 ---- */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_u53(uint32_t a, uint32_t b);
/* purpose 0 = initParticles, 1 = resample. slot 0 = alias uniform, 1..3 = normal pairs */
void orc_rng_normal_pair(uint64_t seed, uint32_t particle_id, uint32_t slot, uint32_t epoch, uint32_t purpose,
                         double* z0, double* z1);
double orc_rng_uniform(uint64_t seed, uint32_t particle_id, uint32_t slot, uint32_t epoch, uint32_t purpose);

/* ---- A0 initParticles / A11 resample on explicit arrays ---- */
void orc_init_particles(const orc_config_t* c, const orc_particle_t* rep, uint32_t id_offset, size_t n_local,
                        orc_particle_t* out);
void orc_resample(const orc_config_t* c, const orc_particle_t* old, size_t n_total, const int32_t* a,
                  const double* q, const orc_particle_t* rep, uint32_t epoch, uint32_t id_offset, size_t n_local,
                  orc_particle_t* out);

/* ---- the tracker (A12 schedule) ---- */
typedef struct orc_tracker orc_tracker_t;
orc_tracker_t* orc_tracker_create(const orc_config_t* c);
void orc_tracker_destroy(orc_tracker_t* t);
int orc_tracker_set_reference(orc_tracker_t* t, const orc_point_t* pts, size_t n);
int orc_tracker_set_trans(orc_tracker_t* t, const float m[16]);
int orc_tracker_set_input(orc_tracker_t* t, const orc_point_t* pts, size_t n); /* borrowed */
int orc_tracker_compute(orc_tracker_t* t);  /* 0 ok; 1 = no input (PCL_ERROR + early return) */
void orc_tracker_get_result(const orc_tracker_t* t, orc_particle_t* out);
size_t orc_tracker_get_particles(const orc_tracker_t* t, orc_particle_t* out, size_t cap);
int orc_tracker_set_particles(orc_tracker_t* t, const orc_particle_t* p, size_t n);
double orc_tracker_fit_ratio(const orc_tracker_t* t);
/* tests only: use these P row-major 4x4 matrices instead of toEigenMatrix(particle) in eval_weights
 * (isolates the float descent/coherence arithmetic from libm-vs-ocml sin/cos ulp differences) */
void orc_tracker_set_matrix_override(orc_tracker_t* t, const float* m16);
/* tests only: 0 (default) = A1 with cosf / sinf as PCL; 1 = sin / cos in double rounded to float, the way the product's
 * device code forms the matrix -- identical matrices on both sides, for bit-level comparison of long tracking runs */
void orc_tracker_set_trig_mode(orc_tracker_t* t, int mode);
/* tests only: 0 (default) = PCL's sequential weight sum (double) and weighted mean (float); 1 = the order the product
 * specifies for its parallel reductions: adjacent-pair tree over the index range padded to a power of two, in double */
void orc_tracker_set_sum_mode(orc_tracker_t* t, int mode);
/* the two stages of sum mode 1 on explicit arrays (same exp / division as orc_normalize_weights and the same products as
 * orc_weighted_mean; only the order of the additions differs: adjacent-pair tree in double) */
void orc_normalize_weights_tree(float* w, size_t n, double alpha, double* fit_ratio);
void orc_weighted_mean_tree(const orc_particle_t* p, size_t n, orc_particle_t* rep);
/* tests of the particle-sharded host logic: crop with this box (the reduction over all ranks) instead of
 * the box of the given particles; bbox_only stops eval_weights after calcBoundingBox */
void orc_tracker_set_bbox_override(orc_tracker_t* t, const double* bbox6);
void orc_tracker_set_bbox_only(orc_tracker_t* t, int on);

/* Stage hook: the deterministic chain A1-A7 of one weight() call on explicit particles.
 * Any output pointer may be NULL.
 *   raw_w[P]            -(float)val per particle (before normalizeWeight)
 *   nn_idx[P*M], nn_d2[P*M]   approx-NN index INTO THE CROPPED CLOUD and its squared distance
 *   crop_idx[cap]       indices (into the input cloud) of the points surviving the crop, in order
 *   bbox[6]             x_min,x_max,y_min,y_max,z_min,z_max of calcBoundingBox
 * returns the number of cropped points. */
size_t orc_tracker_eval_weights(orc_tracker_t* t, const orc_particle_t* particles, size_t P, float* raw_w,
                                int32_t* nn_idx, float* nn_d2, int32_t* crop_idx, size_t crop_cap,
                                double bbox[6], int* octree_depth, double octree_bounds[6],
                                uint64_t* scan_queries, uint64_t* scan_points);

/* per-stage wall time of the last compute(), seconds: [0]=transform [1]=bbox+crop [2]=octree build
 * [3]=coherence [4]=normalize [5]=resample [6]=update */
void orc_tracker_stage_times(const orc_tracker_t* t, double s[7]);

/* KLDAdaptiveParticleFilterTracker::normalQuantile (kld_adaptive_particle_filter.h): despite its name the
 * polynomial normal CDF of CACM Algorithm 209; and calcKLBound(k) with z = normalQuantile(delta) */
double orc_kld_normal_quantile(double u);
double orc_kld_bound(int k, double delta, double epsilon);
/* KLDAdaptiveParticleFilterTracker::resample (impl/kld_adaptive_particle_filter.hpp): draws until the KL bound
 * is met; out has room for cfg->kld_max_particles; returns the new particle count */
size_t orc_kld_resample(const orc_config_t* c, const orc_particle_t* old, size_t n_old, const int32_t* a,
                        const double* q, const orc_particle_t* motion, uint32_t epoch, orc_particle_t* out,
                        int32_t* bins_out /* 6 per particle, may be NULL */, int32_t* k_out /* distinct bins */);

/* ---- input filters in front of the tracker (pft_oracle_filters.c) ---- */
/* PassThrough with a field name (0 = x, 1 = y, 2 = z), inclusive limits, keep_organized = false; returns #kept */
size_t orc_pass_through(const orc_point_t* pts, size_t n, int field, float lo, float hi, int negative,
                        int32_t* out_idx);
/* ApproximateVoxelGrid (history table of hist_size entries, flush on collision); out: capacity n; returns #out */
size_t orc_approx_voxel_grid(const orc_point_t* pts, size_t n, const float leaf[3], uint32_t hist_size,
                             orc_point_t* out);
/* VoxelGrid (exact, output sorted by voxel index; in-voxel summation in input order); returns #out or (size_t)-1 */
size_t orc_voxel_grid(const orc_point_t* pts, size_t n, const float leaf[3], orc_point_t* out);

/* ---- host-side steps around the trackers (pft_oracle_app.c) ---- */
/* removeZeroPoints (/root/reference/src/auto_tracking.cpp:577-595); out: capacity n; returns #kept */
size_t orc_remove_zero_points(const orc_point_t* in, size_t n, orc_point_t* out);
/* pcl::compute3DCentroid<PointT, float>; returns the number of points used (0: centroid untouched) */
size_t orc_compute_3d_centroid(const orc_point_t* pts, size_t n, int is_dense, float centroid[4]);
/* :663-668 re-centre the model on its centroid; trans16 = the matrix handed to setTrans */
void orc_recentre_model(const orc_point_t* in, size_t n, const float centroid[4], orc_point_t* out, float trans16[16]);
/* drawResult + viz_cb (:309-316, :432-433): full-resolution model moved by the result pose (z - 5 mm) and its centroid */
void orc_object_position(const orc_point_t* reference_full, size_t n, const orc_particle_t* result, orc_point_t* moved,
                         float centroid[4]);

#ifdef __cplusplus
}
#endif
#endif
