// pft_api.hip -- the extern "C" boundary declared in include/pft.h: handle lifetime, HBM buffers,
// the A12 schedule (ParticleFilterOMPTracker::computeTracking) as a chain of kernel launches on one
// HIP stream, and the test / profiling hooks.  No CPU compute path exists here: every stage is a kernel.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "pft_internal.h"

#define HIPCHK(t, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (t)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return PFT_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

struct EvPair {
  hipEvent_t a, b;
};

struct pft_tracker {
  pft_config cfg;
  PftParams prm;
  PftDev dev;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
  std::string err;

  // buffers
  pft_point_xyzrgba* d_ref_raw = nullptr;
  float4 *d_ref_xyz = nullptr, *d_ref_hsv = nullptr, *d_ref_box = nullptr;
  size_t bbox_part_cap = 0;
  uint32_t ref_cap = 0;
  pft_point_xyzrgba* d_in_raw = nullptr;
  float4* d_in_pts = nullptr;
  uint32_t in_cap = 0, N = 0;
  const pft_point_xyzrgba* raw_pending = nullptr;  // input handed over but not yet packed: the next crop does it
  pft_particle* d_part[2] = {nullptr, nullptr};
  int cur = 0;
  float* d_mats = nullptr;
  float* d_bbox_part = nullptr;
  float* d_bbox6 = nullptr;
  uint32_t* d_crop_counts = nullptr;
  unsigned long long* d_crop_slots = nullptr;
  uint32_t crop_epoch = 0;  // tag of the last one-pass crop launch (0 = what freshly allocated slots hold)
  float4* d_crop_pts = nullptr;
  int32_t* d_crop_idx = nullptr;
  uint32_t* d_words = nullptr;
  uint32_t max_words = 0;
  uint16_t* d_jump = nullptr;
  uint32_t* d_ref_perm = nullptr;
  float4* d_leaf_pts = nullptr;
  uint32_t *d_leaf_order = nullptr, *d_pt_node = nullptr, *d_pt_key = nullptr, *d_pt_tmp = nullptr;
  unsigned long long* d_pt_key64 = nullptr;
  SortBufs sort = {};
  uint32_t* h_stat = nullptr;   // pinned, device-visible
  int force_builder = 0;        // PFT_FORCE_BUILDER: 1 single workgroup, 2 sorted
  int force_npass = 0;          // test hook (pft_debug_set_limits): radix passes of the sorted builder, 0 = from the last depth
  uint32_t inject_error = 0;    // test hook (pft_debug_inject_error): bits OR-ed into PftHeader::error after the next crop
  double* d_partial = nullptr;
  int32_t* d_alias_list = nullptr;
  double* d_alias_pref = nullptr;
  double* d_pop_part = nullptr;
  uint32_t *d_eg_start = nullptr, *d_eg_cnt = nullptr, *d_eg_tile = nullptr;
  uint32_t *d_ec_slot = nullptr, *d_ec_cells = nullptr, *d_ec_count = nullptr, *d_ec_base = nullptr;
  float4* d_ec_list = nullptr;
  uint32_t ec_pool_cap = 0;  // entries of d_ec_list
  // exact-NN mode, cell-sorted queries (sized by the largest particle count x reference size evaluated so far)
  uint32_t *d_eq_cellq = nullptr, *d_eq_nq = nullptr, *d_eq_qbase = nullptr, *d_eq_bbase = nullptr, *d_eq_fill = nullptr,
           *d_eq_blk = nullptr, *d_eq_tiles = nullptr;
  float4* d_eq_sorted = nullptr;
  double* d_eq_out = nullptr;
  uint32_t eq_cap = 0, eq_blk_cap = 0, eq_tiles_cap = 0;
  uint32_t* d_kld_table = nullptr;
  int32_t* d_kld_bins = nullptr;
  uint32_t dbg_builds = 0;
  uint32_t Pcap = 0;  // particle capacity of the buffers (== P_total unless KLD-adaptive)
  uint32_t* d_alias_pos = nullptr;
  float* d_raw_w = nullptr;
  PftHeader* d_hdr = nullptr;
  int32_t* d_nn_idx = nullptr;
  float* d_nn_d2 = nullptr;
  size_t nn_cap = 0;
  // debug scratch
  pft_particle* d_dbg_part = nullptr;
  size_t dbg_part_cap = 0;
  PftHeader* d_dbg_hdr = nullptr;
  float* d_dbg_f = nullptr;
  size_t dbg_f_cap = 0;
  // dist binding
  void *bound_bbox6 = nullptr, *bound_shard = nullptr, *bound_gathered = nullptr;

  // pft_debug_state_save / _restore: a checkpoint of the filter state in HBM (bench.py's stationary workload, tests)
  pft_particle* sv_part = nullptr;
  int32_t* sv_alias_list = nullptr;
  double* sv_alias_pref = nullptr;
  uint32_t* sv_alias_pos = nullptr;
  PftHeader* sv_hdr = nullptr;
  int sv_cur = 0;
  uint32_t sv_epoch = 0;
  bool sv_changed = false, sv_valid = false;

  // state
  bool has_ref = false, has_input = false, initialized = false, changed = false;
  uint32_t resample_epoch = 0;
  float trans[16];

  // frame graph (PFT_GRAPH=1, own stream only): the launches of a steady-state frame are captured and replayed as one
  // hipGraph; the instantiated graph is updated in place while the launch sequence keeps its shape
  bool use_graph = false;
  hipGraphExec_t graph_exec = nullptr;
  uint32_t graph_frames = 0, graph_rebuilds = 0;

  // profiling
  bool prof = false;
  std::vector<EvPair> ev[PFT_K_COUNT];
  std::vector<EvPair> ev_free;
  double prof_ms[PFT_K_COUNT] = {0};
  uint64_t prof_n[PFT_K_COUNT] = {0};
};

static const char* k_names[PFT_K_COUNT] = {"resample", "aabb", "crop", "octree", "likelihood", "population", "pack"};

extern "C" const char* pft_kernel_name(int id) { return (id >= 0 && id < PFT_K_COUNT) ? k_names[id] : "?"; }

extern "C" const char* pft_status_string(int s) {
  switch (s) {
    case PFT_OK: return "ok";
    case PFT_ERR_INVALID_ARG: return "invalid argument";
    case PFT_ERR_NO_INPUT: return "no input cloud";
    case PFT_ERR_NO_REFERENCE: return "no reference cloud";
    case PFT_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case PFT_ERR_HIP: return "HIP error";
    case PFT_ERR_CAPACITY: return "capacity exceeded";
    case PFT_ERR_STATE: return "invalid state";
  }
  return "unknown";
}

extern "C" void pft_config_default(pft_config* c) {
  memset(c, 0, sizeof(*c));
  c->abi_version = PFT_ABI_VERSION;
  c->device_id = 0;
  c->stream = nullptr;
  c->stream_is_external = 0;
  c->particle_num = 400;
  c->iteration_num = 2;
  for (int k = 0; k < 6; k++) {
    c->step_noise_cov[k] = 0.015 * 0.015;
    c->initial_noise_cov[k] = 0.00001;
    c->initial_noise_mean[k] = 0.0;
  }
  c->step_noise_cov[3] *= 40.0;
  c->step_noise_cov[4] *= 40.0;
  c->step_noise_cov[5] *= 40.0;
  c->alpha = 15.0;
  c->resample_likelihood_thr = 0.0;
  c->max_distance = 0.1;
  c->octree_resolution = 0.01;
  c->distance_weight = 1.0;
  c->hsv_weight = 0.1;
  c->h_weight = 1.0;
  c->s_weight = 1.0;
  c->v_weight = 0.0;
  c->hsv_pcl180_argorder = 1;
  c->use_normal = 0;
  c->seed = 1;
  c->rank = 0;
  c->world_size = 1;
  c->kld_adaptive = 0;            // the north_star path is the fixed tracker; auto_tracking.cpp:207-222 for the KLD one
  c->maximum_particle_num = 500;  // :209
  c->kld_delta = 0.99;            // :210
  c->kld_epsilon = 0.2;           // :211
  for (int k = 0; k < 6; k++) c->kld_bin_size[k] = 0.1;  // :212-219
  c->motion_ratio = 0.25;
  c->exact_nearest = 0;  // ApproxNearestPairPointCloudCoherence, as the reference runs (:235-236)
}

// KLDAdaptiveParticleFilterTracker::normalQuantile (kld_adaptive_particle_filter.h): despite its name the polynomial
// normal CDF of CACM Algorithm 209; host-side double arithmetic, handed to the resample kernel as a constant
extern "C" double pft_kld_normal_quantile(double u) {
  static const double a[9] = {1.24818987e-4, -1.075204047e-3, 5.198775019e-3, -0.019198292004, 0.059054035642,
                              -0.151968751364, 0.319152932694, -0.5319230073, 0.797884560593};
  static const double b[15] = {-4.5255659e-5, 1.5252929e-4, -1.9538132e-5, -6.76904986e-4, 1.390604284e-3,
                               -7.9462082e-4, -2.034254874e-3, 6.549791214e-3, -0.010557625006, 0.011630447319,
                               -9.279453341e-3, 5.353579108e-3, -2.141268741e-3, 5.35310849e-4, 9.99936657524e-1};
  double w, y, z;
  if (u == 0.) return 0.5;
  y = u / 2.0;
  if (y < -6.) return 0.0;
  if (y > 6.) return 1.0;
  if (y < 0.) y = -y;
  if (y < 1.) {
    w = y * y;
    z = a[0];
    for (int i = 1; i < 9; i++) z = z * w + a[i];
    z *= (y * 2.0);
  } else {
    y -= 2.0;
    z = b[0];
    for (int i = 1; i < 15; i++) z = z * y + b[i];
  }
  if (u < 0.0) return (1. - z) / 2.0;
  return (1. + z) / 2.0;
}

extern "C" double pft_kld_bound(int k, double delta, double epsilon) {
  const double z = pft_kld_normal_quantile(delta);
  const double chi = 1.0 - 2.0 / (9.0 * (k - 1)) + sqrt(2.0 / (9.0 * (k - 1))) * z;
  return ((k - 1.0) / 2.0 / epsilon) * chi * chi * chi;
}

// ---- host-side A1 / A0 helpers (toEigenMatrix / toState), float like PCL ----
extern "C" void pft_to_matrix(const pft_particle* p, float m[16]) {
  float A = cosf(p->yaw), B = sinf(p->yaw), C = cosf(p->pitch), D = sinf(p->pitch);
  float E = cosf(p->roll), F = sinf(p->roll), DE = D * E, DF = D * F;
  m[0] = A * C;  m[1] = A * DF - B * E;  m[2] = B * F + A * DE;  m[3] = p->x;
  m[4] = B * C;  m[5] = A * E + B * DF;  m[6] = B * DE - A * F;  m[7] = p->y;
  m[8] = -D;     m[9] = C * F;           m[10] = C * E;          m[11] = p->z;
  m[12] = 0;     m[13] = 0;              m[14] = 0;              m[15] = 1;
}

extern "C" void pft_to_state(const float m[16], pft_particle* out) {
  memset(out, 0, sizeof(*out));
  out->x = m[3];
  out->y = m[7];
  out->z = m[11];
  out->w = 1.0f;
  out->roll = atan2f(m[9], m[10]);
  out->pitch = asinf(-m[8]);
  out->yaw = atan2f(m[4], m[0]);
}

// ---- profiling helpers ----
struct ProfScope {
  pft_tracker* t;
  int id;
  EvPair p;
  bool on;
  ProfScope(pft_tracker* t_, int id_) : t(t_), id(id_), on(t_->prof) {
    if (!on) return;
    if (!t->ev_free.empty()) {
      p = t->ev_free.back();
      t->ev_free.pop_back();
    } else {
      hipEventCreate(&p.a);
      hipEventCreate(&p.b);
    }
    hipEventRecord(p.a, t->stream);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(p.b, t->stream);
    t->ev[id].push_back(p);
  }
};

static void prof_collect(pft_tracker* t) {
  hipStreamSynchronize(t->stream);
  for (int k = 0; k < PFT_K_COUNT; k++) {
    for (auto& p : t->ev[k]) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        t->prof_ms[k] += ms;
        t->prof_n[k] += 1;
      }
      t->ev_free.push_back(p);
    }
    t->ev[k].clear();
  }
}

extern "C" int pft_profile_enable(pft_tracker* t, int on) {
  if (!t) return PFT_ERR_INVALID_ARG;
  if (!on && t->prof) prof_collect(t);
  t->prof = on != 0;
  return PFT_OK;
}
extern "C" int pft_profile_get(pft_tracker* t, int id, double* total_ms, uint64_t* launches) {
  if (!t || id < 0 || id >= PFT_K_COUNT) return PFT_ERR_INVALID_ARG;
  prof_collect(t);
  if (total_ms) *total_ms = t->prof_ms[id];
  if (launches) *launches = t->prof_n[id];
  return PFT_OK;
}
extern "C" int pft_profile_reset(pft_tracker* t) {
  if (!t) return PFT_ERR_INVALID_ARG;
  prof_collect(t);
  for (int k = 0; k < PFT_K_COUNT; k++) {
    t->prof_ms[k] = 0;
    t->prof_n[k] = 0;
  }
  return PFT_OK;
}

// ---- allocation ----
template <typename T>
static hipError_t dalloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T));
}
template <typename T>
static void dfree(T*& p) {
  if (p) hipFree(p);
  p = nullptr;
}

static void sync_dev(pft_tracker* t) {
  PftDev& d = t->dev;
  d.ref_xyz = t->d_ref_xyz;
  d.ref_box = t->d_ref_box;
  d.ref_hsv = t->d_ref_hsv;
  d.in_pts = t->d_in_pts;
  d.N = t->N;
  d.part_cur = t->d_part[t->cur];
  d.part_all = t->bound_gathered ? static_cast<pft_particle*>(t->bound_gathered) : t->d_part[t->cur];
  d.mats = t->d_mats;
  d.bbox_part = t->d_bbox_part;
  d.bbox_grid = (uint32_t)t->num_cus;
  d.bbox_part_cap = (uint32_t)t->bbox_part_cap;
  d.bbox6 = t->bound_bbox6 ? static_cast<float*>(t->bound_bbox6) : t->d_bbox6;
  d.crop_counts = t->d_crop_counts;
  d.crop_slots = t->d_crop_slots;
  d.crop_pts = t->d_crop_pts;
  d.crop_idx = t->d_crop_idx;
  d.words = t->d_words;
  d.max_words = t->max_words;
  d.jump = t->d_jump;
  d.ref_perm = t->d_ref_perm;
  d.leaf_pts = t->d_leaf_pts;
  d.leaf_order = t->d_leaf_order;
  d.pt_node = t->d_pt_node;
  d.pt_key = t->d_pt_key;
  d.pt_tmp = t->d_pt_tmp;
  d.pt_key64 = t->d_pt_key64;
  d.host_stat = t->h_stat;
  d.partial = t->d_partial;
  d.alias_list = t->d_alias_list;
  d.alias_pref = t->d_alias_pref;
  d.pop_part = t->d_pop_part;
  d.p_active = t->prm.kld ? &t->d_hdr->p_active : nullptr;
  d.eg_start = t->d_eg_start;
  d.eg_cnt = t->d_eg_cnt;
  d.eg_tile = t->d_eg_tile;
  d.eg_cap = t->d_eg_start ? PFT_EG_CAP : 0u;
  d.ec_slot = t->d_ec_slot;
  d.ec_cells = t->d_ec_cells;
  d.ec_count = t->d_ec_count;
  d.ec_base = t->d_ec_base;
  d.ec_list = t->d_ec_list;
  d.ec_pool_cap = t->ec_pool_cap;
  d.eq_cellq = t->d_eq_cellq;
  d.eq_nq = t->d_eq_nq;
  d.eq_qbase = t->d_eq_qbase;
  d.eq_bbase = t->d_eq_bbase;
  d.eq_fill = t->d_eq_fill;
  d.eq_blk = t->d_eq_blk;
  d.eq_sorted = t->d_eq_sorted;
  d.eq_out = t->d_eq_out;
  d.eq_cap = t->eq_cap;
  d.eq_blk_cap = t->eq_blk_cap;
  d.eq_tiles = t->d_eq_tiles;
  d.eq_tiles_cap = t->eq_tiles_cap;
  d.kld_table = t->d_kld_table;
  d.kld_bins = t->d_kld_bins;
  d.alias_pos = t->d_alias_pos;
  d.raw_w = t->d_raw_w;
  d.hdr = t->d_hdr;
  d.nn_idx = t->d_nn_idx;
  d.nn_d2 = t->d_nn_d2;
}

// Device-side failures (PftHeader::error: octree capacity, depth / growth overflow, the one-pass crop's bounded wait)
// make the likelihood launch of that iteration run without a target -- all weights zero, the update degenerates to
// the unweighted mean.  The likelihood kernel mirrors the flags into pinned host memory; every host synchronisation
// point calls this AFTER the stream has drained and hands the failure to the caller (the reference's caller looks for
// one: auto_tracking.cpp:692-696).  The flags are per iteration, so the next compute() starts clean.
static int check_device_error(pft_tracker* t) {
  volatile uint32_t* hs = t->h_stat;
  const uint32_t e = hs ? hs[3] : 0u;
  if (!e) return PFT_OK;
  hs[3] = 0u;
  std::string m = "device error flag(s) raised since the last check:";
  if (e & 1u) m += " [bit0] octree node capacity exceeded (more than 8 x input points + 64 words, or 2^24 nodes);";
  if (e & 2u) m += " [bit1] octree depth / bounding-box growth steps exceeded (PFT_MAX_DEPTH 30, PFT_MAX_GROW 40);";
  if (e & 4u) m += " [bit2] the one-pass crop gave up waiting for a predecessor workgroup;";
  if (e & 16u)
    m += " [bit4] a device-scope barrier of the population kernel timed out (its workgroups did not become co-resident "
         "within the spin limit): that iteration's normalisation, weighted mean and alias table were NOT written -- the "
         "weights, the result pose and the resampling table are those of the last completed iteration;";
  if (e & ~23u) m += " [other] " + std::to_string(e & ~23u) + ";";
  if (e & 7u) m += " the iteration(s) with bit0-2 ran without a target cloud (all likelihoods zero)";
  if (e & 16u) {  // the barrier counters of an interrupted launch are cleared before the next one (the launch resets them itself
                  // when all its workgroups get through; this covers a launch that did not)
    hipStreamSynchronize(t->stream);
    hipMemsetAsync(reinterpret_cast<char*>(t->d_hdr) + offsetof(PftHeader, pop_bar), 0, sizeof(((PftHeader*)nullptr)->pop_bar), t->stream);
  }
  t->err = m;
  return (e & (4u | 16u)) ? PFT_ERR_HIP : PFT_ERR_CAPACITY;
}

static int ensure_input_capacity(pft_tracker* t, uint32_t n) {
  if (n <= t->in_cap) return PFT_OK;
  if ((uint64_t)n * 8ull + 64ull >= (1ull << 24)) {
    t->err = "input cloud too large for 24-bit octree child offsets";
    return PFT_ERR_CAPACITY;
  }
  hipStreamSynchronize(t->stream);
  t->in_cap = 0;  // a failing allocation below leaves null buffers: they must not look usable
  dfree(t->d_in_raw); dfree(t->d_in_pts); dfree(t->d_crop_counts); dfree(t->d_crop_slots); dfree(t->d_crop_pts); dfree(t->d_crop_idx);
  dfree(t->d_words); dfree(t->d_leaf_pts); dfree(t->d_leaf_order); dfree(t->d_pt_node); dfree(t->d_pt_key);
  dfree(t->d_pt_tmp); dfree(t->d_pt_key64);
  dfree(t->sort.keys[0]); dfree(t->sort.keys[1]); dfree(t->sort.vals[0]); dfree(t->sort.vals[1]);
  dfree(t->sort.hist); dfree(t->sort.tile_cnt); dfree(t->sort.tile_box);
  uint32_t cap = n;
  t->max_words = cap * 8u + 64u;
  HIPCHK(t, dalloc(&t->d_in_raw, cap));
  HIPCHK(t, dalloc(&t->d_in_pts, cap));
  HIPCHK(t, dalloc(&t->d_crop_counts, (size_t)(cap / 1024 + 2)));
  HIPCHK(t, dalloc(&t->d_crop_slots, (size_t)(cap / 1024 + 2)));
  HIPCHK(t, hipMemsetAsync(t->d_crop_slots, 0, (size_t)(cap / 1024 + 2) * sizeof(unsigned long long), t->stream));
  HIPCHK(t, dalloc(&t->d_crop_pts, cap));
  HIPCHK(t, dalloc(&t->d_crop_idx, cap));
  HIPCHK(t, dalloc(&t->d_words, t->max_words));
  HIPCHK(t, dalloc(&t->d_leaf_pts, cap));
  HIPCHK(t, dalloc(&t->d_leaf_order, cap));
  HIPCHK(t, dalloc(&t->d_pt_node, cap));
  HIPCHK(t, dalloc(&t->d_pt_key, (size_t)cap * 3));
  HIPCHK(t, dalloc(&t->d_pt_tmp, cap));
  HIPCHK(t, dalloc(&t->d_pt_key64, cap));
  t->sort.ntiles = (cap + 1023u) / 1024u;
  HIPCHK(t, dalloc(&t->sort.keys[0], cap));
  HIPCHK(t, dalloc(&t->sort.keys[1], cap));
  HIPCHK(t, dalloc(&t->sort.vals[0], cap));
  HIPCHK(t, dalloc(&t->sort.vals[1], cap));
  HIPCHK(t, dalloc(&t->sort.hist, (size_t)256 * t->sort.ntiles + 256));
  HIPCHK(t, dalloc(&t->sort.tile_cnt, (size_t)t->sort.ntiles * (PFT_MAX_DEPTH + 2)));
  HIPCHK(t, dalloc(&t->sort.tile_box, (size_t)t->sort.ntiles * 6));
  t->in_cap = cap;
  return PFT_OK;
}

extern "C" int pft_create(const pft_config* cfg, pft_tracker** out) {
  if (!cfg || !out) return PFT_ERR_INVALID_ARG;
  *out = nullptr;
  if (cfg->abi_version != PFT_ABI_VERSION || cfg->particle_num <= 0 || cfg->iteration_num <= 0 ||
      cfg->world_size <= 0 || cfg->rank < 0 || cfg->rank >= cfg->world_size || cfg->use_normal != 0 ||
      cfg->particle_num % cfg->world_size != 0 || !(cfg->octree_resolution > 0))
    return PFT_ERR_INVALID_ARG;
  if (cfg->particle_num > PFT_MAX_PARTICLES) return PFT_ERR_CAPACITY;
  if (cfg->kld_adaptive) {
    if (cfg->world_size != 1 || cfg->maximum_particle_num <= 0 || !(cfg->kld_epsilon > 0)) return PFT_ERR_INVALID_ARG;
    for (int k = 0; k < 6; k++)
      if (!(cfg->kld_bin_size[k] > 0)) return PFT_ERR_INVALID_ARG;
    // the population stage of a KLD tracker runs in the single-workgroup kernel
    if (cfg->maximum_particle_num >= PFT_POPM_MIN || cfg->particle_num >= PFT_POPM_MIN) return PFT_ERR_CAPACITY;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device_id >= ndev) return PFT_ERR_NO_DEVICE;
  if (hipSetDevice(cfg->device_id) != hipSuccess) return PFT_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return PFT_ERR_NO_DEVICE;

  pft_tracker* t = new pft_tracker();
  t->cfg = *cfg;
  t->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (cfg->stream_is_external) {
    t->stream = static_cast<hipStream_t>(cfg->stream);
  } else {
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
      delete t;
      return PFT_ERR_HIP;
    }
    t->own_stream = true;
  }
  for (int i = 0; i < 16; i++) t->trans[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  {
    const char* ge = getenv("PFT_GRAPH");
    t->use_graph = ge && ge[0] == '1' && t->own_stream;
  }

  PftParams& p = t->prm;
  memset(&p, 0, sizeof(p));
  p.alpha = cfg->alpha;
  p.maxd2 = cfg->max_distance * cfg->max_distance;
  p.res = cfg->octree_resolution;
  p.dist_w = cfg->distance_weight;
  p.hsv_w = cfg->hsv_weight;
  p.h_w = (float)cfg->h_weight;
  p.s_w = (float)cfg->s_weight;
  p.v_w = (float)cfg->v_weight;
  p.hsv_argorder = cfg->hsv_pcl180_argorder;
  for (int k = 0; k < 6; k++) {
    p.step_sigma[k] = sqrt(cfg->step_noise_cov[k]);
    p.init_sigma[k] = sqrt(cfg->initial_noise_cov[k]);
    p.init_mean[k] = cfg->initial_noise_mean[k];
  }
  p.seed_lo = (uint32_t)cfg->seed;
  p.seed_hi = (uint32_t)(cfg->seed >> 32);
  p.P_total = (uint32_t)cfg->particle_num;
  p.P_local = p.P_total / (uint32_t)cfg->world_size;
  p.id_offset = p.P_local * (uint32_t)cfg->rank;
  p.M = 0;
  p.nchunk = 1;
  p.ref_chunk = PFT_REF_CHUNK;
  p.split_last = 0;
  p.kld = cfg->kld_adaptive ? 1u : 0u;
  p.kld_max = (uint32_t)cfg->maximum_particle_num;
  p.kld_z = pft_kld_normal_quantile(cfg->kld_delta);
  p.kld_eps = cfg->kld_epsilon;
  for (int k = 0; k < 6; k++) p.kld_bin[k] = (float)cfg->kld_bin_size[k];
  p.motion_ratio = cfg->motion_ratio;
  // capacity in particles: the KLD variant grows / shrinks between particle_num and maximum_particle_num
  t->Pcap = p.kld && p.kld_max > p.P_total ? p.kld_max : p.P_total;

  const size_t Pl = p.kld ? t->Pcap : p.P_local, Pt = t->Pcap;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(dalloc(&t->d_part[0], Pt));  // sized P_total so part_all can alias a shard buffer when world_size == 1
  A(dalloc(&t->d_part[1], Pt));
  A(dalloc(&t->d_mats, Pl * 12));
  t->bbox_part_cap = std::max<size_t>((size_t)t->num_cus, (Pl * 4u + 255u) / 256u);  // (one partial per workgroup of k_aabb / of the fused resample)
  A(dalloc(&t->d_bbox_part, t->bbox_part_cap * 6));
  A(dalloc(&t->d_bbox6, 8));
  A(dalloc(&t->d_jump, (size_t)1 << (3 * PFT_JUMP_MAX_LEVEL)));
  A(dalloc(&t->d_alias_list, 2 * Pt));
  A(dalloc(&t->d_alias_pref, 2 * Pt));
  A(dalloc(&t->d_pop_part, (size_t)PFT_POPM_MAX_WGS * 16));
  A(dalloc(&t->d_alias_pos, Pt));
  A(dalloc(&t->d_raw_w, Pt));
  if (cfg->exact_nearest) {
    A(dalloc(&t->d_eg_start, (size_t)PFT_EG_CAP + 1));
    A(dalloc(&t->d_eg_cnt, (size_t)PFT_EG_CAP));
    A(dalloc(&t->d_eg_tile, (size_t)PFT_EG_CAP / 2048 + 2));
    A(dalloc(&t->d_ec_slot, (size_t)PFT_EG_CAP));
    A(dalloc(&t->d_ec_cells, (size_t)PFT_EC_SLOTS));
    A(dalloc(&t->d_ec_count, (size_t)PFT_EC_SLOTS));
    A(dalloc(&t->d_ec_base, (size_t)PFT_EC_SLOTS));
    t->ec_pool_cap = PFT_EC_POOL_MIN;
    A(dalloc(&t->d_ec_list, (size_t)t->ec_pool_cap));
    A(dalloc(&t->d_eq_cellq, (size_t)PFT_EG_CAP));
    A(dalloc(&t->d_eq_nq, (size_t)PFT_EC_SLOTS));
    A(dalloc(&t->d_eq_qbase, (size_t)PFT_EC_SLOTS));
    A(dalloc(&t->d_eq_bbase, (size_t)PFT_EC_SLOTS));
    A(dalloc(&t->d_eq_fill, (size_t)PFT_EC_SLOTS));
  }
  if (p.kld) {
    A(dalloc(&t->d_kld_table, (size_t)6 * p.kld_max + 128));
    A(dalloc(&t->d_kld_bins, (size_t)6 * p.kld_max));
  }
  A(hipHostMalloc(reinterpret_cast<void**>(&t->h_stat), 8 * sizeof(uint32_t), hipHostMallocMapped));
  if (t->h_stat) for (int i = 0; i < 8; i++) t->h_stat[i] = 0;  // [0..3]: pft_debug_get_host_stat; [4]: exact-NN pool demand
  {
    const char* fb = getenv("PFT_FORCE_BUILDER");
    t->force_builder = fb ? (!strcmp(fb, "single") ? 1 : (!strcmp(fb, "sorted") ? 2 : 0)) : 0;
  }
  A(dalloc(&t->d_hdr, 1));
  A(dalloc(&t->d_dbg_hdr, 1));
  if (e == hipSuccess) e = hipMemsetAsync(t->d_hdr, 0, sizeof(PftHeader), t->stream);
  if (e == hipSuccess) e = hipMemsetAsync(t->d_dbg_hdr, 0, sizeof(PftHeader), t->stream);
  if (e != hipSuccess) {
    pft_destroy(t);
    return PFT_ERR_HIP;
  }
  if (cfg->max_input_points) {
    int r = ensure_input_capacity(t, cfg->max_input_points);
    if (r != PFT_OK) {
      pft_destroy(t);
      return r;
    }
  }
  sync_dev(t);
  *out = t;
  return PFT_OK;
}

extern "C" void pft_destroy(pft_tracker* t) {
  if (!t) return;
  if (t->stream) hipStreamSynchronize(t->stream);
  if (t->graph_exec) hipGraphExecDestroy(t->graph_exec);
  for (int k = 0; k < PFT_K_COUNT; k++)
    for (auto& p : t->ev[k]) {
      hipEventDestroy(p.a);
      hipEventDestroy(p.b);
    }
  for (auto& p : t->ev_free) {
    hipEventDestroy(p.a);
    hipEventDestroy(p.b);
  }
  dfree(t->d_ref_raw); dfree(t->d_ref_xyz); dfree(t->d_ref_hsv); dfree(t->d_ref_box);
  dfree(t->d_in_raw); dfree(t->d_in_pts);
  dfree(t->d_part[0]); dfree(t->d_part[1]); dfree(t->d_mats); dfree(t->d_bbox_part); dfree(t->d_bbox6);
  dfree(t->d_crop_counts); dfree(t->d_crop_slots); dfree(t->d_crop_pts); dfree(t->d_crop_idx); dfree(t->d_words); dfree(t->d_jump); dfree(t->d_ref_perm);
  dfree(t->d_leaf_pts); dfree(t->d_leaf_order); dfree(t->d_pt_node); dfree(t->d_pt_key); dfree(t->d_pt_tmp);
  dfree(t->d_pt_key64); dfree(t->sort.keys[0]); dfree(t->sort.keys[1]); dfree(t->sort.vals[0]); dfree(t->sort.vals[1]);
  dfree(t->sort.hist); dfree(t->sort.tile_cnt); dfree(t->sort.tile_box); if (t->h_stat) hipHostFree(t->h_stat);
  dfree(t->d_partial); dfree(t->d_alias_list); dfree(t->d_alias_pos); dfree(t->d_raw_w);
  dfree(t->d_alias_pref); dfree(t->d_pop_part); dfree(t->d_kld_table); dfree(t->d_kld_bins); dfree(t->d_eg_start); dfree(t->d_eg_cnt); dfree(t->d_eg_tile); dfree(t->d_ec_slot); dfree(t->d_ec_cells); dfree(t->d_ec_count); dfree(t->d_ec_base); dfree(t->d_ec_list); dfree(t->d_eq_cellq); dfree(t->d_eq_nq); dfree(t->d_eq_qbase); dfree(t->d_eq_bbase); dfree(t->d_eq_fill); dfree(t->d_eq_blk); dfree(t->d_eq_tiles); dfree(t->d_eq_sorted); dfree(t->d_eq_out); dfree(t->d_hdr); dfree(t->d_nn_idx); dfree(t->d_nn_d2); dfree(t->d_dbg_part);
  dfree(t->d_dbg_hdr); dfree(t->d_dbg_f);
  dfree(t->sv_part); dfree(t->sv_alias_list); dfree(t->sv_alias_pref); dfree(t->sv_alias_pos); dfree(t->sv_hdr);
  if (t->own_stream && t->stream) hipStreamDestroy(t->stream);
  delete t;
}

extern "C" const char* pft_last_error_string(const pft_tracker* t) { return t ? t->err.c_str() : "null handle"; }

extern "C" int pft_synchronize(pft_tracker* t) {
  if (!t) return PFT_ERR_INVALID_ARG;
  HIPCHK(t, hipStreamSynchronize(t->stream));
  return check_device_error(t);
}

extern "C" int pft_set_reference(pft_tracker* t, const pft_point_xyzrgba* pts, size_t n) {
  if (!t || (!pts && n)) return PFT_ERR_INVALID_ARG;
  if (n > 0x7fffffffu) return PFT_ERR_CAPACITY;
  hipSetDevice(t->cfg.device_id);
  if (n > t->ref_cap) {
    hipStreamSynchronize(t->stream);
    dfree(t->d_ref_raw); dfree(t->d_ref_xyz); dfree(t->d_ref_hsv); dfree(t->d_ref_box); dfree(t->d_partial); dfree(t->d_ref_perm);
    HIPCHK(t, dalloc(&t->d_ref_raw, n));
    HIPCHK(t, dalloc(&t->d_ref_perm, n));
    HIPCHK(t, dalloc(&t->d_ref_xyz, n));
    HIPCHK(t, dalloc(&t->d_ref_hsv, n));
    HIPCHK(t, dalloc(&t->d_ref_box, n));
    t->ref_cap = (uint32_t)n;
  }
  t->prm.M = (uint32_t)n;
  {
    // Work item of the likelihood kernels = (particle, ref_chunk reference points).  256 points amortise the per-item
    // cost best when there are plenty of items; with few particles (the reference's own 400-500) smaller items keep
    // all CUs busy: aim at two items per resident wave.
    const uint64_t pl = t->prm.kld ? t->Pcap : t->prm.P_local;
    const uint64_t want_items = 2ull * (uint64_t)PFT_LIK_WGS_PER_CU * (uint64_t)t->num_cus * (PFT_LIK_THREADS / 64u);
    uint64_t c = pl * (uint64_t)n / (want_items ? want_items : 1);
    c = (c / 64u) * 64u;
    if (c < 64u) c = 64u;
    if (c > PFT_REF_CHUNK) c = PFT_REF_CHUNK;
    t->prm.ref_chunk = (uint32_t)c;
  }
  t->prm.nchunk = (uint32_t)((n + t->prm.ref_chunk - 1) / t->prm.ref_chunk);
  if (t->prm.nchunk == 0) t->prm.nchunk = 1;
  // the items a launch ends with decide how long its last waves run alone: the last chunk is cut in two halves, and
  // the halves of all particles are handed out after the full-size items (approximate search only)
  t->prm.split_last = 0;
  if (!t->cfg.exact_nearest && t->prm.ref_chunk >= 128u && n % t->prm.ref_chunk == 0 && t->prm.nchunk >= 2u) {
    t->prm.split_last = t->prm.ref_chunk >= 256u ? 4u : 2u;  // sub-items of at least 64 points (one round of a wave)
    t->prm.nchunk += t->prm.split_last - 1u;
  }
  dfree(t->d_partial);
  HIPCHK(t, dalloc(&t->d_partial, (size_t)(t->prm.kld ? t->Pcap : t->prm.P_local) * t->prm.nchunk));
  if (n) {
    // The reference cloud is stored in Morton (Z-curve) order: the 64 lanes of a wave then query
    // neighbouring space, so their octree paths and leaf records share LDS words and cache lines.  Every
    // per-particle result is a sum over all reference points, so the order is free; ref_perm maps back.
    std::vector<uint32_t> perm(n);
    std::vector<pft_point_xyzrgba> sorted(n);
    {
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (size_t i = 0; i < n; i++) {
        const float c[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int a = 0; a < 3; a++)
          if (std::isfinite(c[a])) {
            lo[a] = std::min(lo[a], c[a]);
            hi[a] = std::max(hi[a], c[a]);
          }
      }
      float ext = 0.0f;
      for (int a = 0; a < 3; a++)
        if (hi[a] >= lo[a]) ext = std::max(ext, hi[a] - lo[a]);
      const float scale = ext > 0.0f ? 1023.0f / ext : 0.0f;
      std::vector<uint64_t> code(n);
      for (size_t i = 0; i < n; i++) {
        const float c[3] = {pts[i].x, pts[i].y, pts[i].z};
        uint64_t m = 0;
        uint32_t q[3];
        for (int a = 0; a < 3; a++) {
          float v = std::isfinite(c[a]) ? (c[a] - lo[a]) * scale : 0.0f;
          q[a] = (uint32_t)std::min(1023.0f, std::max(0.0f, v));
        }
        for (int b = 9; b >= 0; b--)
          m = (m << 3) | (uint64_t)((((q[0] >> b) & 1u) << 2) | (((q[1] >> b) & 1u) << 1) | ((q[2] >> b) & 1u));
        code[i] = (m << 32) | (uint64_t)i;  // index in the low bits: stable
      }
      std::sort(code.begin(), code.end());
      for (size_t i = 0; i < n; i++) {
        perm[i] = (uint32_t)(code[i] & 0xffffffffu);
        sorted[i] = pts[perm[i]];
      }
    }
    pts = sorted.data();
    HIPCHK(t, hipMemcpyAsync(t->d_ref_perm, perm.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    HIPCHK(t, hipMemcpyAsync(t->d_ref_raw, pts, n * sizeof(pft_point_xyzrgba), hipMemcpyHostToDevice, t->stream));
    pftk_pack_reference(t->stream, t->d_ref_raw, (uint32_t)n, t->prm.hsv_argorder, t->d_ref_xyz, t->d_ref_hsv);
    // A3's input: the points that can be extreme in some rigidly transformed coordinate (the hull's vertices and what lies
    // within the float evaluation's reach of its facets); PFT_AABB_FULL=1 keeps every point (cross-check, A/B timing)
    std::vector<uint32_t> keep;
    if (getenv("PFT_AABB_FULL")) {
      keep.resize(n);
      for (size_t i = 0; i < n; i++) keep[i] = (uint32_t)i;
    } else {
      pft_aabb_support_subset(pts, n, keep);
    }
    std::vector<float4> box(keep.size());
    for (size_t i = 0; i < keep.size(); i++) box[i] = make_float4(pts[keep[i]].x, pts[keep[i]].y, pts[keep[i]].z, 0.0f);
    t->prm.M_box = (uint32_t)keep.size();
    HIPCHK(t, hipMemcpyAsync(t->d_ref_box, box.data(), box.size() * sizeof(float4), hipMemcpyHostToDevice, t->stream));
    HIPCHK(t, hipStreamSynchronize(t->stream));
  } else {
    t->prm.M_box = 0;
  }
  t->has_ref = true;
  if (t->h_stat) t->h_stat[0] = t->h_stat[1] = 0;  // new scene: forget the builder hints (unknown depth = all radix passes)
  sync_dev(t);
  return PFT_OK;
}

extern "C" int pft_set_trans(pft_tracker* t, const float m[16]) {
  if (!t || !m) return PFT_ERR_INVALID_ARG;
  memcpy(t->trans, m, sizeof(float) * 16);
  return PFT_OK;
}

static int set_input_common(pft_tracker* t, const void* src, size_t n, bool device) {
  if (!t || (!src && n)) return PFT_ERR_INVALID_ARG;
  if (n > 0x7fffffffu) return PFT_ERR_CAPACITY;
  hipSetDevice(t->cfg.device_id);
  int r = ensure_input_capacity(t, (uint32_t)n);
  if (r != PFT_OK) return r;
  t->N = (uint32_t)n;
  if (n) {
    const pft_point_xyzrgba* dsrc = static_cast<const pft_point_xyzrgba*>(src);
    if (!device) {
      HIPCHK(t, hipMemcpyAsync(t->d_in_raw, src, n * sizeof(pft_point_xyzrgba), hipMemcpyHostToDevice, t->stream));
      dsrc = t->d_in_raw;
    }
    // the 16-byte records (d_in_pts) are formed by the first crop of the frame, which reads the 32-byte layout
    t->raw_pending = dsrc;
    if (!device) HIPCHK(t, hipStreamSynchronize(t->stream));  // the host buffer is only borrowed for this call
  }
  if (!n) t->raw_pending = nullptr;
  t->has_input = n > 0;
  sync_dev(t);
  return PFT_OK;
}

extern "C" int pft_set_input(pft_tracker* t, const pft_point_xyzrgba* pts, size_t n) {
  return set_input_common(t, pts, n, false);
}
extern "C" int pft_set_input_device(pft_tracker* t, const void* device_pts, size_t n) {
  return set_input_common(t, device_pts, n, true);
}

// ---- the stages ----
static void stage_init_particles(pft_tracker* t) {
  pft_particle rep;
  pft_to_state(t->trans, &rep);
  rep.weight = 1.0f / (float)t->prm.P_total;
  sync_dev(t);
  ProfScope ps(t, PFT_K_RESAMPLE);
  pftk_init_particles(t->stream, t->prm, rep, t->d_part[t->cur], t->d_mats, t->d_hdr);
  t->initialized = true;
  t->changed = false;
  t->resample_epoch = 0;
}

static void stage_resample(pft_tracker* t) {
  sync_dev(t);
  pft_particle* out = t->d_part[1 - t->cur];
  {
    ProfScope ps(t, PFT_K_RESAMPLE);
    if (t->prm.kld)
      pftk_resample_kld(t->stream, t->prm, t->dev, t->resample_epoch, out, nullptr, nullptr, nullptr);
    else
      pftk_resample(t->stream, t->prm, t->dev, t->resample_epoch, out);
  }
  t->resample_epoch++;
  t->cur = 1 - t->cur;
  sync_dev(t);
}

// A2+A3 (fused; transformed clouds are never materialised)
static void stage_aabb(pft_tracker* t, const PftDev& d, uint32_t np, bool finalize) {
  ProfScope ps(t, PFT_K_AABB);
  pftk_aabb(t->stream, t->prm, d, np, finalize);
}
// A11 + A1 + A2 + A3 of a steady-state iteration of the fixed-size tracker: ONE launch when the box's support subset is
// small (pft_hull.hip), the quads that draw the particles fold their boxes; PFT_SPLIT_RESAMPLE=1 keeps the two launches
// (cross-check -- identical bits --, A/B timing).  The KLD variant, whose resample is a grid-wide loop of its own, and the
// first iteration after pft_set_particles / init take the separate kernels.
static void stage_resample_aabb(pft_tracker* t, bool finalize) {
  // (read per call, not latched: tests/test_gpu_parity.py toggles them inside one process)
  const bool split = getenv("PFT_SPLIT_RESAMPLE") != nullptr || getenv("PFT_RESAMPLE_ONE_LANE") != nullptr;
  if (!t->prm.kld && t->changed && !split) {
    sync_dev(t);
    uint32_t nparts;
    {
      ProfScope ps(t, PFT_K_RESAMPLE);
      nparts = pftk_resample_box(t->stream, t->prm, t->dev, t->resample_epoch, t->d_part[1 - t->cur]);
    }
    if (nparts) {
      t->resample_epoch++;
      t->cur = 1 - t->cur;
      sync_dev(t);
      t->dev.bbox_grid = nparts;  // (the consumers of the partials: the crop kernel, k_bbox_final)
      if (finalize) pftk_bbox_final(t->stream, t->dev);
      return;
    }
  }
  if (t->changed) stage_resample(t);
  sync_dev(t);
  stage_aabb(t, t->dev, t->prm.kld ? t->Pcap : t->prm.P_local, finalize);
}
// A4, A5, A6+A7
__global__ void k_inject_error(PftHeader* hdr, uint32_t bits) { hdr->error |= bits; }

static void stage_crop_octree_likelihood(pft_tracker* t, const PftDev& d, uint32_t np, bool debug_nn,
                                         bool bbox_from_partials, bool keep_point_keys = false) {
  {
    ProfScope ps(t, PFT_K_CROP);
    if (++t->crop_epoch == 0) t->crop_epoch = 1;
    pftk_crop(t->stream, t->prm, d, bbox_from_partials, t->crop_epoch, t->raw_pending);
    t->raw_pending = nullptr;
    if (t->inject_error & ~16u) {  // test hook: what a failing crop / builder would leave behind (bit 4 belongs to the
                                   // population launch: pft_compute raises it there)
      hipLaunchKernelGGL(k_inject_error, dim3(1), dim3(1), 0, t->stream, t->d_hdr, t->inject_error & ~16u);
      t->inject_error &= 16u;
    }
  }
  if (t->cfg.exact_nearest) {  // NearestPairPointCloudCoherence: uniform grid + true nearest neighbour, no octree
    PftDev dq = d;
    {
      // the cell-sorted query arrays grow to the largest evaluation seen (16 + 8 bytes per query, up to 2^30 queries;
      // beyond that, or if the allocation fails, the per-query kernel serves)
      const unsigned long long nq = (unsigned long long)np * t->prm.M;
      if (nq > t->eq_cap && nq <= (1ull << 30)) {
        dfree(t->d_eq_sorted); dfree(t->d_eq_out); dfree(t->d_eq_blk); dfree(t->d_eq_tiles);
        t->eq_cap = t->eq_blk_cap = t->eq_tiles_cap = 0;
        const size_t nblk = (size_t)(nq / 64u) + PFT_EC_SLOTS + 64u;
        size_t ppw = ((size_t)np + (size_t)t->num_cus - 1u) / (size_t)t->num_cus;  // as pftk_likelihood_exact tiles the particles
        ppw = ppw < 1u ? 1u : (ppw > 32u ? 32u : ppw);
        const size_t ntiles = ((size_t)np + ppw - 1u) / ppw + 1u;
        if (dalloc(&t->d_eq_sorted, (size_t)nq) == hipSuccess && dalloc(&t->d_eq_out, (size_t)nq) == hipSuccess &&
            dalloc(&t->d_eq_blk, nblk) == hipSuccess) {
          t->eq_cap = (uint32_t)nq;
          t->eq_blk_cap = (uint32_t)nblk;
          // the per-tile count tables are optional (k_eq_scatter counts its tile again without them)
          const size_t want = ntiles < 4096u ? ntiles : 4096u;
          if (dalloc(&t->d_eq_tiles, want * 2u * 8192u) == hipSuccess) t->eq_tiles_cap = (uint32_t)want;
          else (void)hipGetLastError();
        } else {
          (void)hipGetLastError();
          dfree(t->d_eq_sorted); dfree(t->d_eq_out); dfree(t->d_eq_blk);
        }
      }
      // the candidate pool follows the demand of the previous iteration (pinned status word 4, read without
      // synchronising; a pool that is too small costs time -- its overflow cells keep the ring search -- never the result)
      {
        const volatile uint32_t* hs = t->h_stat;
        const uint32_t demand = hs ? hs[4] : 0u;
        if (demand > t->ec_pool_cap && t->ec_pool_cap < PFT_EC_POOL) {
          unsigned long long want = (unsigned long long)demand + demand / 4u;  // a quarter of head-room
          uint32_t cap = t->ec_pool_cap;
          while (cap < want && cap < PFT_EC_POOL) cap <<= 1;
          hipStreamSynchronize(t->stream);
          float4* grown = nullptr;
          if (dalloc(&grown, (size_t)cap) == hipSuccess) {
            dfree(t->d_ec_list);
            t->d_ec_list = grown;
            t->ec_pool_cap = cap;
          } else {
            (void)hipGetLastError();  // keep the smaller pool
          }
        }
        dq.ec_list = t->d_ec_list;
        dq.ec_pool_cap = t->ec_pool_cap;
        t->dev.ec_list = t->d_ec_list;
        t->dev.ec_pool_cap = t->ec_pool_cap;
      }
      dq.eq_sorted = t->d_eq_sorted;
      dq.eq_out = t->d_eq_out;
      dq.eq_blk = t->d_eq_blk;
      dq.eq_cap = t->eq_cap;
      dq.eq_blk_cap = t->eq_blk_cap;
      dq.eq_tiles = t->d_eq_tiles;
      dq.eq_tiles_cap = t->eq_tiles_cap;
    }
    {
      ProfScope ps(t, PFT_K_OCTREE);
      pftk_exact_grid(t->stream, t->prm, dq);  // (dq: the copy that carries the current candidate pool)
    }
    ProfScope ps(t, PFT_K_LIKELIHOOD);
    pftk_likelihood_exact(t->stream, t->prm, dq, np, debug_nn, t->num_cus);
    return;
  }
  bool leaf_indirect = false;  // (the sorted builder and the rescue launch behind it always write the leaf records)
  {
    ProfScope ps(t, PFT_K_OCTREE);
    PftDev db = d;
    if (!keep_point_keys) db.pt_key = nullptr;  // the per-point keys are a test hook (pft_debug_get_point_keys after pft_eval_weights)
    // the single-workgroup builder is fastest for small crops, the sorted many-workgroup builder scales; both
    // are correct for any size.  The choice uses the crop size / depth of the PREVIOUS iteration, read from
    // pinned memory without synchronising.
    const volatile uint32_t* hs = t->h_stat;
    const uint32_t last_n = hs ? hs[0] : 0u, last_depth = hs ? hs[1] : 0u;
    bool sorted = last_n > PFT_SORTED_BUILD_MIN;
    if (t->force_builder == 1) sorted = false;
    if (t->force_builder == 2) sorted = true;
#ifdef PFT_DIAG
    // diagnostic build only (results are wrong while set): PFT_DEBUG_SKIP_OCTREE=1 reuses the tree of the
    // previous build after the first 8 builds, to measure the builder's true share of a frame
    static const bool skip_env = getenv("PFT_DEBUG_SKIP_OCTREE") != nullptr;
    if (skip_env && ++t->dbg_builds > 8) {
    } else
#endif
    if (sorted) {
      // 8-bit passes for 3 bits per level; one level of head-room over the last depth (k_so_scan flags an error
      // if the tree turned out deeper than the passes cover)
      // (a deeper tree than the passes cover is rebuilt by the rescue launch behind the sorted builder: the guess
      // costs time when it is wrong, never the iteration)
      int npass = last_depth > 0u ? (int)((3u * (last_depth + 1u) + 7u) / 8u) : 8;
      if (t->force_npass > 0) npass = t->force_npass;
      if (npass > 8) npass = 8;
      pftk_octree_sorted(t->stream, t->prm, db, t->sort, d.N, npass);
    }
    else
      // leaf records followed through leaf_order by the likelihood kernel instead of being copied: +0.32 ps per query
      // there (5.4 us at 8 192 x 2 048), -3 us per build and one launch less here: pays below ~9 million queries (the
      // reference's own 400-500 particles: 0.202 -> 0.196 ms per frame)
      leaf_indirect = pftk_octree(t->stream, t->prm, db, last_n, (unsigned long long)np * t->prm.M <= 8000000ull);
  }
  {
    ProfScope ps(t, PFT_K_LIKELIHOOD);
    pftk_likelihood(t->stream, t->prm, d, np, debug_nn, t->num_cus, leaf_indirect);
  }
}

static int check_ready(pft_tracker* t) {
  if (!t) return PFT_ERR_INVALID_ARG;
  hipSetDevice(t->cfg.device_id);
  if (!t->has_input || t->N == 0) return PFT_ERR_NO_INPUT;  // PCL: PCL_ERROR + early return
  if (!t->has_ref) return PFT_ERR_NO_REFERENCE;
  return PFT_OK;
}

extern "C" int pft_compute(pft_tracker* t) {
  int r = check_ready(t);
  if (r != PFT_OK) return r;
  if (t->cfg.world_size != 1) {
    t->err = "pft_compute on a sharded handle: drive the pft_dist_* phases instead";
    return PFT_ERR_STATE;
  }
  if (!t->initialized) stage_init_particles(t);
  // Steady-state frames as ONE graph launch (opt-in): every launch below is recorded instead of issued, and the recorded
  // graph updates the instantiated one in place (kernel arguments such as the epochs and the particle-buffer parity
  // change from frame to frame, the node sequence only when the builder choice does: then it is instantiated anew).
  // The first frames run directly: they set the kernels' one-time attributes, which must not happen during capture.
  // Nothing that allocates or sets a one-time kernel attribute may run during capture: the exact-NN mode (its query
  // arrays grow with np * M inside the stage) is never captured, and a capture that fails for any other reason -- a
  // builder or rescue kernel whose attribute is set on first use, say -- switches the handle back to direct launches
  // and runs THIS frame directly: the host-side state the recorded pass advanced is put back first.
  auto run_iterations = [&]() {
    for (int it = 0; it < t->cfg.iteration_num; it++) {
      stage_resample_aabb(t, false);
      // KLD variant: launches are sized for the capacity, the kernels take particle_num_ from PftHeader::p_active
      const uint32_t np = t->prm.kld ? t->Pcap : t->prm.P_local;
      stage_crop_octree_likelihood(t, t->dev, np, false, true);
      if (t->inject_error & 16u) {  // test hook: a population barrier that timed out in some workgroup
        hipLaunchKernelGGL(k_inject_error, dim3(1), dim3(1), 0, t->stream, t->d_hdr, 16u);
        t->inject_error &= ~16u;
      }
      {
        ProfScope ps(t, PFT_K_POPULATION);
        // raw weights from the partial sums, then weight()'s normalizeWeight(); use_change_detector_ == false
        // => changed_ = true => update(); the alias prefix form feeds the next resample: one launch
        pftk_population(t->stream, t->prm, t->dev, t->prm.kld ? t->Pcap : t->prm.P_total, 1, 1, 1, 1);
      }
      t->changed = true;
    }
  };
  const bool graphed = t->use_graph && !t->cfg.exact_nearest && t->changed && !t->prof && t->graph_frames++ >= 2u;
  if (!graphed) {
    run_iterations();
  } else {
    const int cur0 = t->cur;
    const uint32_t epoch0 = t->resample_epoch, crop0 = t->crop_epoch, grid0 = t->dev.bbox_grid;
    const pft_point_xyzrgba* raw0 = t->raw_pending;
    hipError_t ge = hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal);
    hipGraph_t g = nullptr;
    if (ge == hipSuccess) {
      run_iterations();
      ge = hipStreamEndCapture(t->stream, &g);
    }
    if (ge == hipSuccess && g) {
      bool ready = false;
      if (t->graph_exec) {
        hipGraphNode_t bad = nullptr;
        hipGraphExecUpdateResult res;
        ready = hipGraphExecUpdate(t->graph_exec, g, &bad, &res) == hipSuccess;
        if (!ready) {
          (void)hipGetLastError();
          hipGraphExecDestroy(t->graph_exec);
          t->graph_exec = nullptr;
        }
      }
      if (!ready) {
        ge = hipGraphInstantiate(&t->graph_exec, g, nullptr, nullptr, 0);
        t->graph_rebuilds++;
        ready = ge == hipSuccess;
      }
      if (ready) ge = hipGraphLaunch(t->graph_exec, t->stream);
    } else if (ge == hipSuccess) {
      ge = hipErrorUnknown;
    }
    if (g) hipGraphDestroy(g);
    if (ge != hipSuccess) {  // none of the frame's work was issued: back to direct launches, for this frame and for good
      (void)hipGetLastError();
      if (t->graph_exec) {
        hipGraphExecDestroy(t->graph_exec);
        t->graph_exec = nullptr;
      }
      t->use_graph = false;
      t->cur = cur0;
      t->resample_epoch = epoch0;
      t->crop_epoch = crop0;
      t->dev.bbox_grid = grid0;
      t->raw_pending = raw0;
      sync_dev(t);
      run_iterations();
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    t->err = std::string("kernel launch: ") + hipGetErrorString(e);
    return PFT_ERR_HIP;
  }
  return PFT_OK;
}

// ---- multi-GPU phases ----
extern "C" int pft_dist_bind(pft_tracker* t, void* bbox6_dev, void* shard_dev, void* gathered_dev) {
  if (!t || !bbox6_dev || !shard_dev || !gathered_dev) return PFT_ERR_INVALID_ARG;
  t->bound_bbox6 = bbox6_dev;
  t->bound_shard = shard_dev;
  t->bound_gathered = gathered_dev;
  sync_dev(t);
  return PFT_OK;
}

extern "C" int pft_dist_begin_frame(pft_tracker* t) {
  int r = check_ready(t);
  if (r != PFT_OK) return r;
  if (!t->bound_gathered) {
    t->err = "pft_dist_bind not called";
    return PFT_ERR_STATE;
  }
  if (!t->initialized) stage_init_particles(t);
  return PFT_OK;
}

extern "C" int pft_dist_phase_a(pft_tracker* t, int iteration) {
  (void)iteration;
  int r = check_ready(t);
  if (r != PFT_OK) return r;
  if (!t->initialized) return PFT_ERR_STATE;
  stage_resample_aabb(t, true);
  return PFT_OK;
}

extern "C" int pft_dist_phase_b(pft_tracker* t) {
  int r = check_ready(t);
  if (r != PFT_OK) return r;
  sync_dev(t);
  stage_crop_octree_likelihood(t, t->dev, t->prm.P_local, false, false);
  {
    ProfScope ps(t, PFT_K_POPULATION);
    // raw weights from the partial sums, written with their particles straight into the all-gather's send buffer
    pftk_finalize_raw(t->stream, t->prm, t->dev, t->prm.P_local, nullptr, static_cast<pft_particle*>(t->bound_shard));
  }
  return PFT_OK;
}

extern "C" int pft_dist_phase_c(pft_tracker* t) {
  int r = check_ready(t);
  if (r != PFT_OK) return r;
  sync_dev(t);
  {
    ProfScope ps(t, PFT_K_POPULATION);
    pftk_population(t->stream, t->prm, t->dev, t->prm.P_total, 0, 1, 1, 1);
  }
  t->changed = true;
  return PFT_OK;
}

// ---- accessors ----
extern "C" int pft_get_result(pft_tracker* t, pft_particle* out) {
  if (!t || !out) return PFT_ERR_INVALID_ARG;
  HIPCHK(t, hipMemcpyAsync(out, &t->d_hdr->rep, sizeof(pft_particle), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  return check_device_error(t);  // the pose is written either way; a failed iteration makes it the unweighted mean
}

extern "C" int pft_get_fit_ratio(pft_tracker* t, double* out) {
  if (!t || !out) return PFT_ERR_INVALID_ARG;
  HIPCHK(t, hipMemcpyAsync(out, &t->d_hdr->fit_ratio, sizeof(double), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  return check_device_error(t);
}

extern "C" int pft_get_particles(pft_tracker* t, pft_particle* out, size_t cap, size_t* n) {
  if (!t) return PFT_ERR_INVALID_ARG;
  size_t P = t->initialized ? t->prm.P_total : 0;
  if (P && t->prm.kld) {  // particle_num_ of the KLD variant lives on the device
    uint32_t pa = 0;
    sync_dev(t);
    HIPCHK(t, hipMemcpyAsync(&pa, &t->d_hdr->p_active, sizeof(pa), hipMemcpyDeviceToHost, t->stream));
    HIPCHK(t, hipStreamSynchronize(t->stream));
    P = pa;
  }
  if (n) *n = P;
  if (!out || !P) return PFT_OK;
  size_t c = cap < P ? cap : P;
  sync_dev(t);
  HIPCHK(t, hipMemcpyAsync(out, t->dev.part_all, c * sizeof(pft_particle), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  return check_device_error(t);
}

extern "C" int pft_set_particles(pft_tracker* t, const pft_particle* p, size_t n) {
  if (!t || !p) return PFT_ERR_INVALID_ARG;
  if (t->prm.kld ? (n == 0 || n > t->Pcap) : (n != t->prm.P_local)) return PFT_ERR_INVALID_ARG;
  HIPCHK(t, hipMemcpyAsync(t->d_part[t->cur], p, n * sizeof(pft_particle), hipMemcpyHostToDevice, t->stream));
  if (t->prm.kld) {  // particle_num_ of the KLD variant lives on the device
    const uint32_t pa = (uint32_t)n;
    HIPCHK(t, hipMemcpyAsync(&t->d_hdr->p_active, &pa, sizeof(pa), hipMemcpyHostToDevice, t->stream));
    HIPCHK(t, hipStreamSynchronize(t->stream));
  }
  pftk_pose_to_matrix(t->stream, t->d_part[t->cur], (uint32_t)n, t->d_mats);
  pft_particle rep;
  pft_to_state(t->trans, &rep);
  rep.weight = 1.0f / (float)t->prm.P_total;
  HIPCHK(t, hipMemcpyAsync(&t->d_hdr->rep, &rep, sizeof(rep), hipMemcpyHostToDevice, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  t->initialized = true;
  t->changed = false;
  if (t->h_stat) t->h_stat[0] = t->h_stat[1] = 0;  // the spread of the particles decides the crop: forget the builder hints
  return PFT_OK;
}

// ---- test hooks ----
extern "C" int pft_debug_set_limits(pft_tracker* t, uint32_t max_words, int sorted_npass) {
  if (!t) return PFT_ERR_INVALID_ARG;
  hipStreamSynchronize(t->stream);
  if (max_words) {
    if (t->in_cap == 0 || max_words > t->in_cap * 8u + 64u) return PFT_ERR_CAPACITY;  // only ever below the allocation
    t->max_words = max_words;
  }
  t->force_npass = sorted_npass;
  sync_dev(t);
  return PFT_OK;
}
// ---- checkpoint of the filter state (between frames): the whole population with its weights, the alias prefix form the
// next resample draws from, the header (representative state, motion, KLD particle count) and the host-side schedule
// state.  Restoring is ONE kernel on the handle's stream, so a benchmark can replay the same frame again and again.
struct CopySeg {
  const uint32_t* src;
  uint32_t* dst;
  uint32_t n;  // 32-bit words
};
struct CopySegs {
  CopySeg s[5];
};
__global__ __launch_bounds__(256) void k_copy_segments(CopySegs cs) {
  const CopySeg g = cs.s[blockIdx.y];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) g.dst[i] = g.src[i];
}

static void state_segments(pft_tracker* t, bool save, CopySegs* cs) {
  const uint32_t P = t->Pcap;
  const void* live[5] = {t->dev.part_all, t->d_alias_list, t->d_alias_pref, t->d_alias_pos, t->d_hdr};
  void* saved[5] = {t->sv_part, t->sv_alias_list, t->sv_alias_pref, t->sv_alias_pos, t->sv_hdr};
  const uint32_t words[5] = {P * 8u, P * 2u, P * 4u, P, (uint32_t)(sizeof(PftHeader) / 4u)};
  for (int k = 0; k < 5; k++) {
    cs->s[k].src = static_cast<const uint32_t*>(save ? live[k] : saved[k]);
    cs->s[k].dst = static_cast<uint32_t*>(save ? saved[k] : const_cast<void*>(live[k]));
    cs->s[k].n = words[k];
  }
}

extern "C" int pft_debug_state_save(pft_tracker* t) {
  if (!t) return PFT_ERR_INVALID_ARG;
  if (!t->initialized) {
    t->err = "pft_debug_state_save before the first compute";
    return PFT_ERR_STATE;
  }
  hipSetDevice(t->cfg.device_id);
  if (!t->sv_part) {
    const size_t P = t->Pcap;
    HIPCHK(t, dalloc(&t->sv_part, P));
    HIPCHK(t, dalloc(&t->sv_alias_list, 2 * P));
    HIPCHK(t, dalloc(&t->sv_alias_pref, 2 * P));
    HIPCHK(t, dalloc(&t->sv_alias_pos, P));
    HIPCHK(t, dalloc(&t->sv_hdr, 1));
  }
  sync_dev(t);
  CopySegs cs;
  state_segments(t, true, &cs);
  hipLaunchKernelGGL(k_copy_segments, dim3(64, 5), dim3(256), 0, t->stream, cs);
  HIPCHK(t, hipStreamSynchronize(t->stream));
  t->sv_cur = t->cur;
  t->sv_epoch = t->resample_epoch;
  t->sv_changed = t->changed;
  t->sv_valid = true;
  return check_device_error(t);
}

extern "C" int pft_debug_state_restore(pft_tracker* t) {
  if (!t) return PFT_ERR_INVALID_ARG;
  if (!t->sv_valid) {
    t->err = "pft_debug_state_restore without a saved state";
    return PFT_ERR_STATE;
  }
  hipSetDevice(t->cfg.device_id);
  t->cur = t->sv_cur;
  t->resample_epoch = t->sv_epoch;
  t->changed = t->sv_changed;
  sync_dev(t);
  CopySegs cs;
  state_segments(t, false, &cs);
  hipLaunchKernelGGL(k_copy_segments, dim3(64, 5), dim3(256), 0, t->stream, cs);
  return PFT_OK;
}

extern "C" int pft_debug_inject_error(pft_tracker* t, uint32_t bits) {
  if (!t) return PFT_ERR_INVALID_ARG;
  t->inject_error = bits;
  return PFT_OK;
}
extern "C" int pft_debug_get_host_stat(pft_tracker* t, uint32_t out4[4]) {
  if (!t || !out4 || !t->h_stat) return PFT_ERR_INVALID_ARG;
  for (int i = 0; i < 4; i++) out4[i] = ((volatile uint32_t*)t->h_stat)[i];
  return PFT_OK;
}
static int ensure_dbg_part(pft_tracker* t, size_t n) {
  if (n > t->dbg_part_cap) {
    hipStreamSynchronize(t->stream);
    dfree(t->d_dbg_part);
    HIPCHK(t, dalloc(&t->d_dbg_part, n));
    t->dbg_part_cap = n;
  }
  return PFT_OK;
}
static int ensure_dbg_f(pft_tracker* t, size_t n) {
  if (n > t->dbg_f_cap) {
    hipStreamSynchronize(t->stream);
    dfree(t->d_dbg_f);
    HIPCHK(t, dalloc(&t->d_dbg_f, n));
    t->dbg_f_cap = n;
  }
  return PFT_OK;
}

extern "C" int pft_eval_weights(pft_tracker* t, const pft_particle* particles, size_t P, float* raw_w,
                                int32_t* nn_idx, float* nn_d2) {
  int r = check_ready(t);
  if (r != PFT_OK) return r;
  if (!particles || !P) return PFT_ERR_INVALID_ARG;
  if (P > (t->prm.kld ? t->Pcap : t->prm.P_local)) {
    t->err = "pft_eval_weights: " + std::to_string(P) + " particles handed over, the handle's buffers hold " +
             std::to_string(t->prm.kld ? t->Pcap : t->prm.P_local) + " (particle_num / maximum_particle_num of this rank)";
    return PFT_ERR_CAPACITY;
  }
  r = ensure_dbg_part(t, P);
  if (r != PFT_OK) return r;
  r = ensure_dbg_f(t, P);
  if (r != PFT_OK) return r;
  const bool want_nn = nn_idx || nn_d2;
  const size_t pairs = P * (size_t)t->prm.M;
  if (want_nn && pairs > t->nn_cap) {
    hipStreamSynchronize(t->stream);
    dfree(t->d_nn_idx);
    dfree(t->d_nn_d2);
    HIPCHK(t, dalloc(&t->d_nn_idx, pairs));
    HIPCHK(t, dalloc(&t->d_nn_d2, pairs));
    t->nn_cap = pairs;
  }
  sync_dev(t);
  PftDev d = t->dev;
  d.p_active = nullptr;  // explicit particle count (a KLD handle's device-side count does not apply here)
  d.part_cur = t->d_dbg_part;
  d.part_all = t->d_dbg_part;
  d.bbox6 = t->d_bbox6;
  HIPCHK(t, hipMemcpyAsync(t->d_dbg_part, particles, P * sizeof(pft_particle), hipMemcpyHostToDevice, t->stream));
  HIPCHK(t, hipMemsetAsync(&t->d_hdr->stat_queries, 0, (2 + 32) * sizeof(unsigned long long), t->stream));
  pftk_pose_to_matrix(t->stream, t->d_dbg_part, (uint32_t)P, t->d_mats);
  stage_aabb(t, d, (uint32_t)P, false);
  stage_crop_octree_likelihood(t, d, (uint32_t)P, want_nn, true, true);
  pftk_finalize_raw(t->stream, t->prm, d, (uint32_t)P, t->d_dbg_f, nullptr);
  if (raw_w) HIPCHK(t, hipMemcpyAsync(raw_w, t->d_dbg_f, P * sizeof(float), hipMemcpyDeviceToHost, t->stream));
  if (nn_idx) HIPCHK(t, hipMemcpyAsync(nn_idx, t->d_nn_idx, pairs * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
  if (nn_d2) HIPCHK(t, hipMemcpyAsync(nn_d2, t->d_nn_d2, pairs * sizeof(float), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  // the matrices of the live particles were overwritten: restore them
  if (t->initialized) pftk_pose_to_matrix(t->stream, t->d_part[t->cur], t->prm.P_local, t->d_mats);
  HIPCHK(t, hipGetLastError());
  return check_device_error(t);
}

static int read_hdr(pft_tracker* t, PftHeader* h) {
  HIPCHK(t, hipMemcpyAsync(h, t->d_hdr, sizeof(PftHeader), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  return PFT_OK;
}

extern "C" int pft_debug_get_bbox(pft_tracker* t, float bbox[6]) {
  if (!t || !bbox) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  memcpy(bbox, h.bbox, sizeof(float) * 6);
  return PFT_OK;
}

extern "C" int pft_debug_get_crop(pft_tracker* t, int32_t* idx, size_t cap, size_t* n) {
  if (!t) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  if (n) *n = h.n_crop;
  size_t c = cap < h.n_crop ? cap : h.n_crop;
  if (idx && c) {
    HIPCHK(t, hipMemcpy(idx, t->d_crop_idx, c * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return PFT_OK;
}

extern "C" int pft_debug_get_octree(pft_tracker* t, int32_t* depth, double mn[3], double mx[3], uint32_t* n_leaves,
                                    uint32_t* n_nodes) {
  if (!t) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  if (h.error) {
    t->err = "octree build error flag " + std::to_string(h.error);
    return PFT_ERR_CAPACITY;
  }
  if (depth) *depth = h.depth;
  if (mn) memcpy(mn, h.omin, sizeof(double) * 3);
  if (mx) memcpy(mx, h.omax, sizeof(double) * 3);
  if (n_leaves) *n_leaves = h.n_leaves;
  if (n_nodes) *n_nodes = h.n_words;
  return PFT_OK;
}

extern "C" int pft_debug_get_point_keys(pft_tracker* t, uint32_t* keys3, size_t cap_points) {
  if (!t || !keys3) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  size_t c = cap_points < h.n_crop ? cap_points : h.n_crop;
  if (c) HIPCHK(t, hipMemcpy(keys3, t->d_pt_key, c * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PFT_OK;
}

extern "C" int pft_debug_get_ticks(pft_tracker* t, uint64_t* ticks32) {
  if (!t || !ticks32) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  for (int i = 0; i < 32; i++) ticks32[i] = h.ticks[i];
  return PFT_OK;
}

extern "C" int pft_debug_aabb_support_subset(const pft_point_xyzrgba* pts, size_t n, uint32_t* keep, size_t* n_keep) {
  if ((!pts && n) || !keep || !n_keep) return PFT_ERR_INVALID_ARG;
  std::vector<uint32_t> k;
  pft_aabb_support_subset(pts, n, k);
  for (size_t i = 0; i < k.size(); i++) keep[i] = k[i];
  *n_keep = k.size();
  return PFT_OK;
}

extern "C" int pft_debug_get_descent_stats(pft_tracker* t, uint64_t* dbg32) {
  if (!t || !dbg32) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  for (int i = 0; i < 32; i++) dbg32[i] = h.dbg[i];
  dbg32[30] = t->prm.M_box;  // reference points the box is taken over (pft_hull.hip), of
  dbg32[31] = t->prm.M;
  if (t->cfg.exact_nearest) {  // the exact-NN mode's bookkeeping of the last iteration (tools/exact_nn_bench.py)
    dbg32[8] = h.ec_nslots;
    dbg32[9] = h.ec_pool_used;
    dbg32[10] = h.eq_totals & 0xffffffffull;
    dbg32[11] = h.eq_totals >> 32;
    dbg32[12] = h.eg_ncells;
    dbg32[13] = h.n_crop;
  }
  return PFT_OK;
}

extern "C" int pft_debug_get_scan_stats(pft_tracker* t, uint64_t* queries, uint64_t* scanned) {
  if (!t) return PFT_ERR_INVALID_ARG;
  PftHeader h;
  int r = read_hdr(t, &h);
  if (r != PFT_OK) return r;
  if (queries) *queries = h.stat_queries;
  if (scanned) *scanned = h.stat_scanned;
  return PFT_OK;
}

// population stages on explicit arrays (temporary buffers; the live state is not touched)
struct DbgPop {
  pft_particle* part = nullptr;
  int32_t* a = nullptr;
  double* q = nullptr;
  int32_t* list = nullptr;
  double* pref = nullptr;
  uint32_t* pos = nullptr;
  ~DbgPop() {
    dfree(part); dfree(a); dfree(q); dfree(list); dfree(pref); dfree(pos);
  }
};

static int dbg_population(pft_tracker* t, std::vector<pft_particle>& host, int norm, int mean, int alias, DbgPop& b,
                          PftHeader* hout) {
  const size_t n = host.size();
  if (n > PFT_MAX_PARTICLES) return PFT_ERR_CAPACITY;
  HIPCHK(t, dalloc(&b.part, n));
  HIPCHK(t, dalloc(&b.a, n));
  HIPCHK(t, dalloc(&b.q, n));
  HIPCHK(t, dalloc(&b.list, 2 * n));
  HIPCHK(t, dalloc(&b.pref, 2 * n));
  HIPCHK(t, dalloc(&b.pos, n));
  HIPCHK(t, hipMemcpyAsync(b.part, host.data(), n * sizeof(pft_particle), hipMemcpyHostToDevice, t->stream));
  HIPCHK(t, hipMemsetAsync(t->d_dbg_hdr, 0, sizeof(PftHeader), t->stream));
  sync_dev(t);
  PftDev d = t->dev;
  d.p_active = nullptr;  // explicit particle count (a KLD handle's device-side count does not apply here)
  d.part_all = b.part;
  d.alias_list = b.list;
  d.alias_pref = b.pref;
  d.alias_pos = b.pos;
  d.hdr = t->d_dbg_hdr;
  pftk_population(t->stream, t->prm, d, (uint32_t)n, 0, norm, mean, alias);
  if (alias) pftk_alias_materialize(t->stream, d, (uint32_t)n, b.a, b.q);
  HIPCHK(t, hipMemcpyAsync(host.data(), b.part, n * sizeof(pft_particle), hipMemcpyDeviceToHost, t->stream));
  if (hout) HIPCHK(t, hipMemcpyAsync(hout, t->d_dbg_hdr, sizeof(PftHeader), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(t, hipStreamSynchronize(t->stream));
  HIPCHK(t, hipGetLastError());
  return PFT_OK;
}

extern "C" int pft_debug_normalize(pft_tracker* t, float* w, size_t n, double* fit_ratio) {
  if (!t || !w || !n) return PFT_ERR_INVALID_ARG;
  std::vector<pft_particle> h(n);
  memset(h.data(), 0, n * sizeof(pft_particle));
  for (size_t i = 0; i < n; i++) h[i].weight = w[i];
  DbgPop b;
  PftHeader hd;
  int r = dbg_population(t, h, 1, 0, 0, b, &hd);
  if (r != PFT_OK) return r;
  for (size_t i = 0; i < n; i++) w[i] = h[i].weight;
  if (fit_ratio) *fit_ratio = hd.fit_ratio;
  return PFT_OK;
}

extern "C" int pft_debug_alias(pft_tracker* t, const float* w, size_t n, int32_t* a, double* q) {
  if (!t || !w || !n || !a || !q) return PFT_ERR_INVALID_ARG;
  std::vector<pft_particle> h(n);
  memset(h.data(), 0, n * sizeof(pft_particle));
  for (size_t i = 0; i < n; i++) h[i].weight = w[i];
  DbgPop b;
  int r = dbg_population(t, h, 0, 0, 1, b, nullptr);
  if (r != PFT_OK) return r;
  HIPCHK(t, hipMemcpy(a, b.a, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPCHK(t, hipMemcpy(q, b.q, n * sizeof(double), hipMemcpyDeviceToHost));
  return PFT_OK;
}

extern "C" int pft_debug_weighted_mean(pft_tracker* t, const pft_particle* p, size_t n, pft_particle* out) {
  if (!t || !p || !n || !out) return PFT_ERR_INVALID_ARG;
  std::vector<pft_particle> h(p, p + n);
  DbgPop b;
  PftHeader hd;
  int r = dbg_population(t, h, 0, 1, 0, b, &hd);
  if (r != PFT_OK) return r;
  *out = hd.rep;
  return PFT_OK;
}

extern "C" int pft_debug_init_particles(pft_tracker* t, const pft_particle* rep, uint32_t id_offset, size_t n_local,
                                        pft_particle* out) {
  if (!t || !rep || !out || !n_local) return PFT_ERR_INVALID_ARG;
  pft_particle* d = nullptr;
  HIPCHK(t, dalloc(&d, n_local));
  PftParams p = t->prm;
  p.id_offset = id_offset;
  p.P_local = (uint32_t)n_local;
  pftk_init_particles(t->stream, p, *rep, d, nullptr, nullptr);
  hipError_t e = hipMemcpyAsync(out, d, n_local * sizeof(pft_particle), hipMemcpyDeviceToHost, t->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  HIPCHK(t, e);
  return PFT_OK;
}

extern "C" int pft_debug_resample(pft_tracker* t, const pft_particle* old, size_t n_total, const int32_t* a,
                                  const double* q, const pft_particle* rep, uint32_t epoch, uint32_t id_offset,
                                  size_t n_local, pft_particle* out) {
  if (!t || !old || !a || !q || !rep || !out || !n_total || !n_local) return PFT_ERR_INVALID_ARG;
  pft_particle *d_old = nullptr, *d_out = nullptr;
  int32_t* d_a = nullptr;
  double* d_q = nullptr;
  hipError_t e = dalloc(&d_old, n_total);
  if (e == hipSuccess) e = dalloc(&d_out, n_local);
  if (e == hipSuccess) e = dalloc(&d_a, n_total);
  if (e == hipSuccess) e = dalloc(&d_q, n_total);
  if (e == hipSuccess) e = hipMemcpyAsync(d_old, old, n_total * sizeof(pft_particle), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_a, a, n_total * sizeof(int32_t), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_q, q, n_total * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemsetAsync(t->d_dbg_hdr, 0, sizeof(PftHeader), t->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&t->d_dbg_hdr->rep, rep, sizeof(pft_particle), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) {
    PftParams p = t->prm;
    p.id_offset = id_offset;
    p.P_local = (uint32_t)n_local;
    p.P_total = (uint32_t)n_total;
    pftk_resample_table(t->stream, p, d_old, d_a, d_q, t->d_dbg_hdr, epoch, d_out);
    e = hipMemcpyAsync(out, d_out, n_local * sizeof(pft_particle), hipMemcpyDeviceToHost, t->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d_old); hipFree(d_out); hipFree(d_a); hipFree(d_q);
  HIPCHK(t, e);
  return PFT_OK;
}

extern "C" int pft_debug_pose_to_matrix(pft_tracker* t, const pft_particle* p, size_t n, float* m12) {
  if (!t || !p || !n || !m12) return PFT_ERR_INVALID_ARG;
  pft_particle* d = nullptr;
  float* dm = nullptr;
  hipError_t e = dalloc(&d, n);
  if (e == hipSuccess) e = dalloc(&dm, n * 12);
  if (e == hipSuccess) e = hipMemcpyAsync(d, p, n * sizeof(pft_particle), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) {
    pftk_pose_to_matrix(t->stream, d, (uint32_t)n, dm);
    e = hipMemcpyAsync(m12, dm, n * 12 * sizeof(float), hipMemcpyDeviceToHost, t->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d); hipFree(dm);
  HIPCHK(t, e);
  return PFT_OK;
}

extern "C" int pft_debug_kld_resample(pft_tracker* t, const pft_particle* old, size_t n_old, const int32_t* a,
                                      const double* q, const pft_particle* motion, uint32_t epoch, pft_particle* out,
                                      int32_t* bins6, uint32_t* n_out, uint32_t* k_out) {
  if (!t || !old || !n_old || !a || !q || !motion || !out || !n_out) return PFT_ERR_INVALID_ARG;
  if (!t->prm.kld) return PFT_ERR_STATE;
  const uint32_t maxn = t->prm.kld_max;
  pft_particle *d_old = nullptr, *d_out = nullptr;
  int32_t *d_a = nullptr, *d_bins = nullptr;
  double* d_q = nullptr;
  PftHeader* h = new PftHeader();
  memset(h, 0, sizeof(*h));
  h->p_active = (uint32_t)n_old;
  h->motion = *motion;
  hipError_t e = dalloc(&d_old, n_old);
  if (e == hipSuccess) e = dalloc(&d_out, maxn);
  if (e == hipSuccess) e = dalloc(&d_a, n_old);
  if (e == hipSuccess) e = dalloc(&d_q, n_old);
  if (e == hipSuccess) e = dalloc(&d_bins, (size_t)6 * maxn);
  if (e == hipSuccess) e = hipMemcpyAsync(d_old, old, n_old * sizeof(pft_particle), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_a, a, n_old * sizeof(int32_t), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_q, q, n_old * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(t->d_dbg_hdr, h, sizeof(PftHeader), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) {
    PftDev d = t->dev;
    d.part_all = d_old;
    d.hdr = t->d_dbg_hdr;
    d.mats = nullptr;
    pftk_resample_kld(t->stream, t->prm, d, epoch, d_out, d_a, d_q, d_bins);
    e = hipMemcpyAsync(h, t->d_dbg_hdr, sizeof(PftHeader), hipMemcpyDeviceToHost, t->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  if (e == hipSuccess) {
    *n_out = h->p_active;
    if (k_out) *k_out = h->kld_k;
    e = hipMemcpy(out, d_out, (size_t)h->p_active * sizeof(pft_particle), hipMemcpyDeviceToHost);
    if (e == hipSuccess && bins6) e = hipMemcpy(bins6, d_bins, (size_t)h->p_active * 6 * sizeof(int32_t), hipMemcpyDeviceToHost);
  }
  hipFree(d_old); hipFree(d_out); hipFree(d_a); hipFree(d_q); hipFree(d_bins);
  delete h;
  HIPCHK(t, e);
  return PFT_OK;
}
