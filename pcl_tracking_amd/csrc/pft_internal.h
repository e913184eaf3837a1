// pft_internal.h -- device-side data layout + kernel launch prototypes (gfx950 only).
//
// HBM layout of one tracker (all arrays allocated once, resident for the handle's lifetime):
//   ref_xyz / ref_hsv   float4[M]   reference points {x,y,z,-} and precomputed {fh,fs,fv,-}   (A7b, once)
//   in_pts              float4[N]   input cloud {x,y,z,rgba-bits}                              (per frame)
//   part[2]             particle[P_local] double-buffered shard; part_all = all P (gathered)
//   mats                float[12*P_local] row-major 3x4 per particle                           (A1)
//   bbox_part           float[6*grid] per-workgroup AABB partials -> bbox6 {-min xyz, max xyz}  (A3)
//   crop_pts            float4[N]   cropped points {x,y,z,h|s<<8|v<<16}, input order            (A4)
//   words               u32[]       linearised octree: branch = mask | child_base<<8, levels contiguous,
//                                   leaf = start offset into leaf_pts (+ one sentinel)          (A5)
//   (per-level per-axis voxel-centre tables: formed in LDS by the likelihood kernel from depth + box)      (A6)
//   leaf_pts            float4[N]   cropped points in leaf order (insertion order inside a leaf)
//   partial             double[P_local*nchunk] per (particle, reference chunk) likelihood sums   (A7)
//   alias_list/pref/pos prefix-sum form of the Walker alias table over all P particles          (A9)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/pft.h"

#define PFT_MAX_DEVICES 64      // per-device caches of one-time kernel attributes
#define PFT_MAX_DEPTH 30
#define PFT_TABLE_MAX_DEPTH 10
#define PFT_MAX_GROW 40
#define PFT_JUMP_MAX_LEVEL 4     // 2^12 cells x u16 = 8 KiB of LDS in the likelihood kernel
#define PFT_REF_CHUNK 256        // reference points per likelihood work item (upper limit; PftParams::ref_chunk)
#ifndef PFT_BUILD_THREADS
#define PFT_BUILD_THREADS 1024
#endif
#ifndef PFT_LIK_GROUPS
#define PFT_LIK_GROUPS 64        // likelihood kernel: groups of workgroups that share a dynamic work-item counter
#endif
#ifndef PFT_LIK_CTR_STRIDE
#define PFT_LIK_CTR_STRIDE 16u    // uint32 between two groups' counters
#endif
#ifndef PFT_LIK_THREADS
#define PFT_LIK_THREADS 1024   // likelihood workgroup size
#endif
#ifndef PFT_LIK_WGS_PER_CU
#define PFT_LIK_WGS_PER_CU 2    // resident likelihood workgroups per CU (each gets 1/N of the LDS)
#endif
#define PFT_POP_THREADS 1024
#define PFT_SORTED_BUILD_MIN 18000  // cropped points (last iteration) above which the sorted builder is used
#define PFT_EG_CAP (1u << 21)     // exact-NN mode: grid cells (8 MB of cell starts)
#define PFT_EC_SLOTS (1u << 19)  // exact-NN mode: candidate lists per iteration (cells hit by queries)
#define PFT_EC_POOL (1u << 24)   // exact-NN mode: candidate entries of all lists together (16 B each), upper limit
#define PFT_EC_POOL_MIN (1u << 20)  // first allocation (16 MB); the pool then follows the demand of the previous iteration
#define PFT_POPM_THREADS 256
#define PFT_POPM_ITEMS 16   // many-workgroup population path: 4096 particles per workgroup
#define PFT_POPM_MAX_WGS 256
#define PFT_POPM_MIN (16 * PFT_POP_THREADS + 1)  // above the register-resident single-workgroup range
#define PFT_MAX_PARTICLES (PFT_POPM_MAX_WGS * PFT_POPM_THREADS * PFT_POPM_ITEMS)  // 1 048 576

struct PftParams {  // immutable per handle, passed by value to kernels
  double alpha;
  double maxd2;          // max_distance * max_distance (double, as PCL compares)
  double res;            // octree resolution
  double dist_w, hsv_w;
  float h_w, s_w, v_w;
  int hsv_argorder;
  double step_sigma[6], init_sigma[6], init_mean[6];
  uint32_t seed_lo, seed_hi;
  uint32_t P_total, P_local, id_offset;
  uint32_t M, nchunk;
  uint32_t M_box;      // reference points that can be extreme in a rigidly transformed coordinate (pft_hull.hip): the box's input
  uint32_t ref_chunk;  // reference points per likelihood work item: 64 .. PFT_REF_CHUNK, smaller when there are few particles
  uint32_t split_last;  // s > 0: the last ref_chunk points of the cloud form s items of ref_chunk / s points, handed out last (shorter tail)
  // KLD-adaptive variant (KLDAdaptiveParticleFilterOMPTracker, auto_tracking.cpp:207-222)
  uint32_t kld;          // 1: the particle count changes at every resample and lives in PftHeader::p_active
  uint32_t kld_max;      // maximum_particle_number_
  double kld_z;          // normalQuantile(delta_), evaluated on the host
  double kld_eps;        // epsilon_
  float kld_bin[6];      // bin_size_ (a ParticleXYZRPY upstream: floats)
  double motion_ratio;   // motion_ratio_
};

struct PftHeader {  // lives in HBM; written by kernels, read by later kernels (and by the host for debug)
  float bbox[6];    // x_min,x_max,y_min,y_max,z_min,z_max
  uint32_t n_crop;
  uint32_t error;   // per iteration (cleared by the crop kernel): bit0 octree capacity exceeded, bit1 depth / growth steps
                    // exceeded, bit2 one-pass crop gave up waiting, bit3 (internal, transient) sorted builder's radix passes
                    // too few -> the rescue launch rebuilds, bit4 a device-scope barrier of the population kernel timed out
  double omin[3], omax[3];
  int32_t depth;
  int32_t use_table;
  uint32_t n_words;   // words used incl. leaf level + sentinel
  uint32_t n_leaves;
  uint32_t leaf_start;  // index of first leaf word
  uint32_t lvl_start[PFT_MAX_DEPTH + 3];
  int32_t n_grow;
  int32_t build_path;   // 1 = register/LDS-resident builder, 0 = generic builder (diagnostic)
  int32_t leaf_indirect;  // 1: leaf_pts was NOT written for this tree; the likelihood kernel reads crop_pts[leaf_order[pos]]
  // fast descent (pft_likelihood.hip): direct-index table of the level-J nodes and the safety margin
  int32_t jump_level;   // J (0 = no table): jump[kx | ky<<J | kz<<2J] = 1 + index of the node inside level J
  float margin_cells;   // a query closer than this (in leaf cells) to a cell face takes the exact generic step
  float ominf[3];       // (float) omin
  float inv_res;        // (float)(1/res)
  // growth history of the box replay (read by the key kernel of the sorted builder)
  uint32_t grow_idx[PFT_MAX_GROW];
  uint32_t grow_shift[PFT_MAX_GROW];      // bit a set: min of axis a lowered by the old side
  uint32_t grow_old_depth[PFT_MAX_GROW];
  double grow_min[PFT_MAX_GROW + 1][3];   // [e] = box minimum valid for points inserted in epoch e
  double fit_ratio;
  pft_particle rep;
  pft_particle motion;
  uint32_t alias_m, alias_nh;   // sizes of the small / large lists
  // exact-nearest-neighbour mode (pft_exact_nn.hip): uniform grid over the crop box
  float eg_g, eg_inv_g, eg_min[3];
  int32_t eg_dim[3];
  uint32_t eg_ncells;
  uint32_t p_active;            // KLD variant: current particle_num_ (written by init / k_resample_kld)
  uint32_t kld_k;               // KLD variant: distinct bins of the last resample (diagnostic)
  unsigned long long stat_queries, stat_scanned;
  unsigned long long dbg[32];    // debug-variant likelihood statistics (tools/descent_stats.py)
  unsigned long long ticks[32];  // wall_clock64() (100 MHz) at phase boundaries: [0..15] octree, [16..31] population
  // likelihood kernel: per group of workgroups, the next work item of the group's range (one counter per 64-byte line;
  // reset by the crop kernel of the same iteration)
  alignas(128) uint32_t lik_ctr[PFT_LIK_GROUPS * PFT_LIK_CTR_STRIDE];
  // exact-NN mode: running totals taken with returning atomics by every wave of k_ec_slots / k_ec_build.  In a cache line
  // of their own: next to the grid geometry, which the same waves READ, 22 000 atomics cost 540 us (25 ns each)
  alignas(128) uint32_t ec_nslots;  // grid cells hit by at least one query this iteration (candidate lists)
  uint32_t ec_pool_used;            // candidate entries allotted so far
  unsigned long long eq_totals;     // (64-query blocks << 32) | queries of the cells allotted so far (k_ec_slots)
  alignas(128) uint32_t crop_ticket;  // one-pass crop: workgroups take their logical index here (the last one resets it)
  uint32_t pop_bar[4];   // population kernel: arrival counters of its three device-scope barriers + "done" (self-resetting)
};

struct PftDev {  // device pointers (host-side struct, passed by value)
  const float4* ref_xyz;
  const float4* ref_hsv;
  const float4* ref_box;    // [M_box] {x, y, z, .} of the box's support subset
  const float4* in_pts;
  uint32_t N;
  pft_particle* part_cur;   // shard being evaluated
  pft_particle* part_all;   // all P particles (== part_cur when world_size == 1)
  float* mats;
  float* bbox_part;
  uint32_t bbox_grid;
  uint32_t bbox_part_cap;   // partial boxes bbox_part has room for
  float* bbox6;             // {-xmin,-ymin,-zmin,xmax,ymax,zmax}: max-reducible across ranks
  uint32_t* crop_counts;
  unsigned long long* crop_slots;  // one-pass crop: per workgroup (launch epoch << 32) | kept points, published with atomics
  float4* crop_pts;
  int32_t* crop_idx;
  uint32_t* words;
  uint32_t max_words;
  uint16_t* jump;           // [2^(3*PFT_JUMP_MAX_LEVEL)]
  const uint32_t* ref_perm; // sorted reference position -> index in the caller's reference cloud
  float4* leaf_pts;
  uint32_t* leaf_order;
  uint32_t* pt_node;
  uint32_t* pt_key;     // 3 per point (final-frame keys; read back by the debug hook)
  unsigned long long* pt_key64;  // packed keys of the generic (HBM-resident) builder path
  uint32_t* pt_tmp;
  double* partial;
  int32_t* alias_list;   // [0,P): small list, [P,2P): large list
  double* alias_pref;    // [0,P): running deficit, [P,2P): running excess
  uint32_t* alias_pos;   // [P]
  float* raw_w;          // [P_local] raw likelihood weights, w = -(float) sum of the particle's partial sums (k_finalize_raw)
  double* pop_part;      // [PFT_POPM_MAX_WGS][16] per-workgroup values that cross workgroups in the population kernel
  PftHeader* hdr;
  const uint32_t* p_active;  // KLD variant: &hdr->p_active (kernels take the particle count from here), else null
  uint32_t* eg_start;        // exact-NN mode: [eg_cap + 1] first slot of every grid cell
  uint32_t* eg_cnt;          // exact-NN mode: [eg_cap] cell counts / fill cursors
  uint32_t* eg_tile;         // exact-NN mode: per-2048-cell tile sums
  uint32_t eg_cap;           // exact-NN mode: cells available
  uint32_t* ec_slot;         // exact-NN mode: [eg_cap] 0 = no query in this cell, 1 = hit (before the slots are allotted), 0xffffffff = no list, else list slot + 2
  uint32_t* ec_cells;        // exact-NN mode: [PFT_EC_SLOTS] cell of every list slot
  uint32_t* ec_count;        // exact-NN mode: [PFT_EC_SLOTS] candidates of the list (0xffffffff: the pool was full, no list)
  uint32_t* ec_base;         // exact-NN mode: [PFT_EC_SLOTS] first entry of the list in ec_list
  float4* ec_list;           // exact-NN mode: [ec_pool_cap] candidates {x, y, z, position in leaf_pts}
  uint32_t ec_pool_cap;      // entries of ec_list: grows with the demand of the previous iteration, at most PFT_EC_POOL
  // exact-NN mode, queries sorted by grid cell (one wave then walks ONE candidate list for 64 queries):
  uint32_t* eq_cellq;        // [eg_cap] queries per grid cell this iteration
  uint32_t* eq_nq;           // [PFT_EC_SLOTS] queries of the slot's cell that are searched through its list
  uint32_t* eq_qbase;        // [PFT_EC_SLOTS] first sorted query of the slot
  uint32_t* eq_bbase;        // [PFT_EC_SLOTS] first 64-query block of the slot
  uint32_t* eq_fill;         // [PFT_EC_SLOTS] fill cursor of the scatter
  uint32_t* eq_blk;          // [eq_blk_cap] slot of every block
  float4* eq_sorted;         // [eq_cap] {qx, qy, qz, query id = particle * M + reference position}
  double* eq_out;            // [eq_cap] by query id: the point coherence of the pair (0: no neighbour inside the gate)
  uint32_t eq_cap, eq_blk_cap;  // 0: the per-query kernel is used instead
  uint32_t* eq_tiles;        // [eq_tiles_cap][2 * 8192] per tile of particles: the LDS count table of k_ec_mark (cells, counts)
  uint32_t eq_tiles_cap;
  uint32_t* kld_table;       // KLD variant: open-addressing table of first occurrences, 2 x pow2(kld_max) entries
  int32_t* kld_bins;         // KLD variant: 6 ints per candidate
  uint32_t* host_stat;  // pinned host memory, device-visible: [0] last n_crop, [1] last octree depth (read by the
                        // host WITHOUT synchronising, to pick the builder for the next iteration: a wrong guess costs
                        // time, never correctness), [2] error flags of the last failed iteration, [3] error flags
                        // accumulated since the host last looked (checked and cleared at the host's sync points)
  int32_t* nn_idx;      // debug only
  float* nn_d2;
};

// launchers
void pftk_pack_reference(hipStream_t s, const pft_point_xyzrgba* d_pts, uint32_t n, int argorder, float4* xyz,
                         float4* hsv);
void pftk_pack_input(hipStream_t s, const pft_point_xyzrgba* d_pts, uint32_t n, float4* out);
void pftk_init_particles(hipStream_t s, const PftParams& p, pft_particle rep, pft_particle* out, float* mats,
                         PftHeader* hdr);
void pftk_resample(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out);
void pftk_bbox_final(hipStream_t s, const PftDev& d);
uint32_t pftk_resample_box(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out);
// debug: resample from an explicit (a, q) table instead of the prefix-sum form
void pftk_resample_table(hipStream_t s, const PftParams& p, const pft_particle* old, const int32_t* a,
                         const double* q, const PftHeader* hdr, uint32_t epoch, pft_particle* out);
void pftk_pose_to_matrix(hipStream_t s, const pft_particle* p, uint32_t n, float* mats);
void pftk_aabb(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool finalize);
void pft_aabb_support_subset(const pft_point_xyzrgba* pts, size_t n, std::vector<uint32_t>& keep);  // pft_hull.hip (host)
// NearestPairPointCloudCoherence mode: uniform grid over the cropped cloud, then the likelihood with the true NN
void pftk_exact_grid(hipStream_t s, const PftParams& p, const PftDev& d);
void pftk_likelihood_exact(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool debug_nn,
                           int num_cus);
// KLD variant: draws up to p.kld_max candidates from d.part_all[0 .. p_active) (alias prefix form, or the explicit
// table a/q when given), keeps the prefix the KL bound asks for, writes particles + matrices and the new p_active
void pftk_resample_kld(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out,
                       const int32_t* table_a, const double* table_q, int32_t* bins_out);
// epoch: a value that differs from the handle's previous crop launch (non-zero); tags the per-workgroup counts of the one-pass crop
// raw: the input in PCL's 32-byte layout when its 16-byte records have not been formed yet (first crop of a frame), else null
void pftk_crop(hipStream_t s, const PftParams& p, const PftDev& d, bool bbox_from_partials, uint32_t epoch,
               const pft_point_xyzrgba* raw);
// returns true when the leaf records were left for the likelihood kernel to follow through leaf_order (its INDIRECT form)
bool pftk_octree(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t expected_points, bool allow_indirect = true);
// no-op launch unless the sorted builder flagged "radix passes too few" (error bit 3): then the single-workgroup build
void pftk_octree_rescue(hipStream_t s, const PftParams& p, const PftDev& d);
struct SortBufs {
  unsigned long long* keys[2];
  uint32_t* vals[2];
  uint32_t* hist;      // [256][ntiles]
  uint32_t* tile_cnt;  // [ntiles][PFT_MAX_DEPTH + 2]
  float* tile_box;     // [ntiles][6] AABB of each 1024-point tile of the cropped cloud
  uint32_t ntiles;
};
// many-workgroup builder for large cropped clouds (pft_octree_sorted.hip); npass = 4 (depth <= 10) or 8
void pftk_octree_sorted(hipStream_t s, const PftParams& p, const PftDev& d, const SortBufs& sb, uint32_t n_pad,
                        int npass);
void pftk_likelihood(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool debug_nn,
                     int num_cus, bool leaf_indirect = false);
// shard (nullable): sharded handles -- the particles with their raw weights also go into the all-gather's send buffer
void pftk_finalize_raw(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles,
                       float* raw_out /*nullable*/, pft_particle* shard /*nullable*/);
// normalise + update + alias prefix form over part_all[0..n) in one launch; from_partials != 0: the raw weights are
// first formed from the likelihood partial sums (single-GPU path: fuses k_finalize_raw)
void pftk_population(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n, int from_partials,
                     int do_normalize, int do_mean, int do_alias);
// debug: materialise the (a, q) table from the prefix-sum form
void pftk_alias_materialize(hipStream_t s, const PftDev& d, uint32_t n, int32_t* a, double* q);
int pftk_max_lds_bytes();
int pftk_cur_device();  // current HIP device ordinal, clamped to [0, PFT_MAX_DEVICES)
