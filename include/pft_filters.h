/*
 * pft_filters.h -- C ABI of the per-frame input filters that run in front of the tracker
 * (SURVEY.md section 8f row 1), exported by the same libpft_hip.so as pft.h.
 *
 * Reference interfaces replaced (cmaestre/pcl_tracking, PCL 1.8.0 classes):
 *   filterPassThrough  /root/reference/src/auto_tracking.cpp:536-547   pcl::PassThrough<PointXYZRGBA>
 *                      (setFilterFieldName("z"), setFilterLimits(0, 10), setKeepOrganized(false)), called :637
 *   gridSampleApprox   /root/reference/src/auto_tracking.cpp:563-575   pcl::ApproximateVoxelGrid<PointXYZRGBA>
 *                      (setLeafSize(0.01 x3)), called per tracked frame :683
 *   gridSample         /root/reference/src/auto_tracking.cpp:549-561   pcl::VoxelGrid<PointXYZRGBA>
 *                      (setLeafSize(0.01 x3)), model preparation :672 and the waiting frames :641
 *
 * One handle runs PassThrough and / or one of the two voxel grids as ONE device pipeline over a cloud of
 * 32-byte PCL points; the output cloud stays in HBM (pft_filter_output_device) so it can be handed to
 * pft_set_input_device() without touching the host.  Results equal the sequential PCL algorithms (the
 * history-table flush order of ApproximateVoxelGrid included); there is no CPU path.
 */
#ifndef PFT_FILTERS_H
#define PFT_FILTERS_H

#include "pft.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { PFT_VOXEL_NONE = 0, PFT_VOXEL_APPROX = 1, PFT_VOXEL_EXACT = 2 };

typedef struct pft_filter_config {
  uint32_t abi_version;       /* PFT_ABI_VERSION */
  int32_t device_id;
  void* stream;               /* hipStream_t to run on when stream_is_external, else the handle creates one */
  int32_t stream_is_external;
  /* PassThrough (auto_tracking.cpp:536-547) */
  int32_t pass_enable;
  int32_t pass_field;         /* setFilterFieldName: 0 = "x", 1 = "y", 2 = "z" */
  float pass_min, pass_max;   /* setFilterLimits, inclusive */
  int32_t pass_negative;      /* setFilterLimitsNegative */
  /* voxel grid (auto_tracking.cpp:549-575) */
  int32_t voxel_mode;         /* PFT_VOXEL_NONE / _APPROX (ApproximateVoxelGrid) / _EXACT (VoxelGrid) */
  float leaf_size[3];         /* setLeafSize */
  uint32_t approx_hist_size;  /* ApproximateVoxelGrid history table entries, power of two <= 2048 (PCL: 512) */
  uint32_t max_points;        /* initial capacity; grows on demand */
} pft_filter_config;

typedef struct pft_filter pft_filter;

/* PassThrough z in [0, 10] + ApproximateVoxelGrid(0.01): what the reference runs per tracked frame */
void pft_filter_default_config(pft_filter_config* cfg);
int pft_filter_create(const pft_filter_config* cfg, pft_filter** out);
void pft_filter_destroy(pft_filter* f);
const char* pft_filter_last_error_string(const pft_filter* f);

/* setInputCloud + filter(): run the pipeline over n points in host / device memory.  Returns when the
 * output count is known (one stream synchronisation); the output cloud is then valid in HBM. */
int pft_filter_apply(pft_filter* f, const pft_point_xyzrgba* host_points, size_t n);
int pft_filter_apply_device(pft_filter* f, const pft_point_xyzrgba* device_points, size_t n);

/* n_pass: points that survived PassThrough (all finite-or-not points when it is disabled);
 * n_out: points of the output cloud */
int pft_filter_counts(const pft_filter* f, size_t* n_pass, size_t* n_out);
int pft_filter_output_device(const pft_filter* f, const pft_point_xyzrgba** device_points, size_t* n_out);
int pft_filter_get_output(pft_filter* f, pft_point_xyzrgba* host_out, size_t capacity, size_t* n_out);
/* indices (into the input cloud) of the points PassThrough kept, in order (pcl::PassThrough::filter(indices)) */
int pft_filter_get_pass_indices(pft_filter* f, int32_t* host_idx, size_t capacity, size_t* n_pass);
/* GPU time of the last apply (HIP events on the handle's stream), milliseconds */
int pft_filter_last_ms(const pft_filter* f, double* ms);

#ifdef __cplusplus
}
#endif
#endif
