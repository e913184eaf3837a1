// pft_filters.hip -- the per-frame input filters in front of the tracker (include/pft_filters.h):
//   pcl::PassThrough            (PCL 1.8.0 filters/impl/passthrough.hpp;           auto_tracking.cpp:536-547)
//   pcl::ApproximateVoxelGrid   (PCL 1.8.0 filters/impl/approximate_voxel_grid.hpp; auto_tracking.cpp:563-575)
//   pcl::VoxelGrid              (PCL 1.8.0 filters/impl/voxel_grid.hpp;            auto_tracking.cpp:549-561)
// as one chain of kernels on one HIP stream.  HBM-bound byte/integer work (N x 32 B in, a few N x 4 B
// side arrays); at a 518 400-point frame the chain is launch-latency bound.
//
// ApproximateVoxelGrid is a SEQUENTIAL algorithm upstream: a small history table holds one open voxel per
// hash entry, a point of another voxel that hashes to an occupied entry flushes it to the output, entries
// still open at the end are flushed in table order.  Its output (which points are merged, float sums in
// arrival order, output order) is reproduced exactly by observing that table entries are independent:
//   1. stable-sort the points by hash entry (arrival order kept inside an entry);
//   2. inside an entry a new run starts wherever the voxel (ix,iy,iz) changes: run = one output point,
//      summed sequentially in arrival order by one thread;
//   3. a run that is followed by another run in the same entry was flushed by that run's first point c:
//      its output slot is the number of such "trigger" points before c in the input; the last run of
//      entry h is flushed at the end: slot = #triggers + #non-empty entries below h.
// VoxelGrid: bounds -> voxel index -> stable sort -> one CentroidPoint per run, output in index order.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "pft_device_utils.h"
#include "../../include/pft_filters.h"

#define F_TILE 1024u
#define F_THREADS 256
#define F_BINS 256
#define F_INVALID 0xFFFFFFFFu
#define F_MAX_HIST 2048u

struct FParams {
  uint32_t n;
  int pass_enable, pass_field, pass_negative;
  float pass_min, pass_max;
  float inv[3];
  uint32_t hist_mask;
  int mode;
};

struct FHdr {
  uint32_t n_pass, n_out, n_trig, n_present;
  int32_t minb[3], div[3], mul[3];
  int32_t leaf_too_small;
};

struct FDev {
  const pft_point_xyzrgba* in;
  pft_point_xyzrgba* out;
  uint32_t* key[2];
  uint32_t* val[2];
  uint16_t* key16;      // approximate grid: table entry of every point (0xFFFF = dropped), input order
  float4* spt;          // approximate grid: {x, y, z, rgba bits} in table-entry order
  uint32_t* trig_bits;  // approximate grid: bit i set = point i flushed the previous voxel of its table entry
  uint32_t* word_pref;  // exclusive prefix of popcount(trig_bits[w])
  uint8_t* passf;
  uint8_t* head;
  uint32_t* hist;
  uint32_t* tile_trig;
  uint32_t* tile_pass;
  uint32_t* tile_head;
  uint32_t* bucket;  // [F_MAX_HIST + 1]: present flag, then rank
  float* bpart;      // [ntiles][6]
  int32_t* pass_idx;
  FHdr* hdr;
  uint32_t* host_stat;
};

// static_cast<int>(floor(v)) as x86 evaluates it (cvttss2si: NaN / out of range -> INT_MIN)
__device__ __forceinline__ int floor_to_int(float v) {
  const float f = floorf(v);
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int)0x80000000;
  return (int)f;
}

__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
}

__device__ __forceinline__ bool f_passes(const FParams& p, float x, float y, float z) {
  if (!p.pass_enable) return true;
  if (!finite3(x, y, z)) return false;
  const float v = p.pass_field == 0 ? x : (p.pass_field == 1 ? y : z);
  if (!p.pass_negative) return !(v < p.pass_min || v > p.pass_max);
  return !(v >= p.pass_min && v <= p.pass_max);
}

// ---- PassThrough alone and VoxelGrid, stage 1: PassThrough decision, per-tile counts (and bounds) ----
__global__ __launch_bounds__(F_THREADS) void k_f_classify(FParams p, FDev d) {
  __shared__ uint32_t su[20];
  __shared__ float sf[6][20];
  const uint32_t t = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) d.tile_head[t] = 0;
  uint32_t cnt = 0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < 4; k++) {
    const uint32_t i = t * F_TILE + k * F_THREADS + tid;
    if (i >= p.n) break;
    const float4 q = *reinterpret_cast<const float4*>(d.in + i);
    bool pass = f_passes(p, q.x, q.y, q.z);
    uint32_t kk = F_INVALID;
    if (p.mode == PFT_VOXEL_EXACT) {
      pass = pass && finite3(q.x, q.y, q.z);  // the grid skips non-finite points (is_dense == false)
      if (pass) {
        mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
        mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
      }
    } else if (pass) {
      kk = 0;
    }
    d.key[0][i] = kk;
    d.val[0][i] = i;
    d.passf[i] = pass ? 1 : 0;
    cnt += pass ? 1u : 0u;
  }
  uint32_t tot;
  block_excl_scan<uint32_t>(cnt, su, &tot);
  if (tid == 0) d.tile_pass[t] = tot;
  if (p.mode == PFT_VOXEL_EXACT) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float lo = block_reduce<float>(mn[a], sf[a], OpMinF(), INFINITY);
      const float hi = block_reduce<float>(mx[a], sf[3 + a], OpMaxF(), -INFINITY);
      if (tid == 0) {
        d.bpart[t * 6 + a] = lo;
        d.bpart[t * 6 + 3 + a] = hi;
      }
    }
  }
}

// VoxelGrid: getMinMax3D over the tiles, min_b / div_b / divb_mul and the "leaf size too small" test
__global__ __launch_bounds__(F_THREADS) void k_f_bounds(FParams p, FDev d, uint32_t ntiles) {
  __shared__ float sf[6][20];
  const uint32_t tid = threadIdx.x;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t t = tid; t < ntiles; t += F_THREADS)
    for (int a = 0; a < 3; a++) {
      mn[a] = fminf(mn[a], d.bpart[t * 6 + a]);
      mx[a] = fmaxf(mx[a], d.bpart[t * 6 + 3 + a]);
    }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    mn[a] = block_reduce<float>(mn[a], sf[a], OpMinF(), INFINITY);
    mx[a] = block_reduce<float>(mx[a], sf[3 + a], OpMaxF(), -INFINITY);
  }
  if (tid == 0) {
    FHdr* h = d.hdr;
    h->leaf_too_small = 0;
    if (mn[0] <= mx[0]) {
      long long dd[3];
      for (int a = 0; a < 3; a++) dd[a] = (long long)((mx[a] - mn[a]) * p.inv[a]) + 1;
      // the three factors are <= 2^31 each in every case that passes; checked stepwise to stay inside 64 bits
      const bool too_small = dd[0] > 0x7fffffffLL || dd[1] > 0x7fffffffLL || dd[2] > 0x7fffffffLL ||
                             dd[0] * dd[1] > 0x7fffffffLL || dd[0] * dd[1] * dd[2] > 0x7fffffffLL;
      h->leaf_too_small = too_small ? 1 : 0;
      for (int a = 0; a < 3; a++) {
        h->minb[a] = floor_to_int(mn[a] * p.inv[a]);
        h->div[a] = floor_to_int(mx[a] * p.inv[a]) - h->minb[a] + 1;
      }
      h->mul[0] = 1;
      h->mul[1] = h->div[0];
      h->mul[2] = h->div[0] * h->div[1];
    }
  }
}

__global__ __launch_bounds__(F_THREADS) void k_f_keys_exact(FParams p, FDev d) {
  const uint32_t i = blockIdx.x * F_THREADS + threadIdx.x;
  if (i >= p.n || !d.passf[i]) return;
  const FHdr* h = d.hdr;
  if (h->leaf_too_small) return;
  const float4 q = *reinterpret_cast<const float4*>(d.in + i);
  const int i0 = (int)(floorf(q.x * p.inv[0]) - (float)h->minb[0]);
  const int i1 = (int)(floorf(q.y * p.inv[1]) - (float)h->minb[1]);
  const int i2 = (int)(floorf(q.z * p.inv[2]) - (float)h->minb[2]);
  d.key[0][i] = (uint32_t)(i0 * h->mul[0] + i1 * h->mul[1] + i2 * h->mul[2]);
}

// ---- stable LSD radix sort of (key, val), 8-bit digit, one wave per 1024-element tile ----
__global__ __launch_bounds__(64) void k_f_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                  uint32_t* __restrict__ hist, uint32_t ntiles) {
  __shared__ uint32_t h[F_BINS];
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  for (int b = lane; b < F_BINS; b += 64) h[b] = 0;
  __syncthreads();
  const uint32_t base = t * F_TILE;
#pragma unroll 4
  for (uint32_t c = 0; c < F_TILE / 64; c++) {
    const uint32_t i = base + c * 64 + lane;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xffu], 1u);
  }
  __syncthreads();
  for (int b = lane; b < F_BINS; b += 64) hist[(size_t)b * ntiles + t] = h[b];
}

// one workgroup per bin: exclusive scan of that bin's per-tile counts, bin total to hist[nbins*ntiles + b]
__global__ __launch_bounds__(F_THREADS) void k_f_rs_scan(uint32_t* __restrict__ hist, uint32_t ntiles, uint32_t nbins) {
  __shared__ uint32_t scr[20];
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  uint32_t* row = hist + (size_t)b * ntiles;
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < ntiles; t0 += F_THREADS) {
    const uint32_t t = t0 + tid;
    const uint32_t v = t < ntiles ? row[t] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<uint32_t>(v, scr, &tot);
    if (t < ntiles) row[t] = carry + ex;
    carry += tot;
  }
  if (tid == 0) hist[(size_t)nbins * ntiles + b] = carry;
}

__global__ __launch_bounds__(64) void k_f_rs_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                     uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                     int shift, const uint32_t* __restrict__ offs, uint32_t ntiles) {
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  const uint32_t base = t * F_TILE;
  __shared__ uint32_t cnt[F_BINS];
  {  // bin bases = exclusive scan of the 256 bin totals (4 consecutive bins per lane) + this tile's offset in the bin
    const uint32_t* tot = offs + (size_t)F_BINS * ntiles;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[k] = tot[lane * 4 + k];
      sum += v[k];
    }
    uint32_t run = wave_incl_scan(sum) - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      cnt[lane * 4 + k] = run + offs[(size_t)(lane * 4 + k) * ntiles + t];
      run += v[k];
    }
  }
  __syncthreads();
  for (uint32_t c = 0; c < F_TILE / 64; c++) {
    const uint32_t i = base + c * 64 + lane;
    const bool valid = i < n;
    const uint32_t key = valid ? kin[i] : 0u;
    const uint32_t val = valid ? vin[i] : 0u;
    const uint32_t dig = (key >> shift) & 0xffu;
    // peers = lanes of this chunk with the same digit; lanes past the end form their own class
    unsigned long long peers = __ballot(valid);
    if (!valid) peers = ~peers;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long m = __ballot((dig >> b) & 1u);
      peers &= ((dig >> b) & 1u) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t rank = __popcll(peers & lt);
    const uint32_t dst0 = cnt[dig];
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) cnt[dig] = dst0 + (uint32_t)__popcll(peers);  // leader advances the running offset
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      kout[dst0 + rank] = key;
      vout[dst0 + rank] = val;
    }
  }
}

// VoxelGrid: run heads in sorted order, per-tile head counts
__global__ __launch_bounds__(F_THREADS) void k_f_heads_exact(FParams p, FDev d, const uint32_t* __restrict__ skey) {
  __shared__ uint32_t su[20];
  const uint32_t t = blockIdx.x, tid = threadIdx.x;
  uint32_t cnt = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t j = t * F_TILE + k * F_THREADS + tid;
    if (j >= p.n) break;
    const uint32_t key = skey[j];
    const bool head = key != F_INVALID && (j == 0 || skey[j - 1] != key);
    d.head[j] = head ? 1 : 0;
    cnt += head ? 1u : 0u;
  }
  uint32_t tot;
  block_excl_scan<uint32_t>(cnt, su, &tot);
  if (tid == 0) d.tile_head[t] = tot;
}

// in-place exclusive scan of a[0..n) by one workgroup; returns the total to every thread
__device__ uint32_t scan_array(uint32_t* a, uint32_t n, uint32_t* scr) {
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < n; t0 += blockDim.x) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < n ? a[t] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<uint32_t>(v, scr, &tot);
    if (t < n) a[t] = carry + ex;
    carry += tot;
  }
  return carry;
}

// one workgroup: the small scans (per-tile counts, table-entry ranks) and the output counts
__global__ __launch_bounds__(1024) void k_f_scan_small(FParams p, FDev d, uint32_t ntiles) {
  __shared__ uint32_t scr[20];
  const uint32_t n_pass = scan_array(d.tile_pass, ntiles, scr);
  uint32_t n_out = n_pass, n_trig = 0, n_present = 0;
  if (p.mode == PFT_VOXEL_EXACT) {
    n_out = scan_array(d.tile_head, ntiles, scr);
    if (d.hdr->leaf_too_small) n_out = 0;
  }
  if (threadIdx.x == 0) {
    d.hdr->n_pass = n_pass;
    d.hdr->n_out = n_out;
    d.hdr->n_trig = n_trig;
    d.hdr->n_present = n_present;
    d.host_stat[0] = n_pass;
    d.host_stat[1] = n_out;
    d.host_stat[2] = (uint32_t)d.hdr->leaf_too_small;
  }
}

__device__ __forceinline__ void store_point(pft_point_xyzrgba* o, float x, float y, float z, uint32_t rgba) {
  float4* q = reinterpret_cast<float4*>(o);
  q[0] = make_float4(x, y, z, 1.0f);
  q[1] = make_float4(__uint_as_float(rgba), 0.0f, 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------------
// ApproximateVoxelGrid (+ PassThrough) in six launches:
//   k_fa_classify  table entry of every point, per-tile histogram over the entries, zeroed trigger bits
//   k_f_rs_scan    per entry: exclusive scan over the tiles, entry totals
//   k_fa_scatter   stable counting sort by entry: {x, y, z, rgba} and the input index in entry order
//   k_fa_heads     run heads; the first point of every later run of an entry is a flush trigger (bit set)
//   k_fa_ranks     one workgroup: trigger ranks per 32-point word, ranks of the non-empty entries, counts
//   k_fa_emit_tile one thread per run, points staged in LDS: centroid in arrival order, written to its flush slot
__device__ __forceinline__ uint32_t cell_entry(const FParams& p, float x, float y, float z, int& ix, int& iy, int& iz) {
  ix = floor_to_int(x * p.inv[0]);
  iy = floor_to_int(y * p.inv[1]);
  iz = floor_to_int(z * p.inv[2]);
  return ((uint32_t)ix * 7171u + (uint32_t)iy * 3079u + (uint32_t)iz * 4231u) & p.hist_mask;
}

__global__ __launch_bounds__(F_THREADS) void k_fa_classify(FParams p, FDev d, uint32_t ntiles) {
  __shared__ uint32_t h[F_MAX_HIST];
  __shared__ uint32_t su[20];
  const uint32_t t = blockIdx.x, tid = threadIdx.x, nb = p.hist_mask + 1u;
  for (uint32_t b = tid; b < nb; b += F_THREADS) h[b] = 0;
  if (tid < F_TILE / 32) d.trig_bits[t * (F_TILE / 32) + tid] = 0;
  __syncthreads();
  uint32_t cnt = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t i = t * F_TILE + k * F_THREADS + tid;
    if (i >= p.n) break;
    const float4 q = *reinterpret_cast<const float4*>(d.in + i);
    const bool pass = f_passes(p, q.x, q.y, q.z);
    uint32_t key = 0xFFFFu;
    if (pass) {
      int ix, iy, iz;
      key = cell_entry(p, q.x, q.y, q.z, ix, iy, iz);
      atomicAdd(&h[key], 1u);
      cnt++;
    }
    d.key16[i] = (uint16_t)key;
    d.passf[i] = pass ? 1 : 0;
  }
  __syncthreads();
  for (uint32_t b = tid; b < nb; b += F_THREADS) d.hist[(size_t)b * ntiles + t] = h[b];
  uint32_t tot;
  block_excl_scan<uint32_t>(cnt, su, &tot);
  if (tid == 0) d.tile_pass[t] = tot;
}

__global__ __launch_bounds__(64) void k_fa_scatter(FParams p, FDev d, uint32_t ntiles, int bits) {
  __shared__ uint32_t cnt[F_MAX_HIST];
  const uint32_t t = blockIdx.x, lane = threadIdx.x, nb = p.hist_mask + 1u;
  const uint32_t* tot = d.hist + (size_t)nb * ntiles;
  {  // entry bases = exclusive scan of the entry totals (consecutive entries per lane) + this tile's offset
    const uint32_t per = nb >= 64u ? nb / 64u : 1u;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) {
      const uint32_t b = lane * per + k;
      if (b < nb) sum += tot[b];
    }
    const uint32_t inc = wave_incl_scan(sum);
    uint32_t run = inc - sum;
    for (uint32_t k = 0; k < per; k++) {
      const uint32_t b = lane * per + k;
      if (b < nb) {
        cnt[b] = run + d.hist[(size_t)b * ntiles + t];
        run += tot[b];
      }
    }
    if (t == 0 && lane == 63) d.hdr->n_pass = inc;  // points that entered the grid
  }
  __syncthreads();
  const uint32_t base = t * F_TILE;
  constexpr int NC = F_TILE / 64;
  uint32_t keys[NC], dst[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) {  // all keys of the tile in flight at once
    const uint32_t i = base + c * 64 + lane;
    keys[c] = i < p.n ? (uint32_t)d.key16[i] : 0xFFFFu;
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {  // destinations, chunk after chunk (arrival order inside an entry)
    uint32_t key = keys[c];
    const bool valid = key != 0xFFFFu;
    if (!valid) key = 0;
    unsigned long long peers = __ballot(valid);
    if (!valid) peers = ~peers;
    for (int b = 0; b < bits; b++) {
      const unsigned long long m = __ballot((key >> b) & 1u);
      peers &= ((key >> b) & 1u) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t rank = __popcll(peers & lt);
    const uint32_t dst0 = cnt[key];
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) cnt[key] = dst0 + (uint32_t)__popcll(peers);  // leader advances the running offset
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dst[c] = valid ? dst0 + rank : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {  // the moves are independent of each other
    if (dst[c] != 0xFFFFFFFFu) {
      const uint32_t i = base + c * 64 + lane;
      const float4* q = reinterpret_cast<const float4*>(d.in + i);
      const float4 a = q[0];
      d.spt[dst[c]] = make_float4(a.x, a.y, a.z, q[1].x);
      d.val[0][dst[c]] = i;
    }
  }
}

// The same stable counting sort with four waves per tile (table sizes up to 512): every wave ranks its own four
// 64-point chunks with ballots, the per-chunk counts of every entry go to LDS, one pass over the 16 chunks turns them
// into per-chunk offsets, and all 1024 moves are issued together.  No wave waits for another wave's running offsets.
#define FS_MAXB 512u
__global__ __launch_bounds__(F_THREADS) void k_fa_scatter4(FParams p, FDev d, uint32_t ntiles, int bits) {
  constexpr uint32_t NCH = F_TILE / 64u;       // chunks per tile
  constexpr uint32_t CPW = NCH / (F_THREADS / 64u);  // chunks per wave
  __shared__ uint32_t cnt[FS_MAXB];
  __shared__ uint32_t ccount[NCH * FS_MAXB / 4u];  // u8 [chunk][entry]: points of the entry in the chunk
  __shared__ uint16_t cbase[NCH * FS_MAXB];        // u16 [chunk][entry]: points of the entry in earlier chunks of the tile
  __shared__ uint32_t su[20];
  const uint32_t t = blockIdx.x, tid = threadIdx.x, nb = p.hist_mask + 1u;
  const int lane = lane_id(), w = wave_id();
  const uint32_t* tot = d.hist + (size_t)nb * ntiles;
  // keys of the wave's chunks (in flight while the bases are computed)
  const uint32_t base = t * F_TILE;
  uint32_t keys[CPW];
#pragma unroll
  for (uint32_t c = 0; c < CPW; c++) {
    const uint32_t i = base + (w * CPW + c) * 64u + lane;
    keys[c] = i < p.n ? (uint32_t)d.key16[i] : 0xFFFFu;
  }
  for (uint32_t k = tid; k < NCH * FS_MAXB / 4u; k += F_THREADS) ccount[k] = 0u;
  {  // entry bases = exclusive scan of the entry totals (consecutive entries per thread) + this tile's offset
    const uint32_t per = (nb + F_THREADS - 1u) / F_THREADS;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) {
      const uint32_t b = tid * per + k;
      if (b < nb) sum += tot[b];
    }
    uint32_t total;
    uint32_t run = block_excl_scan<uint32_t>(sum, su, &total);  // (its barriers also publish the zeroed counts)
    for (uint32_t k = 0; k < per; k++) {
      const uint32_t b = tid * per + k;
      if (b < nb) {
        cnt[b] = run + d.hist[(size_t)b * ntiles + t];
        run += tot[b];
      }
    }
    if (t == 0 && tid == 0) d.hdr->n_pass = total;  // points that entered the grid
  }
  uint32_t rank[CPW];
  uint8_t* cc8 = reinterpret_cast<uint8_t*>(ccount);
#pragma unroll
  for (uint32_t c = 0; c < CPW; c++) {
    uint32_t key = keys[c];
    const bool valid = key != 0xFFFFu;
    if (!valid) key = 0;
    unsigned long long peers = __ballot(valid);
    if (!valid) peers = ~peers;
    for (int b = 0; b < bits; b++) {
      const unsigned long long m = __ballot((key >> b) & 1u);
      peers &= ((key >> b) & 1u) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    rank[c] = __popcll(peers & lt);
    if (valid && rank[c] == 0) cc8[(w * CPW + c) * FS_MAXB + key] = (uint8_t)__popcll(peers);
    keys[c] = valid ? key : 0xFFFFu;
  }
  __syncthreads();
  for (uint32_t b = tid; b < nb; b += F_THREADS) {  // per entry: offsets of the chunks inside the tile
    uint32_t run = 0;
#pragma unroll
    for (uint32_t c = 0; c < NCH; c++) {
      cbase[c * FS_MAXB + b] = (uint16_t)run;
      run += cc8[c * FS_MAXB + b];
    }
  }
  __syncthreads();
  uint32_t dst[CPW];
#pragma unroll
  for (uint32_t c = 0; c < CPW; c++) {
    const uint32_t key = keys[c];
    dst[c] = key != 0xFFFFu ? cnt[key] + cbase[(w * CPW + c) * FS_MAXB + key] + rank[c] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (uint32_t c = 0; c < CPW; c++) {
    if (dst[c] != 0xFFFFFFFFu) {
      const uint32_t i = base + (w * CPW + c) * 64u + lane;
      const float4* q = reinterpret_cast<const float4*>(d.in + i);
      const float4 a = q[0];
      d.spt[dst[c]] = make_float4(a.x, a.y, a.z, q[1].x);
      d.val[0][dst[c]] = i;
    }
  }
}

__global__ __launch_bounds__(F_THREADS) void k_fa_heads(FParams p, FDev d) {
  const uint32_t j = blockIdx.x * F_THREADS + threadIdx.x;
  const uint32_t nv = d.hdr->n_pass;
  if (j > nv) return;
  if (j == nv) {  // sentinel: ends the last run
    d.head[j] = 2;
    return;
  }
  const float4 a = d.spt[j];
  int ax, ay, az;
  const uint32_t ea = cell_entry(p, a.x, a.y, a.z, ax, ay, az);
  uint32_t flag = 2;  // first run of its table entry
  if (j > 0) {
    const float4 b = d.spt[j - 1];
    int bx, by, bz;
    const uint32_t eb = cell_entry(p, b.x, b.y, b.z, bx, by, bz);
    if (eb == ea) flag = (ax != bx || ay != by || az != bz) ? 1u : 0u;
  }
  d.head[j] = (uint8_t)flag;
  if (flag == 1) {  // this point's arrival flushed the previous voxel of the entry
    const uint32_t i = d.val[0][j];
    atomicOr(&d.trig_bits[i >> 5], 1u << (i & 31u));
  }
}

__global__ __launch_bounds__(1024) void k_fa_ranks(FParams p, FDev d, uint32_t ntiles) {
  __shared__ uint32_t scr[20];
  const uint32_t tid = threadIdx.x, nb = p.hist_mask + 1u;
  const uint32_t nwords_pad = ntiles * (F_TILE / 32u);
  // exclusive prefix of the trigger popcounts, 16 consecutive words per thread per round
  uint32_t carry = 0;
  for (uint32_t w0 = 0; w0 < nwords_pad; w0 += 1024u * 16u) {
    const uint32_t wb = w0 + tid * 16u;
    uint32_t c[16], sum = 0;
    // 16-byte accesses (the arrays are padded to whole tiles = multiples of 32 words, zero-filled by classify)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (wb + 4 * k < nwords_pad) v = *reinterpret_cast<const uint4*>(d.trig_bits + wb + 4 * k);
      c[4 * k + 0] = __popc(v.x);
      c[4 * k + 1] = __popc(v.y);
      c[4 * k + 2] = __popc(v.z);
      c[4 * k + 3] = __popc(v.w);
      sum += c[4 * k] + c[4 * k + 1] + c[4 * k + 2] + c[4 * k + 3];
    }
    uint32_t tot;
    uint32_t ex = carry + block_excl_scan<uint32_t>(sum, scr, &tot);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint4 o;
      o.x = ex;
      o.y = o.x + c[4 * k];
      o.z = o.y + c[4 * k + 1];
      o.w = o.z + c[4 * k + 2];
      ex = o.w + c[4 * k + 3];
      if (wb + 4 * k < nwords_pad) *reinterpret_cast<uint4*>(d.word_pref + wb + 4 * k) = o;
    }
    carry += tot;
  }
  const uint32_t n_trig = carry;
  // ranks of the non-empty table entries (two consecutive entries per thread)
  const uint32_t* tot = d.hist + (size_t)nb * ntiles;
  const uint32_t b0 = tid * 2u;
  const uint32_t f0 = (b0 < nb && tot[b0]) ? 1u : 0u, f1 = (b0 + 1u < nb && tot[b0 + 1u]) ? 1u : 0u;
  uint32_t n_present;
  const uint32_t ex = block_excl_scan<uint32_t>(f0 + f1, scr, &n_present);
  if (b0 < nb) d.bucket[b0] = ex;
  if (b0 + 1u < nb) d.bucket[b0 + 1u] = ex + f0;
  if (tid == 0) {
    d.hdr->n_trig = n_trig;
    d.hdr->n_present = n_present;
    d.hdr->n_out = n_trig + n_present;
    d.hdr->leaf_too_small = 0;
    d.host_stat[0] = d.hdr->n_pass;
    d.host_stat[1] = n_trig + n_present;
    d.host_stat[2] = 0;
  }
}

struct RunOut {
  float sx, sy, sz, sr, sg, sb;
};

__device__ __forceinline__ void emit_run(const FParams& p, const FDev& d, const RunOut& s, uint32_t count, float x, float y,
                                         float z, uint32_t next_flag, uint32_t next_pos) {
  const float cnt = (float)count;
  const int rgb = ((int)(s.sr / cnt)) << 16 | ((int)(s.sg / cnt)) << 8 | ((int)(s.sb / cnt));
  uint32_t pos;
  if (next_flag == 1) {  // flushed when the next voxel of this table entry arrived: rank of that trigger point
    const uint32_t i = d.val[0][next_pos];
    pos = d.word_pref[i >> 5] + __popc(d.trig_bits[i >> 5] & ((1u << (i & 31u)) - 1u));
  } else {  // still open at the end: flushed in table order
    int ix, iy, iz;
    pos = d.hdr->n_trig + d.bucket[cell_entry(p, x, y, z, ix, iy, iz)];
  }
  store_point(d.out + pos, s.sx / cnt, s.sy / cnt, s.sz / cnt, (uint32_t)rgb);
}

// Centroids, one workgroup per 1024 consecutive sorted points: the tile is loaded coalesced into LDS, the run heads
// inside it are compacted into a list, and one THREAD per run walks its points in LDS (sequential float adds in
// arrival order, as upstream; runs are 6 points long on average but reach ~200 on a depth frame).  The tile's last
// run continues into the following tiles: the workgroup fetches 256 points at a time and one thread keeps adding.
// (Measured at 518 400 points: 14 us; a thread per run reading HBM directly 90 us, a wave per 64 points with 64
// lane-broadcast steps 26 us.)
#define FE_TILE 1024u
#define FE_THREADS (FE_TILE / 4u)
__global__ __launch_bounds__(FE_THREADS) void k_fa_emit_tile(FParams p, FDev d) {
  __shared__ float4 spts[FE_TILE];
  __shared__ uint32_t shead[FE_TILE / 4];
  __shared__ uint16_t srun[FE_TILE];
  __shared__ uint32_t su[20];
  const uint32_t base = blockIdx.x * FE_TILE, tid = threadIdx.x;
  // The tile, its head flags and the first 256 points after it are requested before the point count is known (the
  // buffers are padded to whole tiles; the tile index is below the input count): one memory round trip less.
  float4 lq[FE_TILE / FE_THREADS];
#pragma unroll
  for (uint32_t k = 0; k < FE_TILE / FE_THREADS; k++) lq[k] = d.spt[base + k * FE_THREADS + tid];
  uint32_t hw = *reinterpret_cast<const uint32_t*>(d.head + base + tid * 4u);  // four consecutive points
  const uint32_t jh = base + FE_TILE + tid;
  uint32_t h = jh <= p.n ? (uint32_t)d.head[jh] : 2u;
  float4 cq = jh < p.n ? d.spt[jh] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const uint32_t nv = d.hdr->n_pass;
  if (base >= nv) return;
  const uint32_t cnt_tile = min(FE_TILE, nv - base);
#pragma unroll
  for (uint32_t k = 0; k < FE_TILE / FE_THREADS; k++) spts[k * FE_THREADS + tid] = lq[k];
  // positions past the last point never start a run (their flags are stale)
#pragma unroll
  for (uint32_t b = 0; b < 4; b++)
    if (tid * 4u + b >= cnt_tile) hw &= ~(0xffu << (8u * b));
  if (jh > nv || cnt_tile < FE_TILE) h = 2u;  // head[nv] is the sentinel; a short tile ends at it
  shead[tid] = hw;
  const uint32_t f0 = (hw & 0xffu) ? 1u : 0u, f1 = (hw & 0xff00u) ? 1u : 0u, f2 = (hw & 0xff0000u) ? 1u : 0u,
                 f3 = (hw >> 24) ? 1u : 0u;
  uint32_t R;
  uint32_t o = block_excl_scan<uint32_t>(f0 + f1 + f2 + f3, su, &R);
  if (f0) srun[o++] = (uint16_t)(tid * 4u);
  if (f1) srun[o++] = (uint16_t)(tid * 4u + 1u);
  if (f2) srun[o++] = (uint16_t)(tid * 4u + 2u);
  if (f3) srun[o++] = (uint16_t)(tid * 4u + 3u);
  __syncthreads();
  const uint8_t* hb = reinterpret_cast<const uint8_t*>(shead);
#define FE_ACC(q)                                 \
  do {                                            \
    const uint32_t c_ = __float_as_uint((q).w);   \
    s.sx += (q).x;                                \
    s.sy += (q).y;                                \
    s.sz += (q).z;                                \
    s.sr += (float)((c_ >> 16) & 255u);           \
    s.sg += (float)((c_ >> 8) & 255u);            \
    s.sb += (float)(c_ & 255u);                   \
  } while (0)
// adds stay in arrival order; four LDS reads are in flight per round
#define FE_WALK(arr, from, to)                                                     \
  do {                                                                             \
    uint32_t l_ = (from);                                                          \
    for (; l_ + 4u <= (to); l_ += 4u) {                                            \
      const float4 q0 = arr[l_], q1 = arr[l_ + 1u], q2 = arr[l_ + 2u], q3 = arr[l_ + 3u]; \
      FE_ACC(q0);                                                                  \
      FE_ACC(q1);                                                                  \
      FE_ACC(q2);                                                                  \
      FE_ACC(q3);                                                                  \
    }                                                                              \
    for (; l_ < (to); l_++) {                                                      \
      const float4 q0 = arr[l_];                                                   \
      FE_ACC(q0);                                                                  \
    }                                                                              \
  } while (0)
  for (uint32_t r = tid; r + 1u < R; r += FE_THREADS) {  // the runs that end inside the tile
    const uint32_t s0 = srun[r], e0 = srun[r + 1u];
    RunOut s = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const float4 first = spts[s0];
    FE_WALK(spts, s0, e0);
    emit_run(p, d, s, e0 - s0, first.x, first.y, first.z, hb[e0], base + e0);
  }
  if (R == 0) return;  // (workgroup-uniform) the whole tile belongs to a run of an earlier tile
  // The tile's last run may continue in the following tiles: the workgroup fetches 256 points at a time into LDS and
  // finds the next head; one thread keeps adding in arrival order.
  __shared__ float4 cpts[FE_THREADS];
  __shared__ uint32_t sfirst[2], shf;
  const uint32_t owner = FE_THREADS - 1u;
  RunOut s = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const uint32_t s0 = srun[R - 1u];
  const float4 first = spts[s0];
  if (tid == owner) FE_WALK(spts, s0, cnt_tile);
  if (tid == 0) sfirst[0] = sfirst[1] = FE_THREADS;
  __syncthreads();
  uint32_t npos = base + cnt_tile;
  for (uint32_t it = 0;; it++) {
    if (it > 0) {
      const uint32_t jj = npos + tid;
      h = jj <= nv ? (uint32_t)d.head[jj] : 2u;
      if (jj < nv) cq = d.spt[jj];
    }
    cpts[tid] = cq;
    if (h) atomicMin(&sfirst[it & 1u], tid);
    __syncthreads();
    const uint32_t f = sfirst[it & 1u];
    if (tid == f) shf = h;
    if (tid == owner) {
      sfirst[(it + 1u) & 1u] = FE_THREADS;
      FE_WALK(cpts, 0u, f);
    }
    __syncthreads();
    npos += f;
    if (f < FE_THREADS) break;
  }
#undef FE_WALK
#undef FE_ACC
  if (tid == owner) emit_run(p, d, s, npos - (base + s0), first.x, first.y, first.z, shf, npos);
}

// VoxelGrid: one CentroidPoint per voxel, output in voxel-index order.  Same shape as k_fa_emit_tile: the 1024 sorted
// entries of a tile are gathered (through the sorted input indices) into LDS by all threads at once, one thread per
// run adds in sorted order (= input order inside a voxel), the tile's last run continues 256 entries at a time.
// Entries at or after n_pass are the dropped points (key F_INVALID sorts last).
struct RunOutA {
  float sx, sy, sz, sr, sg, sb, sa;
};

__global__ __launch_bounds__(FE_THREADS) void k_f_emit_exact(FParams p, FDev d, const uint32_t* __restrict__ sval) {
  __shared__ float4 spts[FE_TILE];
  __shared__ uint16_t srun[FE_TILE];
  __shared__ uint32_t su[20];
  __shared__ float4 cpts[FE_THREADS];
  __shared__ uint32_t sfirst[2];
  const uint32_t t = blockIdx.x, base = t * FE_TILE, tid = threadIdx.x;
  const uint32_t nv = d.hdr->n_pass;
  if (base >= nv || d.hdr->leaf_too_small) return;
  const uint32_t cnt_tile = min(FE_TILE, nv - base);
#pragma unroll
  for (uint32_t k = 0; k < FE_TILE / FE_THREADS; k++) {
    const uint32_t l = k * FE_THREADS + tid;
    if (l < cnt_tile) {
      const float4* q = reinterpret_cast<const float4*>(d.in + sval[base + l]);
      const float4 a = q[0];
      spts[l] = make_float4(a.x, a.y, a.z, q[1].x);
    }
  }
  uint32_t f[4], cnt = 0;
#pragma unroll
  for (uint32_t b = 0; b < 4; b++) {
    const uint32_t l = tid * 4u + b;
    f[b] = l < cnt_tile ? (uint32_t)d.head[base + l] : 0u;
    cnt += f[b];
  }
  uint32_t R;
  uint32_t o = block_excl_scan<uint32_t>(cnt, su, &R);
#pragma unroll
  for (uint32_t b = 0; b < 4; b++)
    if (f[b]) srun[o++] = (uint16_t)(tid * 4u + b);
  __syncthreads();
  const uint32_t out0 = d.tile_head[t];
#define FX_ACC(q)                               \
  do {                                          \
    const uint32_t c_ = __float_as_uint((q).w); \
    s.sx += (q).x;                              \
    s.sy += (q).y;                              \
    s.sz += (q).z;                              \
    s.sr += (float)((c_ >> 16) & 255u);         \
    s.sg += (float)((c_ >> 8) & 255u);          \
    s.sb += (float)(c_ & 255u);                 \
    s.sa += (float)(c_ >> 24);                  \
  } while (0)
#define FX_WALK(arr, from, to)                                                            \
  do {                                                                                    \
    uint32_t l_ = (from);                                                                 \
    for (; l_ + 4u <= (to); l_ += 4u) {                                                   \
      const float4 q0 = arr[l_], q1 = arr[l_ + 1u], q2 = arr[l_ + 2u], q3 = arr[l_ + 3u]; \
      FX_ACC(q0);                                                                         \
      FX_ACC(q1);                                                                         \
      FX_ACC(q2);                                                                         \
      FX_ACC(q3);                                                                         \
    }                                                                                     \
    for (; l_ < (to); l_++) {                                                             \
      const float4 q0 = arr[l_];                                                          \
      FX_ACC(q0);                                                                         \
    }                                                                                     \
  } while (0)
#define FX_EMIT(slot, count)                                                                                     \
  do {                                                                                                           \
    const float c = (float)(count);                                                                              \
    const uint32_t rgba =                                                                                        \
        (uint32_t)(s.sa / c) << 24 | (uint32_t)(s.sr / c) << 16 | (uint32_t)(s.sg / c) << 8 | (uint32_t)(s.sb / c); \
    store_point(d.out + (slot), s.sx / c, s.sy / c, s.sz / c, rgba);                                             \
  } while (0)
  for (uint32_t r = tid; r + 1u < R; r += FE_THREADS) {  // the runs that end inside the tile
    const uint32_t s0 = srun[r], e0 = srun[r + 1u];
    RunOutA s = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    FX_WALK(spts, s0, e0);
    FX_EMIT(out0 + r, e0 - s0);
  }
  if (R == 0) return;  // (workgroup-uniform) the whole tile belongs to a run of an earlier tile
  const uint32_t owner = FE_THREADS - 1u;
  RunOutA s = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const uint32_t s0 = srun[R - 1u];
  if (tid == owner) FX_WALK(spts, s0, cnt_tile);
  if (tid == 0) sfirst[0] = sfirst[1] = FE_THREADS;
  __syncthreads();
  uint32_t npos = base + cnt_tile;
  for (uint32_t it = 0;; it++) {
    const uint32_t jj = npos + tid;
    const uint32_t h = jj < nv ? (uint32_t)d.head[jj] : 1u;  // the end of the kept points ends the last run
    if (jj < nv) {
      const float4* q = reinterpret_cast<const float4*>(d.in + sval[jj]);
      const float4 a = q[0];
      cpts[tid] = make_float4(a.x, a.y, a.z, q[1].x);
    }
    if (h) atomicMin(&sfirst[it & 1u], tid);
    __syncthreads();
    const uint32_t fpos = sfirst[it & 1u];
    if (tid == owner) {
      sfirst[(it + 1u) & 1u] = FE_THREADS;
      FX_WALK(cpts, 0u, fpos);
    }
    __syncthreads();
    npos += fpos;
    if (fpos < FE_THREADS) break;
  }
  if (tid == owner) FX_EMIT(out0 + R - 1u, npos - (base + s0));
#undef FX_EMIT
#undef FX_WALK
#undef FX_ACC
}

// PassThrough alone: the kept points, stable (sorted position j < n_pass holds input index sval[j])
__global__ __launch_bounds__(F_THREADS) void k_f_gather(FParams p, FDev d, const uint32_t* __restrict__ sval) {
  const uint32_t j = blockIdx.x * F_THREADS + threadIdx.x;
  if (j >= d.hdr->n_pass) return;
  const float4* q = reinterpret_cast<const float4*>(d.in + sval[j]);
  float4* o = reinterpret_cast<float4*>(d.out + j);
  o[0] = q[0];
  o[1] = q[1];
}

__global__ __launch_bounds__(1024) void k_f_scan_tiles(FDev d, uint32_t ntiles) {
  __shared__ uint32_t scr[20];
  scan_array(d.tile_pass, ntiles, scr);
}

// indices kept by PassThrough, in input order (pcl::PassThrough::filter(std::vector<int>&))
__global__ __launch_bounds__(F_THREADS) void k_f_pass_indices(FParams p, FDev d) {
  __shared__ uint32_t su[20];
  const uint32_t t = blockIdx.x, tid = threadIdx.x;
  const uint32_t i0 = t * F_TILE + tid * 4;
  uint32_t f[4], cnt = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    f[k] = (i0 + k < p.n) ? d.passf[i0 + k] : 0u;
    cnt += f[k];
  }
  uint32_t tot;
  uint32_t ex = d.tile_pass[t] + block_excl_scan<uint32_t>(cnt, su, &tot);
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (f[k]) d.pass_idx[ex++] = (int32_t)(i0 + k);
}

// ---------------------------------------------------------------------------------------------------
// host side
#define FCHK(f, call)                                                  \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) {                                            \
      (f)->err = std::string(#call) + ": " + hipGetErrorString(e_);    \
      return PFT_ERR_HIP;                                              \
    }                                                                  \
  } while (0)

struct pft_filter {
  pft_filter_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  size_t cap = 0;
  uint32_t ntiles = 0;
  pft_point_xyzrgba* d_in_own = nullptr;
  FDev d = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  size_t n_in = 0, n_pass = 0, n_out = 0;
  int leaf_too_small = 0;
  bool have_result = false;
  bool tile_pass_scanned = false;
  double last_ms = 0.0;
};

template <typename T>
static hipError_t falloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T));
}
template <typename T>
static void ffree(T*& p) {
  if (p) hipFree((void*)p);
  p = nullptr;
}

static void free_buffers(pft_filter* f) {
  FDev& d = f->d;
  ffree(f->d_in_own); ffree(d.out); ffree(d.key[0]); ffree(d.key[1]); ffree(d.val[0]); ffree(d.val[1]);
  ffree(d.key16); ffree(d.spt); ffree(d.trig_bits); ffree(d.word_pref); ffree(d.passf); ffree(d.head); ffree(d.hist);
  ffree(d.tile_trig); ffree(d.tile_pass); ffree(d.tile_head); ffree(d.bpart); ffree(d.pass_idx);
  f->cap = 0;
}

static int ensure_capacity(pft_filter* f, size_t n) {
  if (n <= f->cap) return PFT_OK;
  if (f->stream) FCHK(f, hipStreamSynchronize(f->stream));
  free_buffers(f);
  size_t cap = n < 1024 ? 1024 : n;
  cap = (cap + 1023) / 1024 * 1024;
  const uint32_t nt = (uint32_t)(cap / 1024);
  FDev& d = f->d;
  FCHK(f, falloc(&f->d_in_own, cap));
  FCHK(f, falloc(&d.out, cap));
  for (int k = 0; k < 2; k++) {
    FCHK(f, falloc(&d.key[k], cap));
    FCHK(f, falloc(&d.val[k], cap));
  }
  FCHK(f, falloc(&d.key16, cap));
  FCHK(f, falloc(&d.spt, cap + 8));
  FCHK(f, falloc(&d.trig_bits, cap / 32));
  FCHK(f, falloc(&d.word_pref, cap / 32));
  FCHK(f, falloc(&d.passf, cap));
  FCHK(f, falloc(&d.head, cap + 8));
  FCHK(f, falloc(&d.hist, (size_t)F_MAX_HIST * nt + F_MAX_HIST));
  FCHK(f, falloc(&d.tile_trig, nt));
  FCHK(f, falloc(&d.tile_pass, nt));
  FCHK(f, falloc(&d.tile_head, nt));
  FCHK(f, falloc(&d.bpart, (size_t)nt * 6));
  FCHK(f, falloc(&d.pass_idx, cap));
  f->cap = cap;
  return PFT_OK;
}

extern "C" void pft_filter_default_config(pft_filter_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->abi_version = PFT_ABI_VERSION;
  c->pass_enable = 1;  // auto_tracking.cpp:536-547
  c->pass_field = 2;
  c->pass_min = 0.0f;
  c->pass_max = 10.0f;
  c->voxel_mode = PFT_VOXEL_APPROX;  // auto_tracking.cpp:563-575
  c->leaf_size[0] = c->leaf_size[1] = c->leaf_size[2] = 0.01f;
  c->approx_hist_size = 512;  // PCL 1.8.0 approximate_voxel_grid.h: histsize_ (512)
  c->max_points = 960 * 540;  // Kinect2 qhd, auto_tracking.cpp:775
}

extern "C" int pft_filter_create(const pft_filter_config* cfg, pft_filter** out) {
  if (!cfg || !out) return PFT_ERR_INVALID_ARG;
  *out = nullptr;
  if (cfg->abi_version != PFT_ABI_VERSION) return PFT_ERR_INVALID_ARG;
  if (cfg->pass_field < 0 || cfg->pass_field > 2) return PFT_ERR_INVALID_ARG;
  if (cfg->voxel_mode < PFT_VOXEL_NONE || cfg->voxel_mode > PFT_VOXEL_EXACT) return PFT_ERR_INVALID_ARG;
  if (cfg->voxel_mode != PFT_VOXEL_NONE)
    for (int k = 0; k < 3; k++)
      if (!(cfg->leaf_size[k] > 0.0f)) return PFT_ERR_INVALID_ARG;
  const uint32_t hs = cfg->approx_hist_size;
  if (cfg->voxel_mode == PFT_VOXEL_APPROX && (hs == 0 || (hs & (hs - 1)) || hs > F_MAX_HIST)) return PFT_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PFT_ERR_NO_DEVICE;  // no CPU path
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return PFT_ERR_INVALID_ARG;
  if (hipSetDevice(cfg->device_id) != hipSuccess) return PFT_ERR_NO_DEVICE;
  pft_filter* f = new pft_filter();
  f->cfg = *cfg;
  if (cfg->stream_is_external) {
    f->stream = reinterpret_cast<hipStream_t>(cfg->stream);
  } else {
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) {
      delete f;
      return PFT_ERR_HIP;
    }
    f->own_stream = true;
  }
  bool ok = hipEventCreate(&f->ev0) == hipSuccess && hipEventCreate(&f->ev1) == hipSuccess &&
            falloc(&f->d.bucket, F_MAX_HIST + 1) == hipSuccess && falloc(&f->d.hdr, 1) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&f->d.host_stat), 4 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess;
  if (ok) ok = hipMemset(f->d.hdr, 0, sizeof(FHdr)) == hipSuccess;
  if (ok) ok = ensure_capacity(f, cfg->max_points ? cfg->max_points : 1024) == PFT_OK;
  if (!ok) {
    pft_filter_destroy(f);
    return PFT_ERR_HIP;
  }
  *out = f;
  return PFT_OK;
}

extern "C" void pft_filter_destroy(pft_filter* f) {
  if (!f) return;
  if (f->stream) hipStreamSynchronize(f->stream);
  free_buffers(f);
  ffree(f->d.bucket);
  ffree(f->d.hdr);
  if (f->d.host_stat) hipHostFree(f->d.host_stat);
  if (f->ev0) hipEventDestroy(f->ev0);
  if (f->ev1) hipEventDestroy(f->ev1);
  if (f->own_stream && f->stream) hipStreamDestroy(f->stream);
  delete f;
}

extern "C" const char* pft_filter_last_error_string(const pft_filter* f) { return f ? f->err.c_str() : "null handle"; }

static int run_pipeline(pft_filter* f, const pft_point_xyzrgba* d_in, size_t n) {
  const pft_filter_config& c = f->cfg;
  hipStream_t s = f->stream;
  FDev d = f->d;
  d.in = d_in;
  FParams p;
  p.n = (uint32_t)n;
  p.pass_enable = c.pass_enable;
  p.pass_field = c.pass_field;
  p.pass_negative = c.pass_negative;
  p.pass_min = c.pass_min;
  p.pass_max = c.pass_max;
  // setLeafSize: inverse_leaf_size_ = Eigen::Array3f::Ones() / leaf_size_.array()  (float division on the host)
  for (int k = 0; k < 3; k++) p.inv[k] = c.voxel_mode != PFT_VOXEL_NONE ? 1.0f / c.leaf_size[k] : 0.0f;
  p.hist_mask = c.approx_hist_size ? c.approx_hist_size - 1u : 0u;
  p.mode = c.voxel_mode;
  const uint32_t ntiles = (uint32_t)((n + F_TILE - 1) / F_TILE);
  const uint32_t nblk = (uint32_t)((n + F_THREADS - 1) / F_THREADS);
  f->ntiles = ntiles;
  FCHK(f, hipEventRecord(f->ev0, s));
  if (p.mode == PFT_VOXEL_APPROX) {
    const uint32_t nb = p.hist_mask + 1u;
    int bits = 0;
    while ((1u << bits) < nb) bits++;
    hipLaunchKernelGGL(k_fa_classify, dim3(ntiles), dim3(F_THREADS), 0, s, p, d, ntiles);
    hipLaunchKernelGGL(k_f_rs_scan, dim3(nb), dim3(F_THREADS), 0, s, d.hist, ntiles, nb);
    static const bool one_wave_scatter = getenv("PFT_FILTER_SCATTER1") != nullptr;  // A/B timing only
    if (nb <= FS_MAXB && !one_wave_scatter)
      hipLaunchKernelGGL(k_fa_scatter4, dim3(ntiles), dim3(F_THREADS), 0, s, p, d, ntiles, bits);
    else
      hipLaunchKernelGGL(k_fa_scatter, dim3(ntiles), dim3(64), 0, s, p, d, ntiles, bits);
    hipLaunchKernelGGL(k_fa_heads, dim3((uint32_t)((n + 1 + F_THREADS - 1) / F_THREADS)), dim3(F_THREADS), 0, s, p, d);
    hipLaunchKernelGGL(k_fa_ranks, dim3(1), dim3(1024), 0, s, p, d, ntiles);
    hipLaunchKernelGGL(k_fa_emit_tile, dim3((uint32_t)((n + FE_TILE - 1) / FE_TILE)), dim3(FE_THREADS), 0, s, p, d);
    f->tile_pass_scanned = false;
  } else {
    hipLaunchKernelGGL(k_f_classify, dim3(ntiles), dim3(F_THREADS), 0, s, p, d);
    int npass = 1;
    if (p.mode == PFT_VOXEL_EXACT) {
      npass = 4;
      hipLaunchKernelGGL(k_f_bounds, dim3(1), dim3(F_THREADS), 0, s, p, d, ntiles);
      hipLaunchKernelGGL(k_f_keys_exact, dim3(nblk), dim3(F_THREADS), 0, s, p, d);
    }
    int cur = 0;
    for (int pass = 0; pass < npass; pass++) {
      const int shift = 8 * pass;
      hipLaunchKernelGGL(k_f_rs_hist, dim3(ntiles), dim3(64), 0, s, d.key[cur], p.n, shift, d.hist, ntiles);
      hipLaunchKernelGGL(k_f_rs_scan, dim3(F_BINS), dim3(F_THREADS), 0, s, d.hist, ntiles, (uint32_t)F_BINS);
      hipLaunchKernelGGL(k_f_rs_scatter, dim3(ntiles), dim3(64), 0, s, d.key[cur], d.val[cur], d.key[1 - cur],
                         d.val[1 - cur], p.n, shift, d.hist, ntiles);
      cur = 1 - cur;
    }
    const uint32_t* skey = d.key[cur];
    const uint32_t* sval = d.val[cur];
    if (p.mode == PFT_VOXEL_EXACT) {
      hipLaunchKernelGGL(k_f_heads_exact, dim3(ntiles), dim3(F_THREADS), 0, s, p, d, skey);
      hipLaunchKernelGGL(k_f_scan_small, dim3(1), dim3(1024), 0, s, p, d, ntiles);
      hipLaunchKernelGGL(k_f_emit_exact, dim3(ntiles), dim3(FE_THREADS), 0, s, p, d, sval);
    } else {
      hipLaunchKernelGGL(k_f_scan_small, dim3(1), dim3(1024), 0, s, p, d, ntiles);
      hipLaunchKernelGGL(k_f_gather, dim3(nblk), dim3(F_THREADS), 0, s, p, d, sval);
    }
    f->tile_pass_scanned = true;
  }
  FCHK(f, hipEventRecord(f->ev1, s));
  FCHK(f, hipGetLastError());
  FCHK(f, hipStreamSynchronize(s));
  float ms = 0.0f;
  FCHK(f, hipEventElapsedTime(&ms, f->ev0, f->ev1));
  f->last_ms = ms;
  f->n_in = n;
  f->n_pass = f->d.host_stat[0];
  f->n_out = f->d.host_stat[1];
  f->leaf_too_small = (int)f->d.host_stat[2];
  f->have_result = true;
  if (f->leaf_too_small) {
    // PCL warns "Leaf size is too small for the input dataset" and hands the input cloud through unchanged
    FCHK(f, hipMemcpyAsync(f->d.out, d_in, n * sizeof(pft_point_xyzrgba), hipMemcpyDeviceToDevice, s));
    FCHK(f, hipStreamSynchronize(s));
    f->n_out = n;
  }
  return PFT_OK;
}

static int apply_common(pft_filter* f, const pft_point_xyzrgba* pts, size_t n, bool on_device) {
  if (!f || (!pts && n)) return PFT_ERR_INVALID_ARG;
  if (n > 0x7fffffffu) return PFT_ERR_CAPACITY;
  f->have_result = false;
  FCHK(f, hipSetDevice(f->cfg.device_id));
  if (n == 0) {  // empty input cloud: empty output
    f->n_in = f->n_pass = f->n_out = 0;
    f->leaf_too_small = 0;
    f->last_ms = 0.0;
    f->have_result = true;
    return PFT_OK;
  }
  int r = ensure_capacity(f, n);
  if (r != PFT_OK) return r;
  const pft_point_xyzrgba* d_in = pts;
  if (!on_device) {
    FCHK(f, hipMemcpyAsync(f->d_in_own, pts, n * sizeof(pft_point_xyzrgba), hipMemcpyHostToDevice, f->stream));
    d_in = f->d_in_own;
  }
  return run_pipeline(f, d_in, n);
}

extern "C" int pft_filter_apply(pft_filter* f, const pft_point_xyzrgba* host_points, size_t n) {
  return apply_common(f, host_points, n, false);
}

extern "C" int pft_filter_apply_device(pft_filter* f, const pft_point_xyzrgba* device_points, size_t n) {
  return apply_common(f, device_points, n, true);
}

extern "C" int pft_filter_counts(const pft_filter* f, size_t* n_pass, size_t* n_out) {
  if (!f) return PFT_ERR_INVALID_ARG;
  if (!f->have_result) return PFT_ERR_STATE;
  if (n_pass) *n_pass = f->n_pass;
  if (n_out) *n_out = f->n_out;
  return PFT_OK;
}

extern "C" int pft_filter_output_device(const pft_filter* f, const pft_point_xyzrgba** device_points, size_t* n_out) {
  if (!f || !device_points || !n_out) return PFT_ERR_INVALID_ARG;
  if (!f->have_result) return PFT_ERR_STATE;
  *device_points = f->d.out;
  *n_out = f->n_out;
  return PFT_OK;
}

extern "C" int pft_filter_get_output(pft_filter* f, pft_point_xyzrgba* host_out, size_t capacity, size_t* n_out) {
  if (!f || !n_out) return PFT_ERR_INVALID_ARG;
  if (!f->have_result) return PFT_ERR_STATE;
  *n_out = f->n_out;
  if (f->n_out > capacity) return PFT_ERR_CAPACITY;
  if (f->n_out) {
    if (!host_out) return PFT_ERR_INVALID_ARG;
    FCHK(f, hipMemcpyAsync(host_out, f->d.out, f->n_out * sizeof(pft_point_xyzrgba), hipMemcpyDeviceToHost, f->stream));
    FCHK(f, hipStreamSynchronize(f->stream));
  }
  return PFT_OK;
}

extern "C" int pft_filter_get_pass_indices(pft_filter* f, int32_t* host_idx, size_t capacity, size_t* n_pass) {
  if (!f || !n_pass) return PFT_ERR_INVALID_ARG;
  if (!f->have_result) return PFT_ERR_STATE;
  *n_pass = f->n_pass;
  if (f->n_pass > capacity) return PFT_ERR_CAPACITY;
  if (f->n_pass) {
    if (!host_idx) return PFT_ERR_INVALID_ARG;
    FParams p = {};
    p.n = (uint32_t)f->n_in;
    if (!f->tile_pass_scanned) {
      hipLaunchKernelGGL(k_f_scan_tiles, dim3(1), dim3(1024), 0, f->stream, f->d, f->ntiles);
      f->tile_pass_scanned = true;
    }
    hipLaunchKernelGGL(k_f_pass_indices, dim3(f->ntiles), dim3(F_THREADS), 0, f->stream, p, f->d);
    FCHK(f, hipMemcpyAsync(host_idx, f->d.pass_idx, f->n_pass * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
    FCHK(f, hipStreamSynchronize(f->stream));
  }
  return PFT_OK;
}

extern "C" int pft_filter_last_ms(const pft_filter* f, double* ms) {
  if (!f || !ms) return PFT_ERR_INVALID_ARG;
  if (!f->have_result) return PFT_ERR_STATE;
  *ms = f->last_ms;
  return PFT_OK;
}
