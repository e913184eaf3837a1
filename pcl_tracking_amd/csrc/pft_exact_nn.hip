// pft_exact_nn.hip -- NearestPairPointCloudCoherence (PCL 1.8.0 tracking/impl/nearest_pair_point_cloud_coherence.hpp),
// the alternative to ApproxNearestPair... that /root/reference/src/auto_tracking.cpp leaves commented out at
// :237-238, :249: per transformed reference point the TRUE nearest neighbour in the cropped
// cloud (search_->nearestKSearch(q, 1, ...), squared distance in float), gated by d2 < max_distance^2, then the same
// point coherences as the approximate variant.
//
// Only neighbours inside the gate contribute, so the search structure is a uniform grid over the crop box with a cell
// of two octree leaves (2 cm): counting sort of the cropped points by cell (small kernels, every iteration).  The
// queries of an iteration fall into a few ten thousand of its cells; for each of those cells the points that can be
// the nearest neighbour of ANY query inside it are collected once (k_ec_mark / k_ec_slots / k_ec_build: "candidate
// lists", about 90 candidates per query on the bench workload instead of the thousands a per-query search of the
// 10 cm gate visits), the queries are sorted by cell (k_eq_scatter), and a wave walks ONE list for 64 queries of one cell
// with the candidate in scalar registers (k_eq_search; k_eq_reduce sums the pairs' values per particle).  Nearest =
// smallest float pointSquaredDist, equal distances -> lowest cloud index (upstream leaves ties to std::sort).  The
// per-query list walk (k_likelihood_exact, PFT_EXACT_PER_QUERY=1) and the per-query shell search (rings of grid rows
// of growing radius: queries outside the grid, cells that found the list pool full, PFT_EXACT_SHELLS_ONLY=1) remain
// as cross-checks and fallbacks.  No octree is built in this mode.  Kernels per iteration: k_eg_* (grid, 7 launches),
// k_ec_mark, k_ec_slots, k_ec_build, k_eq_scatter, k_eq_search, k_eq_reduce.
#include "pft_device_utils.h"

#define EG_TILE 2048u

__device__ __forceinline__ int eg_cell1(float v, float mn, float inv_g, int dim) {
  int c = (int)floorf((v - mn) * inv_g);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// grid geometry from the crop box; the cell doubles until the grid fits the cell arrays
__global__ void k_eg_setup(PftParams prm, PftDev d) {
  PftHeader* h = d.hdr;
  if (h->error) h->n_crop = 0u;  // a failed crop (bit 2) leaves no target: reported through host_stat by the likelihood kernel
  const uint32_t n = h->n_crop;
  float g = (float)(2.0 * prm.res);
  int dim[3] = {1, 1, 1};
  if (n > 0) {
    for (;;) {
      unsigned long long cells = 1;
      for (int a = 0; a < 3; a++) {
        const float ext = h->bbox[2 * a + 1] - h->bbox[2 * a];
        float q = floorf(ext / g);
        if (!(q >= 0.0f)) q = 0.0f;
        if (q > 4.0e6f) q = 4.0e6f;
        dim[a] = (int)q + 1;
        cells *= (unsigned long long)dim[a];
      }
      if (cells <= (unsigned long long)d.eg_cap) break;
      g *= 2.0f;
    }
  }
  h->eg_g = g;
  h->eg_inv_g = 1.0f / g;
  for (int a = 0; a < 3; a++) {
    h->eg_dim[a] = dim[a];
    h->eg_min[a] = h->bbox[2 * a];
  }
  h->eg_ncells = n > 0 ? (uint32_t)(dim[0] * dim[1] * dim[2]) : 0u;
  h->ec_nslots = 0u;
  h->ec_pool_used = 0u;
  h->eq_totals = 0ull;
}

__global__ __launch_bounds__(256) void k_eg_zero(PftDev d) {
  const uint32_t nc = d.hdr->eg_ncells;
  for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < min(nc, (blockIdx.x + 1u) * 1024u); i += 256u) {
    d.eg_cnt[i] = 0u;
    d.ec_slot[i] = 0u;
    d.eq_cellq[i] = 0u;
  }
}

__device__ __forceinline__ uint32_t eg_cell_of(const PftHeader* h, float x, float y, float z) {
  const int cx = eg_cell1(x, h->eg_min[0], h->eg_inv_g, h->eg_dim[0]);
  const int cy = eg_cell1(y, h->eg_min[1], h->eg_inv_g, h->eg_dim[1]);
  const int cz = eg_cell1(z, h->eg_min[2], h->eg_inv_g, h->eg_dim[2]);
  return (uint32_t)((cz * h->eg_dim[1] + cy) * h->eg_dim[0] + cx);
}

__global__ __launch_bounds__(256) void k_eg_count(PftDev d) {
  const PftHeader* h = d.hdr;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= h->n_crop) return;
  const float4 p = d.crop_pts[i];
  atomicAdd(&d.eg_cnt[eg_cell_of(h, p.x, p.y, p.z)], 1u);
}

// exclusive scan of the cell counts: tile sums, one workgroup over the tiles, tile-local scan + base
__global__ __launch_bounds__(256) void k_eg_tsum(PftDev d) {
  __shared__ uint32_t scr[20];
  const uint32_t nc = d.hdr->eg_ncells, t = blockIdx.x;
  if (t * EG_TILE >= nc) return;
  uint32_t s = 0;
  for (uint32_t i = t * EG_TILE + threadIdx.x; i < min(nc, (t + 1u) * EG_TILE); i += 256u) s += d.eg_cnt[i];
  uint32_t tot;
  block_excl_scan<uint32_t>(s, scr, &tot);
  if (threadIdx.x == 0) d.eg_tile[t] = tot;
}

__global__ __launch_bounds__(1024) void k_eg_tscan(PftDev d) {
  __shared__ uint32_t scr[20];
  const uint32_t nc = d.hdr->eg_ncells;
  const uint32_t ntiles = (nc + EG_TILE - 1u) / EG_TILE;
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < ntiles; t0 += 1024u) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < ntiles ? d.eg_tile[t] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<uint32_t>(v, scr, &tot);
    if (t < ntiles) d.eg_tile[t] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) d.eg_start[nc] = carry;  // == n_crop
}

__global__ __launch_bounds__(256) void k_eg_apply(PftDev d) {
  __shared__ uint32_t scr[20];
  const uint32_t nc = d.hdr->eg_ncells, t = blockIdx.x;
  if (t * EG_TILE >= nc) return;
  const uint32_t i0 = t * EG_TILE + threadIdx.x * 8u;  // 8 consecutive cells per thread
  uint32_t c[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c[k] = (i0 + k < nc) ? d.eg_cnt[i0 + k] : 0u;
    s += c[k];
  }
  uint32_t tot;
  uint32_t run = d.eg_tile[t] + block_excl_scan<uint32_t>(s, scr, &tot);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (i0 + k < nc) {
      d.eg_start[i0 + k] = run;
      d.eg_cnt[i0 + k] = 0u;  // becomes the fill cursor of the scatter
    }
    run += c[k];
  }
}

__global__ __launch_bounds__(256) void k_eg_scatter(PftDev d) {
  const PftHeader* h = d.hdr;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= h->n_crop) return;
  const float4 p = d.crop_pts[i];
  const uint32_t c = eg_cell_of(h, p.x, p.y, p.z);
  const uint32_t pos = d.eg_start[c] + atomicAdd(&d.eg_cnt[c], 1u);  // order inside a cell is irrelevant (see the query)
  d.leaf_pts[pos] = p;    // the octree's leaf arrays are free in this mode
  d.leaf_order[pos] = i;
}

void pftk_exact_grid(hipStream_t s, const PftParams& p, const PftDev& d) {
  const uint32_t ntiles = (d.eg_cap + EG_TILE - 1u) / EG_TILE;
  const uint32_t nb = (d.N + 255u) / 256u;
  hipLaunchKernelGGL(k_eg_setup, dim3(1), dim3(1), 0, s, p, d);
  hipLaunchKernelGGL(k_eg_zero, dim3((d.eg_cap + 1023u) / 1024u), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_eg_count, dim3(nb ? nb : 1), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_eg_tsum, dim3(ntiles), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_eg_tscan, dim3(1), dim3(1024), 0, s, d);
  hipLaunchKernelGGL(k_eg_apply, dim3(ntiles), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_eg_scatter, dim3(nb ? nb : 1), dim3(256), 0, s, d);
}

// ---- candidate lists ----
// The queries of one iteration (P x M of them) fall into a few ten thousand grid cells.  For a cell with centre m and
// half diagonal r, let D = distance from m to the nearest cloud point.  A query q of that cell has its nearest
// neighbour no farther than D + r, so every point that can be the nearest neighbour (or tie with it) of ANY query of
// the cell lies within D + 2r of m -- and only neighbours inside the gate count, so within gate + r as well.  These
// few dozen to few hundred points are collected ONCE per cell (one wave per cell, lanes over grid rows); a query then
// only walks its cell's list.  The lists live in one pool (PFT_EC_POOL entries); cells that find it full and queries
// outside the grid take the shell search below.
#define EC_NOLIST 0xffffffffu

// unclamped cell coordinates of a query; false if it lies outside the grid
__device__ __forceinline__ bool eg_query_cell(const PftHeader* h, float qx, float qy, float qz, int& cx, int& cy, int& cz) {
  cx = (int)floorf((qx - h->eg_min[0]) * h->eg_inv_g);
  cy = (int)floorf((qy - h->eg_min[1]) * h->eg_inv_g);
  cz = (int)floorf((qz - h->eg_min[2]) * h->eg_inv_g);
  return cx >= 0 && cx < h->eg_dim[0] && cy >= 0 && cy < h->eg_dim[1] && cz >= 0 && cz < h->eg_dim[2];
}

// ---- the queries of a workgroup's tile of particles, aggregated by cell in LDS ----
// Sixteen million global atomics (one per query) take 1.5 ms; the queries of neighbouring particles fall into the same
// few thousand cells, so a workgroup first counts its tile in an LDS hash table (cell -> count) and then issues one
// global atomic per cell it has seen.  A cell that finds its probe window full is counted with global atomics directly
// (wide particle clouds); the window is a function of the table's final state only, so every query of a cell takes
// the same route in every pass.
#define EQ_TAB 8192u  // entries; 64 KiB with the values
#define EQ_EMPTY 0xffffffffu
#define EQ_PROBES 48

static_assert(EQ_TAB == 8192u, "eq_hash yields 13 bits");
__device__ __forceinline__ uint32_t eq_hash(uint32_t c) { return (c * 2654435761u) >> 19; }  // 13 bits

__device__ __forceinline__ uint32_t eq_insert(uint32_t* key, uint32_t c) {
  uint32_t hh = eq_hash(c);
  for (int probe = 0; probe < EQ_PROBES; probe++) {
    uint32_t k = key[hh];
    if (k == EQ_EMPTY) k = atomicCAS(&key[hh], EQ_EMPTY, c);
    if (k == c || k == EQ_EMPTY) return hh;
    hh = (hh + 1u) & (EQ_TAB - 1u);
  }
  return EQ_TAB;
}

__device__ __forceinline__ uint32_t eq_find(const uint32_t* key, uint32_t c) {
  uint32_t hh = eq_hash(c);
  for (int probe = 0; probe < EQ_PROBES; probe++) {
    const uint32_t k = key[hh];
    if (k == c) return hh;
    if (k == EQ_EMPTY) return EQ_TAB;
    hh = (hh + 1u) & (EQ_TAB - 1u);
  }
  return EQ_TAB;
}

// one pass of a workgroup over the queries of its particles [p0, p1): f(particle, reference position, qx, qy, qz)
template <class F>
__device__ __forceinline__ void eq_tile_queries(const PftParams& prm, const PftDev& d, uint32_t p0, uint32_t p1, F&& f) {
  const uint32_t M = prm.M, nblk = (M + 511u) / 512u;
  const uint32_t lane = (uint32_t)lane_id(), nw = blockDim.x >> 6;
  const uint32_t n_items = (p1 - p0) * nblk;
  for (uint32_t item_v = wave_id(); item_v < n_items; item_v += nw) {
    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_v);
    const uint32_t pi = p0 + item / nblk, ch = item % nblk;
    float T[12];
    load_matrix(d.mats, pi, T);
    const uint32_t jend = min(M, (ch + 1u) * 512u);
    for (uint32_t j = ch * 512u + lane; j < jend; j += WAVE) {
      const float4 r = d.ref_xyz[j];
      float qx, qy, qz;
      xform(T, r.x, r.y, r.z, qx, qy, qz);
      f(pi, j, qx, qy, qz);
    }
  }
}

// every query counts itself in its cell
__global__ __launch_bounds__(1024) void k_ec_mark(PftParams prm, PftDev d, uint32_t n_particles, uint32_t ppw) {
  extern __shared__ __attribute__((aligned(16))) uint32_t eq_lds[];
  uint32_t* key = eq_lds;
  uint32_t* val = eq_lds + EQ_TAB;
  const PftHeader* h = d.hdr;
  if (h->n_crop == 0) return;
  if (d.p_active) n_particles = *d.p_active;
  const uint32_t p0 = min(n_particles, blockIdx.x * ppw), p1 = min(n_particles, p0 + ppw);
  if (p0 >= p1) return;
  for (uint32_t e = threadIdx.x; e < EQ_TAB; e += blockDim.x) {
    key[e] = EQ_EMPTY;
    val[e] = 0u;
  }
  __syncthreads();
  const int dx_ = h->eg_dim[0], dy_ = h->eg_dim[1];
  eq_tile_queries(prm, d, p0, p1, [&](uint32_t, uint32_t, float qx, float qy, float qz) {
    int cx, cy, cz;
    if (!eg_query_cell(h, qx, qy, qz, cx, cy, cz)) return;
    const uint32_t c = (uint32_t)((cz * dy_ + cy) * dx_ + cx);
    const uint32_t e = eq_insert(key, c);
    if (e < EQ_TAB)
      atomicAdd(&val[e], 1u);
    else
      atomicAdd(&d.eq_cellq[c], 1u);
  });
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < EQ_TAB; e += blockDim.x)
    if (key[e] != EQ_EMPTY && val[e]) atomicAdd(&d.eq_cellq[key[e]], val[e]);
  // the table is kept for k_eq_scatter, which would otherwise count the same tile again
  if (d.eq_tiles && blockIdx.x < d.eq_tiles_cap) {
    uint32_t* out = d.eq_tiles + (size_t)blockIdx.x * 2u * EQ_TAB;
    for (uint32_t e = threadIdx.x; e < 2u * EQ_TAB; e += blockDim.x) out[e] = eq_lds[e];
  }
}

// the cells with queries get list slots (order irrelevant) -- and, for the cell-sorted search, their segment of the
// sorted query array and their 64-query blocks: three running totals, one 64-bit and one 32-bit atomic per wave
__global__ __launch_bounds__(256) void k_ec_slots(PftDev d) {
  PftHeader* h = d.hdr;
  const uint32_t nc = h->n_crop ? h->eg_ncells : 0u;
  const int lane = lane_id();
  for (uint32_t c0 = blockIdx.x * 256u; c0 < nc; c0 += gridDim.x * 256u) {
    const uint32_t c = c0 + threadIdx.x;
    const uint32_t nq = c < nc ? d.eq_cellq[c] : 0u;
    const bool hit = nq != 0u;
    const unsigned long long m = __ballot(hit);
    if (!m) continue;
    const int first = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(&h->ec_nslots, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, first);
    const uint32_t my_slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const bool placed = hit && my_slot < PFT_EC_SLOTS;  // (a cell beyond the slot capacity gets no list and no segment)
    const uint32_t nb = placed ? (nq + 63u) >> 6 : 0u;
    const unsigned long long mine = ((unsigned long long)nb << 32) | (placed ? nq : 0u);  // (blocks << 32) | queries
    const unsigned long long incl = wave_incl_scan(mine);
    unsigned long long qb = 0;
    if (lane == WAVE - 1) qb = atomicAdd(&h->eq_totals, incl);
    qb = __shfl(qb, WAVE - 1) + incl - mine;
    if (hit) {
      const uint32_t sl = my_slot;
      if (placed) {
        d.ec_cells[sl] = c;
        d.eq_nq[sl] = nq;
        d.eq_qbase[sl] = (uint32_t)qb;
        d.eq_bbase[sl] = (uint32_t)(qb >> 32);
        d.eq_fill[sl] = 0u;
        d.ec_slot[c] = sl + 2u;
        if (d.eq_blk)
          for (uint32_t bq = 0; bq < nb; bq++)
            if ((uint32_t)(qb >> 32) + bq < d.eq_blk_cap) d.eq_blk[(uint32_t)(qb >> 32) + bq] = sl;
      } else {
        d.ec_slot[c] = EC_NOLIST;
      }
    }
  }
}

// ---- a wave over the points of many row segments ----
// A lane per row with its own loop over the row's points leaves the wave waiting for its longest row, one dependent
// load after the other.  Instead: the lanes fetch the segments' [start, end) (all loads of a round in flight), a prefix
// sum over the lengths lays the points out in one flat index space, and the lanes walk THAT, 64 points per round; a
// point's segment is found by binary search in the prefix array (LDS).  `seg(e, s0, s1)` yields segment e (s0 >= s1:
// empty), `pt4(live[4], pos[4])` takes four points per lane; both are called by all lanes of the wave.
#define EC_SEGS 64
struct SegScratch {
  uint32_t s0[EC_SEGS], pre[EC_SEGS + 1];
};
template <class SegF, class PtF>
__device__ __forceinline__ void wave_segments_points(SegScratch& S, int nseg, SegF&& seg, PtF&& pt4) {
  const int lane = lane_id();
  for (int e0 = 0; e0 < nseg; e0 += EC_SEGS) {
    const int n = min(EC_SEGS, nseg - e0);
    for (int e = lane; e < n; e += WAVE) {
      uint32_t a = 0, b = 0;
      seg(e0 + e, a, b);
      S.s0[e] = a;
      S.pre[e] = b > a ? b - a : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // exclusive prefix: a contiguous run of entries per lane, one wave scan
    const int per = (n + WAVE - 1) / WAVE, a0 = min(n, lane * per), a1 = min(n, a0 + per);
    uint32_t mine = 0;
    for (int e = a0; e < a1; e++) mine += S.pre[e];
    const uint32_t incl = wave_incl_scan(mine);
    uint32_t run = incl - mine;
    for (int e = a0; e < a1; e++) {
      const uint32_t len = S.pre[e];
      S.pre[e] = run;
      run += len;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, WAVE - 1);
    if (lane == 0) S.pre[n] = total;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // four points per lane and round, their searches and loads independent of one another (a round is a chain of LDS
    // and memory latencies; the rounds of one wave do not overlap by themselves)
    for (uint32_t i0 = 0; i0 < total; i0 += 4u * WAVE) {
      uint32_t pos[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t i = i0 + (uint32_t)(u * WAVE + lane);
        live[u] = i < total;
        const uint32_t ii = live[u] ? i : 0u;
        // the last entry e with pre[e] <= i (empty segments repeat a value: the LAST of them is the one with points)
        int lo = 0;
#pragma unroll
        for (int step = EC_SEGS / 2; step > 0; step >>= 1)
          if (lo + step < n && S.pre[lo + step] <= ii) lo += step;
        pos[u] = S.s0[lo] + (ii - S.pre[lo]);
      }
      pt4(live, pos);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// -DPFT_EC_TIMING: per-phase wave time of k_ec_build (100 MHz ticks, summed over the waves) in PftHeader::dbg[16..]
#ifdef PFT_EC_TIMING
#define EC_TICK(k) do { const unsigned long long now_ = wall_clock64(); if (lane == 0) atomicAdd(&d.hdr->dbg[16 + (k)], now_ - tick_); tick_ = now_; } while (0)
#else
#define EC_TICK(k) do { } while (0)
#endif

// one wave per list: D, then the points within min(D + 2r, gate + r) of the cell centre.
// The cells are sorted (z, y, x)-major, so the cells of one z and a RANGE of y -- all x -- are one contiguous piece
// of the sorted cloud: a neighbourhood of (2K + 1)^3 cells is 2K + 1 such slabs, two loads of cell starts each, and the
// slabs' points are read with consecutive lanes on consecutive records.  A slab holds points outside the x range of the
// neighbourhood too (the cloud is sparse: a few points per grid row), which the distance tests sort out; per-row
// segments trimmed in x -- two scattered loads of cell starts per row, 225 rows around a cell -- were bound by the
// cache-line throughput of their divergent loads (93 % of the wave cycles in s_waitcnt).
#define EC_STAGE 448u  // staged candidates per wave (28 KiB for the four waves)
#define EC_CHUNK 512u  // list-pool entries a wave takes at a time
__global__ __launch_bounds__(256) void k_ec_build(PftParams prm, PftDev d) {
  __shared__ uint32_t wcnt[4];
  __shared__ float4 stage[4][EC_STAGE];
  __shared__ SegScratch segs[4];
  const PftHeader* h = d.hdr;
  const uint32_t ns = min(h->ec_nslots, (uint32_t)PFT_EC_SLOTS);
  const float g = h->eg_g, inv_g = h->eg_inv_g, hh = 0.5f * h->eg_g;
  const int dx_ = h->eg_dim[0], dy_ = h->eg_dim[1], dz_ = h->eg_dim[2];
  const float r = hh * 1.7320508f + 1.0e-4f;  // half diagonal of a cell + slack for the float cell assignment
  const float gate = sqrtf((float)prm.maxd2);
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + w, tw = gridDim.x * nw;
  const int Kmax = (int)ceilf((gate + r) * inv_g) + 1;
  const float gmin0 = h->eg_min[0], gmin1 = h->eg_min[1], gmin2 = h->eg_min[2];
  uint32_t chunk_base = 0, chunk_left = 0;  // (lane 0) this wave's piece of the list pool
  for (uint32_t s = gw; s < ns; s += tw) {  // wave-uniform
    const uint32_t c = d.ec_cells[s];
    const int cx = (int)(c % (uint32_t)dx_), cy = (int)((c / (uint32_t)dx_) % (uint32_t)dy_),
              cz = (int)(c / (uint32_t)(dx_ * dy_));
#ifdef PFT_EC_TIMING
    unsigned long long tick_ = wall_clock64();
#endif
    const float mx = gmin0 + ((float)cx + 0.5f) * g, my = gmin1 + ((float)cy + 0.5f) * g, mz = gmin2 + ((float)cz + 0.5f) * g;
    // slab e of the neighbourhood of K cells: z = cz - K + e, y within ky(e) cells of cy, every x
    auto slab = [&](int z, int ky, uint32_t& a, uint32_t& bnd) {
      if (z < 0 || z >= dz_) return;
      const int y0 = max(cy - ky, 0), y1 = min(cy + ky, dy_ - 1);
      if (y0 > y1) return;
      a = d.eg_start[(uint32_t)((z * dy_ + y0) * dx_)];
      bnd = d.eg_start[(uint32_t)((z * dy_ + y1) * dx_ + dx_)];
    };
    // ---- D: cubes of 2, 4 and Kmax cells around the cell until the nearest point is inside the cube's reach ----
    float best = INFINITY;
    bool found = false;
    for (int K = min(2, Kmax);; K = K < 4 ? min(4, Kmax) : Kmax) {
      wave_segments_points(
          segs[w], 2 * K + 1, [&](int e, uint32_t& a, uint32_t& bnd) { slab(cz - K + e, K, a, bnd); },
          [&](const bool* live, const uint32_t* pos) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = d.leaf_pts[live[u] ? pos[u] : 0u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const float dd = (p[u].x - mx) * (p[u].x - mx) + ((p[u].y - my) * (p[u].y - my) + (p[u].z - mz) * (p[u].z - mz));
              best = live[u] ? fminf(best, dd) : best;
            }
          });
      const float wb = wave_min(best);
      const float reach = (float)K * g + hh;  // every point outside the cube is at least this far from the centre
      found = wb <= reach * reach * 0.999f;
      if (found || K >= Kmax) break;  // (Kmax * g + hh >= gate + r: nothing farther matters)
    }
    EC_TICK(0);
    const float D = sqrtf(wave_min(best));
    if (!found || !(D - r <= gate)) {  // no query of this cell has a neighbour inside the gate
      if (lane == 0) d.ec_count[s] = 0u;
      EC_TICK(4);
      continue;
    }
    const float T = fminf(D + 2.0f * r, gate + r) + 1.0e-4f, T2 = T * T;
    // ---- the list: the slabs within T of the centre, each trimmed in y.  One walk: the candidates are staged in LDS (and
    // counted), then allotted in the pool and copied out; a list longer than the staging area is walked a second time ----
    const int KT = (int)ceilf((T + hh) * inv_g);
    uint32_t total = 0, base = 0;
    if (lane == 0) wcnt[w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int pass = 0; pass < 2; pass++) {
      wave_segments_points(
          segs[w], 2 * KT + 1,
          [&](int e, uint32_t& a, uint32_t& bnd) {
            const int oz = e - KT;
            const float lz = fmaxf((float)abs(oz) * g - hh, 0.0f) * 0.9999f;  // the slab is at least this far away in z
            if (lz * lz > T2) return;
            slab(cz + oz, (int)floorf((sqrtf(T2 - lz * lz) + hh) * inv_g) + 1, a, bnd);
          },
          [&](const bool* live, const uint32_t* pos) {
            float4 pp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pp[u] = d.leaf_pts[live[u] ? pos[u] : 0u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const float4 p = pp[u];
              const float ex = p.x - mx, ey = p.y - my, ez = p.z - mz;
              if (live[u] && ex * ex + (ey * ey + ez * ez) <= T2) {
                const uint32_t k = atomicAdd(&wcnt[w], 1u);
                const float4 e = make_float4(p.x, p.y, p.z, __uint_as_float(pos[u]));
                if (pass == 1)
                  d.ec_list[base + k] = e;
                else if (k < EC_STAGE)
                  stage[w][k] = e;
              }
            }
          });
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      EC_TICK(2);
      if (pass == 0) {
        total = wcnt[w];
        if (lane == 0) {
          // the pool is handed out in chunks: one returning atomic per list on the header's counter -- 22 000 of them on
          // one cache line, which every wave also reads -- was four fifths of this kernel's time (540 against 120 us)
          if (total > chunk_left) {
            if (total > EC_CHUNK / 2u) {
              base = atomicAdd(&d.hdr->ec_pool_used, total);  // a long list: its own piece, the chunk stays
            } else {
              chunk_base = atomicAdd(&d.hdr->ec_pool_used, EC_CHUNK);
              chunk_left = EC_CHUNK;
              base = chunk_base;
              chunk_base += total;
              chunk_left -= total;
            }
          } else {
            base = chunk_base;
            chunk_base += total;
            chunk_left -= total;
          }
          wcnt[w] = 0u;
        }
        base = (uint32_t)__shfl((int)base, 0);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (base + total > d.ec_pool_cap || base + total < base) {  // (wave-uniform) the pool is full: no list for this cell
          total = EC_NOLIST;
          break;
        }
        if (total <= EC_STAGE) {  // (wave-uniform) the usual case: copy the staged entries out
          for (uint32_t k = (uint32_t)lane; k < total; k += WAVE) d.ec_list[base + k] = stage[w][k];
          break;
        }
      }
    }
    if (lane == 0) {
      d.ec_count[s] = total;
      d.ec_base[s] = base;
    }
    EC_TICK(3);
  }
}

struct EgHit {
  float best;
  uint32_t bi;
  float4 bt;
};

__device__ __forceinline__ void eg_take(const PftDev& d, uint32_t pos, float qx, float qy, float qz, EgHit& b) {
  const float4 p = d.leaf_pts[pos];
  const float ex = p.x - qx, ey = p.y - qy, ez = p.z - qz;
  const float dd = ex * ex + (ey * ey + ez * ez);  // pointSquaredDist
  if (dd <= b.best) {
    const uint32_t idx = d.leaf_order[pos];
    if (dd < b.best || idx < b.bi) {  // equal distances: the lowest index
      b.best = dd;
      b.bi = idx;
      b.bt = p;
    }
  }
}

// Shells of growing Chebyshev radius rr around the query's cell (cells not clamped: a query outside the box starts its
// rings outside).  The cells of one (z, y) row are contiguous in the sorted order, so a row segment costs two loads of
// cell starts however many cells it spans: rows on the shell's faces are scanned over their whole x range, inner rows
// only at their two end cells.  A row whose cells are all farther than the best distance so far is skipped before
// anything is loaded (a point in a row at offset o is at least (|o| - 1) cells away on that axis), and once shell rr is
// done everything closer than rr * g has been seen.
__device__ void eg_shell_search(const PftDev& d, int cqx, int cqy, int cqz, float g, int dx_, int dy_, int dz_, int R,
                                float gate_f, float qx, float qy, float qz, EgHit& b) {
  const float gg = g * 0.9999f;
  for (int rr = 0; rr <= R; rr++) {
    for (int oz = -rr; oz <= rr; oz++) {
      const int cz = cqz + oz;
      if (cz < 0 || cz >= dz_) continue;
      const float lz = (float)max(abs(oz) - 1, 0) * gg;
      for (int oy = -rr; oy <= rr; oy++) {
        const int cy = cqy + oy;
        if (cy < 0 || cy >= dy_) continue;
        const float ly = (float)max(abs(oy) - 1, 0) * gg;
        const float lyz = ly * ly + lz * lz;
        const bool face = (abs(oz) == rr) || (abs(oy) == rr);
        const float lx = face ? 0.0f : (float)max(rr - 1, 0) * gg;
        if (lyz + lx * lx > fminf(b.best, gate_f)) continue;
        const uint32_t row = (uint32_t)((cz * dy_ + cy) * dx_);
        const int nseg = (face || rr == 0) ? 1 : 2;
        for (int sgi = 0; sgi < nseg; sgi++) {
          int x0, x1;
          if (nseg == 1) {
            x0 = max(cqx - rr, 0);
            x1 = min(cqx + rr, dx_ - 1);
          } else {
            x0 = x1 = sgi == 0 ? cqx - rr : cqx + rr;
            if (x0 < 0 || x0 >= dx_) continue;
          }
          if (x0 > x1) continue;
          const uint32_t s0 = d.eg_start[row + (uint32_t)x0], s1 = d.eg_start[row + (uint32_t)x1 + 1u];
          for (uint32_t pos = s0; pos < s1; pos++) eg_take(d, pos, qx, qy, qz, b);
        }
      }
    }
    const float reach = (float)rr * gg;  // everything closer than this has been seen
    if (b.best < reach * reach || reach * reach > gate_f) break;
  }
}

// ---- A7 with the exact nearest neighbour ----
struct CohParams {
  double wd, whsv, maxd2;
  float hw, sw, vw;
};

// DistanceCoherence x HSVColorCoherence of one pair (A7a, A7b, as in pft_likelihood.hip); 0 when the query has no
// neighbour inside the gate.  bt = the neighbour's record {x, y, z, packed hsv}, j = the reference point's position.
__device__ __forceinline__ double exact_pair_value(const PftDev& d, const CohParams& cp, const float* lut_h,
                                                   const float* lut_s, float qx, float qy, float qz, float best,
                                                   uint32_t bi, const float4& bt, uint32_t j) {
  if (!(bi != 0xffffffffu && (double)best < cp.maxd2)) return 0.0;
  const float ex = qx - bt.x, ey = qy - bt.y, ez = qz - bt.z;
  const float n2 = (ex * ex + ey * ey) + ez * ez;
  const double dist = (double)sqrt_rn_coherence(n2);
  const double A = 1.0 + dist * dist * cp.wd;
  const float4 rh = d.ref_hsv[j];
  const uint32_t pk = __float_as_uint(bt.w);
  const float th = lut_h[pk & 0xffu], ts = lut_s[(pk >> 8) & 0xffu], tv = lut_s[(pk >> 16) & 0xffu];
  const float hd1 = fabsf(rh.x - th);
  float hd2;
  if (rh.x < th)
    hd2 = fabsf(1.0f + rh.x - th);
  else
    hd2 = fabsf(1.0f + th - rh.x);
  float h_diff;
  if (hd1 < hd2)
    h_diff = cp.hw * hd1 * hd1;
  else
    h_diff = cp.hw * hd2 * hd2;
  const float s_diff = cp.sw * (rh.y - ts) * (rh.y - ts);
  const float v_diff = cp.vw * (rh.z - tv) * (rh.z - tv);
  const double Bq = 1.0 + cp.whsv * (double)(h_diff + s_diff + v_diff);
  return 1.0 / (A * Bq);
}

template <bool DEBUG_NN>
__global__ __launch_bounds__(256) void k_likelihood_exact(PftParams prm, PftDev d, uint32_t n_particles, int use_lists) {
  __shared__ float lut_h[256], lut_s[256];
  const PftHeader* h = d.hdr;
  if (d.p_active) n_particles = *d.p_active;
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    lut_h[i] = (float)i / 180.0f;
    lut_s[i] = (float)i / 255.0f;
  }
  __syncthreads();
  // the candidate pool follows the demand: entries asked for in this iteration (may exceed the capacity: the cells that
  // found it full kept the ring search), read by the host without synchronising before the next iteration
  if (blockIdx.x == 0 && threadIdx.x == 0 && d.host_stat) d.host_stat[4] = h->ec_pool_used;
  if (h->error && blockIdx.x == 0 && threadIdx.x == 0 && d.host_stat) {  // as k_likelihood: surfaced at the next host sync
    d.host_stat[2] = h->error;
    d.host_stat[3] |= h->error;
  }
  const uint32_t n_crop = h->n_crop;
  const float g = h->eg_g, inv_g = h->eg_inv_g;
  const int dx_ = h->eg_dim[0], dy_ = h->eg_dim[1], dz_ = h->eg_dim[2];
  const double maxd2 = prm.maxd2;
  const float gate_f = (float)maxd2 * 1.0001f;  // pruning bound; the gate itself is the double comparison below
  const int R = (int)ceilf(sqrtf((float)maxd2) * inv_g) + 1;
  const CohParams cp = {prm.dist_w, prm.hsv_w, maxd2, prm.h_w, prm.s_w, prm.v_w};
  const uint32_t M = prm.M, nchunk = prm.nchunk;
  const int lane = lane_id(), nw = blockDim.x >> 6;
  const uint32_t n_items = n_particles * nchunk;
  // work items handed out dynamically inside groups of workgroups, as in pft_likelihood.hip (list lengths vary a lot)
  const uint32_t G = min((uint32_t)PFT_LIK_GROUPS, gridDim.x), grp = blockIdx.x % G;
  const uint32_t gq = gridDim.x / G, gr = gridDim.x % G;
  const uint32_t wgs_in_grp = gq + (grp < gr ? 1u : 0u), waves_in_grp = wgs_in_grp * (uint32_t)nw;
  const uint32_t wgs_before = grp * gq + min(grp, gr);
  const uint32_t it_begin = (uint32_t)((unsigned long long)n_items * wgs_before / gridDim.x);
  const uint32_t it_end = (uint32_t)((unsigned long long)n_items * (wgs_before + wgs_in_grp) / gridDim.x);
  uint32_t* ctr = &d.hdr->lik_ctr[grp * PFT_LIK_CTR_STRIDE];
  for (uint32_t item_v = it_begin + (blockIdx.x / G) * (uint32_t)nw + wave_id(); item_v < it_end;) {
    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_v);
    const uint32_t pi = item / nchunk, ch = item % nchunk;
    float T[12];
    load_matrix(d.mats, pi, T);
    double val = 0.0;
    const uint32_t jend = min(M, (ch + 1) * prm.ref_chunk);
    for (uint32_t j = ch * prm.ref_chunk + lane; j < jend; j += WAVE) {
      const float4 r = d.ref_xyz[j];
      float qx, qy, qz;
      xform(T, r.x, r.y, r.z, qx, qy, qz);
      EgHit b;
      b.best = INFINITY;
      b.bi = 0xffffffffu;
      b.bt = make_float4(0, 0, 0, 0);
      if (n_crop > 0) {
        int cqx, cqy, cqz;
        const bool inside = eg_query_cell(h, qx, qy, qz, cqx, cqy, cqz);
        uint32_t cnt = EC_NOLIST, slot = 0;
        if (use_lists && inside) {
          const uint32_t sl = d.ec_slot[(uint32_t)((cqz * dy_ + cqy) * dx_ + cqx)];
          if (sl >= 2u && sl != EC_NOLIST) {
            slot = sl - 2u;
            cnt = d.ec_count[slot];
          }
        }
        if (DEBUG_NN) {  // list statistics (tools/exact_nn_bench.py): queries, served by a list, candidates walked
          atomicAdd(&d.hdr->dbg[0], 1ull);
          if (cnt != EC_NOLIST) {
            atomicAdd(&d.hdr->dbg[1], 1ull);
            atomicAdd(&d.hdr->dbg[2], (unsigned long long)cnt);
          }
          if (!inside) atomicAdd(&d.hdr->dbg[3], 1ull);
          int mc = cnt != EC_NOLIST ? (int)cnt : 100000;
          for (int o = 32; o > 0; o >>= 1) mc = max(mc, __shfl_xor(mc, o));
          if (lane == __ffsll((long long)__ballot(1)) - 1) {
            atomicAdd(&d.hdr->dbg[4], 1ull);
            if (mc < 100000) atomicAdd(&d.hdr->dbg[5], (unsigned long long)mc);
            else atomicAdd(&d.hdr->dbg[6], 1ull);
          }
        }
        if (cnt != EC_NOLIST) {  // the cell's candidate list holds every possible in-gate nearest neighbour
          // four candidates in flight per round; entries past the end repeat the last one (harmless: a repeat is
          // neither closer nor, with the same index, earlier)
          const float4* list = d.ec_list + d.ec_base[slot];
          uint32_t bpos = 0xffffffffu;
          for (uint32_t k = 0; k < cnt; k += 4u) {
            const float4 c0 = list[k], c1 = list[min(k + 1u, cnt - 1u)], c2 = list[min(k + 2u, cnt - 1u)],
                         c3 = list[min(k + 3u, cnt - 1u)];
#define EC_TEST(C)                                                                         \
  {                                                                                        \
    const float ex = (C).x - qx, ey = (C).y - qy, ez = (C).z - qz;                         \
    const float dd = ex * ex + (ey * ey + ez * ez);                                        \
    const uint32_t pos = __float_as_uint((C).w);                                           \
    if (dd == b.best && pos != bpos) { /* equal distances (rare): the lowest cloud index */ \
      if (d.leaf_order[pos] < d.leaf_order[bpos]) bpos = pos;                              \
    }                                                                                      \
    const bool lt = dd < b.best;                                                           \
    b.best = lt ? dd : b.best;                                                             \
    bpos = lt ? pos : bpos;                                                                \
  }
            EC_TEST(c0);
            EC_TEST(c1);
            EC_TEST(c2);
            EC_TEST(c3);
#undef EC_TEST
          }
          if (bpos != 0xffffffffu) {
            b.bi = d.leaf_order[bpos];
            b.bt = d.leaf_pts[bpos];
          }
        } else {
          eg_shell_search(d, cqx, cqy, cqz, g, dx_, dy_, dz_, R, gate_f, qx, qy, qz, b);
        }
      }
      const float best = b.best;
      const uint32_t bi = b.bi;
      const float4 bt = b.bt;
      if (DEBUG_NN) {
        const size_t o = (size_t)pi * M + d.ref_perm[j];
        const bool in_gate = bi != 0xffffffffu && (double)best < maxd2;
        d.nn_idx[o] = in_gate ? (int32_t)bi : -1;  // neighbours outside the gate are not searched for
        d.nn_d2[o] = in_gate ? best : INFINITY;
      }
      val += exact_pair_value(d, cp, lut_h, lut_s, qx, qy, qz, best, bi, bt, j);
    }
    val = wave_sum(val);
    if (lane == 0) d.partial[(size_t)pi * nchunk + ch] = val;
    uint32_t nx = 0;
    if (lane == 0) nx = it_begin + waves_in_grp + atomicAdd(ctr, 1u);
    item_v = (uint32_t)__shfl((int)nx, 0);
  }
}


// ---- the same search with the queries sorted by grid cell ----
// k_likelihood_exact above walks a list per LANE: 64 lanes, some 40 different lists, every candidate a divergent
// 16-byte gather -- the kernel is bound by those gathers, not by arithmetic.  Sorted by cell, 64 queries of ONE cell
// share a wave: the candidate is wave-uniform (scalar loads), a test is a dozen VALU instructions and no memory
// instruction.  k_ec_mark has counted the queries per cell, k_ec_slots has given every such cell a segment of the sorted
// array and its 64-query blocks, k_eq_scatter writes {q, id} into the segments (order inside a segment is
// irrelevant: a query's result goes to out[id]), k_eq_search is one wave per block, k_eq_reduce sums out[] per
// (particle, chunk) in the order of the per-query kernel -- the two paths give the same bits.

template <bool DEBUG_NN>
__device__ __forceinline__ void eq_finish(const PftDev& d, const CohParams& cp, const float* lut_h, const float* lut_s,
                                          uint32_t M, uint32_t id, float qx, float qy, float qz, float best, uint32_t bi,
                                          const float4& bt) {
  const uint32_t pi = id / M, j = id - pi * M;
  if (DEBUG_NN) {
    const size_t o = (size_t)pi * M + d.ref_perm[j];
    const bool in_gate = bi != 0xffffffffu && (double)best < cp.maxd2;
    d.nn_idx[o] = in_gate ? (int32_t)bi : -1;  // neighbours outside the gate are not searched for
    d.nn_d2[o] = in_gate ? best : INFINITY;
  }
  d.eq_out[id] = exact_pair_value(d, cp, lut_h, lut_s, qx, qy, qz, best, bi, bt, j);
}

// every query goes into its cell's segment -- or is answered on the spot: no cloud, a cell whose list is empty (no
// neighbour inside the gate for any of its queries), or no list at all (outside the grid, pool full: shell search).
// A workgroup counts its tile per cell in LDS as k_ec_mark does, reserves a piece of each cell's segment with ONE global
// atomic, and hands out the positions inside the piece with LDS atomics.
#define EQ_INLINE_EMPTY 0xfffffffeu
#define EQ_INLINE_SHELL 0xffffffffu
template <bool DEBUG_NN>
__global__ __launch_bounds__(1024) void k_eq_scatter(PftParams prm, PftDev d, uint32_t n_particles, uint32_t ppw) {
  extern __shared__ __attribute__((aligned(16))) uint32_t eq_lds[];
  uint32_t* key = eq_lds;
  uint32_t* val = eq_lds + EQ_TAB;
  float* lut_h = reinterpret_cast<float*>(eq_lds + 2u * EQ_TAB);
  float* lut_s = lut_h + 256;
  const PftHeader* h = d.hdr;
  if (d.p_active) n_particles = *d.p_active;
  const uint32_t p0 = min(n_particles, blockIdx.x * ppw), p1 = min(n_particles, p0 + ppw);
  if (p0 >= p1) return;
  for (uint32_t e = threadIdx.x; e < EQ_TAB; e += blockDim.x) {
    key[e] = EQ_EMPTY;
    val[e] = 0u;
  }
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    lut_h[i] = (float)i / 180.0f;
    lut_s[i] = (float)i / 255.0f;
  }
  __syncthreads();
  const uint32_t n_crop = h->n_crop;
  const float g = h->eg_g, inv_g = h->eg_inv_g;
  const int dx_ = h->eg_dim[0], dy_ = h->eg_dim[1], dz_ = h->eg_dim[2];
  const float gate_f = (float)prm.maxd2 * 1.0001f;
  const int R = (int)ceilf(sqrtf((float)prm.maxd2) * inv_g) + 1;
  const CohParams cp = {prm.dist_w, prm.hsv_w, prm.maxd2, prm.h_w, prm.s_w, prm.v_w};
  const uint32_t M = prm.M;
  // where a cell's queries go: a segment position, or one of the two answers on the spot
  auto route_of_cell = [&](uint32_t c, uint32_t& slot) -> uint32_t {
    const uint32_t sl = d.ec_slot[c];
    if (sl < 2u || sl == EC_NOLIST) return EQ_INLINE_SHELL;
    slot = sl - 2u;
    const uint32_t cnt = d.ec_count[slot];
    return cnt == EC_NOLIST ? EQ_INLINE_SHELL : (cnt == 0u ? EQ_INLINE_EMPTY : 0u);
  };
  if (n_crop > 0) {
    if (d.eq_tiles && blockIdx.x < d.eq_tiles_cap) {  // the tile's cells and counts as k_ec_mark left them
      const uint32_t* in = d.eq_tiles + (size_t)blockIdx.x * 2u * EQ_TAB;
      for (uint32_t e = threadIdx.x; e < 2u * EQ_TAB; e += blockDim.x) eq_lds[e] = in[e];
    } else {
      eq_tile_queries(prm, d, p0, p1, [&](uint32_t, uint32_t, float qx, float qy, float qz) {
        int cx, cy, cz;
        if (!eg_query_cell(h, qx, qy, qz, cx, cy, cz)) return;
        const uint32_t e = eq_insert(key, (uint32_t)((cz * dy_ + cy) * dx_ + cx));
        if (e < EQ_TAB) atomicAdd(&val[e], 1u);
      });
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < EQ_TAB; e += blockDim.x) {
      if (key[e] == EQ_EMPTY) continue;
      uint32_t slot = 0;
      const uint32_t route = route_of_cell(key[e], slot);
      val[e] = route ? route : d.eq_qbase[slot] + atomicAdd(&d.eq_fill[slot], val[e]);
    }
    __syncthreads();
  }
  eq_tile_queries(prm, d, p0, p1, [&](uint32_t pi, uint32_t j, float qx, float qy, float qz) {
    const uint32_t id = pi * M + j;
    EgHit b;
    b.best = INFINITY;
    b.bi = 0xffffffffu;
    b.bt = make_float4(0, 0, 0, 0);
    if (n_crop > 0) {
      int cqx, cqy, cqz;
      uint32_t route = EQ_INLINE_SHELL, pos = 0;
      if (eg_query_cell(h, qx, qy, qz, cqx, cqy, cqz)) {
        const uint32_t c = (uint32_t)((cqz * dy_ + cqy) * dx_ + cqx);
        const uint32_t e = eq_find(key, c);
        if (e < EQ_TAB) {
          route = val[e];
          if (route < EQ_INLINE_EMPTY) pos = atomicAdd(&val[e], 1u);  // (positions stay far below the two sentinels: <= 2^30)
        } else {  // a cell that found its probe window full: as without the table
          uint32_t slot = 0;
          route = route_of_cell(c, slot);
          if (!route) pos = d.eq_qbase[slot] + atomicAdd(&d.eq_fill[slot], 1u);
        }
      }
      if (route < EQ_INLINE_EMPTY) {
        if (pos < d.eq_cap) d.eq_sorted[pos] = make_float4(qx, qy, qz, __uint_as_float(id));
        return;
      }
      if (route == EQ_INLINE_SHELL) eg_shell_search(d, cqx, cqy, cqz, g, dx_, dy_, dz_, R, gate_f, qx, qy, qz, b);
    }
    eq_finish<DEBUG_NN>(d, cp, lut_h, lut_s, M, id, qx, qy, qz, b.best, b.bi, b.bt);
  });
}

typedef float __attribute__((ext_vector_type(4))) eq_f4;
typedef const __attribute__((address_space(4))) eq_f4* eq_const_list;  // wave-uniform address: scalar loads

// one wave per block of 64 queries of one cell
template <bool DEBUG_NN>
__global__ __launch_bounds__(256) void k_eq_search(PftParams prm, PftDev d) {
  __shared__ float lut_h[256], lut_s[256];
  const PftHeader* h = d.hdr;
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    lut_h[i] = (float)i / 180.0f;
    lut_s[i] = (float)i / 255.0f;
  }
  __syncthreads();
  const uint32_t nb = min((uint32_t)(h->eq_totals >> 32), d.eq_blk_cap);
  const CohParams cp = {prm.dist_w, prm.hsv_w, prm.maxd2, prm.h_w, prm.s_w, prm.v_w};
  const uint32_t M = prm.M;
  const uint32_t lane = (uint32_t)lane_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + wave_id(), tw = gridDim.x * nw;
  for (uint32_t bv = gw; bv < nb; bv += tw) {
    const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv);
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.eq_blk[blk]);
    if (slot >= PFT_EC_SLOTS) continue;  // (never written by k_ec_slots)
    const uint32_t j0 = (blk - d.eq_bbase[slot]) * 64u, nq = d.eq_nq[slot];
    const uint32_t n = min(64u, nq - j0);
    const uint32_t cnt = d.ec_count[slot];
    if (cnt == EC_NOLIST || cnt == 0u) continue;  // (uniform) these cells' queries were answered by k_eq_scatter
    const eq_const_list list = (eq_const_list)(uintptr_t)(d.ec_list + d.ec_base[slot]);
    const bool live = lane < n;
    const float4 rec = d.eq_sorted[min((uint32_t)(d.eq_qbase[slot] + j0 + (live ? lane : 0u)), d.eq_cap - 1u)];
    const float qx = rec.x, qy = rec.y, qz = rec.z;
    float best = INFINITY;
    uint32_t bpos = 0xffffffffu;
    bool tie = false;  // some candidate was exactly as far as the best before it: settled after the loop
#define EQ_TEST(C)                                                  \
  {                                                                 \
    const float ex = (C).x - qx, ey = (C).y - qy, ez = (C).z - qz;  \
    const float dd = ex * ex + (ey * ey + ez * ez); /* pointSquaredDist */ \
    tie |= dd == best;                                              \
    const bool lt = dd < best;                                      \
    best = lt ? dd : best;                                          \
    bpos = lt ? __float_as_uint((C).w) : bpos;                      \
  }
    uint32_t k = 0;
    for (; k + 4u <= cnt; k += 4u) {  // four scalar loads in flight
      const eq_f4 c0 = list[k], c1 = list[k + 1u], c2 = list[k + 2u], c3 = list[k + 3u];
      EQ_TEST(c0);
      EQ_TEST(c1);
      EQ_TEST(c2);
      EQ_TEST(c3);
    }
    for (; k < cnt; k++) {
      const eq_f4 c = list[k];
      EQ_TEST(c);
    }
#undef EQ_TEST
    if (tie) {  // equal distances (rare): among the candidates at the minimum, the lowest cloud index
      uint32_t bidx = d.leaf_order[bpos];
      for (uint32_t kk = 0; kk < cnt; kk++) {
        const eq_f4 c = list[kk];
        const float ex = c.x - qx, ey = c.y - qy, ez = c.z - qz;
        const float dd = ex * ex + (ey * ey + ez * ez);
        if (dd == best) {
          const uint32_t pos = __float_as_uint(c.w), idx = d.leaf_order[pos];
          if (idx < bidx) {
            bidx = idx;
            bpos = pos;
          }
        }
      }
    }
    if (live) {
      uint32_t bi = 0xffffffffu;
      float4 bt = make_float4(0, 0, 0, 0);
      if (bpos != 0xffffffffu) {
        bi = d.leaf_order[bpos];
        bt = d.leaf_pts[bpos];
      }
      eq_finish<DEBUG_NN>(d, cp, lut_h, lut_s, M, __float_as_uint(rec.w), qx, qy, qz, best, bi, bt);
    }
  }
}

// partial[particle][chunk] = the sum of its queries' values, lane by lane and then across the wave exactly as
// k_likelihood_exact accumulates them
__global__ __launch_bounds__(256) void k_eq_reduce(PftParams prm, PftDev d, uint32_t n_particles) {
  const PftHeader* h = d.hdr;
  if (d.p_active) n_particles = *d.p_active;
  // the candidate pool follows the demand: entries asked for in this iteration (may exceed the capacity: the cells that
  // found it full kept the ring search), read by the host without synchronising before the next iteration
  if (blockIdx.x == 0 && threadIdx.x == 0 && d.host_stat) d.host_stat[4] = h->ec_pool_used;
  if (h->error && blockIdx.x == 0 && threadIdx.x == 0 && d.host_stat) {  // as k_likelihood: surfaced at the next host sync
    d.host_stat[2] = h->error;
    d.host_stat[3] |= h->error;
  }
  const uint32_t M = prm.M, nchunk = prm.nchunk;
  const uint32_t lane = (uint32_t)lane_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + wave_id(), tw = gridDim.x * nw;
  const uint32_t n_items = n_particles * nchunk;
  for (uint32_t item = gw; item < n_items; item += tw) {
    const uint32_t pi = item / nchunk, ch = item % nchunk;
    double val = 0.0;
    const uint32_t jend = min(M, (ch + 1u) * prm.ref_chunk);
    for (uint32_t j = ch * prm.ref_chunk + lane; j < jend; j += WAVE) {
      const double v = d.eq_out[(size_t)pi * M + j];
      if (v != 0.0) val += v;  // (the per-query kernel adds only the pairs inside the gate)
    }
    val = wave_sum(val);
    if (lane == 0) d.partial[(size_t)pi * nchunk + ch] = val;
  }
}

void pftk_likelihood_exact(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool debug_nn,
                           int num_cus) {
  const int use_lists = getenv("PFT_EXACT_SHELLS_ONLY") ? 0 : 1;  // 0: the per-query shell search alone (cross-check, A/B timing)
  // 1: a list walk per lane instead of the cell-sorted search (cross-check -- identical bits --, A/B timing)
  const bool per_query = getenv("PFT_EXACT_PER_QUERY") != nullptr;
  const uint32_t items = n_particles * p.nchunk;
  uint32_t grid = 8u * (uint32_t)num_cus;
  const uint32_t need = (items + 3u) / 4u;
  if (grid > need) grid = need ? need : 1u;
  // tiles of the LDS-aggregated passes: one workgroup per `ppw` particles
  static bool attr_set[PFT_MAX_DEVICES];
  const uint32_t tile_lds = 2u * EQ_TAB * 4u + 2048u;
  const int dev = pftk_cur_device();
  if (!attr_set[dev])
    attr_set[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ec_mark), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eq_scatter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eq_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds) == hipSuccess;
  uint32_t ppw = (n_particles + (uint32_t)num_cus - 1u) / (uint32_t)num_cus;
  ppw = ppw < 1u ? 1u : (ppw > 32u ? 32u : ppw);
  const uint32_t tiles = (n_particles + ppw - 1u) / ppw ? (n_particles + ppw - 1u) / ppw : 1u;
  if (use_lists) {
    hipLaunchKernelGGL(k_ec_mark, dim3(tiles), dim3(1024), tile_lds, s, p, d, n_particles, ppw);
    hipLaunchKernelGGL(k_ec_slots, dim3(4u * (uint32_t)num_cus), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_ec_build, dim3(8u * (uint32_t)num_cus), dim3(256), 0, s, p, d);
  }
  const unsigned long long nq = (unsigned long long)n_particles * p.M;
  if (use_lists && !per_query && d.eq_cap && nq <= d.eq_cap) {
    if (debug_nn) {
      hipLaunchKernelGGL(k_eq_scatter<true>, dim3(tiles), dim3(1024), tile_lds, s, p, d, n_particles, ppw);
      hipLaunchKernelGGL(k_eq_search<true>, dim3(16u * (uint32_t)num_cus), dim3(256), 0, s, p, d);
    } else {
      hipLaunchKernelGGL(k_eq_scatter<false>, dim3(tiles), dim3(1024), tile_lds, s, p, d, n_particles, ppw);
      hipLaunchKernelGGL(k_eq_search<false>, dim3(16u * (uint32_t)num_cus), dim3(256), 0, s, p, d);
    }
    hipLaunchKernelGGL(k_eq_reduce, dim3(grid), dim3(256), 0, s, p, d, n_particles);
    return;
  }
  if (debug_nn)
    hipLaunchKernelGGL(k_likelihood_exact<true>, dim3(grid), dim3(256), 0, s, p, d, n_particles, use_lists);
  else
    hipLaunchKernelGGL(k_likelihood_exact<false>, dim3(grid), dim3(256), 0, s, p, d, n_particles, use_lists);
}
