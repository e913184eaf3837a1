// pft_exact_nn.hip -- NearestPairPointCloudCoherence (PCL 1.8.0 tracking/impl/nearest_pair_point_cloud_coherence.hpp),
// the alternative to ApproxNearestPair... that /root/reference/src/auto_tracking.cpp leaves commented out at
// :237-238, :249: per transformed reference point the TRUE nearest neighbour in the cropped
// cloud (search_->nearestKSearch(q, 1, ...), squared distance in float), gated by d2 < max_distance^2, then the same
// point coherences as the approximate variant.
//
// Only neighbours inside the gate contribute, so the search structure is a uniform grid over the crop box with a cell
// of two octree leaves (2 cm): counting sort of the cropped points by cell (three small kernels per iteration), then
// per query shells of growing radius (in cells) around the query's cell are scanned, row segment by row segment,
// with rows farther than the best distance skipped, until the reach exceeds the best distance found (or the gate).  Nearest = smallest float distance, equal distances -> lowest index
// (upstream leaves ties to std::sort).  No octree is built in this mode.
#include "pft_device_utils.h"

#define EG_TILE 2048u

__device__ __forceinline__ int eg_cell1(float v, float mn, float inv_g, int dim) {
  int c = (int)floorf((v - mn) * inv_g);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// grid geometry from the crop box; the cell doubles until the grid fits the cell arrays
__global__ void k_eg_setup(PftParams prm, PftDev d) {
  PftHeader* h = d.hdr;
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
}

__global__ __launch_bounds__(256) void k_eg_zero(PftDev d) {
  const uint32_t nc = d.hdr->eg_ncells;
  for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < min(nc, (blockIdx.x + 1u) * 1024u); i += 256u) d.eg_cnt[i] = 0u;
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

// ---- A7 with the exact nearest neighbour ----
template <bool DEBUG_NN>
__global__ __launch_bounds__(256) void k_likelihood_exact(PftParams prm, PftDev d, uint32_t n_particles) {
  __shared__ float lut_h[256], lut_s[256];
  const PftHeader* h = d.hdr;
  if (d.p_active) n_particles = *d.p_active;
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    lut_h[i] = (float)i / 180.0f;
    lut_s[i] = (float)i / 255.0f;
  }
  __syncthreads();
  const uint32_t n_crop = h->n_crop;
  const float g = h->eg_g, inv_g = h->eg_inv_g;
  const float mnx = h->eg_min[0], mny = h->eg_min[1], mnz = h->eg_min[2];
  const int dx_ = h->eg_dim[0], dy_ = h->eg_dim[1], dz_ = h->eg_dim[2];
  const double maxd2 = prm.maxd2;
  const float gate_f = (float)maxd2 * 1.0001f;  // pruning bound; the gate itself is the double comparison below
  const int R = (int)ceilf(sqrtf((float)maxd2) * inv_g) + 1;
  const double wd = prm.dist_w, whsv = prm.hsv_w;
  const float hw = prm.h_w, sw = prm.s_w, vw = prm.v_w;
  const uint32_t M = prm.M, nchunk = prm.nchunk;
  const int lane = lane_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + wave_id(), tw = gridDim.x * nw;
  const uint32_t n_items = n_particles * nchunk;
  for (uint32_t item_v = gw; item_v < n_items; item_v += tw) {
    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_v);
    const uint32_t pi = item / nchunk, ch = item % nchunk;
    float T[12];
    load_matrix(d.mats, pi, T);
    double val = 0.0;
    const uint32_t jend = min(M, (ch + 1) * (uint32_t)PFT_REF_CHUNK);
    for (uint32_t j = ch * PFT_REF_CHUNK + lane; j < jend; j += WAVE) {
      const float4 r = d.ref_xyz[j];
      float qx, qy, qz;
      xform(T, r.x, r.y, r.z, qx, qy, qz);
      float best = INFINITY;
      uint32_t bi = 0xffffffffu;
      float4 bt = make_float4(0, 0, 0, 0);
      if (n_crop > 0) {
        // the query's own cell (not clamped: a query outside the box starts its rings outside)
        const int cqx = (int)floorf((qx - mnx) * inv_g), cqy = (int)floorf((qy - mny) * inv_g),
                  cqz = (int)floorf((qz - mnz) * inv_g);
        // Shells of growing Chebyshev radius rr around the query's cell.  The cells of one (z, y) row are contiguous
        // in the sorted order, so a row segment costs two loads of cell starts however many cells it spans: rows on the
        // shell's faces are scanned over their whole x range, inner rows only at their two end cells.  A row whose
        // cells are all farther than the best distance so far is skipped before anything is loaded (a point in a row
        // at offset o is at least (|o| - 1) cells away on that axis), and once shell rr is done everything closer than
        // rr * g has been seen.
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
              if (lyz + lx * lx > fminf(best, gate_f)) continue;
              const uint32_t row = (uint32_t)((cz * dy_ + cy) * dx_);
              // face rows: one segment [cqx-rr, cqx+rr]; inner rows: the two cells cqx-rr and cqx+rr
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
                for (uint32_t pos = s0; pos < s1; pos++) {
                  const float4 p = d.leaf_pts[pos];
                  const float ex = p.x - qx, ey = p.y - qy, ez = p.z - qz;
                  const float dd = ex * ex + (ey * ey + ez * ez);  // pointSquaredDist
                  if (dd <= best) {
                    const uint32_t idx = d.leaf_order[pos];
                    if (dd < best || idx < bi) {  // equal distances: the lowest index
                      best = dd;
                      bi = idx;
                      bt = p;
                    }
                  }
                }
              }
            }
          }
          const float reach = (float)rr * gg;  // everything closer than this has been seen
          if (best < reach * reach || reach * reach > gate_f) break;
        }
      }
      if (DEBUG_NN) {
        const size_t o = (size_t)pi * M + d.ref_perm[j];
        const bool in_gate = bi != 0xffffffffu && (double)best < maxd2;
        d.nn_idx[o] = in_gate ? (int32_t)bi : -1;  // neighbours outside the gate are not searched for
        d.nn_d2[o] = in_gate ? best : INFINITY;
      }
      if (bi != 0xffffffffu && (double)best < maxd2) {
        // DistanceCoherence x HSVColorCoherence: as in pft_likelihood.hip (A7a, A7b)
        const float ex = qx - bt.x, ey = qy - bt.y, ez = qz - bt.z;
        const float n2 = (ex * ex + ey * ey) + ez * ez;
        const double dist = (double)sqrtf(n2);
        const double A = 1.0 + dist * dist * wd;
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
          h_diff = hw * hd1 * hd1;
        else
          h_diff = hw * hd2 * hd2;
        const float s_diff = sw * (rh.y - ts) * (rh.y - ts);
        const float v_diff = vw * (rh.z - tv) * (rh.z - tv);
        const double Bq = 1.0 + whsv * (double)(h_diff + s_diff + v_diff);
        val += 1.0 / (A * Bq);
      }
    }
    val = wave_sum(val);
    if (lane == 0) d.partial[(size_t)pi * nchunk + ch] = val;
  }
}

void pftk_likelihood_exact(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool debug_nn,
                           int num_cus) {
  const uint32_t items = n_particles * p.nchunk;
  uint32_t grid = 8u * (uint32_t)num_cus;
  const uint32_t need = (items + 3u) / 4u;
  if (grid > need) grid = need ? need : 1u;
  if (debug_nn)
    hipLaunchKernelGGL(k_likelihood_exact<true>, dim3(grid), dim3(256), 0, s, p, d, n_particles);
  else
    hipLaunchKernelGGL(k_likelihood_exact<false>, dim3(grid), dim3(256), 0, s, p, d, n_particles);
}
