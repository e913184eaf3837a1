// pft_octree_sorted.hip -- A5 for large cropped clouds: the same linearised octree as pft_octree.hip, built by
// many workgroups.  (pft_octree.hip's single workgroup keeps everything in registers / LDS and wins below
// ~20k points; its cost grows linearly and it leaves 255 CUs idle.)
//
//   k_so_replay   1 workgroup   box-growth replay (inherently sequential), centre tables, header
//   k_so_keys     n/256         final-frame key of every point -> Morton code (level-1 digit on top), value = i
//   radix sort    stable LSD, 8-bit digits, 3 launches per pass: per-tile histogram, scan, scatter.  A tile is
//                 1024 elements owned by ONE wave: chunks of 64 are ranked with ballots (peers with the same
//                 digit), so equal codes keep their input order = PCL's insertion order inside a leaf
//   k_so_count    n/1024        per tile and level: number of nodes that start in the tile (prefix change)
//   k_so_scan     1 workgroup   per-level exclusive scan over tiles, level offsets, header
//   k_so_emit     n/1024        node words (mask by atomicOr from the children, child_base from the node's
//                               first element), leaf starts, leaf-ordered point records, jump table
// The sorted order IS the leaf order: leaf_pts[pos] = crop_pts[value[pos]].
#include "pft_device_utils.h"

#define SO_TILE 1024
#define SO_BINS 256

// ---- replay kernel: the sequential part (same routines as the single-workgroup builder) ----
struct ReplaySh {
  double mn[3], mx[3];
  int depth, ngrow;
  uint32_t cur, err;
  uint32_t u32s[40];
};

__device__ void so_box_grow(ReplaySh& S, PftHeader* hdr, float4 p, uint32_t idx, double res) {
  const double epsd = (double)FLT_EPSILON;
  for (;;) {
    bool lx = p.x < S.mn[0], ly = p.y < S.mn[1], lz = p.z < S.mn[2];
    bool ux = p.x >= S.mx[0], uy = p.y >= S.mx[1], uz = p.z >= S.mx[2];
    if (!(lx || ly || lz || ux || uy || uz)) break;
    int g = S.ngrow;
    if (g >= PFT_MAX_GROW || S.depth >= PFT_MAX_DEPTH) {
      S.err |= 2u;
      break;
    }
    double side = (double)(1u << S.depth) * res;
    hdr->grow_idx[g] = idx;
    hdr->grow_shift[g] = (ux ? 0u : 1u) | (uy ? 0u : 2u) | (uz ? 0u : 4u);
    hdr->grow_old_depth[g] = (uint32_t)S.depth;
    if (!ux) S.mn[0] -= side;
    if (!uy) S.mn[1] -= side;
    if (!uz) S.mn[2] -= side;
    S.depth = S.depth + 1;
    side = (double)(1u << S.depth) * res - epsd;
    S.mx[0] = S.mn[0] + side;
    S.mx[1] = S.mn[1] + side;
    S.mx[2] = S.mn[2] + side;
    hdr->grow_min[g + 1][0] = S.mn[0];
    hdr->grow_min[g + 1][1] = S.mn[1];
    hdr->grow_min[g + 1][2] = S.mn[2];
    S.ngrow = g + 1;
  }
}

__device__ __forceinline__ bool so_violates(float x, float y, float z, const double* mn, const double* mx) {
  return (x < mn[0]) || (y < mn[1]) || (z < mn[2]) || (x >= mx[0]) || (y >= mx[1]) || (z >= mx[2]);
}

// per-tile AABB of the cropped cloud: {min xyz, max xyz}
__global__ __launch_bounds__(256) void k_so_tilebox(PftDev d, float* __restrict__ tile_box) {
  __shared__ float sr[6][4];
  const uint32_t n = d.hdr->n_crop, t = blockIdx.x;
  if (t * SO_TILE >= n) return;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = t * SO_TILE + threadIdx.x; i < min(n, (t + 1) * SO_TILE); i += blockDim.x) {
    const float4 p = d.crop_pts[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
  for (int k = 0; k < 3; k++) {
    const float a = wave_min(mn[k]), b = wave_max(mx[k]);
    if (lane_id() == 0) {
      sr[k][wave_id()] = a;
      sr[3 + k][wave_id()] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = sr[k][0];
    for (int w = 1; w < 4; w++) v = k < 3 ? fminf(v, sr[k][w]) : fmaxf(v, sr[k][w]);
    tile_box[(size_t)t * 6 + k] = v;
  }
}

__global__ __launch_bounds__(1024) void k_so_replay(PftParams prm, PftDev d, const float* __restrict__ tile_box) {
  __shared__ ReplaySh S;
  PftHeader* hdr = d.hdr;
  const uint32_t n = hdr->n_crop, tid = threadIdx.x, nt = blockDim.x;
  const float4* pts = d.crop_pts;
  const double res = prm.res;
  if (tid == 0) {
    S.err = hdr->error & 4u;  // (bit 2: this iteration's one-pass crop gave up waiting for a predecessor)
    S.ngrow = 0;
    S.depth = 0;
    S.cur = 1;
    if (n > 0) {
      // first point: box = p +- res/2, then getKeyBitSize() pads it to depth 1 (side 2*res - eps)
      const float4 p0 = pts[0];
      const double epsd = (double)FLT_EPSILON;
      double lo[3] = {(double)p0.x - res / 2, (double)p0.y - res / 2, (double)p0.z - res / 2};
      double hi[3] = {(double)p0.x + res / 2, (double)p0.y + res / 2, (double)p0.z + res / 2};
      unsigned mk = 0;
      for (int a = 0; a < 3; a++) {
        unsigned k = (unsigned)((hi[a] - lo[a]) / res);
        mk = k > mk ? k : mk;
      }
      unsigned mv = mk > 2u ? mk : 2u;
      double l2 = log((double)mv) / log(2.0);
      unsigned dep = (unsigned)ceil(l2 - (double)FLT_EPSILON);
      if (dep > 32u) dep = 32u;
      double side = (double)(1u << dep) * res - epsd;
      for (int a = 0; a < 3; a++) {
        double over = (side - (hi[a] - lo[a])) / 2.0;
        S.mn[a] = lo[a] - over;
        S.mx[a] = hi[a] + over;
        hdr->grow_min[0][a] = S.mn[a];
      }
      S.depth = (int)dep;
    }
  }
  __syncthreads();
  if (n > 1) {
    // rounds over the per-tile AABBs (k_so_tilebox): first tile whose box sticks out of the current box, then the
    // first offending point inside that tile; a tile without one (its offenders were inserted before the box
    // grew) is skipped
    const uint32_t ntile = (n + SO_TILE - 1) / SO_TILE;
    for (;;) {
      const uint32_t cur = S.cur;
      if (cur >= n) break;
      const double mn[3] = {S.mn[0], S.mn[1], S.mn[2]};
      const double mx[3] = {S.mx[0], S.mx[1], S.mx[2]};
      uint32_t ft = 0xffffffffu;
      for (uint32_t t = cur / SO_TILE + tid; t < ntile; t += nt) {
        const float* b = tile_box + (size_t)t * 6;
        if (so_violates(b[0], b[1], b[2], mn, mx) || so_violates(b[3], b[4], b[5], mn, mx)) {
          ft = t;
          break;
        }
      }
      ft = block_reduce<uint32_t>(ft, S.u32s, OpMinU(), 0xffffffffu);
      if (ft == 0xffffffffu) break;
      uint32_t first = 0xffffffffu;
      {
        const uint32_t i = ft * SO_TILE + tid;  // SO_TILE == blockDim.x
        if (i >= cur && i < n) {
          const float4 p = pts[i];
          if (so_violates(p.x, p.y, p.z, mn, mx)) first = i;
        }
      }
      first = block_reduce<uint32_t>(first, S.u32s + 20, OpMinU(), 0xffffffffu);
      if (tid == 0) {
        if (first == 0xffffffffu) {
          S.cur = (ft + 1) * SO_TILE;
        } else {
          so_box_grow(S, hdr, pts[first], first, res);
          S.cur = first + 1;
        }
      }
      __syncthreads();
      if (S.err) break;
    }
  }
  __syncthreads();
  const int D = S.depth;
  const bool ok = n > 0 && !S.err && D > 0;
  const int use_table = (ok && D <= PFT_TABLE_MAX_DEPTH) ? 1 : 0;
  const int J = (D >= 4 && D <= PFT_TABLE_MAX_DEPTH) ? (D - 1 < PFT_JUMP_MAX_LEVEL ? D - 1 : PFT_JUMP_MAX_LEVEL) : 0;
  // (the per-level voxel-centre tables are formed by the likelihood workgroups themselves, in LDS, from depth + box)
  if (J > 0) {
    uint32_t* jz = reinterpret_cast<uint32_t*>(d.jump);
    for (uint32_t j = tid; j < (1u << (3 * J - 1)); j += nt) jz[j] = 0u;
  }
  if (tid == 0) {
    hdr->error = S.err;
    hdr->depth = D;
    hdr->use_table = use_table;
    hdr->n_grow = S.ngrow;
    hdr->build_path = 2;
    hdr->leaf_indirect = 0;  // this builder writes leaf_pts itself
    hdr->jump_level = (ok && use_table) ? J : 0;
    double maxabs = 0.0;
    for (int a = 0; a < 3; a++) maxabs = fmax(maxabs, fmax(fabs(S.mn[a]), fabs(S.mx[a])));
    const double eta = maxabs * 1.1920928955078125e-07;
    const double s_top = res * (double)(1u << (D > 0 ? D - 1 : 0));
    const double E = 9.0 * eta + 40.0 * 5.9604644775390625e-08 * s_top;
    double mc = 2.0 * E / res + 1.0e-3;
    hdr->margin_cells = (float)(mc < 0.25 ? mc : 1.0);
    for (int a = 0; a < 3; a++) {
      hdr->ominf[a] = (float)S.mn[a];
      hdr->omin[a] = n > 0 ? S.mn[a] : 0.0;
      hdr->omax[a] = n > 0 ? S.mx[a] : 0.0;
    }
    hdr->inv_res = (float)(1.0 / res);
    if (!ok) {
      hdr->n_words = 0;
      hdr->n_leaves = 0;
      hdr->leaf_start = 0;
    }
  }
}

// ---- keys: final-frame key -> Morton code with the level-1 digit (x<<2|y<<1|z) most significant ----
__global__ void k_so_keys(PftParams prm, PftDev d, SortBufs sb, uint32_t n_pad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  const PftHeader* hdr = d.hdr;
  const uint32_t n = hdr->n_crop;
  const int D = hdr->depth, ngrow = hdr->n_grow;
  if (i >= n || hdr->error || D <= 0) return;  // the sort kernels take their size from the header too
  const double res = prm.res, inv_res = 1.0 / prm.res;
  const float4 p = d.crop_pts[i];
  int e = 0;
  while (e < ngrow && hdr->grow_idx[e] <= i) e++;
  const double tq[3] = {(double)p.x - hdr->grow_min[e][0], (double)p.y - hdr->grow_min[e][1],
                        (double)p.z - hdr->grow_min[e][2]};
  uint32_t kk[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double q0 = tq[a] * inv_res;  // == (unsigned)(t / res) unless within 1e-6 of an integer: exact division then
    const double fr = q0 - floor(q0);
    if (fr < 1.0e-6 || fr > 1.0 - 1.0e-6) q0 = tq[a] / res;
    kk[a] = (uint32_t)q0;
  }
  for (int s = e; s < ngrow; s++) {
    const uint32_t sh = hdr->grow_shift[s], od = hdr->grow_old_depth[s];
    if (sh & 1u) kk[0] += 1u << od;
    if (sh & 2u) kk[1] += 1u << od;
    if (sh & 4u) kk[2] += 1u << od;
  }
  if (d.pt_key) {  // test hook pft_debug_get_point_keys; null in pft_compute
    d.pt_key[3 * (size_t)i + 0] = kk[0];
    d.pt_key[3 * (size_t)i + 1] = kk[1];
    d.pt_key[3 * (size_t)i + 2] = kk[2];
  }
  unsigned long long code = 0;
  for (int b = D - 1; b >= 0; b--)
    code = (code << 3) | (unsigned long long)((((kk[0] >> b) & 1u) << 2) | (((kk[1] >> b) & 1u) << 1) | ((kk[2] >> b) & 1u));
  sb.keys[0][i] = code;
  sb.vals[0][i] = i;
}

// ---- stable LSD radix sort, 8-bit digit; one wave owns a tile of 1024 elements ----
__global__ __launch_bounds__(64) void k_rs_hist(const unsigned long long* __restrict__ keys, uint32_t n_pad, int shift,
                                                uint32_t* __restrict__ hist, uint32_t ntiles, const PftHeader* hdr,
                                                int pass) {
  if (pass * 8 >= 3 * hdr->depth && pass > 0) return;  // no significant bits left
  __shared__ uint32_t h[SO_BINS];
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  n_pad = hdr->error ? 0u : hdr->n_crop;  // tiles past the cropped cloud contribute zero counts
  for (int b = lane; b < SO_BINS; b += 64) h[b] = 0;
  __syncthreads();
  const uint32_t base = t * SO_TILE;
#pragma unroll 4
  for (int c = 0; c < SO_TILE / 64; c++) {
    const uint32_t i = base + c * 64 + lane;
    if (i < n_pad) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 0xffu], 1u);
  }
  __syncthreads();
  for (int b = lane; b < SO_BINS; b += 64) hist[(size_t)b * ntiles + t] = h[b];
}

// one workgroup per digit value: exclusive scan of that bin's per-tile counts, bin total to hist[256*ntiles + b]
__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t* __restrict__ hist, uint32_t ntiles, const PftHeader* hdr,
                                                  int pass) {
  if (pass * 8 >= 3 * hdr->depth && pass > 0) return;
  __shared__ uint32_t scr[20];
  const uint32_t b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  uint32_t* row = hist + (size_t)b * ntiles;
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < ntiles; t0 += nt) {
    const uint32_t t = t0 + tid;
    const uint32_t v = t < ntiles ? row[t] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<uint32_t>(v, scr, &tot);
    if (t < ntiles) row[t] = carry + ex;
    carry += tot;
  }
  if (tid == 0) hist[(size_t)SO_BINS * ntiles + b] = carry;
}

__global__ __launch_bounds__(64) void k_rs_scatter(const unsigned long long* __restrict__ kin,
                                                   const uint32_t* __restrict__ vin, unsigned long long* __restrict__ kout,
                                                   uint32_t* __restrict__ vout, uint32_t n_pad, int shift,
                                                   const uint32_t* __restrict__ offs, uint32_t ntiles, const PftHeader* hdr,
                                                   int pass) {
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  const uint32_t base = t * SO_TILE;
  n_pad = hdr->error ? 0u : hdr->n_crop;
  if (base >= n_pad) return;
  if (pass * 8 >= 3 * hdr->depth && pass > 0) {  // identity pass: keep the ping-pong parity fixed
    for (int c = 0; c < SO_TILE / 64; c++) {
      const uint32_t i = base + c * 64 + lane;
      if (i < n_pad) {
        kout[i] = kin[i];
        vout[i] = vin[i];
      }
    }
    return;
  }
  __shared__ uint32_t cnt[SO_BINS];
  {  // bin bases = exclusive scan of the 256 bin totals (4 consecutive bins per lane), + this tile's offset in the bin
    const uint32_t* tot = offs + (size_t)SO_BINS * ntiles;
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
  for (int c = 0; c < SO_TILE / 64; c++) {
    const uint32_t i = base + c * 64 + lane;
    const bool valid = i < n_pad;
    const unsigned long long key = valid ? kin[i] : ~0ull;
    const uint32_t val = valid ? vin[i] : 0u;
    const uint32_t dig = (uint32_t)(key >> shift) & 0xffu;
    // peers = lanes of this chunk with the same digit (8 ballots); invalid lanes form their own class
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

// ---- levels from the sorted codes ----
// first level (1-based) at which element i starts a new node: 1 + number of leading 3-bit digits it shares
// with its predecessor; element 0 starts a node on every level
__device__ __forceinline__ int so_first_new_level(unsigned long long ci, unsigned long long cp, int D, bool first) {
  if (first) return 1;
  const unsigned long long x = ci ^ cp;
  if (x == 0) return D + 1;  // same leaf
  const int hb = 63 - __clzll((long long)x);  // highest differing bit
  const int dig = hb / 3;                     // digit index from the bottom (0 = level D)
  return D - dig;
}

__global__ __launch_bounds__(256) void k_so_count(PftDev d, SortBufs sb, int buf) {
  __shared__ uint32_t cnt[PFT_MAX_DEPTH + 2];
  const PftHeader* hdr = d.hdr;
  const uint32_t n = hdr->n_crop;
  const int D = hdr->depth;
  const uint32_t t = blockIdx.x;
  if (threadIdx.x < PFT_MAX_DEPTH + 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  if (!hdr->error && D > 0) {
    const unsigned long long* keys = sb.keys[buf];
    for (uint32_t i = t * SO_TILE + threadIdx.x; i < min(n, (t + 1) * SO_TILE); i += blockDim.x) {
      const int fl = so_first_new_level(keys[i], i ? keys[i - 1] : 0ull, D, i == 0);
      for (int l = fl; l <= D; l++) atomicAdd(&cnt[l], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < PFT_MAX_DEPTH + 2) sb.tile_cnt[(size_t)t * (PFT_MAX_DEPTH + 2) + threadIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(1024) void k_so_scan(PftDev d, SortBufs sb, uint32_t npass) {
  // one wave per level: exclusive scan over the tiles; then the level offsets
  __shared__ uint32_t tot[PFT_MAX_DEPTH + 2];
  PftHeader* hdr = d.hdr;
  const int D = hdr->depth;
  const int w = wave_id(), lane = lane_id(), nw = blockDim.x >> 6;
  const uint32_t n = hdr->n_crop;
  const uint32_t nt_used = (n + SO_TILE - 1) / SO_TILE;
  for (int l = 1 + w; l <= D; l += nw) {
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < nt_used; t0 += 64) {
      const uint32_t t = t0 + lane;
      uint32_t v = t < nt_used ? sb.tile_cnt[(size_t)t * (PFT_MAX_DEPTH + 2) + l] : 0u;
      uint32_t inc = wave_incl_scan(v);
      if (t < nt_used) sb.tile_cnt[(size_t)t * (PFT_MAX_DEPTH + 2) + l] = run + inc - v;
      run += __shfl(inc, 63);
    }
    if (lane == 0) tot[l] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0 && !hdr->error && D > 0 && n > 0) {
    uint32_t off = 1;
    hdr->lvl_start[0] = 0;
    for (int l = 1; l <= D; l++) {
      hdr->lvl_start[l] = off;
      off += tot[l];
    }
    hdr->lvl_start[D + 1] = off;
    hdr->leaf_start = hdr->lvl_start[D];
    hdr->n_leaves = tot[D];
    hdr->n_words = off + 1;  // + sentinel
    if (d.host_stat) d.host_stat[1] = (uint32_t)D;
    if (3u * (uint32_t)D > 8u * npass) {
      // the host sized the radix passes from the previous iteration's depth and this tree is deeper (the codes are not
      // fully sorted, the counts above mean nothing): not the caller's error -- bit 3 makes k_so_emit stand back and
      // the rescue launch behind it rebuild the tree
      hdr->error |= 8u;
      hdr->n_words = 0;
    } else if (off + 2 > d.max_words || off >= (1u << 24)) {
      hdr->error |= 1u;
      hdr->n_words = 0;
    }
  }
}

__global__ __launch_bounds__(256) void k_so_emit(PftDev d, SortBufs sb, int buf) {
  __shared__ uint32_t scr[20];
  const PftHeader* hdr = d.hdr;
  const uint32_t n = hdr->n_crop;
  const int D = hdr->depth;
  if (hdr->error || D <= 0 || n == 0) return;
  const int J = hdr->jump_level;
  const uint32_t t = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* keys = sb.keys[buf];
  const uint32_t* vals = sb.vals[buf];
  uint32_t* W = d.words;
  // 4 consecutive elements per thread
  unsigned long long code[4];
  int fl[4];
  uint32_t idx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    idx[k] = t * SO_TILE + tid * 4 + k;
    code[k] = idx[k] < n ? keys[idx[k]] : 0ull;
    const unsigned long long prev = idx[k] ? keys[(idx[k] < n ? idx[k] : n) - 1] : 0ull;
    fl[k] = idx[k] < n ? so_first_new_level(code[k], prev, D, idx[k] == 0) : D + 2;
  }
  // leaf-ordered records
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (idx[k] < n) {
      const uint32_t v = vals[idx[k]];
      d.leaf_order[idx[k]] = v;
      d.leaf_pts[idx[k]] = d.crop_pts[v];
    }
  // per level: rank of the nodes that start in this tile; index of the node CONTAINING each element
  uint32_t prev_idx[4] = {0, 0, 0, 0};  // index (inside its level) of the level-(l-1) node containing the element
  for (int l = 1; l <= D; l++) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) c += (fl[k] <= l) ? 1u : 0u;
    uint32_t total;
    uint32_t ex = block_excl_scan<uint32_t>(c, scr, &total);
    uint32_t run = sb.tile_cnt[(size_t)t * (PFT_MAX_DEPTH + 2) + l] + ex;  // nodes of level l before this element
    const uint32_t lvl_l = hdr->lvl_start[l], lvl_p = hdr->lvl_start[l - 1];
    const uint32_t lvl_n = l < D ? hdr->lvl_start[l + 1] : 0u;
    (void)lvl_n;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool starts = fl[k] <= l;
      if (starts) run++;
      const uint32_t my = run - 1u;  // level-l node containing element k (valid when idx < n)
      if (idx[k] < n && starts) {
        const uint32_t digit = (uint32_t)(code[k] >> (3 * (D - l))) & 7u;
        atomicOr(&W[lvl_p + prev_idx[k]], 1u << digit);  // the parent gains this child
        if (l < D) {
          // child_base = index of this node's first child = the level-(l+1) node that starts at the same element;
          // filled in at level l+1 below (the same element starts it)
        } else {
          W[lvl_l + my] = idx[k];  // leaf word = start offset into leaf_pts
        }
        if (l == J && J > 0) {
          const unsigned long long pre = code[k] >> (3 * (D - J));
          uint32_t cx = 0, cy = 0, cz = 0;
          for (int b = 0; b < J; b++) {
            const uint32_t dg = (uint32_t)(pre >> (3 * b)) & 7u;
            cx |= ((dg >> 2) & 1u) << b;
            cy |= ((dg >> 1) & 1u) << b;
            cz |= (dg & 1u) << b;
          }
          d.jump[cx | (cy << J) | (cz << (2 * J))] = (uint16_t)(my + 1u);
        }
        // this element starts the level-l node `my`: it is the first child of its parent iff the parent also
        // starts here (fl <= l-1) -- then the parent's child_base is this node
        if (fl[k] <= l - 1 || l == 1) {
          if (l == 1) {
            if (idx[k] == 0) atomicOr(&W[0], (lvl_l + my) << 8);
          } else {
            atomicOr(&W[lvl_p + prev_idx[k]], (lvl_l + my) << 8);
          }
        }
      }
      prev_idx[k] = my;
    }
  }
  if (t == 0 && tid == 0) W[hdr->lvl_start[D] + hdr->n_leaves] = n;  // sentinel
}

// ---- host side ----
static inline uint32_t so_cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void pftk_octree_sorted(hipStream_t s, const PftParams& p, const PftDev& d, const SortBufs& sb, uint32_t n_pad,
                        int npass) {
  if (n_pad == 0) n_pad = 1;
  const uint32_t ntiles = so_cdiv(n_pad, SO_TILE);
  hipMemsetAsync(d.words, 0, (size_t)d.max_words * sizeof(uint32_t), s);
  hipLaunchKernelGGL(k_so_tilebox, dim3(ntiles), dim3(256), 0, s, d, sb.tile_box);
  hipLaunchKernelGGL(k_so_replay, dim3(1), dim3(SO_TILE), 0, s, p, d, sb.tile_box);
  hipLaunchKernelGGL(k_so_keys, dim3(so_cdiv(n_pad, 256)), dim3(256), 0, s, p, d, sb, n_pad);
  int buf = 0;
  for (int pass = 0; pass < npass; pass++) {
    const int shift = pass * 8;
    hipLaunchKernelGGL(k_rs_hist, dim3(ntiles), dim3(64), 0, s, sb.keys[buf], n_pad, shift, sb.hist, ntiles, d.hdr, pass);
    hipLaunchKernelGGL(k_rs_scan, dim3(SO_BINS), dim3(ntiles >= 512 ? 1024 : 256), 0, s, sb.hist, ntiles, d.hdr, pass);
    hipLaunchKernelGGL(k_rs_scatter, dim3(ntiles), dim3(64), 0, s, sb.keys[buf], sb.vals[buf], sb.keys[1 - buf],
                       sb.vals[1 - buf], n_pad, shift, sb.hist, ntiles, d.hdr, pass);
    buf = 1 - buf;
  }
  hipLaunchKernelGGL(k_so_count, dim3(ntiles), dim3(256), 0, s, d, sb, buf);
  hipLaunchKernelGGL(k_so_scan, dim3(1), dim3(1024), 0, s, d, sb, (uint32_t)npass);
  hipLaunchKernelGGL(k_so_emit, dim3(ntiles), dim3(256), 0, s, d, sb, buf);
  if (npass < 8) pftk_octree_rescue(s, p, d);  // 8 passes cover PFT_MAX_DEPTH: nothing to rescue then
}
