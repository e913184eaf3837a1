// pft_octree.hip -- A5: search::Octree(res).setInputCloud(cropped) == OctreePointCloud::
// addPointsFromInputCloud (PCL 1.8.0 octree/impl/octree_pointcloud.hpp), rebuilt every iteration.
//
// One 1024-thread workgroup (the build is a chain of short dependent phases; per-point state lives in
// registers and the node words in LDS, so a phase costs a barrier, not an HBM round trip):
//   1. replay of the insertion-order-dependent bounding-box growth (adoptBoundingBoxToPoint): rounds of
//      "first point outside the current box" (parallel min-index search) + the serial growth steps
//   2. key of every point in the final key frame (insertion-time key + later root shifts)
//   3. top-down levels: atomicOr of child bits into the parent's mask, exclusive scan of popcounts ->
//      child_base; children of a node are contiguous and ordered by child index
//   4. leaves: counts -> starts (+ sentinel); points ranked by insertion index inside their leaf
//      (OctreeContainerPointIndices keeps push_back order, and the leaf scan keeps the first minimum)
//   (the per-level per-axis voxel-centre tables of genVoxelCenterFromOctreeKey are formed by the likelihood kernel)
// Output in HBM: words[], leaf_pts[], leaf_order[], jump[], header.
#include "pft_device_utils.h"

// Phase stamps (tools/phase_ticks.py) exist in the diagnostic variant only (-DPFT_DIAG): a wall_clock64() is an
// s_memrealtime + s_waitcnt, and the thirty of them in a build sat on this one-workgroup kernel's critical path.
#ifdef PFT_DIAG
#define STAMP(k) do { if (threadIdx.x == 0) d.hdr->ticks[k] = wall_clock64(); } while (0)
#define TICK_NOW() wall_clock64()
#else
#define STAMP(k) do { } while (0)
#define TICK_NOW() 0ull
#endif

struct BuildSh {
  double mn[3], mx[3];
  int depth, ngrow;
  uint32_t cur, err, carry;
  int jump;
  uint32_t u32s[40];
  uint32_t gidx[PFT_MAX_GROW], gshift[PFT_MAX_GROW], gold[PFT_MAX_GROW];
  double gmin[PFT_MAX_GROW + 1][3];
  uint32_t lvl[PFT_MAX_DEPTH + 3];
  // dense top levels (build_tree): per level <= J the occupancy bits in Morton order, their popcount prefix, node counts
  uint32_t dn_bits[PFT_JUMP_MAX_LEVEL + 1][128], dn_pref[PFT_JUMP_MAX_LEVEL + 1][128], dn_cnt[PFT_JUMP_MAX_LEVEL + 1];
};

// first point: box = p +- res/2, then getKeyBitSize() pads it to depth 1 (side 2*res - eps)
__device__ void box_init(BuildSh& S, float4 p0, double res) {
  const double epsd = (double)FLT_EPSILON;
  double lo[3] = {(double)p0.x - res / 2, (double)p0.y - res / 2, (double)p0.z - res / 2};
  double hi[3] = {(double)p0.x + res / 2, (double)p0.y + res / 2, (double)p0.z + res / 2};
  unsigned mk = 0;
  for (int a = 0; a < 3; a++) {
    unsigned k = (unsigned)((hi[a] - lo[a]) / res);
    mk = k > mk ? k : mk;
  }
  unsigned mv = mk > 2u ? mk : 2u;
  // getKeyBitSize: ceil(log2(max key) - eps), at least... mv is 2 for the one-point box (log(2)/log(2) == 1.0 exactly):
  // the two double logarithms are only evaluated in the general case
  double l2 = mv == 2u ? 1.0 : log((double)mv) / log(2.0);
  unsigned dep = (unsigned)ceil(l2 - (double)FLT_EPSILON);
  if (dep > 32u) dep = 32u;
  double side = (double)(1u << dep) * res - epsd;
  for (int a = 0; a < 3; a++) {
    double over = (side - (hi[a] - lo[a])) / 2.0;
    S.mn[a] = lo[a] - over;
    S.mx[a] = hi[a] + over;
    S.gmin[0][a] = S.mn[a];
  }
  S.depth = (int)dep;
}

// adoptBoundingBoxToPoint for one violating point: new root above the old one until the point fits;
// axes without an upper violation extend downwards
__device__ void box_grow(BuildSh& S, float4 p, uint32_t idx, double res) {
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
    S.gidx[g] = idx;
    S.gshift[g] = (ux ? 0u : 1u) | (uy ? 0u : 2u) | (uz ? 0u : 4u);
    S.gold[g] = (uint32_t)S.depth;
    if (!ux) S.mn[0] -= side;
    if (!uy) S.mn[1] -= side;
    if (!uz) S.mn[2] -= side;
    S.depth = S.depth + 1;
    side = (double)(1u << S.depth) * res - epsd;
    S.mx[0] = S.mn[0] + side;
    S.mx[1] = S.mn[1] + side;
    S.mx[2] = S.mn[2] + side;
    S.gmin[g + 1][0] = S.mn[0];
    S.gmin[g + 1][1] = S.mn[1];
    S.gmin[g + 1][2] = S.mn[2];
    S.ngrow = g + 1;
  }
}

__device__ __forceinline__ bool box_violates(float x, float y, float z, const double* mn, const double* mx) {
  return (x < mn[0]) || (y < mn[1]) || (z < mn[2]) || (x >= mx[0]) || (y >= mx[1]) || (z >= mx[2]);
}

// Replay of the growth sequence.  The box after the first few dozen points usually contains everything,
// so: (a) wave 0 alone handles the growth events among the first 1024 points (16 chunks of 64 held in registers, one
// ballot per event, no workgroup barrier); (b) the workgroup then looks for later violators with a per-thread AABB
// quick reject; each remaining event costs one min-index reduction.
#define PFT_REPLAY_HEAD 1024u
__device__ __forceinline__ void box_replay(BuildSh& S, const float4* __restrict__ pts, uint32_t n, double res, unsigned long long* hdr_ticks) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const uint32_t head = n < PFT_REPLAY_HEAD ? n : PFT_REPLAY_HEAD;
  // (b), first half: thread-local AABB of the thread's strided points after the head -- done by the waves that have
  // nothing to do while wave 0 replays the head
  float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const uint32_t wt = tid - WAVE, ws = nt - WAVE;  // the tail is strided over the waves other than wave 0
  auto local_aabb = [&]() {
    for (uint32_t i0 = PFT_REPLAY_HEAD + wt; i0 < n; i0 += 8u * ws) {  // eight loads in flight per round (latency-bound)
      float4 q[8];
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) q[k] = pts[min(i0 + k * ws, n - 1u)];  // (a repeated point changes no minimum)
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) {
        lmn[0] = fminf(lmn[0], q[k].x); lmx[0] = fmaxf(lmx[0], q[k].x);
        lmn[1] = fminf(lmn[1], q[k].y); lmx[1] = fmaxf(lmx[1], q[k].y);
        lmn[2] = fminf(lmn[2], q[k].z); lmx[2] = fmaxf(lmx[2], q[k].z);
      }
    }
  };
  if (tid >= WAVE) local_aabb();
#ifdef PFT_DIAG
  if (tid == WAVE) hdr_ticks[13] = wall_clock64();
#endif
  if (tid < WAVE) {
    float4 q[PFT_REPLAY_HEAD / WAVE];
#pragma unroll
    for (uint32_t c = 0; c < PFT_REPLAY_HEAD / WAVE; c++) q[c] = pts[min(c * WAVE + tid, n - 1u)];
    uint32_t cur = 1;
    bool stop = false;
    // the box lives in registers, identical in every lane: a growth event is replayed by all lanes on the broadcast
    // point (no LDS round trips on the serial path); lane 0 records the event for the key phase
    double mn[3] = {S.mn[0], S.mn[1], S.mn[2]}, mx[3] = {S.mx[0], S.mx[1], S.mx[2]};
    int depth = S.depth, ngrow = S.ngrow;
    uint32_t err = S.err;
    const double epsd = (double)FLT_EPSILON;
#pragma unroll
    for (uint32_t c = 0; c < PFT_REPLAY_HEAD / WAVE; c++) {
      const uint32_t gi = c * WAVE + tid;
      const float4 p = q[c];
      while (!stop) {
        const bool viol = gi >= cur && gi < head && box_violates(p.x, p.y, p.z, mn, mx);
        const unsigned long long bal = __ballot(viol);
        if (!bal) break;
        const int f = __ffsll((long long)bal) - 1;
        const float px = __shfl(p.x, f), py = __shfl(p.y, f), pz = __shfl(p.z, f);
        const uint32_t idx = c * WAVE + (uint32_t)f;
        for (;;) {  // adoptBoundingBoxToPoint, as box_grow()
          const bool lx = px < mn[0], ly = py < mn[1], lz = pz < mn[2];
          const bool ux = px >= mx[0], uy = py >= mx[1], uz = pz >= mx[2];
          if (!(lx || ly || lz || ux || uy || uz)) break;
          if (ngrow >= PFT_MAX_GROW || depth >= PFT_MAX_DEPTH) {
            err |= 2u;
            break;
          }
          double side = (double)(1u << depth) * res;
          if (tid == 0) {
            S.gidx[ngrow] = idx;
            S.gshift[ngrow] = (ux ? 0u : 1u) | (uy ? 0u : 2u) | (uz ? 0u : 4u);
            S.gold[ngrow] = (uint32_t)depth;
          }
          if (!ux) mn[0] -= side;
          if (!uy) mn[1] -= side;
          if (!uz) mn[2] -= side;
          depth++;
          side = (double)(1u << depth) * res - epsd;
          mx[0] = mn[0] + side;
          mx[1] = mn[1] + side;
          mx[2] = mn[2] + side;
          if (tid == 0) {
            S.gmin[ngrow + 1][0] = mn[0];
            S.gmin[ngrow + 1][1] = mn[1];
            S.gmin[ngrow + 1][2] = mn[2];
          }
          ngrow++;
        }
        cur = idx + 1u;
        if (err) stop = true;
      }
    }
    if (tid == 0) {
      for (int a = 0; a < 3; a++) {
        S.mn[a] = mn[a];
        S.mx[a] = mx[a];
      }
      S.depth = depth;
      S.ngrow = ngrow;
      S.err = err;
    }
    if (tid == 0) S.cur = head;
#ifdef PFT_DIAG
    if (tid == 0) hdr_ticks[14] = wall_clock64();
#endif
  }
  __syncthreads();
  if (S.err || n <= PFT_REPLAY_HEAD) return;
  // (b), second half
  for (;;) {
    const uint32_t cur = S.cur;
    const double mn[3] = {S.mn[0], S.mn[1], S.mn[2]};
    const double mx[3] = {S.mx[0], S.mx[1], S.mx[2]};
    uint32_t first = 0xffffffffu;
    if (tid >= WAVE && (box_violates(lmn[0], lmn[1], lmn[2], mn, mx) || box_violates(lmx[0], lmx[1], lmx[2], mn, mx))) {
      uint32_t i0 = PFT_REPLAY_HEAD + wt;
      if (i0 < cur) i0 += ((cur - i0 + ws - 1) / ws) * ws;
      for (uint32_t i = i0; i < n; i += ws) {
        const float4 p = pts[i];
        if (box_violates(p.x, p.y, p.z, mn, mx)) {
          first = i;
          break;
        }
      }
    }
    first = block_reduce<uint32_t>(first, S.u32s, OpMinU(), 0xffffffffu);
    if (first == 0xffffffffu) break;
    if (tid == 0) {
      box_grow(S, pts[first], first, res);
      S.cur = first + 1;
    }
    __syncthreads();
    if (S.err) break;
  }
}

// ---- per-point state: registers (packed 10-bit keys, D <= 10) or HBM arrays (21-bit keys) ----
template <int K>
struct RegStore {
  typedef uint32_t key_t;
  static constexpr int B = 10;
  static constexpr bool MORTON = true;  // 30-bit keys are kept bit-interleaved: a level's child index is one bit-field extract
  uint32_t key[K], node[K];
  __device__ RegStore(const PftDev&) {}
  template <class F>
  __device__ __forceinline__ void each(uint32_t n, F&& f) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      uint32_t i = threadIdx.x + j * PFT_BUILD_THREADS;
      if (i < n) f(i, key[j], node[j]);
      UNROLL_FENCE(j, 2);
    }
  }
};

// registers for the first K*1024 points, L2-resident arrays for the rest: crops a little above the register range
template <int K>
struct HybridStore {
  typedef uint32_t key_t;
  static constexpr int B = 10;
  static constexpr bool MORTON = true;
  uint32_t key[K], node[K];
  uint32_t* gkey;
  uint32_t* gnode;
  __device__ HybridStore(const PftDev& d) : gkey(reinterpret_cast<uint32_t*>(d.pt_key64)), gnode(d.pt_node) {}
  template <class F>
  __device__ __forceinline__ void each(uint32_t n, F&& f) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      uint32_t i = threadIdx.x + j * PFT_BUILD_THREADS;
      if (i < n) f(i, key[j], node[j]);
      UNROLL_FENCE(j, 2);
    }
    for (uint32_t i = K * PFT_BUILD_THREADS + threadIdx.x; i < n; i += PFT_BUILD_THREADS) f(i, gkey[i], gnode[i]);
  }
};

struct GlobStore {
  typedef unsigned long long key_t;
  static constexpr int B = 21;
  static constexpr bool MORTON = false;  // (x << 42 | y << 21 | z)
  unsigned long long* key;
  uint32_t* node;
  __device__ GlobStore(const PftDev& d) : key(d.pt_key64), node(d.pt_node) {}
  template <class F>
  __device__ __forceinline__ void each(uint32_t n, F&& f) {
#pragma unroll 4
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) f(i, key[i], node[i]);
  }
};

// child index (x << 2 | y << 1 | z) of a key at bit `bit` of its three coordinates
template <class KT, int B, bool MORTON>
__device__ __forceinline__ uint32_t key_child(KT k, int bit) {
  if (MORTON) return (uint32_t)(k >> (3 * bit)) & 7u;
  return (uint32_t)(((k >> (2 * B + bit)) & 1) << 2) | (uint32_t)(((k >> (B + bit)) & 1) << 1) |
         (uint32_t)((k >> bit) & 1);
}

// 10 bits -> every third bit (bit b -> 3b)
__device__ __forceinline__ uint32_t spread3_10(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

// LDSW: node words and the leaf scratch list live in LDS (typed pointers: ds_* instructions); otherwise in HBM.
// Returns false if the node words outgrow LDS (the caller then rebuilds in HBM mode).
template <class Store, bool LDSW, bool TMPLDS>
__device__ __forceinline__ bool build_tree(const PftParams& prm, const PftDev& d, BuildSh& S, uint32_t n, uint32_t* lds_words,
                           uint32_t lds_words_cap, uint32_t* lds_tmp, uint32_t* out_leaf_start,
                           uint32_t* out_n_leaves) {
  typedef typename Store::key_t key_t;
  constexpr int B = Store::B;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const float4* pts = d.crop_pts;
  const double res = prm.res, inv_res = 1.0 / prm.res;
  const int D = S.depth, ngrow = S.ngrow;
  Store st(d);
  // direct-index table of the level-J nodes for the likelihood kernel's fast descent; the levels above J are built from
  // the occupancy bits of the level-J cells (below), which the key phase records as it goes
  const int J = (D >= 4 && D <= PFT_TABLE_MAX_DEPTH) ? (D - 1 < PFT_JUMP_MAX_LEVEL ? D - 1 : PFT_JUMP_MAX_LEVEL) : 0;
  if (J > 0) {
    for (uint32_t i = tid; i < 128u; i += nt) S.dn_bits[J][i] = 0u;
    __syncthreads();
  }

  // ---- keys: genOctreeKeyforPoint at insertion time, shifted into the final key frame ----
  const uint32_t last_grow = ngrow > 0 ? S.gidx[ngrow - 1] : 0u;
  st.each(n, [&](uint32_t i, key_t& key, uint32_t& node) {
    float4 p = pts[i];
    int e = ngrow;  // growth epoch of the point: almost every point comes after the last growth event
    if (ngrow > 0 && i < last_grow) {
      e = 0;
      while (e < ngrow && S.gidx[e] <= i) e++;
    }
    // (unsigned)((p - min) / res) in double, as genOctreeKeyforPoint: the product with 1/res agrees with
    // the correctly rounded quotient to ~1e-13, so it is used unless it lands within 1e-6 of an integer
    const double tq[3] = {(double)p.x - S.gmin[e][0], (double)p.y - S.gmin[e][1], (double)p.z - S.gmin[e][2]};
    uint32_t kk[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double q0 = tq[a] * inv_res;
      const double fr = q0 - floor(q0);
      if (fr < 1.0e-6 || fr > 1.0 - 1.0e-6) q0 = tq[a] / res;
      kk[a] = (uint32_t)q0;
    }
    uint32_t kx = kk[0], ky = kk[1], kz = kk[2];
    for (int s = e; s < ngrow; s++) {
      uint32_t sh = S.gshift[s], od = S.gold[s];
      if (sh & 1u) kx += 1u << od;
      if (sh & 2u) ky += 1u << od;
      if (sh & 4u) kz += 1u << od;
    }
    if (d.pt_key) {  // (workgroup-uniform) test hook pft_debug_get_point_keys; null in pft_compute
      d.pt_key[3 * (size_t)i + 0] = kx;
      d.pt_key[3 * (size_t)i + 1] = ky;
      d.pt_key[3 * (size_t)i + 2] = kz;
    }
    if (Store::MORTON)
      key = (key_t)((spread3_10(kx) << 2) | (spread3_10(ky) << 1) | spread3_10(kz));
    else
      key = ((key_t)kx << (2 * B)) | ((key_t)ky << B) | (key_t)kz;
    node = 0;
  });

  STAMP(2);
  uint32_t* W = LDSW ? lds_words : d.words;
  if (tid == 0) S.jump = J;
  // ---- levels 0 .. J-1 without a pass over the points per level: near the root thousands of points share a node, so
  // the occupancy of the level-J cells (at most 8^4) is recorded once in a bit array in Morton order; the child mask
  // of a level-(l-1) node is then literally byte i of the level-l bit array, a node's index inside its level is the
  // rank of its bit, and its children start at the rank of its first child's bit ----
  int l0 = 0;
  if (J > 0) {
    uint32_t (*dn_bits)[128] = S.dn_bits;
    uint32_t (*dn_pref)[128] = S.dn_pref;
    uint32_t* dn_cnt = S.dn_cnt;
    st.each(n, [&](uint32_t, key_t& key, uint32_t& node) {  // Morton index of the point's level-J cell; its occupancy bit
      uint32_t mj = 0;
      if (Store::MORTON)
        mj = (uint32_t)(key >> (3 * (D - J)));  // the top J digits of the interleaved key
      else
        for (int l = 0; l < J; l++) mj = (mj << 3) | key_child<key_t, B, Store::MORTON>(key, D - 1 - l);
      node = mj;
      const uint32_t bt = 1u << (mj & 31u);
      if (!(dn_bits[J][mj >> 5] & bt)) atomicOr(&dn_bits[J][mj >> 5], bt);
    });
    __syncthreads();
    for (int l = J - 1; l >= 0; l--) {  // bit i of level l = "byte i of level l + 1 is not zero" (at most 512 bits: one ballot per wave)
      const uint32_t nbits = 1u << (3 * l);
      const bool pred = tid < nbits && ((dn_bits[l + 1][tid >> 2] >> (8u * (tid & 3u))) & 0xffu) != 0u;
      const unsigned long long m = __ballot(pred);
      if (lane_id() == 0 && (uint32_t)wave_id() * 64u < ((nbits + 63u) & ~63u)) {
        dn_bits[l][2 * wave_id()] = (uint32_t)m;
        dn_bits[l][2 * wave_id() + 1] = (uint32_t)(m >> 32);
      }
      __syncthreads();
    }
    if (wave_id() <= J) {  // one wave per level: exclusive popcount prefix over the level's (at most 128) words
      const int l = wave_id(), ln = lane_id();
      const uint32_t nwords = ((1u << (3 * l)) + 31u) >> 5;
      const uint32_t c0 = (uint32_t)ln < nwords ? __popc(dn_bits[l][ln]) : 0u;
      const uint32_t c1 = (uint32_t)ln + 64u < nwords ? __popc(dn_bits[l][ln + 64]) : 0u;
      const uint32_t i0 = wave_incl_scan(c0);
      const uint32_t t0 = (uint32_t)__shfl((int)i0, 63);
      const uint32_t i1 = wave_incl_scan(c1);
      dn_pref[l][ln] = i0 - c0;
      dn_pref[l][ln + 64] = t0 + i1 - c1;
      if (ln == 63) dn_cnt[l] = t0 + i1;
    }
    __syncthreads();
    if (tid == 0) {
      S.lvl[0] = 0;
      for (int l = 0; l <= J; l++) S.lvl[l + 1] = S.lvl[l] + dn_cnt[l];
    }
    __syncthreads();
    const uint32_t top_end = S.lvl[J + 1];
    if (LDSW && top_end + 2 > lds_words_cap) return false;  // uniform: not even the top levels fit (tiny LDS share)
    if (top_end + 2 > d.max_words) {
      if (tid == 0) S.err |= 1u;
    } else {
#define DN_RANK(l, i) (dn_pref[l][(i) >> 5] + __popc(dn_bits[l][(i) >> 5] & ((1u << ((i) & 31u)) - 1u)))
      for (uint32_t cand = tid; cand < 585u; cand += nt) {  // node words of the levels below J: 1 + 8 + 64 + 512 candidates
        int l = 0;
        uint32_t i = cand;
        while (l < J && i >= (1u << (3 * l))) {
          i -= 1u << (3 * l);
          l++;
        }
        if (l < J && ((dn_bits[l][i >> 5] >> (i & 31u)) & 1u)) {
          const uint32_t mask = (dn_bits[l + 1][i >> 2] >> (8u * (i & 3u))) & 0xffu;
          const uint32_t base = S.lvl[l + 1] + DN_RANK(l + 1, 8u * i);
          W[S.lvl[l] + DN_RANK(l, i)] = mask | (base << 8);
        }
      }
      for (uint32_t j = S.lvl[J] + tid; j < top_end; j += nt) W[j] = 0;
      // the likelihood kernel's direct-index table of the level-J nodes: cell (cx, cy, cz) -> node - first node + 1, 0 = empty
      for (uint32_t i = tid; i < (1u << (3 * J)); i += nt) {
        uint32_t cx = 0, cy = 0, cz = 0;
        for (int b = 0; b < J; b++) {  // digit b (from the leaf side) of the Morton index
          const uint32_t dg = (i >> (3 * b)) & 7u;
          cx |= (dg >> 2) << b;
          cy |= ((dg >> 1) & 1u) << b;
          cz |= (dg & 1u) << b;
        }
        const bool occ = (dn_bits[J][i >> 5] >> (i & 31u)) & 1u;
        d.jump[cx | (cy << J) | (cz << (2 * J))] = occ ? (uint16_t)(DN_RANK(J, i) + 1u) : (uint16_t)0;
      }
      const uint32_t lj = S.lvl[J];
      st.each(n, [&](uint32_t, key_t&, uint32_t& node) { node = lj + DN_RANK(J, node); });
#undef DN_RANK
    }
    __syncthreads();
    l0 = J;
  } else {
    if (tid == 0) {
      W[0] = 0;
      S.lvl[0] = 0;
      S.lvl[1] = 1;
    }
    __syncthreads();
  }

  // ---- levels ----
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t0_, t1_;
  for (int l = l0; l < D && !S.err; l++) {
    const int bit = D - 1 - l;
    t0_ = TICK_NOW();
    st.each(n, [&](uint32_t, key_t& key, uint32_t& node) {
      if (l > l0) {  // move to the level-l node chosen by the previous level's bits
        uint32_t w = W[node];
        uint32_t cp = key_child<key_t, B, Store::MORTON>(key, bit + 1);
        node = (w >> 8) + __popc(w & 0xffu & ((1u << cp) - 1u));
      }
      // near the root thousands of points share a word: test first, so only the first arrivals pay for the
      // (same-address, serialised) LDS atomic
      const uint32_t cb = 1u << key_child<key_t, B, Store::MORTON>(key, bit);
      if (!(W[node] & cb)) atomicOr(&W[node], cb);
    });
    __syncthreads();
    t1_ = TICK_NOW(); tA += t1_ - t0_; t0_ = t1_;
    // child_base: each thread owns a contiguous run of this level's nodes, one workgroup scan per level
    // (two barriers: the scan scratch alternates between levels); the next level's words are zeroed in the
    // same phase as the bases are written
    const uint32_t ls = S.lvl[l], le = S.lvl[l + 1], nl = le - ls;
    const uint32_t per = (nl + nt - 1) / nt;
    const uint32_t a0 = ls + min(nl, tid * per), a1 = ls + min(nl, (tid + 1) * per);
    uint32_t cnt = 0;
    for (uint32_t nd = a0; nd < a1; nd++) cnt += __popc(W[nd] & 0xffu);
    t1_ = TICK_NOW(); tB += t1_ - t0_; t0_ = t1_;
    uint32_t* scr = S.u32s + (l & 1) * 20;
    uint32_t inc = wave_incl_scan(cnt);
    if (lane_id() == WAVE - 1) scr[wave_id()] = inc;
    __syncthreads();
    if (wave_id() == 0) {
      const int nw = (int)(nt >> 6);
      uint32_t t = lane_id() < nw ? scr[lane_id()] : 0u;
      uint32_t ti = wave_incl_scan(t);
      if (lane_id() < nw) scr[lane_id()] = ti - t;
      if (lane_id() == nw - 1) scr[17] = ti;
    }
    __syncthreads();
    t1_ = TICK_NOW(); tC += t1_ - t0_; t0_ = t1_;
    uint32_t base = le + scr[wave_id()] + inc - cnt;
    const uint32_t nend = le + scr[17];
    const bool overflow = nend + 2 > d.max_words || nend >= (1u << 24);
    if (LDSW && !overflow && nend + 2 > lds_words_cap) return false;  // uniform: node words outgrew LDS
    if (overflow) {
      if (tid == 0) S.err |= 1u;
    } else {
      for (uint32_t nd = a0; nd < a1; nd++) {
        const uint32_t wv = W[nd];
        W[nd] = wv | (base << 8);
        base += __popc(wv & 0xffu);
      }
      for (uint32_t j = le + tid; j < nend; j += nt) W[j] = 0;
    }
    if (tid == 0) S.lvl[l + 2] = nend;
    __syncthreads();
    t1_ = TICK_NOW(); tD += t1_ - t0_;
    if (S.err) break;
  }
#ifdef PFT_DIAG
  if (tid == 0) { d.hdr->ticks[9] = tA; d.hdr->ticks[10] = tB; d.hdr->ticks[11] = tC; d.hdr->ticks[12] = tD; }
#endif
  __syncthreads();
  if (S.err || D <= 0) {
    *out_leaf_start = 0;
    *out_n_leaves = 0;
    return true;
  }

  // ---- leaves ----
  STAMP(3);
  const uint32_t leaf_start = S.lvl[D], n_leaves = S.lvl[D + 1] - leaf_start;
  st.each(n, [&](uint32_t, key_t& key, uint32_t& node) {
    uint32_t w = W[node];
    uint32_t c = key_child<key_t, B, Store::MORTON>(key, 0);
    uint32_t leaf = (w >> 8) + __popc(w & 0xffu & ((1u << c) - 1u));
    node = leaf;
    key = (key_t)atomicAdd(&W[leaf], 1u);  // arrival slot inside the leaf (the key is no longer needed)
  });
  __syncthreads();
  {
    const uint32_t per = (n_leaves + nt - 1) / nt;
    const uint32_t a0 = leaf_start + min(n_leaves, tid * per), a1 = leaf_start + min(n_leaves, (tid + 1) * per);
    uint32_t cnt = 0;
    for (uint32_t nd = a0; nd < a1; nd++) cnt += W[nd];
    uint32_t total;
    uint32_t start = block_excl_scan<uint32_t>(cnt, S.u32s, &total);
    for (uint32_t nd = a0; nd < a1; nd++) {
      const uint32_t c = W[nd];
      W[nd] = start;
      start += c;
    }
  }
  if (tid == 0) W[leaf_start + n_leaves] = n;  // sentinel: count(leaf j) = start[j+1] - start[j]
  __syncthreads();
  STAMP(4);
  uint32_t* TMP = TMPLDS ? lds_tmp : d.pt_tmp;
  st.each(n, [&](uint32_t i, key_t& key, uint32_t& node) { TMP[W[node] + (uint32_t)key] = i; });
  __syncthreads();
  STAMP(5);
  st.each(n, [&](uint32_t i, key_t&, uint32_t& node) {
    const uint32_t s = W[node], e = W[node + 1];
    uint32_t rank = 0;
    for (uint32_t k = s; k < e; k++) rank += TMP[k] < i ? 1u : 0u;
    d.leaf_order[s + rank] = i;  // (k_leaf_gather copies the point records: one CU's bandwidth is better spent elsewhere)
  });
  STAMP(6);
  const uint32_t n_words = leaf_start + n_leaves + 1;
  if (LDSW)
    for (uint32_t j = tid; j < n_words; j += nt) d.words[j] = W[j];
  *out_leaf_start = leaf_start;
  *out_n_leaves = n_leaves;
  return true;
}

template <class Store>
__device__ __forceinline__ void build_tree_any(const PftParams& prm, const PftDev& d, BuildSh& S, uint32_t n,
                                               uint32_t* lds_words, uint32_t cap, uint32_t* lds_tmp, uint32_t* ls,
                                               uint32_t* nl) {
  bool done = false;
  if (cap >= 64) {
    if (lds_tmp) done = build_tree<Store, true, true>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
    else done = build_tree<Store, true, false>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
  }
  __syncthreads();
  if (!done) build_tree<Store, false, false>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
}

template <int K>
__device__ __forceinline__ void build_hybrid(const PftParams& prm, const PftDev& d, BuildSh& S, uint32_t n, uint32_t* lds_words,
                             uint32_t cap, uint32_t* lds_tmp, uint32_t* ls, uint32_t* nl, int* path) {
  if (S.depth <= HybridStore<K>::B) {
    *path = 2;
    build_tree_any<HybridStore<K>>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
  } else {
    build_tree_any<GlobStore>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
  }
}

template <int K>
__device__ __forceinline__ void build_regs(const PftParams& prm, const PftDev& d, BuildSh& S, uint32_t n, uint32_t* lds_words,
                           uint32_t cap, uint32_t* lds_tmp, uint32_t* ls, uint32_t* nl, int* path) {
  if (S.depth <= RegStore<K>::B) {
    *path = 1;
    build_tree_any<RegStore<K>>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
  } else {
    build_tree_any<GlobStore>(prm, d, S, n, lds_words, cap, lds_tmp, ls, nl);
  }
}

// rescue != 0: launched behind the sorted builder (pft_octree_sorted.hip); returns at once unless that builder found
// its radix passes too few for the tree's depth (error bit 3), in which case the tree is built here instead
__global__ __launch_bounds__(PFT_BUILD_THREADS) void k_octree_build(PftParams prm, PftDev d, uint32_t lds_bytes,
                                                                    int copy_leaf_pts, int rescue) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ BuildSh S;
  PftHeader* hdr = d.hdr;
  if (rescue && !(hdr->error & 8u)) return;  // (uniform)
  const uint32_t tid = threadIdx.x;
  // (issued beside the header loads, not behind them: one memory latency less at the head of this one-workgroup kernel;
  // record 0 is valid memory whatever n_crop says)
  const float4 p_first = tid == 0 ? d.crop_pts[0] : make_float4(0, 0, 0, 0);
  const uint32_t n = hdr->n_crop;

  STAMP(0);
  if (tid == 0) {
    S.err = hdr->error & 4u;  // (bit 2: this iteration's one-pass crop gave up waiting for a predecessor)
    S.ngrow = 0;
    S.depth = 0;
    S.cur = 1;
    if (n > 0) box_init(S, p_first, prm.res);
  }
  __syncthreads();

  // LDS carve: [leaf scratch list: n words, if the node words (about 1.5 n) still fit beside it][node words]
  uint32_t* lds_u = reinterpret_cast<uint32_t*>(smem);
  const uint32_t lds_words_total = lds_bytes / 4u;
  uint32_t* lds_tmp = nullptr;
  uint32_t tmp_words = 0;
  if ((size_t)n * 5u / 2u + 64u <= lds_words_total) {
    lds_tmp = lds_u;
    tmp_words = (n + 3u) & ~3u;
  }
  uint32_t* lds_words = lds_u + tmp_words;
  const uint32_t cap = lds_words_total - tmp_words;

  uint32_t leaf_start = 0, n_leaves = 0;
  int path = 0;
  if (n > 0) {
    box_replay(S, d.crop_pts, n, prm.res, d.hdr->ticks);
    __syncthreads();
    STAMP(1);
  }
  if (n > 0 && !S.err) {
    if (n <= 4u * PFT_BUILD_THREADS) build_regs<4>(prm, d, S, n, lds_words, cap, lds_tmp, &leaf_start, &n_leaves, &path);
    else if (n <= 8u * PFT_BUILD_THREADS) build_regs<8>(prm, d, S, n, lds_words, cap, lds_tmp, &leaf_start, &n_leaves, &path);
    // 14 points per thread is the last register-resident size without scratch (126 VGPRs; 16 spills 12)
    else if (n <= 14u * PFT_BUILD_THREADS) build_regs<14>(prm, d, S, n, lds_words, cap, lds_tmp, &leaf_start, &n_leaves, &path);
    else if (n <= 18u * PFT_BUILD_THREADS) build_hybrid<8>(prm, d, S, n, lds_words, cap, lds_tmp, &leaf_start, &n_leaves, &path);
    else build_tree_any<GlobStore>(prm, d, S, n, lds_words, cap, lds_tmp, &leaf_start, &n_leaves);
  }
  __syncthreads();
  STAMP(7);
  const int D = S.depth;
  const bool ok = n > 0 && !S.err && D > 0;
  if (copy_leaf_pts == 1 && ok) {  // small crops: the leaf-ordered point records are copied here instead of by k_leaf_gather
    __threadfence_block();    // (a launch costs more than moving a few thousand records through one CU)
    for (uint32_t pos = tid; pos < n; pos += blockDim.x) d.leaf_pts[pos] = d.crop_pts[d.leaf_order[pos]];
  }

  // (the per-level voxel-centre tables are formed by the likelihood workgroups themselves, in LDS, from depth + box)
  const int use_table = (ok && D <= PFT_TABLE_MAX_DEPTH) ? 1 : 0;
  STAMP(8);
  // the header: spread over a few threads of different waves (one thread doing all of it -- two double divisions, a dozen
  // dependent LDS reads, ~50 stores -- was 4 us at the tail of a one-workgroup kernel)
  if (tid == 0) {
    hdr->error = S.err;
    hdr->depth = D;
    hdr->use_table = use_table;
    hdr->n_grow = S.ngrow;
    hdr->build_path = path;
    hdr->leaf_indirect = copy_leaf_pts == 2 ? 1 : 0;  // 2: nobody copies the records, the likelihood kernel follows leaf_order
    if (d.host_stat) d.host_stat[1] = (uint32_t)D;
    hdr->jump_level = (ok && use_table) ? S.jump : 0;
    hdr->n_leaves = ok ? n_leaves : 0;
    hdr->leaf_start = ok ? leaf_start : 0;
    hdr->n_words = ok ? leaf_start + n_leaves + 1 : 0;
  }
  if (tid == 64) {
    // safety margin of the fast descent (DESIGN.md "fast descent"): float rounding of the voxel centres
    // (<= ulp(max |coordinate|)) and of the squared-distance sums (<= 40 u s_top) can only reorder two
    // children when the query is this close to a cell face
    double maxabs = 0.0;
    for (int a = 0; a < 3; a++) maxabs = fmax(maxabs, fmax(fabs(S.mn[a]), fabs(S.mx[a])));
    const double eta = maxabs * 1.1920928955078125e-07;
    const double s_top = prm.res * (double)(1u << (D > 0 ? D - 1 : 0));
    const double E = 9.0 * eta + 40.0 * 5.9604644775390625e-08 * s_top;
    double mc = 2.0 * E / prm.res + 1.0e-3;
    hdr->margin_cells = (float)(mc < 0.25 ? mc : 1.0);  // >= 0.5: fast descent never taken
  }
  if (tid == 128) hdr->inv_res = (float)(1.0 / prm.res);
  if (tid >= 192 && tid < 195) {
    const int a = (int)tid - 192;
    hdr->ominf[a] = (float)S.mn[a];
    hdr->omin[a] = n > 0 ? S.mn[a] : 0.0;
    hdr->omax[a] = n > 0 ? S.mx[a] : 0.0;
  }
  if (tid >= 256 && tid < 256 + PFT_MAX_DEPTH + 3 && (int)tid - 256 <= D + 1) hdr->lvl_start[tid - 256] = S.lvl[tid - 256];
}

// leaf-ordered point records for the likelihood kernel's leaf scan: leaf_pts[pos] = crop_pts[leaf_order[pos]]
__global__ __launch_bounds__(256) void k_leaf_gather(PftDev d) {
  const PftHeader* hdr = d.hdr;
  const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
  if (hdr->error || hdr->depth <= 0 || pos >= hdr->n_crop) return;
  d.leaf_pts[pos] = d.crop_pts[d.leaf_order[pos]];
}

static void pftk_octree_set_attr() {
  static bool attr_set[PFT_MAX_DEVICES];
  const uint32_t lds = ((uint32_t)pftk_max_lds_bytes() - 10240u) & ~15u;
  const int dev = pftk_cur_device();
  if (!attr_set[dev])
    attr_set[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_octree_build),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
}

bool pftk_octree(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t expected_points, bool allow_indirect) {
  // static LDS of the kernel (BuildSh ~2.6 KB, the dense top-level arrays 5 KB): leave 10 KB out of the dynamic request
  const uint32_t lds = ((uint32_t)pftk_max_lds_bytes() - 10240u) & ~15u;
  pftk_octree_set_attr();
  // Who produces the leaf-ordered point records (any choice is correct for any size; the builder records it in the header):
  //   2  nobody: the likelihood kernel reads crop_pts[leaf_order[pos]] -- one more dependent 4-byte load per candidate
  //      (measured +0.32 ps per query: 5.4 us per launch at 8 192 particles x 2 048 points) against a launch of its own
  //      (4.6 us) or scattered stores by the one builder workgroup; the caller allows it when the launch is small
  //      (allow_indirect); PFT_LEAF_INDIRECT=0 keeps the copies, =1 forces the indirection (A/B timing, cross-check)
  //   1  the builder itself (crops of at most 5 000 points, by the previous iteration's size)
  //   0  k_leaf_gather, a launch of many workgroups
  const char* e = getenv("PFT_LEAF_INDIRECT");
  const bool indirect = e ? e[0] == '1' : allow_indirect;
  const int mode = indirect ? 2 : (expected_points <= 5000u ? 1 : 0);
  hipLaunchKernelGGL(k_octree_build, dim3(1), dim3(PFT_BUILD_THREADS), lds, s, p, d, lds, mode, 0);
  // at most PFT_SORTED_BUILD_MIN-ish points reach this builder in practice, but any crop (<= N) is legal
  if (mode == 0)
    hipLaunchKernelGGL(k_leaf_gather, dim3((d.N + 255u) / 256u ? (d.N + 255u) / 256u : 1u), dim3(256), 0, s, d);
  return indirect;
}

// behind the sorted builder: a no-op unless error bit 3 asks for the rebuild (then the whole tree, leaf records included)
void pftk_octree_rescue(hipStream_t s, const PftParams& p, const PftDev& d) {
  pftk_octree_set_attr();
  const uint32_t lds = ((uint32_t)pftk_max_lds_bytes() - 10240u) & ~15u;
  hipLaunchKernelGGL(k_octree_build, dim3(1), dim3(PFT_BUILD_THREADS), lds, s, p, d, lds, 1, 1);
}
