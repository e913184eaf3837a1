// pft_likelihood.hip -- A6 + A7, the dominant kernel: per (particle, reference point)
//   transform (A2, recomputed in registers) -> greedy closest-child-centre descent
//   (OctreePointCloudSearch::approxNearestSearchRecursive, octree/impl/octree_search.hpp) -> leaf scan
//   -> gate d2 < max_distance^2 -> DistanceCoherence x HSVColorCoherence
//   (tracking/impl/approx_nearest_pair_point_cloud_coherence.hpp), summed per particle in double.
//
// One persistent 1024-thread workgroup per CU stages the linearised octree (node words, per-level centre
// tables, the level-J jump table, hue/saturation LUTs) into LDS once; each wave then walks work items
// (particle, chunk of 512 Morton-ordered reference points): the particle's 3x4 matrix is wave-uniform,
// reference points are read coalesced, the descent runs in registers against LDS, leaf records (16 B) are
// gathered from L2, and the per-lane double partial sums are combined with wave shuffles.
//
// Descent.  PCL picks, at every level, the EXISTING child whose voxel centre is closest to the query
// (float squared distances, ties to the lowest child index).  While the child that CONTAINS the query
// exists, that child is the strict minimum, so the step needs only the query's integer key bits:
//   * jump: the level-J ancestor (J <= 5) is looked up directly in a dense table;
//   * fast levels: follow the key bit while the containing child exists;
//   * generic levels: the exact float evaluation of all existing children, as PCL does it.
// The shortcut is taken only when the query is farther than `margin` from every face of the cells
// involved (float rounding of centres and sums can reorder children only inside that band: DESIGN.md
// "fast descent"); otherwise the level is evaluated generically.  Results are bit-identical to the
// all-generic descent (tests/test_gpu_parity.py compares index and distance of every pair).
// Algorithmic bytes per pair-eval: 16 (reference point) + 16 per candidate scanned in the reached leaf.
// (Round 3 experiments that are NOT in this file any more, because their scaffolding -- the loop bodies as lambdas, a ballot
// per trip -- cost the product kernel 2 %: the three per-lane loops with their sparse tails padded to 12 lanes,
// -DPFT_LIK_PAD; measured 178.8 ... 202 us against 176.8, profiles/r03_padding_variants_ab.txt; the code is in the
// repository history, commit "Phase stamps only in the diagnostic variant ...".)
#include "pft_device_utils.h"

#ifndef PFT_LIK_PENALTY
#define PFT_LIK_PENALTY 1
#endif

struct LikCtx {
  const uint32_t* words;   // LDS or HBM
  const float* tab;        // per-axis centre tables, stride per_axis
  uint32_t per_axis;
  const float* lut_h;      // [256] h/180
  const float* lut_s;      // [256] s/255
  const float* pen;        // [256][8] by child mask: 0 for an existing child, +inf for an absent one
  const uint16_t* jump;    // LDS, or null
  int J;
  uint32_t lvlJ_start;
  float margin, near_thr, ominx, ominy, ominz, inv_res, ncell;
  uint32_t leaf0;
  const uint16_t* leaf16;  // LDS: start offsets of the leaves (+ sentinel)
};

// node words when the tree outgrows LDS: the first n_lds words (the top levels, read by every descent) from LDS,
// the rest from HBM / L2
struct HybridWords {
  const __attribute__((address_space(3))) uint32_t* lds;
  const uint32_t* __restrict__ glob;
  uint32_t n_lds;
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return i < n_lds ? lds[i] : glob[i]; }
};

__device__ __forceinline__ double rcp_nr(double x) {
  // 1/x for x in [1, 2): hardware estimate + two Newton steps (full double precision up to ~1 ulp)
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}

// true if the query is within the margin of a face of its cell of 2^sh leaf cells per side
__device__ __forceinline__ bool face_violation(uint32_t kx, uint32_t ky, uint32_t kz, uint32_t lowf, uint32_t highf,
                                               int sh) {
  const uint32_t mx = (1u << sh) - 1u;
  const uint32_t rx = kx & mx, ry = ky & mx, rz = kz & mx;
  bool v = ((lowf & 1u) && rx == 0) || ((highf & 1u) && rx == mx);
  v |= ((lowf & 2u) && ry == 0) || ((highf & 2u) && ry == mx);
  v |= ((lowf & 4u) && rz == 0) || ((highf & 4u) && rz == mx);
  return v;
}

// number of trailing zeros (query near the low face) / trailing ones (near the high face) of a key coordinate: the largest
// cell size, as a power of two, whose face the query's leaf-cell face lies on; -1 when the query is near neither face
__device__ __forceinline__ int near_face_level(uint32_t k, bool low, bool high) {
  const uint32_t t = low ? k : ~k;
  const uint32_t c = min((uint32_t)(t == 0u ? -1 : __builtin_ctz(t)), 31u);  // one v_ffbl_b32 + v_min_u32
  return (low | high) ? (int)c : -1;
}

// LEAF: where the leaf level's start offsets come from -- 0: the node words themselves (W), 1: u16 array in LDS,
// 2: the u32 words in HBM / L2 (the branch levels alone are in LDS)
// INDIRECT (PftHeader::leaf_indirect; compile-time here -- as a run-time flag the branch in the leaf scan cost 4 us per launch
// at the headline size): the builder left the point records where the crop put them, a candidate is
// crop_pts[leaf_order[pos]] and bpos carries the record's index in crop_pts
template <bool USE_TAB, bool FAST, bool DEBUG_NN, int LEAF, bool INDIRECT, typename WordPtr>
__device__ __forceinline__ void likelihood_items(const PftParams& prm, const PftDev& d, const LikCtx& cx, WordPtr W,
                                                 uint32_t n_particles, int D, uint32_t n_crop,
                                                 const double omin[3], int abl) {
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  const uint32_t M = prm.M, nchunk = prm.nchunk;
  // Work distribution.  The cost of an item depends on where its queries land, and with a static round-robin the
  // slowest wave sets the launch time (mean wave busy 219 us, launch 250 us).  The workgroups form PFT_LIK_GROUPS
  // groups (blockIdx % groups: spread over the XCDs); a group owns a contiguous range of items; a wave's first item is
  // static, the following ones come from the group's counter (one atomic per item; a single counter for the whole
  // launch serialises at one L2 line and costs 150 us).
  const uint32_t G = min((uint32_t)PFT_LIK_GROUPS, gridDim.x), grp = blockIdx.x % G;
  const uint32_t gq = gridDim.x / G, gr = gridDim.x % G;  // groups below gr have gq + 1 workgroups, the others gq
  const uint32_t wgs_in_grp = gq + (grp < gr ? 1u : 0u), waves_in_grp = wgs_in_grp * (uint32_t)nw;
  const uint32_t wgs_before = grp * gq + min(grp, gr);
  // the group's share of the particles is proportional to its workgroups; inside the group the full-size chunks of all
  // its particles come first, then (split_last) the small chunks that end the reference cloud
  const uint32_t pb = (uint32_t)((unsigned long long)n_particles * wgs_before / gridDim.x);
  const uint32_t pe = (uint32_t)((unsigned long long)n_particles * (wgs_before + wgs_in_grp) / gridDim.x);
  const uint32_t np_grp = pe - pb, nbig = prm.split_last ? nchunk - prm.split_last : nchunk, nsmall = nchunk - nbig;
  const uint32_t small_len = prm.split_last ? prm.ref_chunk / prm.split_last : 0u;
  const uint32_t it_begin = 0u, it_end = np_grp * nchunk, big_items = np_grp * nbig;
  const uint32_t lw = (blockIdx.x / G) * (uint32_t)nw + (uint32_t)w;
  uint32_t* ctr = &d.hdr->lik_ctr[grp * PFT_LIK_CTR_STRIDE];
  const double res = prm.res;
  const double maxd2 = prm.maxd2;
  const double wd = prm.dist_w, whsv = prm.hsv_w;
  const float hw = prm.h_w, sw = prm.s_w, vw = prm.v_w;
  for (uint32_t item_v = it_begin + lw; item_v < it_end;) {
    // the work item is wave-uniform: said explicitly, so the particle's matrix is fetched with scalar loads and
    // lives in SGPRs
    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_v);
    uint32_t pi, ch, cstart, clen;
    if (item < big_items) {
      pi = pb + item / nbig;
      ch = item % nbig;
      cstart = ch * prm.ref_chunk;
      clen = prm.ref_chunk;
    } else {
      const uint32_t r = item - big_items;
      pi = pb + r / nsmall;
      ch = nbig + r % nsmall;
      cstart = nbig * prm.ref_chunk + (r % nsmall) * small_len;
      clen = small_len;
    }
    float T[12];
    load_matrix(d.mats, pi, T);
    double val = 0.0;
    unsigned long long st_q = 0, st_s = 0;
    const uint32_t jend = min(M, cstart + clen);
    const uint32_t j0 = cstart + lane;
    float4 rnext = j0 < jend ? d.ref_xyz[j0] : make_float4(0, 0, 0, 0);
    for (uint32_t j = j0; j < jend; j += WAVE) {
      const float4 r = rnext;
      if (j + WAVE < jend) rnext = d.ref_xyz[j + WAVE];  // next point's load overlaps this point's descent
      float qx, qy, qz;
      xform(T, r.x, r.y, r.z, qx, qy, qz);
      if (n_crop == 0) {  // empty target: PCL asserts; defined as "no correspondence"
        if (DEBUG_NN) {
          const size_t o = (size_t)pi * M + d.ref_perm[j];
          d.nn_idx[o] = -1;
          d.nn_d2[o] = INFINITY;
        }
        continue;
      }
      // heap-numbered path key of `node` per axis: j = 2^lvl + key (1 at the root); its children's centres are the
      // table entries 2j-2 and 2j-1, and j' = 2j + bit
      uint32_t node = 0, jx = 1, jy = 1, jz = 1;
      int lvl = 0;
      int dbg_fast = 0, dbg_gen = 0, dbg_jump = 0, dbg_hard = 0;
      if (FAST) {
        // integer key of the query in leaf cells (float: error < 1e-4 cells, far inside the margin)
        const float tx = (qx - cx.ominx) * cx.inv_res, ty = (qy - cx.ominy) * cx.inv_res,
                    tz = (qz - cx.ominz) * cx.inv_res;
        const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
        // inside the box on all three axes: smallest coordinate >= 0 and largest < 2^D (two three-operand instructions and
        // two compares instead of six compares)
        const bool inside = (int)(fminf(fminf(tx, ty), tz) >= 0.0f) & (int)(fmaxf(fmaxf(tx, ty), tz) < cx.ncell);
        const float fx = tx - flx, fy = ty - fly, fz = tz - flz;
        const float mg = cx.margin, mh = 1.0f - cx.margin;
        const uint32_t kx = inside ? (uint32_t)flx : 0u, ky = inside ? (uint32_t)fly : 0u,
                       kz = inside ? (uint32_t)flz : 0u;
        // The shortcut may not be taken at a level whose cell has a face within the margin of the query.  The cell
        // of 2^sh leaf cells has its low (high) face on an axis exactly at the query's leaf-cell face iff the low sh
        // key bits are all 0 (all 1): the levels concerned are sh <= V, V = max over the axes that are within
        // the margin of (trailing zeros | trailing ones) of the key.  Queries outside the box take no shortcut.
        // Only 0.4 % of the queries are near a face at all: three waves out of four have none, and skip the per-axis
        // evaluation on a (conservative, slightly wider) test of the largest distance from the cell centre.
        int V = -1;
        const float off = fmaxf(fmaxf(fabsf(fx - 0.5f), fabsf(fy - 0.5f)), fabsf(fz - 0.5f));
        if (__builtin_amdgcn_ballot_w64(off > cx.near_thr)) {
          // (branch-free: v_ffbl_b32 returns -1 for a zero operand, which the unsigned minimum turns into the 31 wanted
          // for key 0; keys never have all 32 bits set)
          const int vx = near_face_level(kx, fx < mg, fx > mh), vy = near_face_level(ky, fy < mg, fy > mh),
                    vz = near_face_level(kz, fz < mg, fz > mh);
          V = max(vx, max(vy, vz));
        }
        V = inside ? V : 31;
        const int lim = min(D, D - 1 - V);  // fast levels are those with lvl < lim
        if (cx.J > 0) {  // (wave-uniform) the jump lands on level J: its cell spans 2^(D-J) leaf cells, D-J > V
          const int sh = D - cx.J;
          const uint32_t e = cx.jump[(kx >> sh) | ((ky >> sh) << cx.J) | ((kz >> sh) << (2 * cx.J))];
          const bool take = (cx.J <= lim) & (e != 0u);  // all ancestors of an existing node exist and contain the query
          node = take ? cx.lvlJ_start + e - 1u : 0u;
          lvl = take ? cx.J : 0;
          if (DEBUG_NN) dbg_jump = take ? 1 : 0;
        }
        // fast levels: follow the key while the child containing the query exists (a divergent loop: a version
        // with a wave-uniform trip count and predicated steps was slower, 280 us against 267)
        while (lvl < lim) {
          const int sh = D - lvl - 1;
          const uint32_t c = (((kx >> sh) & 1u) << 2) | (((ky >> sh) & 1u) << 1) | ((kz >> sh) & 1u);
          const uint32_t wv = W[node];
          if (!((wv >> c) & 1u)) break;
          node = (wv >> 8) + __popc(wv & 0xffu & ((1u << c) - 1u));
          lvl++;
          dbg_fast++;
        }
        const int up = D - lvl;
        const uint32_t top = 1u << lvl;
        jx = (kx >> up) | top; jy = (ky >> up) | top; jz = (kz >> up) | top;
      }
      // ---- generic levels: exact float evaluation of the existing children ----
      if ((abl & 1) && lvl < D) {  // timing ablation only (PFT_ABLATE): skip the generic levels
        node = cx.leaf0;
        lvl = D;
      }
      for (; lvl < D; lvl++) {
        dbg_gen++;
        const uint32_t wv = W[node];
        const uint32_t mask = wv & 0xffu, base = wv >> 8;
        float cx0, cx1, cy0, cy1, cz0, cz1;
        if (USE_TAB) {
          const float2 tx2 = *reinterpret_cast<const float2*>(cx.tab - 2 + 2u * jx);
          const float2 ty2 = *reinterpret_cast<const float2*>(cx.tab + cx.per_axis - 2 + 2u * jy);
          const float2 tz2 = *reinterpret_cast<const float2*>(cx.tab + 2u * cx.per_axis - 2 + 2u * jz);
          cx0 = tx2.x; cx1 = tx2.y; cy0 = ty2.x; cy1 = ty2.y; cz0 = tz2.x; cz1 = tz2.y;
        } else {
          const double vs = res * (double)(1u << (D - lvl - 1));
          const uint32_t pkx = jx - (1u << lvl), pky = jy - (1u << lvl), pkz = jz - (1u << lvl);
          cx0 = (float)(((double)(2u * pkx) + 0.5) * vs + omin[0]);
          cx1 = (float)(((double)(2u * pkx + 1u) + 0.5) * vs + omin[0]);
          cy0 = (float)(((double)(2u * pky) + 0.5) * vs + omin[1]);
          cy1 = (float)(((double)(2u * pky + 1u) + 0.5) * vs + omin[1]);
          cz0 = (float)(((double)(2u * pkz) + 0.5) * vs + omin[2]);
          cz1 = (float)(((double)(2u * pkz + 1u) + 0.5) * vs + omin[2]);
        }
        // pointSquaredDist: Vector3f difference, squaredNorm = x2 + (y2 + z2)
        float dx0 = cx0 - qx, dx1 = cx1 - qx, dy0 = cy0 - qy, dy1 = cy1 - qy, dz0 = cz0 - qz, dz1 = cz1 - qz;
        float X0 = dx0 * dx0, X1 = dx1 * dx1, Y0 = dy0 * dy0, Y1 = dy1 * dy1, Z0 = dz0 * dz0, Z1 = dz1 * dz1;
        const float yz[4] = {Y0 + Z0, Y0 + Z1, Y1 + Z0, Y1 + Z1};
#if PFT_LIK_PENALTY
        // "if (dist >= min) continue" over the existing children in ascending order = the lowest index among the minima.
        // Absent children are priced out by ADDING +inf (d + 0.0f is d exactly): the eight penalties of the node's mask
        // come from LDS in two 16-byte reads, which replaces a bit test per child (and + compare: 54 cycles per level) by
        // an add (21).  Then the minimum (v_min3 tree) and the first index that attains it.
        const float4 pa = *reinterpret_cast<const float4*>(cx.pen + 8u * mask);
        const float4 pb = *reinterpret_cast<const float4*>(cx.pen + 8u * mask + 4u);
        const float d0 = (X0 + yz[0]) + pa.x, d1 = (X0 + yz[1]) + pa.y, d2 = (X0 + yz[2]) + pa.z, d3 = (X0 + yz[3]) + pa.w;
        const float d4 = (X1 + yz[0]) + pb.x, d5 = (X1 + yz[1]) + pb.y, d6 = (X1 + yz[2]) + pb.z, d7 = (X1 + yz[3]) + pb.w;
        float m0, m1, m2, best;
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(d0), "v"(d1), "v"(d2));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(d3), "v"(d4), "v"(d5));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(d6), "v"(d7), "v"(m0));
        asm("v_min_f32 %0, %1, %2" : "=v"(best) : "v"(m1), "v"(m2));
        uint32_t bc = 7u;
        bc = d6 == best ? 6u : bc;
        bc = d5 == best ? 5u : bc;
        bc = d4 == best ? 4u : bc;
        bc = d3 == best ? 3u : bc;
        bc = d2 == best ? 2u : bc;
        bc = d1 == best ? 1u : bc;
        bc = d0 == best ? 0u : bc;
#else
        float best = INFINITY;
        uint32_t bc = 0;
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
          float dc = ((c & 4u) ? X1 : X0) + yz[c & 3u];
          bool ex = (mask >> c) & 1u;
          if (ex && dc < best) {  // "if (dist >= min) continue": ties keep the lowest child index
            best = dc;
            bc = c;
          }
        }
#endif
        if (DEBUG_NN) {  // "hard" step: the per-axis nearest child (the ideal corner) does not exist
          const uint32_t ideal = ((X1 < X0) ? 4u : 0u) | ((Y1 < Y0) ? 2u : 0u) | ((Z1 < Z0) ? 1u : 0u);
          if (!((mask >> ideal) & 1u)) dbg_hard++;
        }
        node = base + __popc(mask & ((1u << bc) - 1u));
        // U10 (switch point): the path key follows the chosen (minimum) child, PCL's `minChildKey`;
        // the upstream variant that handed down `new_key` -- the last existing child iterated -- would take the bits of
        // 31 - clz(mask) here instead of bc (the CPU restatement under oracle/ carries the same note)
        jx = 2u * jx + ((bc >> 2) & 1u);
        jy = 2u * jy + ((bc >> 1) & 1u);
        jz = 2u * jz + (bc & 1u);
      }
      // ---- leaf scan: first strictly-smaller wins (insertion order) ----
      uint32_t ls, le;
      if (LEAF == 1) {  // leaf starts as u16 in LDS (cropped clouds below 65536 points)
        const uint32_t li = node - cx.leaf0;
        ls = cx.leaf16[li];
        le = cx.leaf16[li + 1u];
      } else if (LEAF == 2) {
        ls = d.words[node];
        le = d.words[node + 1];
      } else {
        ls = W[node];
        le = W[node + 1];
      }
      if (abl & 2) le = ls;
      float bd = (abl & 2) ? 1.0e-3f : INFINITY;
      uint32_t bpos = ls;
      if (!INDIRECT) {
        for (uint32_t pos = ls; pos < le; pos += 2) {  // two candidates per round: their gathers overlap
          const bool two = pos + 1 < le;
          const float4 c = d.leaf_pts[pos];
          const float4 c2 = d.leaf_pts[two ? pos + 1 : pos];
          float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
          float dd = dx * dx + (dy * dy + dz * dz);
          bool better = dd < bd;
          bd = better ? dd : bd;
          bpos = better ? pos : bpos;
          dx = c2.x - qx; dy = c2.y - qy; dz = c2.z - qz;
          dd = dx * dx + (dy * dy + dz * dz);
          better = two & (dd < bd);
          bd = better ? dd : bd;
          bpos = better ? pos + 1 : bpos;
        }
      } else {
        for (uint32_t pos = ls; pos < le; pos += 2) {
          const bool two = pos + 1 < le;
          const uint32_t i1 = d.leaf_order[pos], i2 = d.leaf_order[two ? pos + 1 : pos];
          const float4 c = d.crop_pts[i1];
          const float4 c2 = d.crop_pts[i2];
          float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
          float dd = dx * dx + (dy * dy + dz * dz);
          bool better = dd < bd;
          bd = better ? dd : bd;
          bpos = better ? i1 : bpos;
          dx = c2.x - qx; dy = c2.y - qy; dz = c2.z - qz;
          dd = dx * dx + (dy * dy + dz * dz);
          better = two & (dd < bd);
          bd = better ? dd : bd;
          bpos = better ? i2 : bpos;
        }
        // (no candidate beat +inf -- a NaN query --: bpos still holds the leaf position it started from; the direct path
        // reports the leaf's first record then, and so does this one)
        if (!(bd < INFINITY)) bpos = d.leaf_order[min(bpos, n_crop - 1u)];
      }
      // the winner's record (position + packed colour) is fetched again instead of being carried through the loop
      const float4 bt = INDIRECT ? d.crop_pts[bpos] : d.leaf_pts[bpos];
      if (DEBUG_NN) {
        const size_t o = (size_t)pi * M + d.ref_perm[j];
        d.nn_idx[o] = INDIRECT ? (int32_t)bpos : (int32_t)d.leaf_order[bpos];
        d.nn_d2[o] = bd;
        st_q += 1;
        st_s += le - ls;
        // distribution of the descent work: per query (counted per wave with ballots: one atomic per bucket and wave,
        // not per query) and per wave (max over lanes)
        {
          const int first = __ffsll((long long)__ballot(1)) - 1;
          const int gb = dbg_gen < 10 ? dbg_gen : 10, hb = dbg_hard < 4 ? dbg_hard : 4;
          for (int g_ = 0; g_ <= 10; g_++) {
            const unsigned long long m_ = __ballot(gb == g_);
            if (m_ && lane == first) atomicAdd(&d.hdr->dbg[g_], (unsigned long long)__popcll(m_));
          }
          for (int h_ = 0; h_ <= 4; h_++) {
            const unsigned long long m_ = __ballot(hb == h_);
            if (m_ && lane == first) atomicAdd(&d.hdr->dbg[27 + h_], (unsigned long long)__popcll(m_));
          }
          const unsigned long long mj_ = __ballot(dbg_jump != 0);
          if (mj_ && lane == first) atomicAdd(&d.hdr->dbg[11], (unsigned long long)__popcll(mj_));
        }
        int mg_ = dbg_gen, mf_ = dbg_fast, ml_ = (int)(le - ls);
        for (int o = 32; o > 0; o >>= 1) {
          mg_ = max(mg_, __shfl_xor(mg_, o));
          mf_ = max(mf_, __shfl_xor(mf_, o));
          ml_ = max(ml_, __shfl_xor(ml_, o));
        }
        if (lane == __ffsll((long long)__ballot(1)) - 1) {
          atomicAdd(&d.hdr->dbg[12], 1ull);
          atomicAdd(&d.hdr->dbg[13], (unsigned long long)mg_);
          atomicAdd(&d.hdr->dbg[14], (unsigned long long)mf_);
          atomicAdd(&d.hdr->dbg[15], (unsigned long long)ml_);
          atomicAdd(&d.hdr->dbg[16 + (mg_ < 10 ? mg_ : 10)], 1ull);
        }
      }
      // ---- A7: gate + point coherences ----
      if (abl & 4) {
        val += (double)bd;
      } else if ((double)bd < maxd2) {
        // DistanceCoherence: Vector4f norm (SSE3 packet reduction (dx2+dy2)+(dz2+0)); 1/(1 + d*d*w)
        float ex = qx - bt.x, ey = qy - bt.y, ez = qz - bt.z;
        float n2 = (ex * ex + ey * ey) + ez * ez;
        double dist = (double)sqrt_rn_coherence(n2);
        double A = 1.0 + dist * dist * wd;
        // HSVColorCoherence on precomputed (h,s,v): 1/(1 + w * diff2)
        const float4 rh = d.ref_hsv[j];
        const uint32_t pk = __float_as_uint(bt.w);
        const float th = cx.lut_h[pk & 0xffu], ts = cx.lut_s[(pk >> 8) & 0xffu], tv = cx.lut_s[(pk >> 16) & 0xffu];
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
        const float diff2 = h_diff + s_diff + v_diff;
        double Bq = 1.0 + whsv * (double)diff2;
        // (1/A) * (1/B) as one reciprocal of the product: equal to within 2 ulp(double); the per-particle
        // sum is cast to float afterwards (DESIGN.md "numerics")
        val += rcp_nr(A * Bq);
      }
    }
    val = wave_sum_lane63(val);  // (DPP: 18 instructions against 42 for the shuffle butterfly; the total lands in lane 63)
    if (lane == 63) d.partial[(size_t)pi * nchunk + ch] = val;
    {  // (drawing the next item at the START of this one, to hide the atomic's latency, was measured: 188 against 183 us --
       // the launch then ends with waves still holding an item they drew long ago)
      uint32_t nx = 0;
      if (lane == 0) nx = it_begin + waves_in_grp + atomicAdd(ctr, 1u);
      item_v = (uint32_t)__shfl((int)nx, 0);
    }
    if (DEBUG_NN) {
      st_q = wave_sum(st_q);
      st_s = wave_sum(st_s);
      if (lane == 0 && st_q) {
        atomicAdd(&d.hdr->stat_queries, st_q);
        atomicAdd(&d.hdr->stat_scanned, st_s);
      }
    }
  }
}

#ifndef PFT_LIK_REFILL
#define PFT_LIK_REFILL 0  // 1: the while-while work item of pft_likelihood_refill.h (measured experiment, not the product)
#endif
#if PFT_LIK_REFILL
#include "pft_likelihood_refill.h"
#endif

// INDIRECT is a parameter of the KERNEL (the host launches the instantiation that matches the builder mode it chose:
// pftk_octree returns it): both forms inside one kernel doubled its code and its scalar-register spills (76 -> 136) and
// cost the headline launch 2 %
#define PFT_LIK_RUN(UT, FA, LF) likelihood_items<UT, FA, DEBUG_NN, LF, INDIRECT>(prm, d, cx, W, n_particles, D, n_crop, omin, abl)

template <bool DEBUG_NN, bool INDIRECT>
__global__ __launch_bounds__(PFT_LIK_THREADS, DEBUG_NN ? 1 : (PFT_LIK_THREADS * PFT_LIK_WGS_PER_CU) / 256) void k_likelihood(PftParams prm, PftDev d, uint32_t n_particles,
                                                                uint32_t lds_bytes, int flags) {
  const int allow_fast = flags & 1;
#ifdef PFT_DIAG
  const int abl = flags >> 8;  // diagnostic build only (tools/build_variant.py diag -DPFT_DIAG): stage ablation for timing
#else
  constexpr int abl = 0;  // the product build has no way to skip a stage
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const PftHeader* hdr = d.hdr;
  if (d.p_active) n_particles = *d.p_active;  // KLD variant: particle_num_ lives on the device
  const int D = hdr->depth;
  // (the host launches the form the builder recorded; a mismatch would read the wrong array: no target instead)
  const uint32_t n_crop = (hdr->error || D <= 0 || (hdr->leaf_indirect != 0) != INDIRECT) ? 0u : hdr->n_crop;
  // A failed crop / octree build leaves this launch without a target (all weights 0): the flag is mirrored into pinned
  // host memory, and the next host synchronisation point returns it to the caller (pft_get_result & co.)
  if (hdr->error && blockIdx.x == 0 && threadIdx.x == 0 && d.host_stat) {
    d.host_stat[2] = hdr->error;
    d.host_stat[3] |= hdr->error;
  }
  const uint32_t n_words = hdr->n_words;
  const int use_tab = hdr->use_table;
  const double omin[3] = {hdr->omin[0], hdr->omin[1], hdr->omin[2]};
  const uint32_t per_axis = use_tab ? (2u << D) : 0u;
  const bool fast = allow_fast && use_tab && hdr->margin_cells < 0.5f;
  int J = fast ? hdr->jump_level : 0;

  // LDS carve: luts (2 KiB) | centre tables | jump table | node words
  float* pen = reinterpret_cast<float*>(smem);
  constexpr uint32_t pen_bytes = PFT_LIK_PENALTY ? 256u * 8u * 4u : 0u;
  float* lut_h = reinterpret_cast<float*>(smem + pen_bytes);
  float* lut_s = lut_h + 256;
  float* tab = lut_s + 256;
  uint32_t used = pen_bytes + 2048u + 3u * per_axis * 4u;
  used = (used + 15u) & ~15u;
  // node words: branch levels as u32; the leaf level (about 70 % of the words) as u16 start offsets when the
  // cropped cloud has fewer than 65536 points
  const uint32_t leaf_start = hdr->leaf_start, n_leaves = hdr->n_leaves;
  const bool leaf16 = n_crop > 0 && n_crop < 65536u;
  const uint32_t branch_bytes = leaf16 ? leaf_start * 4u : n_words * 4u;
  const uint32_t leaf_bytes = leaf16 ? ((n_leaves + 1u) * 2u + 3u) & ~3u : 0u;
  uint32_t jump_bytes = J > 0 ? (2u << (3 * J)) : 0u;
  // everything fits: jump table + all words.  A little too big: drop the jump table.  Far too big (large crops):
  // keep the jump table and hold only the top levels of the tree in LDS (HybridWords).
  const bool fits_with_jump = (size_t)used + jump_bytes + branch_bytes + leaf_bytes <= (size_t)lds_bytes;
  const bool fits_without_jump = (size_t)used + branch_bytes + leaf_bytes <= (size_t)lds_bytes;
  // (deep trees: 24 KiB of centre tables at depth 10) the branch levels and the jump table in LDS, the leaf starts
  // from L2: 228 us against 242 us without the jump table and 238 us with the words split at an arbitrary index
  // (only with centre tables: a tree deeper than PFT_TABLE_MAX_DEPTH has none and takes the top-levels-in-LDS layout below --
  // found by tools/fuzz_parity.py: this layout used to read the tables regardless, and a 12-level tree over a 41 m crop box
  // sent every query to one leaf)
  const bool branch_only = use_tab && !fits_with_jump && leaf16 && (size_t)used + jump_bytes + branch_bytes <= (size_t)lds_bytes;
  if (!fits_with_jump && fits_without_jump && !branch_only) {
    J = 0;
    jump_bytes = 0;
  }
  uint16_t* ljump = reinterpret_cast<uint16_t*>(smem + used);
  used += jump_bytes;
  uint32_t* lwords = reinterpret_cast<uint32_t*>(smem + used);
  uint16_t* lleaf = reinterpret_cast<uint16_t*>(smem + used + branch_bytes);
  const bool words_in_lds = !branch_only && (size_t)used + branch_bytes + leaf_bytes <= (size_t)lds_bytes;
  const uint32_t n_lds_words = (words_in_lds || branch_only) ? 0u : min(n_words, (lds_bytes - used) / 4u);

  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    lut_h[i] = (float)i / 180.0f;
    lut_s[i] = (float)i / 255.0f;
  }
  if (PFT_LIK_PENALTY)
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) pen[i] = ((i >> 3) >> (i & 7u)) & 1u ? 0.0f : INFINITY;
  // per-level per-axis voxel-centre tables, centre(level l, key k) = (float)((k + 0.5) * res * 2^(D-l) + min) exactly as
  // genVoxelCenterFromOctreeKey; entry 2^l - 2 + k.  Formed here, a few entries per thread, instead of by the one
  // workgroup of the builder (4.6 us there) and a global round trip.
  for (uint32_t e = threadIdx.x; e < 3u * per_axis; e += blockDim.x) {
    const uint32_t a = e / per_axis, r = e - a * per_axis;
    float c = 0.0f;
    if (r + 2u < per_axis) {
      const uint32_t l = 31u - (uint32_t)__clz((int)(r + 2u));
      const uint32_t k = r + 2u - (1u << l);
      const double vs = prm.res * (double)(1u << (D - (int)l));
      c = (float)(((double)k + 0.5) * vs + omin[a]);
    }
    tab[e] = c;
  }
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d.jump);
    uint32_t* dst = reinterpret_cast<uint32_t*>(ljump);
    for (uint32_t i = threadIdx.x; i < jump_bytes / 4u; i += blockDim.x) dst[i] = src[i];
  }
  if (words_in_lds || branch_only) {
    const uint32_t nb = leaf16 ? leaf_start : n_words;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lwords[i] = d.words[i];
    if (leaf16 && !branch_only)
      for (uint32_t i = threadIdx.x; i <= n_leaves; i += blockDim.x) lleaf[i] = (uint16_t)d.words[leaf_start + i];
  } else {
    for (uint32_t i = threadIdx.x; i < n_lds_words; i += blockDim.x) lwords[i] = d.words[i];
  }
  __syncthreads();

  LikCtx cx;
  cx.words = nullptr;
  cx.tab = tab;
  cx.per_axis = per_axis;
  cx.lut_h = lut_h;
  cx.lut_s = lut_s;
  cx.pen = pen;
  cx.jump = ljump;
  cx.J = J;
  cx.lvlJ_start = J > 0 ? hdr->lvl_start[J] : 0u;
  cx.margin = hdr->margin_cells;
  // |f - 0.5| above this: the query may be within `margin` of a face of its leaf cell (a little wider than the exact test)
  cx.near_thr = 0.5f - 1.01f * hdr->margin_cells - 1.0e-6f;
  cx.ominx = hdr->ominf[0];
  cx.ominy = hdr->ominf[1];
  cx.ominz = hdr->ominf[2];
  cx.inv_res = hdr->inv_res;
  cx.ncell = (float)(1u << (D > 0 ? D : 0));
  cx.leaf0 = hdr->leaf_start;
  cx.leaf16 = lleaf;
  if (branch_only) {
    const uint32_t* W = lwords;
    if (fast)
      PFT_LIK_RUN(true, true, 2);
    else
      PFT_LIK_RUN(true, false, 2);
  } else if (words_in_lds && leaf16) {  // node words addressed as LDS (ds_read), not through a generic pointer
    const uint32_t* W = lwords;
    if (fast)
#if PFT_LIK_REFILL
      likelihood_items_refill<DEBUG_NN>(prm, d, cx, W, n_particles, D, n_crop);
#else
      PFT_LIK_RUN(true, true, 1);
#endif
    else if (use_tab)
      PFT_LIK_RUN(true, false, 1);
    else
      PFT_LIK_RUN(false, false, 1);
  } else if (words_in_lds) {
    const uint32_t* W = lwords;
    if (fast)
      PFT_LIK_RUN(true, true, 0);
    else if (use_tab)
      PFT_LIK_RUN(true, false, 0);
    else
      PFT_LIK_RUN(false, false, 0);
  } else {
    HybridWords W;
    W.lds = (const __attribute__((address_space(3))) uint32_t*)lwords;
    W.glob = d.words;
    W.n_lds = n_lds_words;
    if (fast)
      PFT_LIK_RUN(true, true, 0);
    else if (use_tab)
      PFT_LIK_RUN(true, false, 0);
    else
      PFT_LIK_RUN(false, false, 0);
  }
}

static int g_allow_fast = -1;

// diagnostic: resident workgroups per CU the runtime reports for the production likelihood kernel
extern "C" int pft_debug_likelihood_occupancy(void) {
  int nb = -1;
  uint32_t lds = ((uint32_t)pftk_max_lds_bytes() / (uint32_t)PFT_LIK_WGS_PER_CU) & ~255u;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_likelihood<false, false>), PFT_LIK_THREADS, lds) != hipSuccess) return -1;
  return nb;
}

#ifdef PFT_DIAG
// timing experiments only (tools/lik_microbench.py with the diagnostic variant library): bit0 generic levels, bit1 leaf
// scan, bit2 coherence.  Not compiled into the product library.
extern "C" void pft_debug_set_ablate(int mask) {
  if (g_allow_fast < 0) g_allow_fast = 1;
  g_allow_fast = (g_allow_fast & 0xff) | (mask << 8);
}
#endif

void pftk_likelihood(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool debug_nn,
                     int num_cus, bool leaf_indirect) {
  static bool attr_set[PFT_MAX_DEVICES];
  // half of the CU's LDS per workgroup: two 1024-thread workgroups (32 waves, 8 per SIMD) are resident per CU
  uint32_t lds = ((uint32_t)pftk_max_lds_bytes() / (uint32_t)PFT_LIK_WGS_PER_CU) & ~255u;
  const int dev = pftk_cur_device();
  if (!attr_set[dev]) {
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<false, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<true, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<true, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess;
  }
  {  // PFT_GENERIC_DESCENT=1: all-generic descent (A/B and parity cross-check; read per launch: tools/fuzz_parity.py draws it per case)
    const char* e = getenv("PFT_GENERIC_DESCENT");
    if (g_allow_fast < 0) g_allow_fast = 1;
    g_allow_fast = (g_allow_fast & ~1) | ((e && e[0] == '1') ? 0 : 1);
#ifdef PFT_DIAG
    const char* a = getenv("PFT_ABLATE");  // timing experiments only: bit0 generic levels, bit1 leaf scan, bit2 coherence
    if (a) g_allow_fast |= atoi(a) << 8;
#endif
  }
  // (fewer workgroups at small particle counts -- less staging traffic -- was measured: 400 particles 52.2 us per frame with
  // the full grid, 54.9 / 56.2 / 84.2 with 384 / 256 / 128 workgroups)
  uint32_t items = n_particles * p.nchunk;
  uint32_t grid = (uint32_t)PFT_LIK_WGS_PER_CU * (uint32_t)num_cus;
  uint32_t need = (items + (PFT_LIK_THREADS / 64) - 1) / (PFT_LIK_THREADS / 64);
  if (need == 0) need = 1;
  if (grid > need) grid = need;
  if (debug_nn && leaf_indirect)
    hipLaunchKernelGGL((k_likelihood<true, true>), dim3(grid), dim3(PFT_LIK_THREADS), lds, s, p, d, n_particles, lds, g_allow_fast);
  else if (debug_nn)
    hipLaunchKernelGGL((k_likelihood<true, false>), dim3(grid), dim3(PFT_LIK_THREADS), lds, s, p, d, n_particles, lds, g_allow_fast);
  else if (leaf_indirect)
    hipLaunchKernelGGL((k_likelihood<false, true>), dim3(grid), dim3(PFT_LIK_THREADS), lds, s, p, d, n_particles, lds, g_allow_fast);
  else
    hipLaunchKernelGGL((k_likelihood<false, false>), dim3(grid), dim3(PFT_LIK_THREADS), lds, s, p, d, n_particles, lds, g_allow_fast);
}
