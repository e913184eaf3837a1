// pft_likelihood_refill.h -- EXPERIMENT (compile-time variant, -DPFT_LIK_REFILL=1; not in the product build):
// the while-while form of the likelihood work item that VERDICT r2 #1 asked for.
//
// The product kernel (pft_likelihood.hip, likelihood_items) walks a work item as rounds of 64 queries in lockstep: every
// phase of a round costs what its worst lane costs.  Here the 64 lanes are independent walkers inside the item:
//   * a lane that has nothing to do takes the item's next unprocessed reference point
//       j = next + mbcnt(ballot(idle))            (gather of 16 B from the L1-hot reference cloud, no atomics)
//     and runs the query set-up (transform, key, jump, fast levels) -- under the mask of the idle lanes;
//   * generic levels run for the lanes that are in descent;
//   * a lane whose descent is finished PARKS (q, leaf node, j); leaf scan + coherence run as their own phase when enough
//     lanes are parked, when a lane would have to park a second result, or when the item is drained.
// The per-lane double sums are reduced once per item as in the product kernel; which lane evaluates which point depends on
// the item's data only, so the item's value is still a function of the item alone (deterministic), but its summation
// order differs from the lockstep kernel's (last bits of the double sum).
//
// Thresholds (compile-time): PFT_REFILL_FILL idle lanes trigger a refill, PFT_REFILL_LEAF parked lanes a leaf phase.
// tools/proto/refill_sim.py costs this schedule with the phase costs measured in round 2: the set-up and the leaf /
// coherence phases, which run at full width in lockstep, run under partial masks here, and that costs more than the
// generic levels' idle lanes give back (predicted +13 ... +36 %).  Measured: DESIGN.md section 5.
//
// Supported layout: tables + fast descent + node words and u16 leaf starts in LDS (LEAF == 1), the layout of the bench
// workload; every other layout keeps the lockstep function.
#pragma once

#ifndef PFT_REFILL_FILL
#define PFT_REFILL_FILL 32
#endif
#ifndef PFT_REFILL_LEAF
#define PFT_REFILL_LEAF 32
#endif

template <bool DEBUG_NN>
__device__ __forceinline__ void likelihood_items_refill(const PftParams& prm, const PftDev& d, const LikCtx& cx,
                                                        const uint32_t* W, uint32_t n_particles, int D, uint32_t n_crop) {
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  const uint32_t M = prm.M, nchunk = prm.nchunk;
  // work distribution: exactly the product kernel's (groups of workgroups, static first item, then the group's counter)
  const uint32_t G = min((uint32_t)PFT_LIK_GROUPS, gridDim.x), grp = blockIdx.x % G;
  const uint32_t gq = gridDim.x / G, gr = gridDim.x % G;
  const uint32_t wgs_in_grp = gq + (grp < gr ? 1u : 0u), waves_in_grp = wgs_in_grp * (uint32_t)nw;
  const uint32_t wgs_before = grp * gq + min(grp, gr);
  const uint32_t pb = (uint32_t)((unsigned long long)n_particles * wgs_before / gridDim.x);
  const uint32_t pe = (uint32_t)((unsigned long long)n_particles * (wgs_before + wgs_in_grp) / gridDim.x);
  const uint32_t np_grp = pe - pb, nbig = prm.split_last ? nchunk - prm.split_last : nchunk, nsmall = nchunk - nbig;
  const uint32_t small_len = prm.split_last ? prm.ref_chunk / prm.split_last : 0u;
  const uint32_t it_end = np_grp * nchunk, big_items = np_grp * nbig;
  const uint32_t lw = (blockIdx.x / G) * (uint32_t)nw + (uint32_t)w;
  uint32_t* ctr = &d.hdr->lik_ctr[grp * PFT_LIK_CTR_STRIDE];
  const double maxd2 = prm.maxd2;
  const double wd = prm.dist_w, whsv = prm.hsv_w;
  const float hw = prm.h_w, sw = prm.s_w, vw = prm.v_w;
  for (uint32_t item_v = lw; item_v < it_end;) {
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
    const uint32_t jend = min(M, cstart + clen);
    uint32_t nxt = cstart;  // wave-uniform: the item's next unprocessed reference point
    // ---- lane state ----
    bool act = false;     // a query is in this lane (descending while lvl < D, finished and waiting to park at lvl == D)
    bool parked = false;  // a finished query waits for the leaf phase
    float qx = 0.f, qy = 0.f, qz = 0.f, pqx = 0.f, pqy = 0.f, pqz = 0.f;
    uint32_t node = 0, jx = 1, jy = 1, jz = 1, jref = 0, pnode = 0, pj = 0;
    int lvl = 0;
    if (n_crop == 0) {  // empty target: "no correspondence" for every pair
      if (DEBUG_NN)
        for (uint32_t j = cstart + lane; j < jend; j += WAVE) {
          const size_t o = (size_t)pi * M + d.ref_perm[j];
          d.nn_idx[o] = -1;
          d.nn_d2[o] = INFINITY;
        }
      nxt = jend;
    }
    for (;;) {
      // finished queries move to the parking slot when it is free
      if (act && lvl >= D && !parked) {
        pqx = qx; pqy = qy; pqz = qz; pnode = node; pj = jref;
        parked = true;
        act = false;
      }
      const unsigned long long m_idle = __ballot(!act && !parked);
      const unsigned long long m_desc = __ballot(act && lvl < D);
      const unsigned long long m_park = __ballot(parked);
      const unsigned long long m_wait = __ballot(act && lvl >= D);  // finished, parking slot still taken
      const uint32_t n_idle = (uint32_t)__popcll(m_idle), n_park = (uint32_t)__popcll(m_park);
      const uint32_t left = jend - nxt;
      if (left == 0u && !m_desc && !m_park && !m_wait) break;
      // ---- refill: idle lanes take the next points of the item and run the query set-up ----
      if (left > 0u && n_idle > 0u && !m_wait && (n_idle >= (uint32_t)PFT_REFILL_FILL || !m_desc)) {
        const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
        if (!act && !parked && r < left) {
          const uint32_t j = nxt + r;
          const float4 rp = d.ref_xyz[j];
          xform(T, rp.x, rp.y, rp.z, qx, qy, qz);
          jref = j;
          node = 0; lvl = 0;
          const float tx = (qx - cx.ominx) * cx.inv_res, ty = (qy - cx.ominy) * cx.inv_res, tz = (qz - cx.ominz) * cx.inv_res;
          const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
          const bool inside = (int)(fminf(fminf(tx, ty), tz) >= 0.0f) & (int)(fmaxf(fmaxf(tx, ty), tz) < cx.ncell);
          const float fx = tx - flx, fy = ty - fly, fz = tz - flz;
          const float mg = cx.margin, mh = 1.0f - cx.margin;
          const uint32_t kx = inside ? (uint32_t)flx : 0u, ky = inside ? (uint32_t)fly : 0u, kz = inside ? (uint32_t)flz : 0u;
          int V = -1;
          const float off = fmaxf(fmaxf(fabsf(fx - 0.5f), fabsf(fy - 0.5f)), fabsf(fz - 0.5f));
          if (off > cx.near_thr) {
            const int vx = near_face_level(kx, fx < mg, fx > mh), vy = near_face_level(ky, fy < mg, fy > mh),
                      vz = near_face_level(kz, fz < mg, fz > mh);
            V = max(vx, max(vy, vz));
          }
          V = inside ? V : 31;
          const int lim = min(D, D - 1 - V);
          if (cx.J > 0) {
            const int sh = D - cx.J;
            const uint32_t e = cx.jump[(kx >> sh) | ((ky >> sh) << cx.J) | ((kz >> sh) << (2 * cx.J))];
            const bool take = (cx.J <= lim) & (e != 0u);
            node = take ? cx.lvlJ_start + e - 1u : 0u;
            lvl = take ? cx.J : 0;
          }
          while (lvl < lim) {
            const int sh = D - lvl - 1;
            const uint32_t c = (((kx >> sh) & 1u) << 2) | (((ky >> sh) & 1u) << 1) | ((kz >> sh) & 1u);
            const uint32_t wv = W[node];
            if (!((wv >> c) & 1u)) break;
            node = (wv >> 8) + __popc(wv & 0xffu & ((1u << c) - 1u));
            lvl++;
          }
          const int up = D - lvl;
          const uint32_t top = 1u << lvl;
          jx = (kx >> up) | top; jy = (ky >> up) | top; jz = (kz >> up) | top;
          act = true;
        }
        nxt += min(n_idle, left);
        continue;
      }
      // ---- leaf scan + coherence of the parked queries ----
      if (n_park > 0u && (n_park >= (uint32_t)PFT_REFILL_LEAF || m_wait || !m_desc)) {
        if (parked) {
          const uint32_t li = pnode - cx.leaf0;
          const uint32_t ls = cx.leaf16[li], le = cx.leaf16[li + 1u];
          float bd = INFINITY;
          uint32_t bpos = ls;
          for (uint32_t pos = ls; pos < le; pos += 2) {
            const bool two = pos + 1 < le;
            const float4 c = d.leaf_pts[pos];
            const float4 c2 = d.leaf_pts[two ? pos + 1 : pos];
            float dx = c.x - pqx, dy = c.y - pqy, dz = c.z - pqz;
            float dd = dx * dx + (dy * dy + dz * dz);
            bool better = dd < bd;
            bd = better ? dd : bd;
            bpos = better ? pos : bpos;
            dx = c2.x - pqx; dy = c2.y - pqy; dz = c2.z - pqz;
            dd = dx * dx + (dy * dy + dz * dz);
            better = two & (dd < bd);
            bd = better ? dd : bd;
            bpos = better ? pos + 1 : bpos;
          }
          const float4 bt = d.leaf_pts[bpos];
          if (DEBUG_NN) {
            const size_t o = (size_t)pi * M + d.ref_perm[pj];
            d.nn_idx[o] = (int32_t)d.leaf_order[bpos];
            d.nn_d2[o] = bd;
            atomicAdd(&d.hdr->stat_queries, 1ull);
            atomicAdd(&d.hdr->stat_scanned, (unsigned long long)(le - ls));
          }
          if ((double)bd < maxd2) {
            float ex = pqx - bt.x, ey = pqy - bt.y, ez = pqz - bt.z;
            float n2 = (ex * ex + ey * ey) + ez * ez;
            double dist = (double)sqrt_rn_coherence(n2);
            double A = 1.0 + dist * dist * wd;
            const float4 rh = d.ref_hsv[pj];
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
            val += rcp_nr(A * Bq);
          }
          parked = false;
        }
        continue;
      }
      // ---- one generic level for the lanes in descent ----
      if (act && lvl < D) {
        const uint32_t wv = W[node];
        const uint32_t mask = wv & 0xffu, base = wv >> 8;
        const float2 tx2 = *reinterpret_cast<const float2*>(cx.tab - 2 + 2u * jx);
        const float2 ty2 = *reinterpret_cast<const float2*>(cx.tab + cx.per_axis - 2 + 2u * jy);
        const float2 tz2 = *reinterpret_cast<const float2*>(cx.tab + 2u * cx.per_axis - 2 + 2u * jz);
        const float dx0 = tx2.x - qx, dx1 = tx2.y - qx, dy0 = ty2.x - qy, dy1 = ty2.y - qy, dz0 = tz2.x - qz, dz1 = tz2.y - qz;
        const float X0 = dx0 * dx0, X1 = dx1 * dx1, Y0 = dy0 * dy0, Y1 = dy1 * dy1, Z0 = dz0 * dz0, Z1 = dz1 * dz1;
        const float yz[4] = {Y0 + Z0, Y0 + Z1, Y1 + Z0, Y1 + Z1};
        const float4 pa = *reinterpret_cast<const float4*>(cx.pen + 8u * mask);
        const float4 pb2 = *reinterpret_cast<const float4*>(cx.pen + 8u * mask + 4u);
        const float d0 = (X0 + yz[0]) + pa.x, d1 = (X0 + yz[1]) + pa.y, d2 = (X0 + yz[2]) + pa.z, d3 = (X0 + yz[3]) + pa.w;
        const float d4 = (X1 + yz[0]) + pb2.x, d5 = (X1 + yz[1]) + pb2.y, d6 = (X1 + yz[2]) + pb2.z, d7 = (X1 + yz[3]) + pb2.w;
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
        node = base + __popc(mask & ((1u << bc) - 1u));
        jx = 2u * jx + ((bc >> 2) & 1u);
        jy = 2u * jy + ((bc >> 1) & 1u);
        jz = 2u * jz + (bc & 1u);
        lvl++;
      }
    }
    val = wave_sum_lane63(val);
    if (lane == 63) d.partial[(size_t)pi * nchunk + ch] = val;
    {
      uint32_t nx = 0;
      if (lane == 0) nx = waves_in_grp + atomicAdd(ctr, 1u);
      item_v = (uint32_t)__shfl((int)nx, 0);
    }
  }
}
