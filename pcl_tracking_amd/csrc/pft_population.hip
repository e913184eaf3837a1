// pft_population.hip -- the stages that need the whole particle population, one 1024-thread workgroup,
// per-particle values held in registers (K particles per thread), ~10 barriers in total:
//   A8  ParticleFilterTracker::normalizeWeight   (tracking/impl/particle_filter.hpp)
//   A10 ParticleFilterTracker::update            (weighted mean; double tree sum instead of a
//                                                 sequential float sum: DESIGN.md "numerics")
//   A9  genAliasTable, in prefix-sum form: PCL's Walker table has the H stack growing from the front and
//       the L stack from the back of one array, both popped highest-index-first.  With D_i the running
//       deficit (1-q) over the L list and E_k the running excess (q-1) over the H list, small l_i is
//       paired with the first h_k whose E_k >= D_(i-1); h_k drops below 1 at the first i with D_i > E_k
//       (q = 1 + E_k - D_i) and is then itself paired with h_(k+1).  The lists and prefix sums are built
//       here with one fused scan; the (a[k], q[k]) entry of a drawn k is evaluated on demand by the
//       resample kernel (pft_device_utils.h alias_q / alias_a_small).
#include "pft_device_utils.h"

#define STAMP(k) do { if (threadIdx.x == 0) d.hdr->ticks[16 + (k)] = wall_clock64(); } while (0)

struct PopSh {
  double d6[6][16];
  double dmin[16], dmax[16];
  uint32_t u[20];
  double da[20], db[20];
};

template <int K>
__device__ __forceinline__ void population_body(const PftParams& prm, const PftDev& d, uint32_t n, int from_partials, int do_norm,
                                int do_mean, int do_alias, PopSh& S) {
  pft_particle* P = d.part_all;
  const uint32_t tid = threadIdx.x;
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  constexpr uint32_t NT = PFT_POP_THREADS;

  STAMP(0);
  // ---- raw (or given) weights, strided ownership: i = tid + j*1024 ----
  float wr[K];
#pragma unroll
  for (int j = 0; j < K; j++) {
    const uint32_t i = tid + j * NT;
    wr[j] = 0.0f;
    if (i < n) {
      if (from_partials == 2) {  // raw weights already summed by k_finalize_raw
        wr[j] = d.raw_w[i];
      } else if (from_partials) {  // w = -(float) val, val = sum of the per-chunk likelihood partial sums
        double v = 0.0;
        for (uint32_t c = 0; c < prm.nchunk; c++) v += d.partial[(size_t)i * prm.nchunk + c];
        wr[j] = -(float)v;
      } else {
        wr[j] = P[i].weight;
      }
    }
    UNROLL_FENCE(j, 4);
  }

  STAMP(1);
  if (do_norm) {
    double wmin = DBL_MAX, wmax = -DBL_MAX;
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (tid + j * NT < n) {
        double x = (double)wr[j];
        if (wmin > x) wmin = x;
        if (x != 0.0 && wmax < x) wmax = x;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      wmin = fmin(wmin, __shfl_xor(wmin, o));
      wmax = fmax(wmax, __shfl_xor(wmax, o));
    }
    if (lane == 0) {
      S.dmin[w] = wmin;
      S.dmax[w] = wmax;
    }
    __syncthreads();
    wmin = lane < nw ? S.dmin[lane] : DBL_MAX;
    wmax = lane < nw ? S.dmax[lane] : -DBL_MAX;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      wmin = fmin(wmin, __shfl_xor(wmin, o));
      wmax = fmax(wmax, __shfl_xor(wmax, o));
    }
    if (tid == 0) d.hdr->fit_ratio = wmin;
    double sum = 0.0;
    if (wmax != wmin) {
#pragma unroll
      for (int j = 0; j < K; j++) {
        if (tid + j * NT < n) {
          if (wr[j] != 0.0f) wr[j] = (float)exp(1.0 - prm.alpha * ((double)wr[j] - wmin) / (wmax - wmin));
          sum += (double)wr[j];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < K; j++) {
        if (tid + j * NT < n) {
          wr[j] = 1.0f / (float)n;
          sum += (double)wr[j];
        }
      }
    }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) S.dmin[w] = sum;
    __syncthreads();
    sum = lane < nw ? S.dmin[lane] : 0.0;
    sum = wave_sum(sum);
    if (sum != 0.0) {
      const float fs = (float)sum;
#pragma unroll
      for (int j = 0; j < K; j++) wr[j] = wr[j] / fs;
    } else {
#pragma unroll
      for (int j = 0; j < K; j++) wr[j] = 1.0f / (float)n;
    }
  }
  if (do_norm || from_partials) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      const uint32_t i = tid + j * NT;
      if (i < n) P[i].weight = wr[j];
    }
  }

  STAMP(2);
  if (do_mean) {
    double a[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < K; j++) {
      const uint32_t i = tid + j * NT;
      if (i < n) {
        const float4* pp = reinterpret_cast<const float4*>(P + i);
        const float4 lo = pp[0], hi = pp[1];
        const double wgt = (double)wr[j];
        a[0] += (double)lo.x * wgt; a[1] += (double)lo.y * wgt; a[2] += (double)lo.z * wgt;
        a[3] += (double)hi.x * wgt; a[4] += (double)hi.y * wgt; a[5] += (double)hi.z * wgt;
      }
      UNROLL_FENCE(j, 4);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      a[k] = wave_sum(a[k]);
      if (lane == 0) S.d6[k][w] = a[k];
    }
    __syncthreads();
    if (w < 6) {
      double r = lane < nw ? S.d6[w][lane] : 0.0;
      r = wave_sum(r);
      if (lane == 0) S.d6[w][0] = r;
    }
    __syncthreads();
    if (tid == 0) {
      pft_particle orig = d.hdr->rep, r;
      r.x = (float)S.d6[0][0]; r.y = (float)S.d6[1][0]; r.z = (float)S.d6[2][0]; r.w = 1.0f;
      r.roll = (float)S.d6[3][0]; r.pitch = (float)S.d6[4][0]; r.yaw = (float)S.d6[5][0];
      r.weight = 1.0f / (float)n;
      pft_particle m;
      m.x = r.x - orig.x; m.y = r.y - orig.y; m.z = r.z - orig.z; m.w = 1.0f;
      m.roll = r.roll - orig.roll; m.pitch = r.pitch - orig.pitch; m.yaw = r.yaw - orig.yaw;
      m.weight = 0.0f;
      d.hdr->rep = r;
      d.hdr->motion = m;
    }
  }

  STAMP(3);
  if (do_alias) {
    __threadfence_block();
    __syncthreads();  // the normalised weights written above are re-read below with a different ownership
    int32_t* Llist = d.alias_list;
    int32_t* Hlist = d.alias_list + n;
    double* Dp = d.alias_pref;      // inclusive running deficit over the L list
    double* Ep = d.alias_pref + n;  // inclusive running excess over the H list
    // thread t owns reversed positions [t*Kp, (t+1)*Kp): both stacks pop the highest index first
    const uint32_t Kp = (n + NT - 1) / NT;
    const uint32_t r0 = tid * Kp;
    float wq[K];
    uint32_t cntL = 0;
    double defs = 0.0, excs = 0.0;
#pragma unroll
    for (int j = 0; j < K; j++) {
      const uint32_t r = r0 + j;
      wq[j] = 0.0f;
      if ((uint32_t)j < Kp && r < n) {
        wq[j] = P[n - 1 - r].weight;
        const double q = (double)(wq[j] * (float)n);  // float product widened to double
        if (q < 1.0) {
          cntL++;
          defs += 1.0 - q;
        } else {
          excs += q - 1.0;
        }
      }
      UNROLL_FENCE(j, 8);
    }
    STAMP(4);
    // fused exclusive scan of (cntL, defs, excs) over the workgroup
    uint32_t iu = cntL;
    double ia = defs, ib = excs;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      uint32_t nu = __shfl_up(iu, o);
      double na = __shfl_up(ia, o), nb = __shfl_up(ib, o);
      if (lane >= o) {
        iu += nu;
        ia += na;
        ib += nb;
      }
    }
    if (lane == WAVE - 1) {
      S.u[w] = iu;
      S.da[w] = ia;
      S.db[w] = ib;
    }
    __syncthreads();
    if (w == 0) {
      uint32_t tu = lane < nw ? S.u[lane] : 0u;
      double ta = lane < nw ? S.da[lane] : 0.0, tb = lane < nw ? S.db[lane] : 0.0;
      uint32_t su = tu;
      double sa = ta, sb = tb;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t nu = __shfl_up(su, o);
        double na = __shfl_up(sa, o), nb = __shfl_up(sb, o);
        if (lane >= o) {
          su += nu;
          sa += na;
          sb += nb;
        }
      }
      if (lane < nw) {
        S.u[lane] = su - tu;
        S.da[lane] = sa - ta;
        S.db[lane] = sb - tb;
      }
      if (lane == nw - 1) S.u[17] = su;
    }
    __syncthreads();
    uint32_t offL = S.u[w] + iu - cntL;
    double offD = S.da[w] + ia - defs, offE = S.db[w] + ib - excs;
    const uint32_t totL = S.u[17];
    const uint32_t rbeg = r0 < n ? r0 : n;
    uint32_t offH = rbeg - offL;  // larges before me = elements before me - smalls before me
#pragma unroll
    for (int j = 0; j < K; j++) {
      const uint32_t r = r0 + j;
      if ((uint32_t)j < Kp && r < n) {
        const uint32_t i = n - 1 - r;
        const double q = (double)(wq[j] * (float)n);
        if (q < 1.0) {
          offD += 1.0 - q;
          Llist[offL] = (int32_t)i;
          Dp[offL] = offD;
          d.alias_pos[i] = offL;
          offL++;
        } else {
          offE += q - 1.0;
          Hlist[offH] = (int32_t)i;
          Ep[offH] = offE;
          d.alias_pos[i] = offH | 0x80000000u;
          offH++;
        }
      }
    }
    STAMP(5);
    if (tid == 0) {
      d.hdr->alias_m = totL;
      d.hdr->alias_nh = n - totL;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Populations above 16 particles per thread (P > 16384; the replicated population of a multi-GPU run):
// the same stages over many workgroups, four launches, per-workgroup partials combined in workgroup order
// by every workgroup alike (deterministic, identical on every rank).  Workgroup g owns the REVERSED
// positions r in [g*4096, (g+1)*4096), particle i = n-1-r, so that the alias lists (both stacks pop the
// highest index first) come out as one exclusive scan over workgroups.
//   pop_part[g][0..11] = {min, max(!=0), sum, mean x6, cntL, deficit, excess}
enum { PM_MIN = 0, PM_MAX = 1, PM_SUM = 2, PM_MEAN = 3, PM_CNT = 9, PM_DEF = 10, PM_EXC = 11 };
constexpr uint32_t PM_WG = PFT_POPM_THREADS * PFT_POPM_ITEMS;

struct PopmSh {
  double s[3][20];
};

// phase A: raw weights (from the likelihood partial sums if asked) and the min / max partials
__global__ __launch_bounds__(PFT_POPM_THREADS) void k_popm_minmax(PftParams prm, PftDev d, uint32_t n, int from_partials) {
  __shared__ PopmSh S;
  pft_particle* P = d.part_all;
  const uint32_t g = blockIdx.x, tid = threadIdx.x;
  double wmin = DBL_MAX, wmax = -DBL_MAX;
  for (int j = 0; j < PFT_POPM_ITEMS; j++) {
    const uint32_t r = g * PM_WG + j * PFT_POPM_THREADS + tid;
    if (r >= n) break;
    const uint32_t i = n - 1 - r;
    float wf;
    if (from_partials == 2) {
      wf = d.raw_w[i];
      P[i].weight = wf;
    } else if (from_partials) {
      double v = 0.0;
      for (uint32_t c = 0; c < prm.nchunk; c++) v += d.partial[(size_t)i * prm.nchunk + c];
      wf = -(float)v;
      P[i].weight = wf;
    } else {
      wf = P[i].weight;
    }
    const double x = (double)wf;
    if (wmin > x) wmin = x;
    if (x != 0.0 && wmax < x) wmax = x;
  }
  wmin = block_reduce<double>(wmin, S.s[0], OpMinD(), DBL_MAX);
  wmax = block_reduce<double>(wmax, S.s[1], OpMaxD(), -DBL_MAX);
  if (tid == 0) {
    d.pop_part[g * 16 + PM_MIN] = wmin;
    d.pop_part[g * 16 + PM_MAX] = wmax;
  }
}

// phase B: w <- exp(1 - alpha (w - min)/(max - min)) (zeros kept), per-workgroup sums
__global__ __launch_bounds__(PFT_POPM_THREADS) void k_popm_transform(PftParams prm, PftDev d, uint32_t n) {
  __shared__ PopmSh S;
  pft_particle* P = d.part_all;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
  double wmin = tid < G ? d.pop_part[tid * 16 + PM_MIN] : DBL_MAX;
  double wmax = tid < G ? d.pop_part[tid * 16 + PM_MAX] : -DBL_MAX;
  wmin = block_reduce<double>(wmin, S.s[0], OpMinD(), DBL_MAX);
  wmax = block_reduce<double>(wmax, S.s[1], OpMaxD(), -DBL_MAX);
  if (g == 0 && tid == 0) d.hdr->fit_ratio = wmin;
  double sum = 0.0;
  for (int j = 0; j < PFT_POPM_ITEMS; j++) {
    const uint32_t r = g * PM_WG + j * PFT_POPM_THREADS + tid;
    if (r >= n) break;
    const uint32_t i = n - 1 - r;
    float wf = P[i].weight;
    if (wmax != wmin) {
      if (wf != 0.0f) wf = (float)exp(1.0 - prm.alpha * ((double)wf - wmin) / (wmax - wmin));
    } else {
      wf = 1.0f / (float)n;
    }
    P[i].weight = wf;
    sum += (double)wf;
  }
  sum = block_reduce<double>(sum, S.s[2], OpAddD(), 0.0);
  if (tid == 0) d.pop_part[g * 16 + PM_SUM] = sum;
}

// phase C: w <- w / (float) sum; weighted-mean partials; alias partition partials
__global__ __launch_bounds__(PFT_POPM_THREADS) void k_popm_normalize(PftParams prm, PftDev d, uint32_t n, int do_norm,
                                                                    int do_mean, int do_alias) {
  __shared__ PopmSh S;
  pft_particle* P = d.part_all;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
  double sum = 0.0;
  if (do_norm) {
    sum = tid < G ? d.pop_part[tid * 16 + PM_SUM] : 0.0;
    sum = block_reduce<double>(sum, S.s[0], OpAddD(), 0.0);
  }
  const float fs = (float)sum;
  double a[6] = {0, 0, 0, 0, 0, 0};
  double cnt = 0.0, defs = 0.0, excs = 0.0;
  for (int j = 0; j < PFT_POPM_ITEMS; j++) {
    const uint32_t r = g * PM_WG + j * PFT_POPM_THREADS + tid;
    if (r >= n) break;
    const uint32_t i = n - 1 - r;
    const float4* pp = reinterpret_cast<const float4*>(P + i);
    const float4 lo = pp[0], hi = pp[1];
    float wf = hi.w;
    if (do_norm) {
      wf = (sum != 0.0) ? wf / fs : 1.0f / (float)n;
      P[i].weight = wf;
    }
    const double wgt = (double)wf;
    a[0] += (double)lo.x * wgt; a[1] += (double)lo.y * wgt; a[2] += (double)lo.z * wgt;
    a[3] += (double)hi.x * wgt; a[4] += (double)hi.y * wgt; a[5] += (double)hi.z * wgt;
    const double q = (double)(wf * (float)n);
    if (q < 1.0) {
      cnt += 1.0;
      defs += 1.0 - q;
    } else {
      excs += q - 1.0;
    }
  }
  if (do_mean) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double r = block_reduce<double>(a[k], S.s[k % 3], OpAddD(), 0.0);
      if (tid == 0) d.pop_part[g * 16 + PM_MEAN + k] = r;
    }
  }
  if (do_alias) {
    cnt = block_reduce<double>(cnt, S.s[0], OpAddD(), 0.0);
    defs = block_reduce<double>(defs, S.s[1], OpAddD(), 0.0);
    excs = block_reduce<double>(excs, S.s[2], OpAddD(), 0.0);
    if (tid == 0) {
      d.pop_part[g * 16 + PM_CNT] = cnt;
      d.pop_part[g * 16 + PM_DEF] = defs;
      d.pop_part[g * 16 + PM_EXC] = excs;
    }
  }
}

// phase D: the alias lists with their running deficit / excess; workgroup 0 also finishes the mean
__global__ __launch_bounds__(PFT_POPM_THREADS) void k_popm_finish(PftParams prm, PftDev d, uint32_t n, int do_mean,
                                                                 int do_alias) {
  __shared__ PopmSh S;
  __shared__ uint32_t Su[20];
  pft_particle* P = d.part_all;
  const uint32_t g = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
  if (do_mean && g == 0) {
    double m[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double v = tid < G ? d.pop_part[tid * 16 + PM_MEAN + k] : 0.0;
      m[k] = block_reduce<double>(v, S.s[k % 3], OpAddD(), 0.0);
    }
    if (tid == 0) {
      pft_particle orig = d.hdr->rep, r;
      r.x = (float)m[0]; r.y = (float)m[1]; r.z = (float)m[2]; r.w = 1.0f;
      r.roll = (float)m[3]; r.pitch = (float)m[4]; r.yaw = (float)m[5];
      r.weight = 1.0f / (float)n;
      pft_particle mo;
      mo.x = r.x - orig.x; mo.y = r.y - orig.y; mo.z = r.z - orig.z; mo.w = 1.0f;
      mo.roll = r.roll - orig.roll; mo.pitch = r.pitch - orig.pitch; mo.yaw = r.yaw - orig.yaw;
      mo.weight = 0.0f;
      d.hdr->rep = r;
      d.hdr->motion = mo;
    }
  }
  if (!do_alias) return;
  // totals of the workgroups before mine (and of all, for the header)
  const double c_ = tid < G ? d.pop_part[tid * 16 + PM_CNT] : 0.0;
  const double d_ = tid < G ? d.pop_part[tid * 16 + PM_DEF] : 0.0;
  const double e_ = tid < G ? d.pop_part[tid * 16 + PM_EXC] : 0.0;
  const double baseL = block_reduce<double>(tid < g ? c_ : 0.0, S.s[0], OpAddD(), 0.0);
  const double baseD = block_reduce<double>(tid < g ? d_ : 0.0, S.s[1], OpAddD(), 0.0);
  const double baseE = block_reduce<double>(tid < g ? e_ : 0.0, S.s[2], OpAddD(), 0.0);
  if (g == 0) {
    const double totL = block_reduce<double>(c_, S.s[0], OpAddD(), 0.0);
    if (tid == 0) {
      d.hdr->alias_m = (uint32_t)totL;
      d.hdr->alias_nh = n - (uint32_t)totL;
    }
  }
  int32_t* Llist = d.alias_list;
  int32_t* Hlist = d.alias_list + n;
  double* Dp = d.alias_pref;
  double* Ep = d.alias_pref + n;
  // thread t owns the reversed positions [g*4096 + t*16, +16)
  const uint32_t r0 = g * PM_WG + tid * PFT_POPM_ITEMS;
  float wq[PFT_POPM_ITEMS];
  uint32_t cntL = 0;
  double defs = 0.0, excs = 0.0;
#pragma unroll
  for (int j = 0; j < PFT_POPM_ITEMS; j++) {
    const uint32_t r = r0 + j;
    wq[j] = 0.0f;
    if (r < n) {
      wq[j] = P[n - 1 - r].weight;
      const double q = (double)(wq[j] * (float)n);
      if (q < 1.0) {
        cntL++;
        defs += 1.0 - q;
      } else {
        excs += q - 1.0;
      }
    }
    UNROLL_FENCE(j, 8);
  }
  uint32_t tu;
  double ta, tb;
  uint32_t offL = (uint32_t)baseL + block_excl_scan<uint32_t>(cntL, Su, &tu);
  double offD = baseD + block_excl_scan<double>(defs, S.s[0], &ta);
  double offE = baseE + block_excl_scan<double>(excs, S.s[1], &tb);
  uint32_t offH = (r0 < n ? r0 : n) - offL;
#pragma unroll
  for (int j = 0; j < PFT_POPM_ITEMS; j++) {
    const uint32_t r = r0 + j;
    if (r < n) {
      const uint32_t i = n - 1 - r;
      const double q = (double)(wq[j] * (float)n);
      if (q < 1.0) {
        offD += 1.0 - q;
        Llist[offL] = (int32_t)i;
        Dp[offL] = offD;
        d.alias_pos[i] = offL;
        offL++;
      } else {
        offE += q - 1.0;
        Hlist[offH] = (int32_t)i;
        Ep[offH] = offE;
        d.alias_pos[i] = offH | 0x80000000u;
        offH++;
      }
    }
  }
}

__global__ __launch_bounds__(PFT_POP_THREADS) void k_population(PftParams prm, PftDev d, uint32_t n,
                                                               int from_partials, int do_norm, int do_mean,
                                                               int do_alias) {
  __shared__ PopSh S;
  if (d.p_active) n = *d.p_active;  // KLD variant: particle_num_ lives on the device
  const uint32_t per = (n + PFT_POP_THREADS - 1) / PFT_POP_THREADS;
  if (per <= 1) population_body<1>(prm, d, n, from_partials, do_norm, do_mean, do_alias, S);
  else if (per <= 2) population_body<2>(prm, d, n, from_partials, do_norm, do_mean, do_alias, S);
  else if (per <= 4) population_body<4>(prm, d, n, from_partials, do_norm, do_mean, do_alias, S);
  else if (per <= 8) population_body<8>(prm, d, n, from_partials, do_norm, do_mean, do_alias, S);
  else population_body<16>(prm, d, n, from_partials, do_norm, do_mean, do_alias, S);  // launcher: n <= 16384
}

// debug / test hook: the explicit (a, q) table of genAliasTable from the prefix-sum form
__global__ void k_alias_materialize(const pft_particle* __restrict__ P, AliasView v, const PftHeader* __restrict__ hdr,
                                    int32_t* __restrict__ a, double* __restrict__ q) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= v.n) return;
  v.m = hdr->alias_m;
  v.nh = hdr->alias_nh;
  int32_t a_large;
  const double qk = alias_q(v, k, P[k].weight, &a_large);
  q[k] = qk;
  a[k] = (v.pos[k] >> 31) ? a_large : alias_a_small(v, k);
}

void pftk_population(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n, int from_partials,
                     int do_normalize, int do_mean, int do_alias) {
  if (n < PFT_POPM_MIN) {
    hipLaunchKernelGGL(k_population, dim3(1), dim3(PFT_POP_THREADS), 0, s, p, d, n, from_partials, do_normalize,
                       do_mean, do_alias);
    return;
  }
  const uint32_t G = (n + PM_WG - 1) / PM_WG;  // <= PFT_POPM_MAX_WGS (pft_create caps particle_num)
  const dim3 grid(G), block(PFT_POPM_THREADS);
  if (do_normalize || from_partials) hipLaunchKernelGGL(k_popm_minmax, grid, block, 0, s, p, d, n, from_partials);
  if (do_normalize) hipLaunchKernelGGL(k_popm_transform, grid, block, 0, s, p, d, n);
  if (do_normalize || do_mean || do_alias)
    hipLaunchKernelGGL(k_popm_normalize, grid, block, 0, s, p, d, n, do_normalize, do_mean, do_alias);
  if (do_mean || do_alias) hipLaunchKernelGGL(k_popm_finish, grid, block, 0, s, p, d, n, do_mean, do_alias);
}

void pftk_alias_materialize(hipStream_t s, const PftDev& d, uint32_t n, int32_t* a, double* q) {
  AliasView v;
  v.L = d.alias_list;
  v.H = d.alias_list + n;
  v.D = d.alias_pref;
  v.E = d.alias_pref + n;
  v.pos = d.alias_pos;
  v.m = 0;
  v.nh = 0;
  v.n = n;
  hipLaunchKernelGGL(k_alias_materialize, dim3((n + 255) / 256), dim3(256), 0, s, d.part_all, v, d.hdr, a, q);
}
