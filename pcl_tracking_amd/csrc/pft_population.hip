// pft_population.hip -- the stages that need the whole particle population, ONE launch of a few co-resident
// workgroups that meet at three device-scope barriers:
//   A8  ParticleFilterTracker::normalizeWeight   (tracking/impl/particle_filter.hpp)
//   A10 ParticleFilterTracker::update            (weighted mean; DESIGN.md "numerics")
//   A9  genAliasTable, in prefix-sum form: PCL's Walker table has the H stack growing from the front and
//       the L stack from the back of one array, both popped highest-index-first.  With D_i the running
//       deficit (1-q) over the L list and E_k the running excess (q-1) over the H list, small l_i is
//       paired with the first h_k whose E_k >= D_(i-1); h_k drops below 1 at the first i with D_i > E_k
//       (q = 1 + E_k - D_i) and is then itself paired with h_(k+1).  The lists and prefix sums are built
//       here with suffix scans; the (a[k], q[k]) entry of a drawn k is evaluated on demand by the
//       resample kernel (pft_device_utils.h alias_q / alias_a_small).
//
// Layout.  256-thread workgroups; thread t of workgroup g owns the K = 2^a consecutive particles
// (g * 256 + t) * K + j.  Every phase keeps that ownership, so only a few doubles per workgroup cross workgroups:
//   phase 0  raw weights (sum of the likelihood partial sums, fused), min / max(!= 0) per workgroup
//   -- barrier --
//   phase 1  w <- exp(1 - alpha (w - min) / (max - min)) (zeros stay zero), weight sum per workgroup
//   -- barrier --
//   phase 2  w <- w / (float) sum; weighted-pose sums and alias partition totals per workgroup
//   -- barrier --
//   phase 3  workgroup 0 finishes the mean (representative state, motion); every workgroup writes its piece of the
//            alias lists (suffix scans: both stacks pop the highest index first)
//
// SUMMATION ORDER (part of the product's specification, restated by the oracle's test-only sum mode 1): the weight
// sum and the six weighted-pose sums are ADJACENT-PAIR TREES in double over the index range padded with +0.0 to a
// power of two -- T0[i] = x[i], T(k+1)[i] = Tk[2i] + Tk[2i+1], result = the root.  Thread (pairs of its K values), wave
// (xor 1, 2, .. 32), workgroup (waves 0+1, 2+3), launch (workgroups pairwise) all follow that tree, so the result does not
// depend on K, on the number of workgroups, on the number of GPUs, or on which GPU computes it.  PCL adds
// sequentially (weights in double, poses in float): <= 1 ulp(float) on the sum, ~1e-7 on the pose.
#include "pft_device_utils.h"

#ifndef PFT_POPC_POLL_SLEEP
#define PFT_POPC_POLL_SLEEP 1  // s_sleep units (64 cycles) between two polls of a barrier counter
#endif
#ifdef PFT_DIAG  // phase stamps for tools/phase_ticks.py: diagnostic variant only (an s_memrealtime + wait each)
#define STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) d.hdr->ticks[16 + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

#define PFT_POPC_THREADS 256
#define PFT_POPC_SPIN_LIMIT (1u << 22)
// pop_part[g][..]: per-workgroup values that cross workgroups
enum { PP_MIN = 0, PP_MAX = 1, PP_SUM = 2, PP_MEAN = 3, PP_CNT = 9, PP_DEF = 10, PP_EXC = 11 };

struct PopcSh {
  double red[10][4];
  uint32_t u[4];
  double da[4], db[4];
  uint32_t timed_out;  // grid_barrier's verdict for the workgroup
};

// every thread receives the workgroup's adjacent-pair tree sum of each of its NV values
template <int NV>
__device__ __forceinline__ void wg_tree_sum(double (&v)[NV], PopcSh& S) {
  const int lane = lane_id(), w = wave_id();
#pragma unroll
  for (int k = 0; k < NV; k++) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v[k] += __shfl_xor(v[k], o);
  }
  __syncthreads();  // (the scratch may still be read from its previous use)
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) S.red[k][w] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = (S.red[k][0] + S.red[k][1]) + (S.red[k][2] + S.red[k][3]);
}

template <int K>
__device__ __forceinline__ double thread_tree_sum(double (&v)[K]) {
#pragma unroll
  for (int s = 1; s < K; s <<= 1) {
#pragma unroll
    for (int j = 0; j + s < K; j += 2 * s) v[j] += v[j + s];
  }
  return v[0];
}

// The few doubles per workgroup that cross workgroups travel through device-scope atomics (performed at the level all
// XCDs share), not through cached loads / stores: the barrier then needs no L2 write-back or invalidate (a pair of
// __threadfence() costs about 3 us per barrier here; everything bulky stays with the thread that wrote it).
__device__ __forceinline__ void pub_store(double* p, double v) {
  atomicExch(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ double pub_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT));
}

// Device-scope barrier of the launch's G co-resident workgroups (G <= 256 on a 256-CU part; they are tiny).  Thread 0
// of a workgroup has published the workgroup's values with pub_store; it waits for those atomics to be performed,
// arrives, and polls with a bounded spin: a launch that could not get all its workgroups resident in time raises error
// bit 4 instead of hanging the GPU.  Returns false (to every thread of the workgroup) when the barrier timed out here or
// in any other workgroup: the caller then writes NOTHING further -- no weights, no mean, no alias lists from values that
// did not arrive -- so the population keeps the state of the last good launch and the host reports the flag.  (The
// workgroups are tiny -- G <= 128 x 256 threads, ~400 B of LDS -- and fit beside anything but a kernel that holds every
// CU's whole LDS; they then start as that kernel's workgroups retire, which the spin limit, ~0.4 s, outlasts.)
__device__ __forceinline__ bool grid_barrier(PftHeader* hdr, int k, uint32_t G, uint32_t* sh_flag) {
  if (G <= 1u) {
    __syncthreads();
    return true;
  }
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    atomicAdd(&hdr->pop_bar[k], 1u);
    uint32_t spins = 0;
    while (__hip_atomic_load(&hdr->pop_bar[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
      if (++spins > PFT_POPC_SPIN_LIMIT) {
        atomicOr(&hdr->error, 16u);
        break;
      }
      __builtin_amdgcn_s_sleep(PFT_POPC_POLL_SLEEP);
    }
    asm volatile("" ::: "memory");
    *sh_flag = __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16u;
  }
  __syncthreads();
  return *sh_flag == 0u;
}

template <int K>
__device__ __forceinline__ void population_body(const PftParams& prm, const PftDev& d, uint32_t n, int from_partials,
                                                int do_norm, int do_mean, int do_alias, PopcSh& S) {
  pft_particle* P = d.part_all;
  PftHeader* hdr = d.hdr;
  const uint32_t tid = threadIdx.x, g = blockIdx.x, G = gridDim.x;
  const int lane = lane_id(), w = wave_id();
  const uint32_t i0 = (g * PFT_POPC_THREADS + tid) * (uint32_t)K;
  double* part = d.pop_part;

  bool timed_out = false;
  STAMP(0);
  // ---- phase 0: raw (or given) weights ----
  float wr[K];
#pragma unroll
  for (int j = 0; j < K; j++) {
    const uint32_t i = i0 + j;
    wr[j] = 0.0f;
    if (i < n) {
      if (from_partials) {  // w = -(float) val, val = sum of the per-chunk likelihood partial sums, in chunk order
        // (the loads go out eight at a time: one load, wait, add per chunk -- what the plain loop compiles to -- is a chain
        // of nchunk cache latencies, 11 at the headline size and 35 with the 64-point items of a 400-particle filter; the
        // additions stay in chunk order, so the value is the same)
        double v = 0.0;
        const double* row = d.partial + (size_t)i * prm.nchunk;
        for (uint32_t c0 = 0; c0 < prm.nchunk; c0 += 8u) {
          double tv[8];
#pragma unroll
          for (uint32_t k = 0; k < 8u; k++) tv[k] = row[min(c0 + k, prm.nchunk - 1u)];
#pragma unroll
          for (uint32_t k = 0; k < 8u; k++)
            if (c0 + k < prm.nchunk) v += tv[k];
        }
        wr[j] = -(float)v;
        if (!do_norm) P[i].weight = wr[j];
        if (d.raw_w) d.raw_w[i] = wr[j];
      } else {
        wr[j] = P[i].weight;
      }
    }
  }
  if (do_norm) {
    double wmin = DBL_MAX, wmax = -DBL_MAX;
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (i0 + j < n) {
        const double x = (double)wr[j];
        if (wmin > x) wmin = x;
        if (x != 0.0 && wmax < x) wmax = x;
      }
    }
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      wmin = fmin(wmin, __shfl_xor(wmin, o));
      wmax = fmax(wmax, __shfl_xor(wmax, o));
    }
    if (lane == 0) {
      S.da[w] = wmin;
      S.db[w] = wmax;
    }
    __syncthreads();
    if (tid == 0) {
      pub_store(&part[g * 16 + PP_MIN], fmin(fmin(S.da[0], S.da[1]), fmin(S.da[2], S.da[3])));
      pub_store(&part[g * 16 + PP_MAX], fmax(fmax(S.db[0], S.db[1]), fmax(S.db[2], S.db[3])));
    }
    bool live = grid_barrier(hdr, 0, G, &S.timed_out);
    STAMP(1);
    // ---- phase 1: the exponential, the weight sum ----
    wmin = tid < G ? pub_load(&part[tid * 16 + PP_MIN]) : DBL_MAX;
    wmax = tid < G ? pub_load(&part[tid * 16 + PP_MAX]) : -DBL_MAX;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      wmin = fmin(wmin, __shfl_xor(wmin, o));
      wmax = fmax(wmax, __shfl_xor(wmax, o));
    }
    if (lane == 0) {
      S.da[w] = wmin;
      S.db[w] = wmax;
    }
    __syncthreads();
    wmin = fmin(fmin(S.da[0], S.da[1]), fmin(S.da[2], S.da[3]));
    wmax = fmax(fmax(S.db[0], S.db[1]), fmax(S.db[2], S.db[3]));
    if (g == 0 && tid == 0) hdr->fit_ratio = wmin;
    double sv[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
      sv[j] = 0.0;
      if (i0 + j < n) {
        if (wmax != wmin) {
          if (wr[j] != 0.0f) wr[j] = (float)exp(1.0 - prm.alpha * ((double)wr[j] - wmin) / (wmax - wmin));
        } else {
          wr[j] = 1.0f / (float)n;
        }
        sv[j] = (double)wr[j];
      }
    }
    double s1[1] = {thread_tree_sum<K>(sv)};
    wg_tree_sum<1>(s1, S);
    if (tid == 0) pub_store(&part[g * 16 + PP_SUM], s1[0]);
    live = grid_barrier(hdr, 1, G, &S.timed_out) && live;
    STAMP(2);
    // ---- phase 2: normalise ----
    s1[0] = tid < G ? pub_load(&part[tid * 16 + PP_SUM]) : 0.0;
    wg_tree_sum<1>(s1, S);
    const double sum = s1[0];
    const float fs = (float)sum;
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (i0 + j < n) {
        wr[j] = (sum != 0.0) ? wr[j] / fs : 1.0f / (float)n;
        if (live) P[i0 + j].weight = wr[j];
      }
    }
    timed_out = !live;
  }

  // ---- weighted-pose sums and alias partition totals of this workgroup ----
  // (after a timed-out barrier the workgroups still meet at the remaining barriers and at the counter reset below, but
  // write nothing: `timed_out` is workgroup-uniform)
  uint32_t cntL = 0;
  double defs = 0.0, excs = 0.0;
  if (do_mean || do_alias) {
    double tot[9];
    if (do_mean) {
      constexpr int comp[6] = {0, 1, 2, 4, 5, 6};  // x, y, z, roll, pitch, yaw inside the 8-float particle
#pragma unroll
      for (int k = 0; k < 6; k++) {
        double tv[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
          tv[j] = 0.0;
          if (i0 + j < n) tv[j] = (double)reinterpret_cast<const float*>(P + i0 + j)[comp[k]] * (double)wr[j];
        }
        tot[k] = thread_tree_sum<K>(tv);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; k++) tot[k] = 0.0;
    }
    if (do_alias) {
#pragma unroll
      for (int j = 0; j < K; j++) {
        if (i0 + j < n) {
          const double q = (double)(wr[j] * (float)n);  // float product widened to double, as genAliasTable does
          if (q < 1.0) {
            cntL++;
            defs += 1.0 - q;
          } else {
            excs += q - 1.0;
          }
        }
      }
    }
    tot[6] = (double)cntL;
    tot[7] = defs;
    tot[8] = excs;
    wg_tree_sum<9>(tot, S);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 6; k++) pub_store(&part[g * 16 + PP_MEAN + k], tot[k]);
      pub_store(&part[g * 16 + PP_CNT], tot[6]);
      pub_store(&part[g * 16 + PP_DEF], tot[7]);
      pub_store(&part[g * 16 + PP_EXC], tot[8]);
    }
    timed_out = !grid_barrier(hdr, 2, G, &S.timed_out) || timed_out;
    STAMP(3);
    // ---- phase 3: every workgroup combines the workgroups' values the same way ----
    double fin[10];
#pragma unroll
    for (int k = 0; k < 6; k++) fin[k] = tid < G ? pub_load(&part[tid * 16 + PP_MEAN + k]) : 0.0;
    const bool after_me = tid < G && tid > g;  // reversed order: the workgroups with higher indices come first
    fin[9] = tid < G ? pub_load(&part[tid * 16 + PP_CNT]) : 0.0;
    fin[6] = after_me ? fin[9] : 0.0;
    fin[7] = after_me ? pub_load(&part[tid * 16 + PP_DEF]) : 0.0;
    fin[8] = after_me ? pub_load(&part[tid * 16 + PP_EXC]) : 0.0;
    wg_tree_sum<10>(fin, S);
    if (do_mean && g == 0 && tid == 0 && !timed_out) {
      pft_particle orig = hdr->rep, r;
      r.x = (float)fin[0]; r.y = (float)fin[1]; r.z = (float)fin[2]; r.w = 1.0f;
      r.roll = (float)fin[3]; r.pitch = (float)fin[4]; r.yaw = (float)fin[5];
      r.weight = 1.0f / (float)n;
      pft_particle m;
      m.x = r.x - orig.x; m.y = r.y - orig.y; m.z = r.z - orig.z; m.w = 1.0f;
      m.roll = r.roll - orig.roll; m.pitch = r.pitch - orig.pitch; m.yaw = r.yaw - orig.yaw;
      m.weight = 0.0f;
      hdr->rep = r;
      hdr->motion = m;
    }
    if (do_alias && !timed_out) {
      if (g == 0 && tid == 0) {
        hdr->alias_m = (uint32_t)fin[9];
        hdr->alias_nh = n - (uint32_t)fin[9];
      }
      int32_t* Llist = d.alias_list;
      int32_t* Hlist = d.alias_list + n;
      double* Dp = d.alias_pref;      // inclusive running deficit over the L list
      double* Ep = d.alias_pref + n;  // inclusive running excess over the H list
      // both lists run from the highest particle index down: inclusive SUFFIX scans over the threads of the workgroup
      uint32_t iu = cntL;
      double ia = defs, ib = excs;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t nu = __shfl_down(iu, o);
        const double na = __shfl_down(ia, o), nb = __shfl_down(ib, o);
        if (lane + o < WAVE) {
          iu += nu;
          ia += na;
          ib += nb;
        }
      }
      __syncthreads();
      if (lane == 0) {
        S.u[w] = iu;
        S.da[w] = ia;
        S.db[w] = ib;
      }
      __syncthreads();
      uint32_t offL = (uint32_t)fin[6] + iu - cntL;
      double offD = fin[7], offE = fin[8];
      for (int w2 = 3; w2 > w; w2--) {  // the waves after mine, nearest last (one running sum)
        offL += S.u[w2];
        offD += S.da[w2];
        offE += S.db[w2];
      }
      offD += ia - defs;
      offE += ib - excs;
      const uint32_t iend = min(i0 + (uint32_t)K, n);
      const uint32_t before = n - (i0 < n ? iend : n);  // particles with a higher index than mine
      uint32_t offH = before - offL;
#pragma unroll
      for (int j = K - 1; j >= 0; j--) {
        const uint32_t i = i0 + j;
        if (i < n) {
          const double q = (double)(wr[j] * (float)n);
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
  }
  STAMP(4);
  // the last workgroup through resets the barrier counters for the next launch; a failed launch tells the host
  if (G > 1u && tid == 0) {
    const uint32_t done = atomicAdd(&hdr->pop_bar[3], 1u);
    if (done == G - 1u) {
      atomicExch(&hdr->pop_bar[0], 0u);
      atomicExch(&hdr->pop_bar[1], 0u);
      atomicExch(&hdr->pop_bar[2], 0u);
      atomicExch(&hdr->pop_bar[3], 0u);
      const uint32_t e = __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((e & 16u) && d.host_stat) {
        d.host_stat[2] = e;
        d.host_stat[3] |= e;
      }
    }
  }
}

template <int K>
__global__ __launch_bounds__(PFT_POPC_THREADS) void k_population(PftParams prm, PftDev d, uint32_t n, int from_partials,
                                                                int do_norm, int do_mean, int do_alias) {
  __shared__ PopcSh S;
  if (d.p_active) n = *d.p_active;  // KLD variant: particle_num_ lives on the device (the grid covers the capacity)
  population_body<K>(prm, d, n, from_partials, do_norm, do_mean, do_alias, S);
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

// n = particles (KLD variant: the capacity; the kernel reads the live count).  from_partials != 0: the raw weights are
// first formed from the likelihood partial sums of d.partial (fuses k_finalize_raw); d.raw_w, if set, receives them.
#ifndef PFT_POP_WGS
#define PFT_POP_WGS 128u  // workgroups up to which a thread keeps one particle: at 65 536 particles (the replicated population
                          // of an 8-GPU run) 128 workgroups with two particles per thread take 60.9 us per frame, 256 with one 68.3, 64
                          // with four 67.1 (tools/diag/pop_wgs.sh; PFT_POP_MAX_WGS overrides)
#endif
void pftk_population(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n, int from_partials,
                     int do_normalize, int do_mean, int do_alias) {
  if (!n) return;
  // one particle per thread over G workgroups until G would exceed PFT_POP_WGS (the sums are the same adjacent-pair trees for
  // any K and G).  One workgroup with K particles per thread and no device-scope barrier was tried for the reference's
  // own 400-500 particles: slower (the two workgroups' barriers cost less than a second particle per thread).
  static const uint32_t max_wgs = getenv("PFT_POP_MAX_WGS") ? (uint32_t)atoi(getenv("PFT_POP_MAX_WGS")) : PFT_POP_WGS;
  uint32_t K = 1;
  while (K < 16u && (n + PFT_POPC_THREADS * K - 1) / (PFT_POPC_THREADS * K) > max_wgs) K <<= 1;
  while ((n + PFT_POPC_THREADS * K - 1) / (PFT_POPC_THREADS * K) > PFT_POPM_MAX_WGS) K <<= 1;  // n <= PFT_MAX_PARTICLES: K <= 16
  uint32_t G = 1;
  while (G * PFT_POPC_THREADS * K < n) G <<= 1;  // a power of two: the workgroups are the upper levels of the sum trees
  const dim3 grid(G), block(PFT_POPC_THREADS);
  switch (K) {
    case 1: hipLaunchKernelGGL(k_population<1>, grid, block, 0, s, p, d, n, from_partials, do_normalize, do_mean, do_alias); break;
    case 2: hipLaunchKernelGGL(k_population<2>, grid, block, 0, s, p, d, n, from_partials, do_normalize, do_mean, do_alias); break;
    case 4: hipLaunchKernelGGL(k_population<4>, grid, block, 0, s, p, d, n, from_partials, do_normalize, do_mean, do_alias); break;
    case 8: hipLaunchKernelGGL(k_population<8>, grid, block, 0, s, p, d, n, from_partials, do_normalize, do_mean, do_alias); break;
    default: hipLaunchKernelGGL(k_population<16>, grid, block, 0, s, p, d, n, from_partials, do_normalize, do_mean, do_alias); break;
  }
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
