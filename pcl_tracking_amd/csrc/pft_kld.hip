// pft_kld.hip -- KLDAdaptiveParticleFilterTracker::resample (PCL 1.8.0 tracking/impl/kld_adaptive_particle_filter.hpp;
// kld_adaptive_particle_filter.h: calcKLBound, normalQuantile), the tracker the reference runs by default
// (auto_tracking.cpp:207-222 with use_fixed == false, :821).
//
// Upstream is a sequential loop:
//     do { j = sampleWithReplacement(a, q);  x = particles[j];  x.sample(0, step_cov);
//          if (rand()/RAND_MAX < motion_ratio) x = x + motion;  S.push_back(x);
//          bin[i] = (int)(x[i] / bin_size[i]);  if (insertIntoBins(bin, B)) ++k;  ++n;
//     } while (n < max && (k < 2 || n < calcKLBound(k)));
// With a counter-based RNG (sample n draws from its own Philox stream) the candidates do not depend on each
// other, so one workgroup draws all `max` candidates in parallel, finds which of them open a new bin (first
// occurrence of their 6-D bin, through an open-addressing table with atomicMin on the candidate index),
// prefix-sums those flags into k(n), evaluates the stopping rule for every n and keeps the prefix up to the
// first n at which the loop would have stopped.  Same particles, same count, same k as the sequential loop.
// The new particle count stays on the device (PftHeader::p_active): the following kernels read it there, the
// host never waits for it.  Fused with A1 (pose -> matrix) like the fixed tracker's resample.
#include "pft_device_utils.h"

#define KLD_THREADS 1024
#define KLD_EMPTY 0xFFFFFFFFu

__device__ __forceinline__ bool same_bin(const int32_t* a, const int32_t* b) {
  return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3] && a[4] == b[4] && a[5] == b[5];
}

// KLDAdaptiveParticleFilterTracker::calcKLBound (z = normalQuantile(delta_) comes from the host)
__device__ __forceinline__ double kl_bound(uint32_t k, double z, double eps) {
  const double km1 = (double)((int)k - 1);
  const double chi = 1.0 - 2.0 / (9.0 * km1) + sqrt(2.0 / (9.0 * km1)) * z;
  return (km1 / 2.0 / eps) * chi * chi * chi;
}

template <bool TABLE>
__global__ __launch_bounds__(KLD_THREADS) void k_resample_kld(PftParams p, PftDev d, const int32_t* __restrict__ ta,
                                                              const double* __restrict__ tq, uint32_t epoch,
                                                              pft_particle* __restrict__ out, float* __restrict__ mats,
                                                              uint32_t tab_size, int32_t* __restrict__ bins_out) {
  __shared__ uint32_t scr[20];
  __shared__ uint32_t s_stop;
  // up to 1024 candidates (the reference runs 500) the bin table, the bins and the per-candidate slots / counts live in
  // LDS: the phases below are chains of dependent atomics and reads, and an LDS round trip is a tenth of an L2 one
  __shared__ uint32_t s_tab[2048 + 2 * 1024];
  __shared__ int32_t s_bins[6 * 1024];
  __shared__ double cD[256], cE[256];
  PftHeader* hdr = d.hdr;
  const pft_particle* old = d.part_all;
  const uint32_t tid = threadIdx.x, maxn = p.kld_max, n_old = hdr->p_active;
  const bool in_lds = maxn <= 1024u && tab_size <= 2048u;
  uint32_t* tab = in_lds ? s_tab : d.kld_table;  // [tab_size] table, then [maxn] slots, then [maxn] counts
  int32_t* bins = in_lds ? s_bins : d.kld_bins;
  for (uint32_t i = tid; i < tab_size; i += KLD_THREADS) tab[i] = KLD_EMPTY;
  if (tid == 0) s_stop = maxn;
  const pft_particle motion = hdr->motion;
  AliasView v;
  v.L = d.alias_list;
  v.H = d.alias_list + n_old;
  v.D = d.alias_pref;
  v.E = d.alias_pref + n_old;
  v.pos = d.alias_pos;
  v.m = hdr->alias_m;
  v.nh = hdr->alias_nh;
  v.n = n_old;
  if (!TABLE) {  // coarse levels of the two prefix arrays (as in k_resample)
    v.sD = (v.m + 255u) / 256u;
    v.sE = (v.nh + 255u) / 256u;
    if (tid < 256u) {
      if (v.m && tid * v.sD < v.m) cD[tid] = v.D[min((tid + 1u) * v.sD, v.m) - 1u];
      if (v.nh && tid * v.sE < v.nh) cE[tid] = v.E[min((tid + 1u) * v.sE, v.nh) - 1u];
    }
    __syncthreads();
    v.cD = cD;
    v.cE = cE;
  }

  // ---- all candidates ----
  for (uint32_t s = tid; s < maxn; s += KLD_THREADS) {
    uint32_t o[4];
    philox4x32(s, 0, epoch, 2, p.seed_lo, p.seed_hi, o);
    double rU = u53(o[0], o[1]) * (double)n_old;
    int k = (int)rU;
    rU -= k;
    int target;
    if (TABLE) {
      target = (rU < tq[k]) ? k : ta[k];
    } else {
      int32_t a_large;
      const double qk = alias_q(v, (uint32_t)k, old[k].weight, &a_large);
      if (rU < qk)
        target = k;
      else
        target = (v.pos[k] >> 31) ? a_large : alias_a_small(v, (uint32_t)k);
    }
    pft_particle x = old[target];
    const double zero[6] = {0, 0, 0, 0, 0, 0};
    particle_sample(x, p, p.step_sigma, zero, s, epoch, 2);
    philox4x32(s, 4, epoch, 2, p.seed_lo, p.seed_hi, o);
    if (u53(o[0], o[1]) < p.motion_ratio) {  // StateT operator+: the six pose floats
      x.x = x.x + motion.x; x.y = x.y + motion.y; x.z = x.z + motion.z;
      x.roll = x.roll + motion.roll; x.pitch = x.pitch + motion.pitch; x.yaw = x.yaw + motion.yaw;
    }
    out[s] = x;
    if (mats) {
      float m[12];
      pose_to_matrix(x, m);
      store_matrix(mats, s, m);
    }
    int32_t* b = bins + 6 * (size_t)s;
    b[0] = (int32_t)(x.x / p.kld_bin[0]);
    b[1] = (int32_t)(x.y / p.kld_bin[1]);
    b[2] = (int32_t)(x.z / p.kld_bin[2]);
    b[3] = (int32_t)(x.roll / p.kld_bin[3]);
    b[4] = (int32_t)(x.pitch / p.kld_bin[4]);
    b[5] = (int32_t)(x.yaw / p.kld_bin[5]);
  }
  __threadfence_block();
  __syncthreads();

  // ---- first occurrence of every bin: the table slot of a bin ends up holding its smallest candidate ----
  const uint32_t mask = tab_size - 1u;
  for (uint32_t s0 = 0; s0 < maxn; s0 += KLD_THREADS) {
    const uint32_t s = s0 + tid;
    if (s < maxn) {
      const int32_t* b = bins + 6 * (size_t)s;
      uint32_t h = ((uint32_t)b[0] * 73856093u) ^ ((uint32_t)b[1] * 19349663u) ^ ((uint32_t)b[2] * 83492791u) ^
                   ((uint32_t)b[3] * 2654435761u) ^ ((uint32_t)b[4] * 40503u) ^ ((uint32_t)b[5] * 2246822519u);
      h &= mask;
      for (;;) {
        const uint32_t cur = atomicCAS(&tab[h], KLD_EMPTY, s);
        if (cur == KLD_EMPTY) break;  // claimed an empty slot for this bin
        if (same_bin(bins + 6 * (size_t)cur, b)) {
          atomicMin(&tab[h], s);
          break;
        }
        h = (h + 1u) & mask;  // another bin lives here
      }
      tab[tab_size + s] = h;  // its slot, read again after the barrier
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- k(n) by prefix sum, the stopping rule for every n, the first n at which the loop stops ----
  uint32_t carry = 0;
  for (uint32_t s0 = 0; s0 < maxn; s0 += KLD_THREADS) {
    const uint32_t s = s0 + tid;
    uint32_t first = 0;
    if (s < maxn) first = tab[tab[tab_size + s]] == s ? 1u : 0u;
    uint32_t tot;
    const uint32_t k = carry + block_excl_scan<uint32_t>(first, scr, &tot) + first;  // distinct bins among 0..s
    carry += tot;
    if (s < maxn) {
      const uint32_t n = s + 1u;
      const bool cont = n < maxn && (k < 2u || (double)n < kl_bound(k, p.kld_z, p.kld_eps));
      if (!cont) atomicMin(&s_stop, n);
      tab[tab_size + maxn + s] = k;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const uint32_t n = s_stop;
    hdr->p_active = n;
    hdr->kld_k = tab[tab_size + maxn + n - 1u];
  }
  if (bins_out) {
    for (uint32_t i = tid; i < 6u * maxn; i += KLD_THREADS) bins_out[i] = bins[i];
  }
}

void pftk_resample_kld(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out,
                       const int32_t* table_a, const double* table_q, int32_t* bins_out) {
  uint32_t tab_size = 64;
  while (tab_size < 2u * p.kld_max) tab_size <<= 1;
  if (table_a)
    hipLaunchKernelGGL(k_resample_kld<true>, dim3(1), dim3(KLD_THREADS), 0, s, p, d, table_a, table_q, epoch, out,
                       d.mats, tab_size, bins_out);
  else
    hipLaunchKernelGGL(k_resample_kld<false>, dim3(1), dim3(KLD_THREADS), 0, s, p, d, (const int32_t*)nullptr,
                       (const double*)nullptr, epoch, out, d.mats, tab_size, bins_out);
}
