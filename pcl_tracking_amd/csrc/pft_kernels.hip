// pft_kernels.hip -- per-element stages: packing, A0 initParticles, A11 resample (+A1 pose->matrix),
// A2+A3 transform + AABB, A4 crop.  gfx950, wave64, -ffp-contract=off (see pft_device_utils.h).
// Each kernel names the PCL 1.8.0 routine whose results it reproduces.
#include "pft_device_utils.h"

__global__ void k_pack_reference(const pft_point_xyzrgba* __restrict__ pts, uint32_t n, int argorder,
                                 float4* __restrict__ xyz, float4* __restrict__ hsv) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pft_point_xyzrgba p = pts[i];
  uint32_t k = hsv_pack_of_rgba(p.rgba, argorder);
  xyz[i] = make_float4(p.x, p.y, p.z, 1.0f);
  hsv[i] = make_float4((float)(k & 0xff) / 180.0f, (float)((k >> 8) & 0xff) / 255.0f,
                       (float)((k >> 16) & 0xff) / 255.0f, 0.0f);
}

__global__ void k_pack_input(const pft_point_xyzrgba* __restrict__ pts, uint32_t n, float4* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // 32-byte PCL point -> 16-byte record; two dwordx4 loads per lane, one dwordx4 store
  const float4* src = reinterpret_cast<const float4*>(pts + i);
  float4 a = src[0];
  float4 b = src[1];
  out[i] = make_float4(a.x, a.y, a.z, b.x);  // b.x carries the rgba bits
}

// A0  ParticleFilterTracker::initParticles(true) (tracking/impl/particle_filter.hpp)
__global__ void k_init_particles(PftParams p, pft_particle rep, pft_particle* __restrict__ out,
                                 float* __restrict__ mats, PftHeader* hdr) {
  uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li == 0 && hdr) {
    hdr->rep = rep;
    pft_particle z = {0, 0, 0, 1.0f, 0, 0, 0, 0};
    hdr->motion = z;
    hdr->p_active = p.P_total;
  }
  if (li >= p.P_local) return;
  pft_particle q = {0, 0, 0, 1.0f, 0, 0, 0, 0};
  particle_sample(q, p, p.init_sigma, p.init_mean, p.id_offset + li, 0, 0);
  q.x = q.x + rep.x; q.y = q.y + rep.y; q.z = q.z + rep.z;
  q.roll = q.roll + rep.roll; q.pitch = q.pitch + rep.pitch; q.yaw = q.yaw + rep.yaw;
  q.weight = 1.0f / (float)p.P_total;
  out[li] = q;
  if (mats) {
    float m[12];
    pose_to_matrix(q, m);
    store_matrix(mats, li, m);
  }
}

// A11 resampleWithReplacement + sampleWithReplacement, intended semantics: slot 0 is the
// representative state, every other slot an alias draw from the OLD population plus step noise.  The
// alias entry (a[k], q[k]) of the drawn k is evaluated on demand from the prefix-sum form (A9).
// Fused with A1 (pose -> matrix) for the new particle.
template <bool TABLE>
__global__ void k_resample(PftParams p, const pft_particle* __restrict__ old, AliasView v,
                           const int32_t* __restrict__ ta, const double* __restrict__ tq,
                           const PftHeader* __restrict__ hdr, uint32_t epoch, pft_particle* __restrict__ out,
                           float* __restrict__ mats) {
  uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ double cD[256], cE[256];
  if (!TABLE) {  // coarse levels of the two prefix arrays: 8 of the 13 dependent loads of a search become LDS reads
    v.m = hdr->alias_m;
    v.nh = hdr->alias_nh;
    v.sD = (v.m + 255u) / 256u;
    v.sE = (v.nh + 255u) / 256u;
    const uint32_t t = threadIdx.x;
    if (t < 256u) {
      if (v.m && t * v.sD < v.m) cD[t] = v.D[min((t + 1u) * v.sD, v.m) - 1u];
      if (v.nh && t * v.sE < v.nh) cE[t] = v.E[min((t + 1u) * v.sE, v.nh) - 1u];
    }
    __syncthreads();
    v.cD = cD;
    v.cE = cE;
  }
  if (li >= p.P_local) return;
  uint32_t g = p.id_offset + li;
  pft_particle s;
  if (g == 0) {
    s = hdr->rep;
  } else {
    uint32_t o[4];
    philox4x32(g, 0, epoch, 1, p.seed_lo, p.seed_hi, o);
    double rU = u53(o[0], o[1]) * (double)p.P_total;
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
    s = old[target];
    const double zero[6] = {0, 0, 0, 0, 0, 0};
    particle_sample(s, p, p.step_sigma, zero, g, epoch, 1);
  }
  out[li] = s;
  if (mats) {
    float m[12];
    pose_to_matrix(s, m);
    store_matrix(mats, li, m);
  }
}

__global__ void k_pose_to_matrix(const pft_particle* __restrict__ p, uint32_t n, float* __restrict__ mats) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m[12];
  pose_to_matrix(p[i], m);
  store_matrix(mats, i, m);
}

// ------------------------------------------------------------------------------------------------
// A2+A3  calcBoundingBox: min/max over all P x M transformed reference points.  One 1024-thread
// workgroup per CU holds the reference cloud in LDS (16 B/point, conflict-free ds_read_b128); each wave
// takes particles round-robin (3x4 matrix wave-uniform); transformed points are never stored.
// Output: per-workgroup partials {min xyz, max xyz}.
// ------------------------------------------------------------------------------------------------
// (v_min3_f32 / v_max3_f32 by name: fminf / fmaxf carry IEEE quieting that costs a canonicalising v_max per accumulator and
// iteration, and two points then share one instruction per bound; NaN operands are ignored, as getMinMax3D's compares do)
__device__ __forceinline__ float amin3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float amax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// IN_LDS: the reference cloud fits the workgroup's LDS (typed pointer: ds_read_b128, not flat loads); otherwise it is read
// from global memory / L2
template <bool IN_LDS>
__device__ __forceinline__ void aabb_body(const float4* __restrict__ ref, const float4* lref, uint32_t M,
                                          const float* __restrict__ mats, uint32_t n_particles, float* __restrict__ part,
                                          float (*s_red)[16]) {
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + w, tw = gridDim.x * nw;
  if (IN_LDS) {
    float4* wl = const_cast<float4*>(lref);
    for (uint32_t j = threadIdx.x; j < M; j += blockDim.x) wl[j] = ref[j];
    __syncthreads();
  }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t pi = gw; pi < n_particles; pi += tw) {
    float T[12];
    load_matrix(mats, pi, T);
    // x' = ((T0 x + T1 y) + T2 z) + T3: the translation is added once per particle, after the reduction -- float
    // addition of a constant is monotone, so min / max over the points of fl(s + T3) equal fl(min / max s + T3) exactly
    float pmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, pmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    uint32_t j = lane;
    for (; j + WAVE < M; j += 2 * WAVE) {  // two points per step: one three-operand min and max per axis
      const float4 r = IN_LDS ? lref[j] : ref[j], q = IN_LDS ? lref[j + WAVE] : ref[j + WAVE];
      const float x0 = T[0] * r.x + T[1] * r.y + T[2] * r.z, x1 = T[0] * q.x + T[1] * q.y + T[2] * q.z;
      const float y0 = T[4] * r.x + T[5] * r.y + T[6] * r.z, y1 = T[4] * q.x + T[5] * q.y + T[6] * q.z;
      const float z0 = T[8] * r.x + T[9] * r.y + T[10] * r.z, z1 = T[8] * q.x + T[9] * q.y + T[10] * q.z;
      pmn[0] = amin3(pmn[0], x0, x1); pmx[0] = amax3(pmx[0], x0, x1);
      pmn[1] = amin3(pmn[1], y0, y1); pmx[1] = amax3(pmx[1], y0, y1);
      pmn[2] = amin3(pmn[2], z0, z1); pmx[2] = amax3(pmx[2], z0, z1);
    }
    if (j < M) {
      const float4 r = IN_LDS ? lref[j] : ref[j];
      const float x0 = T[0] * r.x + T[1] * r.y + T[2] * r.z, y0 = T[4] * r.x + T[5] * r.y + T[6] * r.z,
                  z0 = T[8] * r.x + T[9] * r.y + T[10] * r.z;
      pmn[0] = amin3(pmn[0], x0, x0); pmx[0] = amax3(pmx[0], x0, x0);
      pmn[1] = amin3(pmn[1], y0, y0); pmx[1] = amax3(pmx[1], y0, y0);
      pmn[2] = amin3(pmn[2], z0, z0); pmx[2] = amax3(pmx[2], z0, z0);
    }
    if (lane < M) {  // (lanes without a point keep their neutral values)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float lo = pmn[k] + T[4 * k + 3], hi = pmx[k] + T[4 * k + 3];
        mn[k] = amin3(mn[k], lo, lo);
        mx[k] = amax3(mx[k], hi, hi);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float a = wave_min(mn[k]), b = wave_max(mx[k]);
    if (lane == 0) {
      s_red[k][w] = a;
      s_red[3 + k][w] = b;
    }
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float a = lane < nw ? s_red[k][lane] : FLT_MAX;
      float b = lane < nw ? s_red[3 + k][lane] : -FLT_MAX;
      a = wave_min(a);
      b = wave_max(b);
      if (lane == 0) {
        part[blockIdx.x * 6 + k] = a;
        part[blockIdx.x * 6 + 3 + k] = b;
      }
    }
  }
}

__global__ __launch_bounds__(1024) void k_aabb(const float4* __restrict__ ref, uint32_t M,
                                               const float* __restrict__ mats, uint32_t n_particles,
                                               float* __restrict__ part, uint32_t lds_points,
                                               const uint32_t* __restrict__ dyn_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_red[6][16];  // (one static block for both bodies: the dynamic request leaves 512 B for it)
  if (dyn_n) n_particles = *dyn_n;  // KLD variant: particle_num_ lives on the device
  if (M <= lds_points)
    aabb_body<true>(ref, reinterpret_cast<const float4*>(smem), M, mats, n_particles, part, s_red);
  else
    aabb_body<false>(ref, nullptr, M, mats, n_particles, part, s_red);
}

// The same resample with FOUR lanes per particle (the product path; k_resample<TABLE> above stays for the explicit-table
// test hook).  One lane's chain -- two binary searches, Philox, three Box-Muller pairs (log, sqrt, sin, cos in double),
// six more double sin / cos for the matrix -- is 7 us of pure latency for 8 192 particles; its pieces are independent:
//   lane 0 of a quad: the alias draw and the gather of the drawn particle
//   lanes 1..3:       one normal pair each (Philox slot = the lane's role)          -- concurrently with lane 0
//   then lanes 1..3:  cos / sin of roll / pitch / yaw of the new pose               -- concurrently
//   all four:         the nine matrix products; lane 0 stores
// Every number is formed by the same operations as in the one-lane kernel: bit-identical results.
// BOX: A2 + A3 in the same launch -- the quad that has drawn a particle also folds the particle's transformed reference
// cloud into the workgroup's box partial.  Worth it because the box needs only the support subset of the cloud
// (pft_hull.hip: 84 of the 2 048 points of the bench's model), 21 points per lane; the per-point expression, the
// translation added after the reduction and the NaN-ignoring min / max are those of k_aabb: the same bits.
#define PFT_BOX_FUSED_MAX 256u  // support points up to which the box is fused into the resample launch
template <bool BOX>
__global__ __launch_bounds__(256) void k_resample4(PftParams p, const pft_particle* __restrict__ old, AliasView v,
                                                   const PftHeader* __restrict__ hdr, uint32_t epoch,
                                                   pft_particle* __restrict__ out, float* __restrict__ mats,
                                                   const float4* __restrict__ box, float* __restrict__ part) {
  __shared__ double cD[256], cE[256];
  __shared__ float4 lbox[BOX ? PFT_BOX_FUSED_MAX : 1u];
  __shared__ float s_red[6][4];
  if (BOX)
    for (uint32_t j = threadIdx.x; j < p.M_box; j += blockDim.x) lbox[j] = box[j];
  v.m = hdr->alias_m;
  v.nh = hdr->alias_nh;
  v.sD = (v.m + 255u) / 256u;
  v.sE = (v.nh + 255u) / 256u;
  {
    const uint32_t t = threadIdx.x;
    if (v.m && t * v.sD < v.m) cD[t] = v.D[min((t + 1u) * v.sD, v.m) - 1u];
    if (v.nh && t * v.sE < v.nh) cE[t] = v.E[min((t + 1u) * v.sE, v.nh) - 1u];
  }
  __syncthreads();
  v.cD = cD;
  v.cE = cE;
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t li = gt >> 2, role = gt & 3u;
  const int lane = lane_id(), q0 = lane & ~3;  // first lane of my quad
  const bool live = li < p.P_local;
  const uint32_t g = p.id_offset + li;
  pft_particle s = {0, 0, 0, 1.0f, 0, 0, 0, 0};
  double z0 = 0.0, z1 = 0.0;
  if (live) {
    if (role == 0u) {
      if (g == 0) {
        s = hdr->rep;
      } else {
        uint32_t o[4];
        philox4x32(g, 0, epoch, 1, p.seed_lo, p.seed_hi, o);
        double rU = u53(o[0], o[1]) * (double)p.P_total;
        const int k = (int)rU;
        rU -= k;
        int32_t a_large;
        const double qk = alias_q(v, (uint32_t)k, old[k].weight, &a_large);
        int target;
        if (rU < qk)
          target = k;
        else
          target = (v.pos[k] >> 31) ? a_large : alias_a_small(v, (uint32_t)k);
        s = old[target];
      }
    } else if (g != 0) {
      normal_pair(p, g, role, epoch, 1, z0, z1);
    }
  }
  // the step noise of ParticleXYZRPY::sample: component += (float)(z * sigma + mean), mean = 0; role r holds the pair of
  // components 2r - 2, 2r - 1 (x y | z roll | pitch yaw)
  const float n0 = (role && g != 0) ? (float)(z0 * p.step_sigma[2u * role - 2u] + 0.0) : 0.0f;
  const float n1 = (role && g != 0) ? (float)(z1 * p.step_sigma[2u * role - 1u] + 0.0) : 0.0f;
  const float nx = __shfl(n0, q0 + 1), ny = __shfl(n1, q0 + 1), nz = __shfl(n0, q0 + 2), nroll = __shfl(n1, q0 + 2),
              npitch = __shfl(n0, q0 + 3), nyaw = __shfl(n1, q0 + 3);
  if (role == 0u && g != 0) {  // (slot 0 of the population is the representative state verbatim: no noise)
    s.x += nx; s.y += ny; s.z += nz;
    s.roll += nroll; s.pitch += npitch; s.yaw += nyaw;
  }
  // A1: lanes 1..3 take one angle each (double cos / sin rounded to float, as pose_to_matrix)
  const float roll = __shfl(s.roll, q0), pitch = __shfl(s.pitch, q0), yaw = __shfl(s.yaw, q0);
  const float ang = role == 1u ? roll : (role == 2u ? pitch : yaw);
  const float ca = (float)cos((double)ang), sa = (float)sin((double)ang);
  const float E = __shfl(ca, q0 + 1), F = __shfl(sa, q0 + 1), C = __shfl(ca, q0 + 2), D = __shfl(sa, q0 + 2),
              A = __shfl(ca, q0 + 3), B = __shfl(sa, q0 + 3);
  const float DE = D * E, DF = D * F;
  const float sx = __shfl(s.x, q0), sy = __shfl(s.y, q0), sz = __shfl(s.z, q0);
  float m[12];  // (every lane of the quad forms the matrix: the box below needs it in all four)
  m[0] = A * C;  m[1] = A * DF - B * E;  m[2] = B * F + A * DE;  m[3] = sx;
  m[4] = B * C;  m[5] = A * E + B * DF;  m[6] = B * DE - A * F;  m[7] = sy;
  m[8] = -D;     m[9] = C * F;           m[10] = C * E;          m[11] = sz;
  if (live && role == 0u) {
    out[li] = s;
    if (mats) store_matrix(mats, li, m);
  }
  if (BOX) {
    float pmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, pmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (live) {
      for (uint32_t j = role; j < p.M_box; j += 4u) {
        const float4 r = lbox[j];
        const float x0 = m[0] * r.x + m[1] * r.y + m[2] * r.z, y0 = m[4] * r.x + m[5] * r.y + m[6] * r.z,
                    z0 = m[8] * r.x + m[9] * r.y + m[10] * r.z;
        pmn[0] = amin3(pmn[0], x0, x0); pmx[0] = amax3(pmx[0], x0, x0);
        pmn[1] = amin3(pmn[1], y0, y0); pmx[1] = amax3(pmx[1], y0, y0);
        pmn[2] = amin3(pmn[2], z0, z0); pmx[2] = amax3(pmx[2], z0, z0);
      }
    }
    const int w = wave_id(), nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      // the translation once per lane that saw a point (monotone: as after the whole reduction), then quad, wave, workgroup
      float lo = (live && role < p.M_box) ? pmn[k] + m[4 * k + 3] : FLT_MAX;
      float hi = (live && role < p.M_box) ? pmx[k] + m[4 * k + 3] : -FLT_MAX;
      lo = wave_min(lo);
      hi = wave_max(hi);
      if (lane == 0) {
        s_red[k][w] = lo;
        s_red[3 + k][w] = hi;
      }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const int k = (int)threadIdx.x;
      float r = s_red[k][0];
      for (int i = 1; i < nw; i++) r = k < 3 ? amin3(r, s_red[k][i], s_red[k][i]) : amax3(r, s_red[k][i], s_red[k][i]);
      part[blockIdx.x * 6 + k] = r;
    }
  }
}

// partials -> b6 = {-xmin,-ymin,-zmin,xmax,ymax,zmax}: one max-reduction (also across ranks) gives the
// global box (x -> -x is exact in float).  Called by one wave per component set.
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, uint32_t nparts, float* s6) {
  // executed by the first 6 waves of a workgroup: wave k reduces component k
  const int w = wave_id(), lane = lane_id();
  if (w < 6) {
    float v = -FLT_MAX;
    for (uint32_t i = lane; i < nparts; i += WAVE) {
      float x = part[i * 6 + w];
      v = fmaxf(v, w < 3 ? -x : x);
    }
    v = wave_max(v);
    if (lane == 0) s6[w] = v;
  }
}

__global__ __launch_bounds__(512) void k_bbox_final(const float* __restrict__ part, uint32_t nparts,
                                                    float* __restrict__ bbox6) {
  __shared__ float s6[6];
  reduce_partials(part, nparts, s6);
  __syncthreads();
  if (threadIdx.x < 6) bbox6[threadIdx.x] = s6[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// A4  cropInputPointCloud: three inclusive PassThrough filters == one stable compaction.
// pass 1 counts per workgroup, pass 2 scatters (order-preserving) and converts colour to packed HSV.
// With FROM_PART the workgroup first folds the AABB partials itself (single-GPU path: no extra launch).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool crop_keep(float4 p, const float* b6) {
  // PassThrough drops non-finite points, keeps  min <= v <= max  per field
  bool fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  float xmin = -b6[0], ymin = -b6[1], zmin = -b6[2];
  return fin && !(p.x < xmin || p.x > b6[3]) && !(p.y < ymin || p.y > b6[4]) && !(p.z < zmin || p.z > b6[5]);
}

template <bool FROM_PART>
__global__ __launch_bounds__(1024) void k_crop_count(const float4* __restrict__ in, uint32_t N,
                                                     const float* __restrict__ bbox6,
                                                     const float* __restrict__ part, uint32_t nparts,
                                                     uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_cnt[16];
  __shared__ float s6[6];
  if (FROM_PART) {
    reduce_partials(part, nparts, s6);
  } else if (threadIdx.x < 6) {
    s6[threadIdx.x] = bbox6[threadIdx.x];
  }
  __syncthreads();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  if (i < N) keep = crop_keep(in[i], s6);
  unsigned long long m = __ballot(keep);
  if (lane_id() == 0) s_cnt[wave_id()] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_cnt[w];
    counts[blockIdx.x] = t;
  }
}

template <bool FROM_PART>
__global__ __launch_bounds__(1024) void k_crop_scatter(const float4* __restrict__ in, uint32_t N,
                                                       const float* __restrict__ bbox6,
                                                       const float* __restrict__ part, uint32_t nparts,
                                                       const uint32_t* __restrict__ counts, int argorder,
                                                       float4* __restrict__ out, int32_t* __restrict__ out_idx,
                                                       PftHeader* __restrict__ hdr, uint32_t* host_stat) {
  __shared__ uint32_t s_scan[20];
  __shared__ uint32_t s_base;
  __shared__ float s6[6];
  if (FROM_PART) {
    reduce_partials(part, nparts, s6);
  } else if (threadIdx.x < 6) {
    s6[threadIdx.x] = bbox6[threadIdx.x];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr->error = 0u;  // error flags are per iteration (the builder sets them)
  // offset of this workgroup = sum of the counts of the workgroups before it
  if (wave_id() == 7) {
    uint32_t t = 0;
    for (uint32_t b = lane_id(); b < blockIdx.x; b += WAVE) t += counts[b];
    t = wave_sum(t);
    if (lane_id() == 0) s_base = t;
  }
  __syncthreads();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  float4 p = make_float4(0, 0, 0, 0);
  if (i < N) {
    p = in[i];
    keep = crop_keep(p, s6);
  }
  uint32_t total;
  uint32_t pos = block_excl_scan<uint32_t>(keep ? 1u : 0u, s_scan, &total);
  if (keep) {
    uint32_t k = hsv_pack_of_rgba(__float_as_uint(p.w), argorder);
    out[s_base + pos] = make_float4(p.x, p.y, p.z, __uint_as_float(k));
    out_idx[s_base + pos] = (int32_t)i;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < PFT_LIK_GROUPS) hdr->lik_ctr[threadIdx.x * PFT_LIK_CTR_STRIDE] = 0u;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    hdr->n_crop = s_base + total;
    if (host_stat) host_stat[0] = s_base + total;
    hdr->bbox[0] = -s6[0]; hdr->bbox[1] = s6[3];
    hdr->bbox[2] = -s6[1]; hdr->bbox[3] = s6[4];
    hdr->bbox[4] = -s6[2]; hdr->bbox[5] = s6[5];
  }
}

// The same compaction in ONE launch.  A workgroup takes its logical index from a ticket (so every workgroup with a lower
// index is already running: waiting for them cannot deadlock, whatever the dispatch order), counts its kept points,
// publishes (launch epoch << 32 | count) with an atomic, and one wave collects the counts of its predecessors with
// atomic loads, spinning on the few that are not there yet (bounded: a failure raises the error flag instead of
// hanging).  50 000 points are 49 workgroups, so the chain is one step deep in practice.
// raw != null (the first crop after pft_set_input*): the points are read in PCL's 32-byte layout and their 16-byte
// records are written to `packed` on the way, for the frame's later crops (fuses k_pack_input: one launch less per frame)
template <bool FROM_PART>
__global__ __launch_bounds__(1024) void k_crop_onepass(const float4* __restrict__ in, uint32_t N,
                                                       const float* __restrict__ bbox6, const float* __restrict__ part,
                                                       uint32_t nparts, unsigned long long* __restrict__ slots,
                                                       uint32_t epoch, int argorder, float4* __restrict__ out,
                                                       int32_t* __restrict__ out_idx, PftHeader* __restrict__ hdr,
                                                       uint32_t* host_stat, const pft_point_xyzrgba* __restrict__ raw,
                                                       float4* __restrict__ packed) {
  __shared__ uint32_t s_scan[20];
  __shared__ uint32_t s_base, s_b;
  __shared__ float s6[6];
  if (threadIdx.x == 0) {
    s_b = atomicAdd(&hdr->crop_ticket, 1u);
    // error flags are per iteration: the first workgroup of the launch clears them (a later workgroup raises bit 2 only
    // after its bounded spin, long after this store; the builder re-reads bit 2 and adds its own bits)
    if (s_b == 0u) hdr->error = 0u;
  }
  if (FROM_PART) {
    reduce_partials(part, nparts, s6);
  } else if (threadIdx.x < 6) {
    s6[threadIdx.x] = bbox6[threadIdx.x];
  }
  __syncthreads();
  const uint32_t b = s_b, nb = gridDim.x;
  const uint32_t i = b * blockDim.x + threadIdx.x;
  bool keep = false;
  float4 p = make_float4(0, 0, 0, 0);
  if (i < N) {
    if (raw) {
      const float4* src = reinterpret_cast<const float4*>(raw + i);
      const float4 a = src[0], c = src[1];
      p = make_float4(a.x, a.y, a.z, c.x);  // c.x carries the rgba bits
      packed[i] = p;
    } else {
      p = in[i];
    }
    keep = crop_keep(p, s6);
  }
  uint32_t total;
  const uint32_t pos = block_excl_scan<uint32_t>(keep ? 1u : 0u, s_scan, &total);
  if (threadIdx.x == 0) atomicExch(&slots[b], ((unsigned long long)epoch << 32) | total);
  if (wave_id() == 7) {
    uint32_t t = 0;
    bool failed = false;
    for (uint32_t q = lane_id(); q < b; q += WAVE) {
      unsigned long long v = 0;
      uint32_t spins = 0;
      for (;;) {
        v = __hip_atomic_load(&slots[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) == epoch) break;
        if (++spins > (1u << 24)) {
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      t += (uint32_t)v;
    }
    t = wave_sum(t);
    if (__ballot(failed) && lane_id() == 0) hdr->error |= 4u;
    if (lane_id() == 0) s_base = t;
  }
  __syncthreads();
  if (keep) {
    uint32_t kk = hsv_pack_of_rgba(__float_as_uint(p.w), argorder);
    out[s_base + pos] = make_float4(p.x, p.y, p.z, __uint_as_float(kk));
    out_idx[s_base + pos] = (int32_t)i;
  }
  if (b == nb - 1 && threadIdx.x < PFT_LIK_GROUPS) hdr->lik_ctr[threadIdx.x * PFT_LIK_CTR_STRIDE] = 0u;
  if (b == nb - 1 && threadIdx.x == 0) {
    atomicExch(&hdr->crop_ticket, 0u);  // every workgroup of this launch has taken its ticket by now
    hdr->n_crop = s_base + total;
    if (host_stat) host_stat[0] = s_base + total;
    hdr->bbox[0] = -s6[0]; hdr->bbox[1] = s6[3];
    hdr->bbox[2] = -s6[1]; hdr->bbox[3] = s6[4];
    hdr->bbox[4] = -s6[2]; hdr->bbox[5] = s6[5];
  }
}

// raw weight of a particle: w = -(float) val  (ApproxNearestPairPointCloudCoherence::computeCoherence).
// shard != null (sharded handles): the particle with its raw weight also goes into the exchange buffer the all-gather reads
__global__ void k_finalize_raw(const double* __restrict__ partial, uint32_t nchunk, uint32_t n,
                               pft_particle* __restrict__ part, float* __restrict__ raw_out,
                               const uint32_t* __restrict__ p_active, pft_particle* __restrict__ shard) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (p_active) n = *p_active;  // KLD variant: particle_num_ lives on the device
  if (i >= n) return;
  double v = 0.0;  // (loads eight at a time, additions in chunk order: as k_population)
  const double* row = partial + (size_t)i * nchunk;
  for (uint32_t c0 = 0; c0 < nchunk; c0 += 8u) {
    double tv[8];
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) tv[k] = row[min(c0 + k, nchunk - 1u)];
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++)
      if (c0 + k < nchunk) v += tv[k];
  }
  float w = -(float)v;
  part[i].weight = w;
  if (raw_out) raw_out[i] = w;
  if (shard) {
    pft_particle q = part[i];
    q.weight = w;
    shard[i] = q;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// cached per device: handles may live on several GPUs of one process (pft_config.device_id)
static int g_max_lds[PFT_MAX_DEVICES];
int pftk_cur_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PFT_MAX_DEVICES) dev = 0;
  return dev;
}
int pftk_max_lds_bytes() {
  const int dev = pftk_cur_device();
  if (g_max_lds[dev] <= 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 65536;
    g_max_lds[dev] = v;
  }
  return g_max_lds[dev];
}

void pftk_pack_reference(hipStream_t s, const pft_point_xyzrgba* d_pts, uint32_t n, int argorder, float4* xyz,
                         float4* hsv) {
  if (!n) return;
  hipLaunchKernelGGL(k_pack_reference, dim3(cdiv(n, 256)), dim3(256), 0, s, d_pts, n, argorder, xyz, hsv);
}
void pftk_pack_input(hipStream_t s, const pft_point_xyzrgba* d_pts, uint32_t n, float4* out) {
  if (!n) return;
  hipLaunchKernelGGL(k_pack_input, dim3(cdiv(n, 256)), dim3(256), 0, s, d_pts, n, out);
}
void pftk_init_particles(hipStream_t s, const PftParams& p, pft_particle rep, pft_particle* out, float* mats,
                         PftHeader* hdr) {
  hipLaunchKernelGGL(k_init_particles, dim3(cdiv(p.P_local ? p.P_local : 1, 256)), dim3(256), 0, s, p, rep, out, mats,
                     hdr);
}
static AliasView alias_view(const PftDev& d, uint32_t n) {
  AliasView v;
  v.L = d.alias_list;
  v.H = d.alias_list + n;
  v.D = d.alias_pref;
  v.E = d.alias_pref + n;
  v.pos = d.alias_pos;
  v.m = 0;
  v.nh = 0;
  v.n = n;
  return v;
}
void pftk_resample(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out) {
  const bool one_lane = getenv("PFT_RESAMPLE_ONE_LANE") != nullptr;  // A/B and cross-check: the one-lane-per-particle kernel (read per call)
  if (one_lane)
    hipLaunchKernelGGL(k_resample<false>, dim3(cdiv(p.P_local, 256)), dim3(256), 0, s, p, d.part_all,
                       alias_view(d, p.P_total), (const int32_t*)nullptr, (const double*)nullptr, d.hdr, epoch, out,
                       d.mats);
  else
    hipLaunchKernelGGL(k_resample4<false>, dim3(cdiv(4u * p.P_local, 256)), dim3(256), 0, s, p, d.part_all,
                       alias_view(d, p.P_total), d.hdr, epoch, out, d.mats, (const float4*)nullptr, (float*)nullptr);
}
// resample + pose -> matrix + box partials in one launch; returns the number of partials written to d.bbox_part (0: the
// support subset is too large for the fused form, nothing was launched)
uint32_t pftk_resample_box(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out) {
  const uint32_t grid = cdiv(4u * p.P_local, 256);
  if (p.M_box == 0u || p.M_box > PFT_BOX_FUSED_MAX || grid > d.bbox_part_cap) return 0u;
  hipLaunchKernelGGL(k_resample4<true>, dim3(grid), dim3(256), 0, s, p, d.part_all, alias_view(d, p.P_total), d.hdr, epoch,
                     out, d.mats, d.ref_box, d.bbox_part);
  return grid;
}
void pftk_resample_table(hipStream_t s, const PftParams& p, const pft_particle* old, const int32_t* a,
                         const double* q, const PftHeader* hdr, uint32_t epoch, pft_particle* out) {
  AliasView v = {};
  v.n = p.P_total;
  hipLaunchKernelGGL(k_resample<true>, dim3(cdiv(p.P_local, 256)), dim3(256), 0, s, p, old, v, a, q, hdr, epoch, out,
                     (float*)nullptr);
}
void pftk_pose_to_matrix(hipStream_t s, const pft_particle* p, uint32_t n, float* mats) {
  if (!n) return;
  hipLaunchKernelGGL(k_pose_to_matrix, dim3(cdiv(n, 256)), dim3(256), 0, s, p, n, mats);
}
void pftk_aabb(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool finalize) {
  static bool attr_set[PFT_MAX_DEVICES];
  const uint32_t lds_max = (uint32_t)pftk_max_lds_bytes() - 512u;
  const int dev = pftk_cur_device();
  if (!attr_set[dev])
    attr_set[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aabb), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_max) == hipSuccess;
  uint32_t lds_points = lds_max / 16u;
  uint32_t lds = p.M_box <= lds_points ? p.M_box * 16u : 0u;
  hipLaunchKernelGGL(k_aabb, dim3(d.bbox_grid), dim3(1024), lds, s, d.ref_box, p.M_box, d.mats, n_particles, d.bbox_part,
                     lds_points, d.p_active);
  if (finalize) hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(512), 0, s, d.bbox_part, d.bbox_grid, d.bbox6);
}
void pftk_bbox_final(hipStream_t s, const PftDev& d) {
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(512), 0, s, d.bbox_part, d.bbox_grid, d.bbox6);
}
void pftk_crop(hipStream_t s, const PftParams& p, const PftDev& d, bool from_part, uint32_t epoch,
               const pft_point_xyzrgba* raw) {
  uint32_t nb = cdiv(d.N ? d.N : 1, 1024);
  const bool two_pass = getenv("PFT_CROP_TWO_PASS") != nullptr;  // A/B timing and cross-check (read per call: tests toggle it)
  float4* packed = const_cast<float4*>(d.in_pts);
  if (!two_pass) {
    if (from_part)
      hipLaunchKernelGGL(k_crop_onepass<true>, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.bbox_part, d.bbox_grid,
                         d.crop_slots, epoch, p.hsv_argorder, d.crop_pts, d.crop_idx, d.hdr, d.host_stat, raw, packed);
    else
      hipLaunchKernelGGL(k_crop_onepass<false>, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.bbox_part, d.bbox_grid,
                         d.crop_slots, epoch, p.hsv_argorder, d.crop_pts, d.crop_idx, d.hdr, d.host_stat, raw, packed);
    return;
  }
  if (raw) pftk_pack_input(s, raw, d.N, packed);  // the two-pass kernels read the 16-byte records
  if (from_part) {
    hipLaunchKernelGGL(k_crop_count<true>, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.bbox_part,
                       d.bbox_grid, d.crop_counts);
    hipLaunchKernelGGL(k_crop_scatter<true>, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.bbox_part,
                       d.bbox_grid, d.crop_counts, p.hsv_argorder, d.crop_pts, d.crop_idx, d.hdr, d.host_stat);
  } else {
    hipLaunchKernelGGL(k_crop_count<false>, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.bbox_part,
                       d.bbox_grid, d.crop_counts);
    hipLaunchKernelGGL(k_crop_scatter<false>, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.bbox_part,
                       d.bbox_grid, d.crop_counts, p.hsv_argorder, d.crop_pts, d.crop_idx, d.hdr, d.host_stat);
  }
}
void pftk_finalize_raw(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, float* raw_out,
                       pft_particle* shard) {
  if (!n_particles) return;
  hipLaunchKernelGGL(k_finalize_raw, dim3(cdiv(n_particles, 256)), dim3(256), 0, s, d.partial, p.nchunk, n_particles,
                     d.part_cur, raw_out, d.p_active, shard);
}
