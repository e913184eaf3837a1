// pft_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64).  No MFMA: the path is
// transform + gather + reduce.  Built with -ffp-contract=off: PCL's float arithmetic on x86-64 has no
// FMA contraction and the greedy octree descent compares float sums, so a fused multiply-add would
// flip near-ties (SURVEY.md section 7 "Discrete argmin flips").
//
// Each kernel names the PCL 1.8.0 routine whose results it reproduces.
#include <float.h>
#include <math.h>

#include "pft_internal.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_mind(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_maxd(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ uint32_t wave_minu(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}

// inclusive scan inside a wave
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    T n = __shfl_up(v, o);
    if (lane >= o) v += n;
  }
  return v;
}

// exclusive scan over the workgroup (blockDim.x multiple of 64, <= 1024). scratch: >= 18 elements.
template <typename T>
__device__ T block_excl_scan(T v, T* scratch, T* total) {
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  T inc = wave_incl_scan(v);
  if (lane == WAVE - 1) scratch[w] = inc;
  __syncthreads();
  if (w == 0) {
    T t = lane < nw ? scratch[lane] : T(0);
    T ti = wave_incl_scan(t);
    if (lane < nw) scratch[lane] = ti - t;
    if (lane == nw - 1) scratch[17] = ti;
  }
  __syncthreads();
  T r = scratch[w] + inc - v;
  *total = scratch[17];
  __syncthreads();
  return r;
}

template <typename T, typename Op>
__device__ T block_reduce(T v, T* scratch, Op op, T identity) {
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  T r = lane < nw ? scratch[lane] : identity;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r = op(r, __shfl_xor(r, o));
  __syncthreads();
  return r;
}
struct OpAddD { __device__ double operator()(double a, double b) const { return a + b; } };
struct OpMinD { __device__ double operator()(double a, double b) const { return fmin(a, b); } };
struct OpMaxD { __device__ double operator()(double a, double b) const { return fmax(a, b); } };
struct OpMinF { __device__ float operator()(float a, float b) const { return fminf(a, b); } };
struct OpMaxF { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMinU { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; } };

// ------------------------------------------------------------------------------------------------
// A7b  RGB2HSV (PCL 1.8.0 tracking/impl/hsv_color_coherence.hpp) -- integer, done once per point
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int div_table(int i) {
  // upstream literal table == round((255 << 12) / i); no rounding ties for i in 1..255
  return i <= 0 ? 0 : __double2int_rn(1044480.0 / (double)i);
}

__device__ __forceinline__ void rgb2hsv_int(int r, int g, int b, int& h, int& s, int& v) {
  const int hsv_shift = 12;
  v = b;
  int vmin = b;
  v = max(v, g);
  v = max(v, r);
  vmin = min(vmin, g);
  vmin = min(vmin, r);
  int diff = v - vmin;
  int vr = v == r ? -1 : 0;
  int vg = v == g ? -1 : 0;
  s = (diff * div_table(v)) >> hsv_shift;
  h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * div_table(diff) * 15 + (1 << (hsv_shift + 6))) >> (7 + hsv_shift);
  h += h < 0 ? 180 : 0;
}

__device__ __forceinline__ uint32_t hsv_pack_of_rgba(uint32_t rgba, int argorder) {
  int Blue = rgba & 0xff, Green = (rgba >> 8) & 0xff, Red = (rgba >> 16) & 0xff;
  int h, s, v;
  if (argorder)
    rgb2hsv_int(Red, Blue, Green, h, s, v);  // RGB2HSV (rgb.Red, rgb.Blue, rgb.Green, ...) as upstream
  else
    rgb2hsv_int(Red, Green, Blue, h, s, v);
  return (uint32_t)h | ((uint32_t)s << 8) | ((uint32_t)v << 16);
}

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

// ------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10 keyed by the seed, counter = (global particle id, slot, epoch, purpose).
// Same specification as the CPU checker uses (DESIGN.md "RNG"); independent implementation.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t o[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    if (r > 0) {
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  unsigned long long m = ((unsigned long long)(a >> 5) << 26) | (unsigned long long)(b >> 6);
  return (double)m * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void normal_pair(const PftParams& p, uint32_t pid, uint32_t slot, uint32_t epoch,
                                            uint32_t purpose, double& z0, double& z1) {
  uint32_t o[4];
  philox4x32(pid, slot, epoch, purpose, p.seed_lo, p.seed_hi, o);
  double u1 = 1.0 - u53(o[0], o[1]);
  double u2 = u53(o[2], o[3]);
  double r = sqrt(-2.0 * log(u1));
  double th = 6.283185307179586 * u2;
  z0 = r * cos(th);
  z1 = r * sin(th);
}

// ParticleXYZRPY::sample(mean, cov): component += (float) N(mean, sqrt(cov)), order x,y,z,roll,pitch,yaw
__device__ __forceinline__ void particle_sample(pft_particle& q, const PftParams& p, const double* sigma,
                                                const double* mean, uint32_t pid, uint32_t epoch, uint32_t purpose) {
  double z[6];
  normal_pair(p, pid, 1, epoch, purpose, z[0], z[1]);
  normal_pair(p, pid, 2, epoch, purpose, z[2], z[3]);
  normal_pair(p, pid, 3, epoch, purpose, z[4], z[5]);
  q.x += (float)(z[0] * sigma[0] + mean[0]);
  q.y += (float)(z[1] * sigma[1] + mean[1]);
  q.z += (float)(z[2] * sigma[2] + mean[2]);
  q.roll += (float)(z[3] * sigma[3] + mean[3]);
  q.pitch += (float)(z[4] * sigma[4] + mean[4]);
  q.yaw += (float)(z[5] * sigma[5] + mean[5]);
}

// A1  pcl::getTransformation (common/impl/eigen.hpp): R = Rz(yaw) Ry(pitch) Rx(roll).
// sin/cos evaluated in double and rounded to float (PCL calls cosf/sinf; both are within 1 ulp).
__device__ __forceinline__ void pose_to_matrix(const pft_particle& q, float* m /*12*/) {
  float A = (float)cos((double)q.yaw), B = (float)sin((double)q.yaw);
  float C = (float)cos((double)q.pitch), D = (float)sin((double)q.pitch);
  float E = (float)cos((double)q.roll), F = (float)sin((double)q.roll);
  float DE = D * E, DF = D * F;
  m[0] = A * C;  m[1] = A * DF - B * E;  m[2] = B * F + A * DE;  m[3] = q.x;
  m[4] = B * C;  m[5] = A * E + B * DF;  m[6] = B * DE - A * F;  m[7] = q.y;
  m[8] = -D;     m[9] = C * F;           m[10] = C * E;          m[11] = q.z;
}

__device__ __forceinline__ void store_matrix(float* mats, uint32_t i, const float* m) {
  float4* dst = reinterpret_cast<float4*>(mats + 12 * (size_t)i);
  dst[0] = make_float4(m[0], m[1], m[2], m[3]);
  dst[1] = make_float4(m[4], m[5], m[6], m[7]);
  dst[2] = make_float4(m[8], m[9], m[10], m[11]);
}

// A0  ParticleFilterTracker::initParticles(true) (tracking/impl/particle_filter.hpp)
__global__ void k_init_particles(PftParams p, pft_particle rep, pft_particle* __restrict__ out,
                                 float* __restrict__ mats, PftHeader* hdr) {
  uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li == 0 && hdr) {
    hdr->rep = rep;
    pft_particle z = {0, 0, 0, 1.0f, 0, 0, 0, 0};
    hdr->motion = z;
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

// A11 resampleWithReplacement + sampleWithReplacement, intended semantics: slot 0 is
// the representative state, every other slot an alias draw from the OLD population plus step noise.
// Fused with A1 (pose -> matrix) for the new particle.
__global__ void k_resample(PftParams p, const pft_particle* __restrict__ old, const int32_t* __restrict__ a,
                           const double* __restrict__ q, const PftHeader* __restrict__ hdr, uint32_t epoch,
                           pft_particle* __restrict__ out, float* __restrict__ mats) {
  uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
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
    int target = (rU < q[k]) ? k : a[k];
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

// A2  pcl::transformPointCloud (common/impl/transforms.hpp): ((T0*x + T1*y) + T2*z) + T3, float, unfused.
__device__ __forceinline__ void xform(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = T[0] * x + T[1] * y + T[2] * z + T[3];
  oy = T[4] * x + T[5] * y + T[6] * z + T[7];
  oz = T[8] * x + T[9] * y + T[10] * z + T[11];
}

// ------------------------------------------------------------------------------------------------
// A3  calcBoundingBox: min/max over all P x M transformed reference points.  One wave per particle,
// reference points read coalesced (16 B/lane); transformed points are never stored.
// Output: per-workgroup partials {min xyz, max xyz}.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_aabb(const float4* __restrict__ ref, uint32_t M,
                                               const float* __restrict__ mats, uint32_t n_particles,
                                               float* __restrict__ part) {
  __shared__ float s_red[6][16];
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + w, tw = gridDim.x * nw;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t pi = gw; pi < n_particles; pi += tw) {
    float T[12];
    const float4* tp = reinterpret_cast<const float4*>(mats + 12 * (size_t)pi);
    float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
    T[0] = r0.x; T[1] = r0.y; T[2] = r0.z; T[3] = r0.w;
    T[4] = r1.x; T[5] = r1.y; T[6] = r1.z; T[7] = r1.w;
    T[8] = r2.x; T[9] = r2.y; T[10] = r2.z; T[11] = r2.w;
    for (uint32_t j = lane; j < M; j += WAVE) {
      float4 r = ref[j];
      float x, y, z;
      xform(T, r.x, r.y, r.z, x, y, z);
      mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
      mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
      mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
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

// partials -> bbox6 = {-xmin,-ymin,-zmin,xmax,ymax,zmax}: one max-reduction across ranks gives the
// global box (x -> -x is exact in float)
__global__ void k_bbox_final(const float* __restrict__ part, uint32_t nparts, float* __restrict__ bbox6) {
  __shared__ float s[16];
  for (int k = 0; k < 6; k++) {
    float v = -FLT_MAX;
    for (uint32_t i = threadIdx.x; i < nparts; i += blockDim.x) {
      float x = part[i * 6 + k];
      v = fmaxf(v, k < 3 ? -x : x);
    }
    v = block_reduce(v, s, OpMaxF(), -FLT_MAX);
    if (threadIdx.x == 0) bbox6[k] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// A4  cropInputPointCloud: three inclusive PassThrough filters == one stable compaction.
// pass 1 counts per workgroup, pass 2 scatters (order-preserving) and converts colour to packed HSV.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool crop_keep(float4 p, const float* b6) {
  // PassThrough drops non-finite points, keeps  min <= v <= max  per field
  bool fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  float xmin = -b6[0], ymin = -b6[1], zmin = -b6[2];
  return fin && !(p.x < xmin || p.x > b6[3]) && !(p.y < ymin || p.y > b6[4]) && !(p.z < zmin || p.z > b6[5]);
}

__global__ __launch_bounds__(1024) void k_crop_count(const float4* __restrict__ in, uint32_t N,
                                                     const float* __restrict__ bbox6,
                                                     uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_cnt[16];
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  if (i < N) keep = crop_keep(in[i], bbox6);
  unsigned long long m = __ballot(keep);
  if (lane_id() == 0) s_cnt[wave_id()] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_cnt[w];
    counts[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void k_crop_scatter(const float4* __restrict__ in, uint32_t N,
                                                       const float* __restrict__ bbox6,
                                                       const uint32_t* __restrict__ counts, int argorder,
                                                       float4* __restrict__ out, int32_t* __restrict__ out_idx,
                                                       PftHeader* __restrict__ hdr) {
  __shared__ uint32_t s_scan[20];
  __shared__ uint32_t s_base;
  // offset of this workgroup = sum of the counts of the workgroups before it
  if (wave_id() == 0) {
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
    keep = crop_keep(p, bbox6);
  }
  uint32_t total;
  uint32_t pos = block_excl_scan<uint32_t>(keep ? 1u : 0u, s_scan, &total);
  if (keep) {
    uint32_t k = hsv_pack_of_rgba(__float_as_uint(p.w), argorder);
    out[s_base + pos] = make_float4(p.x, p.y, p.z, __uint_as_float(k));
    out_idx[s_base + pos] = (int32_t)i;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    hdr->n_crop = s_base + total;
    hdr->bbox[0] = -bbox6[0]; hdr->bbox[1] = bbox6[3];
    hdr->bbox[2] = -bbox6[1]; hdr->bbox[3] = bbox6[4];
    hdr->bbox[4] = -bbox6[2]; hdr->bbox[5] = bbox6[5];
  }
}

// ------------------------------------------------------------------------------------------------
// A5  search::Octree(res).setInputCloud(cropped) == OctreePointCloud::addPointsFromInputCloud
// (octree/impl/octree_pointcloud.hpp).  One workgroup:
//   1. replay of the insertion-order-dependent bounding-box growth (adoptBoundingBoxToPoint): a few
//      rounds of "first point outside the current box" (parallel min-index search) + serial growth
//   2. keys of every point in the final key frame (insertion-time key + later root shifts)
//   3. top-down level build: atomicOr of child bits, exclusive scan of popcounts -> child_base
//   4. leaves: counts -> starts, points ranked by insertion index inside their leaf
//   5. per-level per-axis voxel-centre tables (genVoxelCenterFromOctreeKey), double -> float
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t child_bits(const uint32_t* key3, int bit) {
  return (((key3[0] >> bit) & 1u) << 2) | (((key3[1] >> bit) & 1u) << 1) | ((key3[2] >> bit) & 1u);
}

__global__ __launch_bounds__(PFT_BUILD_THREADS) void k_octree_build(PftParams prm, PftDev d) {
  __shared__ double s_min[3], s_max[3];
  __shared__ int s_depth, s_ngrow, s_done;
  __shared__ uint32_t s_cur, s_u32[20], s_carry, s_err;
  __shared__ uint32_t s_gidx[PFT_MAX_GROW], s_gshift[PFT_MAX_GROW], s_gold[PFT_MAX_GROW];
  __shared__ double s_gmin[PFT_MAX_GROW + 1][3];
  __shared__ uint32_t s_lvl[PFT_MAX_DEPTH + 3];

  PftHeader* hdr = d.hdr;
  const uint32_t n = hdr->n_crop;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const float4* pts = d.crop_pts;
  const double res = prm.res;
  const double epsd = (double)FLT_EPSILON;

  if (tid == 0) {
    s_err = 0;
    s_ngrow = 0;
    s_done = 0;
    s_depth = 0;
    if (n > 0) {
      // first point: box = p +- res/2, then getKeyBitSize() pads it to depth 1 (side 2*res - eps)
      float4 p0 = pts[0];
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
        s_min[a] = lo[a] - over;
        s_max[a] = hi[a] + over;
        s_gmin[0][a] = s_min[a];
      }
      s_depth = (int)dep;
    }
    s_cur = 1;
  }
  __syncthreads();

  // ---- 1. box growth replay ----
  if (n > 0) {
    for (;;) {
      const uint32_t cur = s_cur;
      const double mnx = s_min[0], mny = s_min[1], mnz = s_min[2];
      const double mxx = s_max[0], mxy = s_max[1], mxz = s_max[2];
      uint32_t first = 0xffffffffu;
      for (uint32_t i = cur + tid; i < n; i += nt) {
        float4 p = pts[i];
        bool viol = (p.x < mnx) || (p.y < mny) || (p.z < mnz) || (p.x >= mxx) || (p.y >= mxy) || (p.z >= mxz);
        if (viol) {
          first = i;
          break;
        }
      }
      first = block_reduce<uint32_t>(first, s_u32, OpMinU(), 0xffffffffu);
      if (first == 0xffffffffu) break;
      if (tid == 0) {
        float4 p = pts[first];
        for (;;) {
          bool lx = p.x < s_min[0], ly = p.y < s_min[1], lz = p.z < s_min[2];
          bool ux = p.x >= s_max[0], uy = p.y >= s_max[1], uz = p.z >= s_max[2];
          if (!(lx || ly || lz || ux || uy || uz)) break;
          int g = s_ngrow;
          if (g >= PFT_MAX_GROW || s_depth >= PFT_MAX_DEPTH) {
            s_err |= 2u;
            break;
          }
          double side = (double)(1u << s_depth) * res;
          s_gidx[g] = first;
          s_gshift[g] = (ux ? 0u : 1u) | (uy ? 0u : 2u) | (uz ? 0u : 4u);
          s_gold[g] = (uint32_t)s_depth;
          if (!ux) s_min[0] -= side;
          if (!uy) s_min[1] -= side;
          if (!uz) s_min[2] -= side;
          s_depth = s_depth + 1;
          side = (double)(1u << s_depth) * res - epsd;
          s_max[0] = s_min[0] + side;
          s_max[1] = s_min[1] + side;
          s_max[2] = s_min[2] + side;
          s_gmin[g + 1][0] = s_min[0];
          s_gmin[g + 1][1] = s_min[1];
          s_gmin[g + 1][2] = s_min[2];
          s_ngrow = g + 1;
        }
        s_cur = first + 1;
      }
      __syncthreads();
      if (s_err) break;
    }
  }
  __syncthreads();
  const int D = s_depth;
  const int ngrow = s_ngrow;

  // ---- 2. keys (genOctreeKeyforPoint at insertion time, shifted into the final key frame) ----
  for (uint32_t i = tid; i < n; i += nt) {
    float4 p = pts[i];
    int e = 0;
    while (e < ngrow && s_gidx[e] <= i) e++;
    uint32_t kx = (uint32_t)(((double)p.x - s_gmin[e][0]) / res);
    uint32_t ky = (uint32_t)(((double)p.y - s_gmin[e][1]) / res);
    uint32_t kz = (uint32_t)(((double)p.z - s_gmin[e][2]) / res);
    for (int s = e; s < ngrow; s++) {
      uint32_t sh = s_gshift[s], od = s_gold[s];
      if (sh & 1u) kx += 1u << od;
      if (sh & 2u) ky += 1u << od;
      if (sh & 4u) kz += 1u << od;
    }
    d.pt_key[3 * (size_t)i + 0] = kx;
    d.pt_key[3 * (size_t)i + 1] = ky;
    d.pt_key[3 * (size_t)i + 2] = kz;
    d.pt_node[i] = 0;
  }
  uint32_t* words = d.words;
  if (tid == 0) {
    words[0] = 0;
    s_lvl[0] = 0;
    s_lvl[1] = 1;
  }
  __syncthreads();

  // ---- 3. levels ----
  for (int l = 0; l < D && n > 0 && !s_err; l++) {
    const int bit = D - 1 - l;
    // (a) children masks; the step that moves a point to its level-l node is fused in
    for (uint32_t i = tid; i < n; i += nt) {
      uint32_t key[3] = {d.pt_key[3 * (size_t)i], d.pt_key[3 * (size_t)i + 1], d.pt_key[3 * (size_t)i + 2]};
      uint32_t node = d.pt_node[i];
      if (l > 0) {
        uint32_t w = words[node];
        uint32_t cprev = child_bits(key, bit + 1);
        node = (w >> 8) + __popc(w & 0xffu & ((1u << cprev) - 1u));
        d.pt_node[i] = node;
      }
      atomicOr(&words[node], 1u << child_bits(key, bit));
    }
    __threadfence_block();
    __syncthreads();
    // (b) child_base by exclusive scan of popcounts over this level's nodes
    const uint32_t ls = s_lvl[l], le = s_lvl[l + 1];
    if (tid == 0) s_carry = le;
    __syncthreads();
    for (uint32_t t0 = ls; t0 < le; t0 += nt) {
      uint32_t node = t0 + tid;
      uint32_t cnt = node < le ? __popc(words[node] & 0xffu) : 0u;
      uint32_t total;
      uint32_t ex = block_excl_scan<uint32_t>(cnt, s_u32, &total);
      uint32_t base = s_carry + ex;
      if (node < le) {
        if (base + cnt > d.max_words - 2 || base >= (1u << 24)) atomicOr(&s_err, 1u);
        else words[node] |= base << 8;
      }
      __syncthreads();
      if (tid == 0) s_carry += total;
      __syncthreads();
    }
    const uint32_t nend = s_carry;
    if (tid == 0) s_lvl[l + 2] = nend;
    if (!s_err)
      for (uint32_t j = le + tid; j < nend && j < d.max_words; j += nt) words[j] = 0;
    __threadfence_block();
    __syncthreads();
  }

  // ---- 4. leaves ----
  uint32_t leaf_start = 0, n_leaves = 0;
  if (n > 0 && !s_err && D > 0) {
    leaf_start = s_lvl[D];
    n_leaves = s_lvl[D + 1] - leaf_start;
    // move points to their leaf, count
    for (uint32_t i = tid; i < n; i += nt) {
      uint32_t key[3] = {d.pt_key[3 * (size_t)i], d.pt_key[3 * (size_t)i + 1], d.pt_key[3 * (size_t)i + 2]};
      uint32_t w = words[d.pt_node[i]];
      uint32_t c = child_bits(key, 0);
      uint32_t leaf = (w >> 8) + __popc(w & 0xffu & ((1u << c) - 1u));
      d.pt_node[i] = leaf;
      atomicAdd(&words[leaf], 1u);
    }
    for (uint32_t j = tid; j < n_leaves; j += nt) d.leaf_cursor[j] = 0;
    __threadfence_block();
    __syncthreads();
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n_leaves; t0 += nt) {
      uint32_t j = t0 + tid;
      uint32_t cnt = j < n_leaves ? words[leaf_start + j] : 0u;
      uint32_t total;
      uint32_t ex = block_excl_scan<uint32_t>(cnt, s_u32, &total);
      if (j < n_leaves) words[leaf_start + j] = s_carry + ex;
      __syncthreads();
      if (tid == 0) s_carry += total;
      __syncthreads();
    }
    if (tid == 0) words[leaf_start + n_leaves] = n;  // sentinel: count(leaf j) = start[j+1] - start[j]
    __threadfence_block();
    __syncthreads();
    // unsorted scatter of indices into each leaf's range ...
    for (uint32_t i = tid; i < n; i += nt) {
      uint32_t leaf = d.pt_node[i];
      uint32_t pos = words[leaf] + atomicAdd(&d.leaf_cursor[leaf - leaf_start], 1u);
      d.pt_tmp[pos] = i;
    }
    __threadfence_block();
    __syncthreads();
    // ... then rank by insertion index inside the leaf (leaf containers keep push_back order)
    for (uint32_t i = tid; i < n; i += nt) {
      uint32_t leaf = d.pt_node[i];
      uint32_t s = words[leaf], e = words[leaf + 1];
      uint32_t rank = 0;
      for (uint32_t k = s; k < e; k++) rank += d.pt_tmp[k] < i ? 1u : 0u;
      d.leaf_order[s + rank] = i;
      d.leaf_pts[s + rank] = pts[i];
    }
  }

  // ---- 5. voxel-centre tables: centre(level l, key k) = (float)((k + 0.5) * res*2^(D-l) + min) ----
  int use_table = (n > 0 && !s_err && D >= 1 && D <= PFT_TABLE_MAX_DEPTH) ? 1 : 0;
  if (use_table) {
    const uint32_t per_axis = (2u << D);  // entries 2^l - 2 + k, l = 1..D
    for (uint32_t e = tid; e < 3u * per_axis; e += nt) {
      uint32_t a = e / per_axis, r = e % per_axis;
      if (r + 2 >= (2u << D)) {  // slots past the last level
        d.centers[e] = 0.0f;
        continue;
      }
      // r = 2^l - 2 + k  ->  l = floor(log2(r + 2)), k = r + 2 - 2^l
      uint32_t l = 31u - __clz(r + 2u);
      uint32_t k = r + 2u - (1u << l);
      double vs = res * (double)(1u << (D - (int)l));
      d.centers[e] = (float)(((double)k + 0.5) * vs + s_min[a]);
    }
  }
  __syncthreads();
  if (tid == 0) {
    hdr->error = s_err;
    hdr->depth = D;
    hdr->use_table = use_table;
    hdr->n_grow = ngrow;
    hdr->n_leaves = n_leaves;
    hdr->leaf_start = leaf_start;
    hdr->n_words = (n > 0 && !s_err && D > 0) ? leaf_start + n_leaves + 1 : 0;
    for (int a = 0; a < 3; a++) {
      hdr->omin[a] = n > 0 ? s_min[a] : 0.0;
      hdr->omax[a] = n > 0 ? s_max[a] : 0.0;
    }
    for (int l = 0; l <= D + 1 && l < PFT_MAX_DEPTH + 3; l++) hdr->lvl_start[l] = s_lvl[l];
    for (int g = 0; g < ngrow; g++) {
      hdr->grow_idx[g] = s_gidx[g];
      hdr->grow_shift[g] = s_gshift[g];
      hdr->grow_old_depth[g] = s_gold[g];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// A6 + A7  the likelihood: per (particle, reference point) transform -> greedy closest-child-centre
// descent (OctreePointCloudSearch::approxNearestSearchRecursive) -> leaf scan -> DistanceCoherence x
// HSVColorCoherence, summed per particle in double.
//
// One persistent 1024-thread workgroup per CU stages the linearised octree (node words + centre
// tables + hue/saturation LUTs) into LDS once; each wave then walks work items (particle, chunk of 512
// reference points): the particle's 3x4 matrix is wave-uniform, reference points are read coalesced,
// the descent runs in registers against LDS, leaf records are gathered from L2, and the per-lane
// double partial sums are combined with wave shuffles.
// ------------------------------------------------------------------------------------------------
struct LikShared {
  const uint32_t* words;   // LDS or global
  const float* tab;        // per-axis centre tables (LDS or global), stride per_axis
  uint32_t per_axis;
  const float* lut_h;      // [256] h/180
  const float* lut_s;      // [256] s/255
};

template <bool USE_TAB, bool DEBUG_NN>
__device__ __forceinline__ void likelihood_items(const PftParams& prm, const PftDev& d, const LikShared& sh,
                                                 uint32_t n_particles, int D, uint32_t n_crop,
                                                 const double omin[3]) {
  const int lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
  const uint32_t gw = blockIdx.x * nw + w, tw = gridDim.x * nw;
  const uint32_t M = prm.M, nchunk = prm.nchunk;
  const uint32_t n_items = n_particles * nchunk;
  const double res = prm.res;
  const double maxd2 = prm.maxd2;
  const double wd = prm.dist_w, whsv = prm.hsv_w;
  const float hw = prm.h_w, sw = prm.s_w, vw = prm.v_w;

  for (uint32_t item = gw; item < n_items; item += tw) {
    const uint32_t pi = item / nchunk, ch = item % nchunk;
    float T[12];
    {
      const float4* tp = reinterpret_cast<const float4*>(d.mats + 12 * (size_t)pi);
      float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
      T[0] = r0.x; T[1] = r0.y; T[2] = r0.z; T[3] = r0.w;
      T[4] = r1.x; T[5] = r1.y; T[6] = r1.z; T[7] = r1.w;
      T[8] = r2.x; T[9] = r2.y; T[10] = r2.z; T[11] = r2.w;
    }
    double val = 0.0;
    unsigned long long st_q = 0, st_s = 0;
    const uint32_t jend = min(M, (ch + 1) * (uint32_t)PFT_REF_CHUNK);
    for (uint32_t j = ch * PFT_REF_CHUNK + lane; j < jend; j += WAVE) {
      const float4 r = d.ref_xyz[j];
      float qx, qy, qz;
      xform(T, r.x, r.y, r.z, qx, qy, qz);
      if (n_crop == 0) {  // empty target: PCL asserts; defined as "no correspondence"
        if (DEBUG_NN) {
          d.nn_idx[(size_t)pi * M + j] = -1;
          d.nn_d2[(size_t)pi * M + j] = INFINITY;
        }
        continue;
      }
      // ---- greedy descent ----
      uint32_t node = 0, kx = 0, ky = 0, kz = 0;
      for (int lvl = 1; lvl <= D; lvl++) {
        const uint32_t wv = sh.words[node];
        const uint32_t mask = wv & 0xffu, base = wv >> 8;
        float cx0, cx1, cy0, cy1, cz0, cz1;
        if (USE_TAB) {
          const uint32_t off = (1u << lvl) - 2u;
          const float2 tx = *reinterpret_cast<const float2*>(sh.tab + off + 2u * kx);
          const float2 ty = *reinterpret_cast<const float2*>(sh.tab + sh.per_axis + off + 2u * ky);
          const float2 tz = *reinterpret_cast<const float2*>(sh.tab + 2u * sh.per_axis + off + 2u * kz);
          cx0 = tx.x; cx1 = tx.y; cy0 = ty.x; cy1 = ty.y; cz0 = tz.x; cz1 = tz.y;
        } else {
          const double vs = res * (double)(1u << (D - lvl));
          cx0 = (float)(((double)(2u * kx) + 0.5) * vs + omin[0]);
          cx1 = (float)(((double)(2u * kx + 1u) + 0.5) * vs + omin[0]);
          cy0 = (float)(((double)(2u * ky) + 0.5) * vs + omin[1]);
          cy1 = (float)(((double)(2u * ky + 1u) + 0.5) * vs + omin[1]);
          cz0 = (float)(((double)(2u * kz) + 0.5) * vs + omin[2]);
          cz1 = (float)(((double)(2u * kz + 1u) + 0.5) * vs + omin[2]);
        }
        // pointSquaredDist: Vector3f difference, squaredNorm = x2 + (y2 + z2)
        float dx0 = cx0 - qx, dx1 = cx1 - qx, dy0 = cy0 - qy, dy1 = cy1 - qy, dz0 = cz0 - qz, dz1 = cz1 - qz;
        float X0 = dx0 * dx0, X1 = dx1 * dx1, Y0 = dy0 * dy0, Y1 = dy1 * dy1, Z0 = dz0 * dz0, Z1 = dz1 * dz1;
        float yz[4] = {Y0 + Z0, Y0 + Z1, Y1 + Z0, Y1 + Z1};
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
        node = base + __popc(mask & ((1u << bc) - 1u));
        kx = 2u * kx + ((bc >> 2) & 1u);
        ky = 2u * ky + ((bc >> 1) & 1u);
        kz = 2u * kz + (bc & 1u);
      }
      // ---- leaf scan: first strictly-smaller wins (insertion order) ----
      const uint32_t ls = sh.words[node], le = sh.words[node + 1];
      float bd = INFINITY;
      uint32_t bpos = ls;
      float4 bt = make_float4(0, 0, 0, 0);
      for (uint32_t pos = ls; pos < le; pos++) {
        const float4 c = d.leaf_pts[pos];
        float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
        float dd = dx * dx + (dy * dy + dz * dz);
        if (dd < bd) {
          bd = dd;
          bpos = pos;
          bt = c;
        }
      }
      if (DEBUG_NN) {
        d.nn_idx[(size_t)pi * M + j] = (int32_t)d.leaf_order[bpos];
        d.nn_d2[(size_t)pi * M + j] = bd;
        st_q += 1;
        st_s += le - ls;
      }
      // ---- A7: gate + point coherences ----
      if ((double)bd < maxd2) {
        // DistanceCoherence: Vector4f norm (SSE3 packet reduction (dx2+dy2)+(dz2+0)), then 1/(1+d*d*w)
        float ex = qx - bt.x, ey = qy - bt.y, ez = qz - bt.z;
        float n2 = (ex * ex + ey * ey) + ez * ez;
        double dist = (double)sqrtf(n2);
        double cd = 1.0 / (1.0 + dist * dist * wd);
        // HSVColorCoherence on precomputed (h,s,v)
        const float4 rh = d.ref_hsv[j];
        const uint32_t pk = __float_as_uint(bt.w);
        const float th = sh.lut_h[pk & 0xffu], ts = sh.lut_s[(pk >> 8) & 0xffu], tv = sh.lut_s[(pk >> 16) & 0xffu];
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
        double chs = 1.0 / (1.0 + whsv * (double)diff2);
        val += cd * chs;
      }
    }
    val = wave_sum(val);
    if (lane == 0) d.partial[(size_t)pi * nchunk + ch] = val;
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

template <bool DEBUG_NN>
__global__ __launch_bounds__(PFT_LIK_THREADS) void k_likelihood(PftParams prm, PftDev d, uint32_t n_particles,
                                                                uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const PftHeader* hdr = d.hdr;
  const int D = hdr->depth;
  const uint32_t n_crop = (hdr->error || D <= 0) ? 0u : hdr->n_crop;
  const uint32_t n_words = hdr->n_words;
  const int use_tab = hdr->use_table;
  const double omin[3] = {hdr->omin[0], hdr->omin[1], hdr->omin[2]};
  const uint32_t per_axis = use_tab ? (2u << D) : 0u;

  // LDS carve: luts (2 KiB) | centre tables | node words
  float* lut_h = reinterpret_cast<float*>(smem);
  float* lut_s = lut_h + 256;
  float* tab = lut_s + 256;
  uint32_t used = 2048u + 3u * per_axis * 4u;
  used = (used + 15u) & ~15u;
  uint32_t* lwords = reinterpret_cast<uint32_t*>(smem + used);
  const bool words_in_lds = (size_t)used + (size_t)n_words * 4u <= (size_t)lds_bytes;

  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    lut_h[i] = (float)i / 180.0f;
    lut_s[i] = (float)i / 255.0f;
  }
  for (uint32_t i = threadIdx.x; i < 3u * per_axis; i += blockDim.x) tab[i] = d.centers[i];
  if (words_in_lds)
    for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) lwords[i] = d.words[i];
  __syncthreads();

  LikShared sh;
  sh.words = words_in_lds ? lwords : d.words;
  sh.tab = tab;
  sh.per_axis = per_axis;
  sh.lut_h = lut_h;
  sh.lut_s = lut_s;
  if (use_tab)
    likelihood_items<true, DEBUG_NN>(prm, d, sh, n_particles, D, n_crop, omin);
  else
    likelihood_items<false, DEBUG_NN>(prm, d, sh, n_particles, D, n_crop, omin);
}

// raw weight of a particle: w = -(float) val  (ApproxNearestPairPointCloudCoherence::computeCoherence)
__global__ void k_finalize_raw(const double* __restrict__ partial, uint32_t nchunk, uint32_t n,
                               pft_particle* __restrict__ part, float* __restrict__ raw_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (uint32_t c = 0; c < nchunk; c++) v += partial[(size_t)i * nchunk + c];
  float w = -(float)v;
  part[i].weight = w;
  if (raw_out) raw_out[i] = w;
}

// ------------------------------------------------------------------------------------------------
// A8 normalizeWeight + A10 update + A9 genAliasTable over the whole population, one workgroup.
// The alias table is the Walker table of PCL's stack discipline (H from the front, L from the back,
// both popped highest-index-first), built in parallel from prefix sums: with D_i the running deficit
// of the L list and E_k the running excess of the H list, small l_i is paired with the first h_k whose
// E_k >= D_(i-1); h_k drops below 1 at the first i with D_i > E_k and is then paired with h_(k+1).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_ge(const double* a, uint32_t n, double x) {
  uint32_t lo = 0, hi = n;  // first index with a[idx] >= x
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] >= x) hi = mid; else lo = mid + 1;
  }
  return lo;
}
__device__ __forceinline__ uint32_t upper_bound_gt(const double* a, uint32_t n, double x) {
  uint32_t lo = 0, hi = n;  // first index with a[idx] > x
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] > x) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__global__ __launch_bounds__(PFT_POP_THREADS) void k_population(PftParams prm, PftDev d, uint32_t n, int do_norm,
                                                               int do_mean, int do_alias) {
  __shared__ double s_d[20];
  __shared__ uint32_t s_u[20];
  pft_particle* P = d.part_all;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;

  if (do_norm) {
    double wmin = DBL_MAX, wmax = -DBL_MAX;
    for (uint32_t i = tid; i < n; i += nt) {
      double w = (double)P[i].weight;
      if (wmin > w) wmin = w;
      if (w != 0.0 && wmax < w) wmax = w;
    }
    wmin = block_reduce<double>(wmin, s_d, OpMinD(), DBL_MAX);
    wmax = block_reduce<double>(wmax, s_d, OpMaxD(), -DBL_MAX);
    if (tid == 0) d.hdr->fit_ratio = wmin;
    if (wmax != wmin) {
      for (uint32_t i = tid; i < n; i += nt) {
        float wf = P[i].weight;
        if (wf != 0.0f) P[i].weight = (float)exp(1.0 - prm.alpha * ((double)wf - wmin) / (wmax - wmin));
      }
    } else {
      for (uint32_t i = tid; i < n; i += nt) P[i].weight = 1.0f / (float)n;
    }
    __threadfence_block();
    __syncthreads();
    double sum = 0.0;
    for (uint32_t i = tid; i < n; i += nt) sum += (double)P[i].weight;
    sum = block_reduce<double>(sum, s_d, OpAddD(), 0.0);
    if (sum != 0.0) {
      const float fs = (float)sum;
      for (uint32_t i = tid; i < n; i += nt) P[i].weight = P[i].weight / fs;
    } else {
      for (uint32_t i = tid; i < n; i += nt) P[i].weight = 1.0f / (float)n;
    }
    __threadfence_block();
    __syncthreads();
  }

  if (do_mean) {
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = tid; i < n; i += nt) {
      pft_particle q = P[i];
      double w = (double)q.weight;
      a[0] += (double)q.x * w; a[1] += (double)q.y * w; a[2] += (double)q.z * w;
      a[3] += (double)q.roll * w; a[4] += (double)q.pitch * w; a[5] += (double)q.yaw * w;
    }
    for (int k = 0; k < 6; k++) a[k] = block_reduce<double>(a[k], s_d, OpAddD(), 0.0);
    if (tid == 0) {
      pft_particle orig = d.hdr->rep, r;
      r.x = (float)a[0]; r.y = (float)a[1]; r.z = (float)a[2]; r.w = 1.0f;
      r.roll = (float)a[3]; r.pitch = (float)a[4]; r.yaw = (float)a[5];
      r.weight = 1.0f / (float)n;
      pft_particle m;
      m.x = r.x - orig.x; m.y = r.y - orig.y; m.z = r.z - orig.z; m.w = 1.0f;
      m.roll = r.roll - orig.roll; m.pitch = r.pitch - orig.pitch; m.yaw = r.yaw - orig.yaw;
      m.weight = 0.0f;
      d.hdr->rep = r;
      d.hdr->motion = m;
    }
    __syncthreads();
  }

  if (do_alias) {
    int32_t* A = d.alias_a;
    double* Q = d.alias_q;
    int32_t* Llist = d.alias_list;
    int32_t* Hlist = d.alias_list + n;
    double* Dp = d.alias_pref;      // inclusive running deficit over the L list
    double* Ep = d.alias_pref + n;  // inclusive running excess over the H list
    // thread t owns reversed positions [t*K, (t+1)*K): stacks pop the highest index first
    const uint32_t K = (n + nt - 1) / nt;
    const uint32_t r0 = tid * K, r1 = min(n, r0 + K);
    uint32_t cntL = 0, cntH = 0;
    double defs = 0.0, excs = 0.0;
    for (uint32_t r = r0; r < r1; r++) {
      uint32_t i = n - 1 - r;
      double q = (double)(P[i].weight * (float)n);  // float product widened to double
      Q[i] = q;
      A[i] = (int32_t)i;
      if (q < 1.0) {
        cntL++;
        defs += 1.0 - q;
      } else {
        cntH++;
        excs += q - 1.0;
      }
    }
    uint32_t totL, totH;
    double totD, totE;
    uint32_t offL = block_excl_scan<uint32_t>(cntL, s_u, &totL);
    uint32_t offH = block_excl_scan<uint32_t>(cntH, s_u, &totH);
    double offD = block_excl_scan<double>(defs, s_d, &totD);
    double offE = block_excl_scan<double>(excs, s_d, &totE);
    for (uint32_t r = r0; r < r1; r++) {
      uint32_t i = n - 1 - r;
      double q = Q[i];
      if (q < 1.0) {
        offD += 1.0 - q;
        Llist[offL] = (int32_t)i;
        Dp[offL] = offD;
        offL++;
      } else {
        offE += q - 1.0;
        Hlist[offH] = (int32_t)i;
        Ep[offH] = offE;
        offH++;
      }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t m = totL, nh = totH;
    if (m > 0 && nh > 0) {
      const double Dm = Dp[m - 1];
      for (uint32_t pos = tid; pos < m; pos += nt) {
        double dprev = pos > 0 ? Dp[pos - 1] : 0.0;
        uint32_t k = lower_bound_ge(Ep, nh, dprev);
        if (k < nh) A[Llist[pos]] = Hlist[k];
      }
      for (uint32_t k = tid; k < nh; k += nt) {
        const double Ek = Ep[k];
        const int32_t hk = Hlist[k];
        uint32_t is = upper_bound_gt(Dp, m, Ek);
        if (is < m) {  // dropped below 1 while absorbing l_is: becomes a small, paired with the next large
          Q[hk] = 1.0 + Ek - Dp[is];
          if (k + 1 < nh) A[hk] = Hlist[k + 1];
        } else {
          const double eprev = k > 0 ? Ep[k - 1] : -1.0;
          if (Dm > eprev) Q[hk] = 1.0 + Ek - Dm;  // the large that was current when L ran empty
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

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
void pftk_resample(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t epoch, pft_particle* out) {
  hipLaunchKernelGGL(k_resample, dim3(cdiv(p.P_local, 256)), dim3(256), 0, s, p, d.part_all, d.alias_a, d.alias_q,
                     d.hdr, epoch, out, d.mats);
}
void pftk_pose_to_matrix(hipStream_t s, const pft_particle* p, uint32_t n, float* mats) {
  if (!n) return;
  hipLaunchKernelGGL(k_pose_to_matrix, dim3(cdiv(n, 256)), dim3(256), 0, s, p, n, mats);
}
void pftk_aabb(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles) {
  hipLaunchKernelGGL(k_aabb, dim3(d.bbox_grid), dim3(1024), 0, s, d.ref_xyz, p.M, d.mats, n_particles, d.bbox_part);
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, d.bbox_part, d.bbox_grid, d.bbox6);
}
void pftk_crop(hipStream_t s, const PftParams& p, const PftDev& d) {
  uint32_t nb = cdiv(d.N ? d.N : 1, 1024);
  hipLaunchKernelGGL(k_crop_count, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.crop_counts);
  hipLaunchKernelGGL(k_crop_scatter, dim3(nb), dim3(1024), 0, s, d.in_pts, d.N, d.bbox6, d.crop_counts,
                     p.hsv_argorder, d.crop_pts, d.crop_idx, d.hdr);
}
void pftk_octree(hipStream_t s, const PftParams& p, const PftDev& d) {
  hipLaunchKernelGGL(k_octree_build, dim3(1), dim3(PFT_BUILD_THREADS), 0, s, p, d);
}

static int g_max_lds = -1;
int pftk_max_lds_bytes() {
  if (g_max_lds < 0) {
    int dev = 0, v = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 65536;
    g_max_lds = v;
  }
  return g_max_lds;
}

void pftk_likelihood(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, bool debug_nn,
                     int num_cus) {
  static bool attr_set = false;
  uint32_t lds = (uint32_t)pftk_max_lds_bytes();
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_likelihood<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  uint32_t items = n_particles * p.nchunk;
  uint32_t grid = (uint32_t)num_cus;
  uint32_t need = cdiv(items ? items : 1, PFT_LIK_THREADS / 64);
  if (grid > need) grid = need;
  if (debug_nn)
    hipLaunchKernelGGL(k_likelihood<true>, dim3(grid), dim3(PFT_LIK_THREADS), lds, s, p, d, n_particles, lds);
  else
    hipLaunchKernelGGL(k_likelihood<false>, dim3(grid), dim3(PFT_LIK_THREADS), lds, s, p, d, n_particles, lds);
}
void pftk_finalize_raw(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n_particles, float* raw_out) {
  if (!n_particles) return;
  hipLaunchKernelGGL(k_finalize_raw, dim3(cdiv(n_particles, 256)), dim3(256), 0, s, d.partial, p.nchunk, n_particles,
                     d.part_cur, raw_out);
}
void pftk_population(hipStream_t s, const PftParams& p, const PftDev& d, uint32_t n, int do_normalize, int do_mean,
                     int do_alias) {
  hipLaunchKernelGGL(k_population, dim3(1), dim3(PFT_POP_THREADS), 0, s, p, d, n, do_normalize, do_mean, do_alias);
}
