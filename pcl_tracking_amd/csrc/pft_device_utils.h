// pft_device_utils.h -- wave64 / workgroup primitives and the small per-element routines shared by the
// kernels (gfx950).  All float arithmetic that PCL does unfused stays unfused (-ffp-contract=off).
#pragma once
#include <float.h>
#include <math.h>

#include "pft_internal.h"

#define WAVE 64
// in fully unrolled per-thread loops: stop the scheduler from hoisting every iteration's loads to the top
// (K x float4 in flight overflows the 128-VGPR budget of a 1024-thread workgroup and spills to scratch)
#define UNROLL_FENCE(j, every) do { if (((j) % (every)) == (every) - 1) __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Sum of a double over the 64 lanes with DPP moves instead of ds_bpermute shuffles: 6 steps of two v_mov_dpp and one
// v_add_f64 (the shuffle form is 7 instructions per step).  Fixed order: inside the quads, inside the rows of 16, then
// row 0 into row 1 / row 2 into row 3 (row_bcast15), then rows 0+1 into rows 2, 3 (row_bcast31).  The total is valid in
// lane 63 ONLY (returned there; other lanes hold partial sums).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo));
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v += dpp_move_f64<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_move_f64<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_move_f64<0x124, 0xf>(v);  // row_ror 4
  v += dpp_move_f64<0x128, 0xf>(v);  // row_ror 8: every lane of a row now holds the row's sum
  v += dpp_move_f64<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3 (lanes of rows 0, 2 add the old value 0)
  v += dpp_move_f64<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
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

// ---- A7b RGB2HSV (PCL 1.8.0 tracking/impl/hsv_color_coherence.hpp): integer, once per point ----
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

// ---- RNG: Philox4x32-10 keyed by the seed, counter = (global particle id, slot, epoch, purpose) ----
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

// Correctly rounded float square root for DistanceCoherence's Vector4f::norm(): the hardware estimate (1 ulp) and the
// round-to-nearest correction of the compiler's own sqrtf expansion, WITHOUT that expansion's rescaling of arguments
// below 2^-96 (six instructions in the likelihood's inner loop): v_sqrt_f32 flushes such an argument to 0, and the only
// consumer here is 1.0 + d*d*w in double, which is 1.0 for every d below 2^-27 / sqrt(w) whatever its last bits.
__device__ __forceinline__ float sqrt_rn_coherence(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float ed = fmaf(-sd, s, x), eu = fmaf(-su, s, x);
  s = ed <= 0.0f ? sd : s;
  s = eu > 0.0f ? su : s;
  return s;
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

__device__ __forceinline__ void load_matrix(const float* mats, uint32_t i, float* T) {
  const float4* tp = reinterpret_cast<const float4*>(mats + 12 * (size_t)i);
  float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
  T[0] = r0.x; T[1] = r0.y; T[2] = r0.z; T[3] = r0.w;
  T[4] = r1.x; T[5] = r1.y; T[6] = r1.z; T[7] = r1.w;
  T[8] = r2.x; T[9] = r2.y; T[10] = r2.z; T[11] = r2.w;
}

// A2  pcl::transformPointCloud (common/impl/transforms.hpp): ((T0*x + T1*y) + T2*z) + T3, float, unfused.
__device__ __forceinline__ void xform(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = T[0] * x + T[1] * y + T[2] * z + T[3];
  oy = T[4] * x + T[5] * y + T[6] * z + T[7];
  oz = T[8] * x + T[9] * y + T[10] * z + T[11];
}

// ---- on-demand entry of PCL's Walker alias table (A9), from the prefix-sum form built by k_population ----
struct AliasView {
  const int32_t* L;   // small list (q < 1), highest index first
  const int32_t* H;   // large list (q >= 1), highest index first
  const double* D;    // inclusive running deficit over L
  const double* E;    // inclusive running excess over H
  const uint32_t* pos;  // per particle: position in its list | large << 31
  uint32_t m, nh, n;
  // optional coarse levels of D and E (every sD-th / sE-th element, in LDS): a search then needs log2(s) dependent
  // global loads instead of log2(n)
  const double* cD = nullptr;
  const double* cE = nullptr;
  uint32_t sD = 0, sE = 0;
};

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

// the same searches through a coarse level c[j] = a[min((j + 1) * s, n) - 1] (the arrays are non-decreasing)
__device__ __forceinline__ uint32_t lower_bound_ge2(const double* a, uint32_t n, double x, const double* c, uint32_t s) {
  if (!c || n == 0) return lower_bound_ge(a, n, x);
  uint32_t lo = 0, hi = (n + s - 1) / s;
  const uint32_t nb = hi;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (c[mid] >= x) hi = mid; else lo = mid + 1;
  }
  if (lo == nb) return n;
  uint32_t b = lo * s, e = min(n, b + s);
  while (b < e) {
    uint32_t mid = (b + e) >> 1;
    if (a[mid] >= x) e = mid; else b = mid + 1;
  }
  return b;
}
__device__ __forceinline__ uint32_t upper_bound_gt2(const double* a, uint32_t n, double x, const double* c, uint32_t s) {
  if (!c || n == 0) return upper_bound_gt(a, n, x);
  uint32_t lo = 0, hi = (n + s - 1) / s;
  const uint32_t nb = hi;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (c[mid] > x) hi = mid; else lo = mid + 1;
  }
  if (lo == nb) return n;
  uint32_t b = lo * s, e = min(n, b + s);
  while (b < e) {
    uint32_t mid = (b + e) >> 1;
    if (a[mid] > x) e = mid; else b = mid + 1;
  }
  return b;
}

// q[k] of the table; *flipped_next receives the alias of a large that dropped below 1 (or -1)
__device__ __forceinline__ double alias_q(const AliasView& v, uint32_t k, float wk, int32_t* alias_if_large) {
  double q0 = (double)(wk * (float)v.n);  // float product widened to double, as genAliasTable does
  *alias_if_large = (int32_t)k;
  const uint32_t pp = v.pos[k];
  if (!(pp >> 31) || v.m == 0) return q0;  // smalls keep their q; without smalls nothing is paired
  const uint32_t pos = pp & 0x7fffffffu;
  const double Ek = v.E[pos];
  const uint32_t is = upper_bound_gt2(v.D, v.m, Ek, v.cD, v.sD);
  if (is < v.m) {  // dropped below 1 while absorbing l_is: becomes a small, paired with the next large
    if (pos + 1 < v.nh) *alias_if_large = v.H[pos + 1];
    return 1.0 + Ek - v.D[is];
  }
  const double eprev = pos > 0 ? v.E[pos - 1] : -1.0;
  const double Dm = v.D[v.m - 1];
  if (Dm > eprev) return 1.0 + Ek - Dm;  // the large that was current when L ran empty
  return q0;
}

// a[k] for a small k: the first large whose running excess covers the deficit accumulated before k
__device__ __forceinline__ int32_t alias_a_small(const AliasView& v, uint32_t k) {
  if (v.nh == 0) return (int32_t)k;
  const uint32_t pos = v.pos[k] & 0x7fffffffu;
  const double dprev = pos > 0 ? v.D[pos - 1] : 0.0;
  const uint32_t kk = lower_bound_ge2(v.E, v.nh, dprev, v.cE, v.sE);
  return kk < v.nh ? v.H[kk] : (int32_t)k;
}
