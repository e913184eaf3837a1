// Issue rate of the VALU instruction kinds the likelihood kernel's child selection is made of (gfx950), as wave64
// instructions per SIMD-cycle at 2.4 GHz: 8 independent chains per lane, 8 waves per SIMD, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
  float c = 1.0001f + threadIdx.x;
  unsigned long long sg = 0;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 1) { REP16(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 2) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
    if (MODE == 3) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
    if (MODE == 4) { REP16(asm volatile("v_cmp_lt_f32 %9, %0, %8\n v_cmp_lt_f32 %9, %1, %8\n v_cmp_lt_f32 %9, %2, %8\n v_cmp_lt_f32 %9, %3, %8\n v_cmp_lt_f32 %9, %4, %8\n v_cmp_lt_f32 %9, %5, %8\n v_cmp_lt_f32 %9, %6, %8\n v_cmp_lt_f32 %9, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "s"(sg));) }
    if (MODE == 5) { REP16(asm volatile("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 6) { REP16(asm volatile("v_bfe_u32 %0, %0, 1, 3\n v_bfe_u32 %1, %1, 1, 3\n v_bfe_u32 %2, %2, 1, 3\n v_bfe_u32 %3, %3, 1, 3\n v_bfe_u32 %4, %4, 1, 3\n v_bfe_u32 %5, %5, 1, 3\n v_bfe_u32 %6, %6, 1, 3\n v_bfe_u32 %7, %7, 1, 3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 7) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 8) { REP16(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 9) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "s"(sg));) }
    if (MODE == 10) { REP16(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
    if (MODE == 11) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
    if (MODE == 12) { REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 13) { REP16(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 14) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 15) { REP16(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 16) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 17) { REP16(asm volatile("v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 18) { REP16(asm volatile("v_cmp_ne_u32_e32 vcc, %0, %8\n v_cmp_ne_u32_e32 vcc, %1, %8\n v_cmp_ne_u32_e32 vcc, %2, %8\n v_cmp_ne_u32_e32 vcc, %3, %8\n v_cmp_ne_u32_e32 vcc, %4, %8\n v_cmp_ne_u32_e32 vcc, %5, %8\n v_cmp_ne_u32_e32 vcc, %6, %8\n v_cmp_ne_u32_e32 vcc, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
    if (MODE == 19) { REP16(asm volatile("v_cmp_ne_u32_e64 %9, %0, %8\n v_cmp_ne_u32_e64 %9, %1, %8\n v_cmp_ne_u32_e64 %9, %2, %8\n v_cmp_ne_u32_e64 %9, %3, %8\n v_cmp_ne_u32_e64 %9, %4, %8\n v_cmp_ne_u32_e64 %9, %5, %8\n v_cmp_ne_u32_e64 %9, %6, %8\n v_cmp_ne_u32_e64 %9, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "s"(sg));) }
    if (MODE == 20) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 21) { REP16(asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 22) { REP16(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 23) { REP16(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 24) { REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 25) { REP16(asm volatile("v_add_f32_e64 %0, %0, %8\n v_add_f32_e64 %1, %1, %8\n v_add_f32_e64 %2, %2, %8\n v_add_f32_e64 %3, %3, %8\n v_add_f32_e64 %4, %4, %8\n v_add_f32_e64 %5, %5, %8\n v_add_f32_e64 %6, %6, %8\n v_add_f32_e64 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (MODE == 26) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
    if (MODE == 27) { REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %1, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %3, %8\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %5, %8\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cmp_lt_f32 s[20:21], %7, %8\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "s20", "s21");) }
    if (MODE == 28) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* d) {
  const int iters = 1000, grid = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = (double)grid * 4 * iters * 16 * 8;  // 4 waves per block
  const double simd_cycles = ms * 1e-3 * 2.4e9 * 256 * 4;
  printf("%-22s %.3f ms  %.2f cycles per wave64 instruction per SIMD  (%.1f T lane-ops/s)\n", name, ms, simd_cycles / wave_insts,
         wave_insts * 64 / ms * 1e-9);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("v_add_f32", d);
  run<7>("v_fma_f32", d);
  run<1>("v_and_b32", d);
  run<8>("v_min_f32", d);
  run<5>("v_min3_f32", d);
  run<6>("v_bfe_u32", d);
  run<2>("v_cndmask_b32 vcc", d);
  run<9>("v_cndmask_b32 sgpr", d);
  run<3>("v_cmp_lt_f32 -> vcc", d);
  run<4>("v_cmp_lt_f32 -> sgpr", d);
  run<10>("v_cndmask_e32 vcc", d);
  run<11>("v_cndmask_e64 vcc", d);
  run<12>("v_mul_f32", d);
  run<13>("v_sub_f32", d);
  run<14>("v_lshlrev_b32", d);
  run<15>("v_or_b32", d);
  run<16>("v_lshl_or_b32", d);
  run<17>("v_and_or_b32", d);
  run<18>("v_cmp_ne_u32_e32 vcc", d);
  run<19>("v_cmp_ne_u32_e64 sgpr", d);
  run<20>("v_add_u32", d);
  run<21>("v_bcnt_u32_b32", d);
  run<22>("v_max_f32", d);
  run<23>("v_min_u32", d);
  run<24>("v_mov_b32", d);
  run<25>("v_add_f32_e64", d);
  run<26>("cmp->vcc + cndmask_e32 (x2)", d);
  run<27>("cmp->sgpr + cndmask_e64 (x2)", d);
  run<28>("cmp->vcc + cndmask_e64 (x2)", d);
  return 0;
}
