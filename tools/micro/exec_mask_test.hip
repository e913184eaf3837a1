// Does a wave64 VALU instruction get cheaper when part of the wave is masked off (gfx950)?  If the SIMD skipped the
// 16-lane passes whose lanes are all inactive, packing the lanes that are still in the generic descent into one quarter of
// the wave would pay; if not, a partially masked instruction costs what a full one costs and only whole-wave exits help.
// Measured: cycles per wave64 instruction per SIMD at 2.4 GHz, 8 independent chains per lane, 8 waves per SIMD, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long mask) {
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
  float c = 1.0001f + threadIdx.x;
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=&s"(saved) : "s"(mask));
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (OP == 1) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "s"(mask));) }
    if (OP == 2) { REP16(asm volatile("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (OP == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %9, %8\n v_fma_f32 %1, %1, %9, %8\n v_fma_f32 %2, %2, %9, %8\n v_fma_f32 %3, %3, %9, %8\n v_fma_f32 %4, %4, %9, %8\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %9, %8\n v_fma_f32 %7, %7, %9, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "v"(0.5f));) }
    if (OP == 4) { REP16(asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31\n v_bfe_u32 %4, %4, 1, 31\n v_bfe_u32 %5, %5, 1, 31\n v_bfe_u32 %6, %6, 1, 31\n v_bfe_u32 %7, %7, 1, 31" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));) }
    if (OP == 5) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");) }
  }
  asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, float* d, unsigned long long mask, const char* mname) {
  const int iters = 1000, grid = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, mask);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = 8;
  const double wave_insts = (double)grid * 4 * iters * 16 * per;
  const double simd_cycles = ms * 1e-3 * 2.4e9 * 256 * 4;
  printf("%-14s exec = %-28s %.3f ms  %.2f cycles per wave64 instruction per SIMD\n", name, mname, ms, simd_cycles / wave_insts);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  const struct { unsigned long long m; const char* n; } masks[] = {
      {~0ull, "all 64 lanes"}, {0xffffffffull, "lanes 0-31"}, {0xffffull, "lanes 0-15"}, {0xffff0000ffffull, "lanes 0-15 + 32-47"},
      {0x1111111111111111ull, "every 4th lane (16)"}, {0xfffull, "lanes 0-11"}, {0xffull, "lanes 0-7"}, {0xff00ull, "lanes 8-15"},
      {0x0101010101010101ull, "every 8th lane (8)"}, {0xfull, "lanes 0-3"}, {0x1ull, "lane 0"}, {0x8000000000000000ull, "lane 63"}};
  for (auto& mk : masks) {
    run<0>("v_add_f32", d, mk.m, mk.n);
    run<1>("v_cndmask_b32", d, mk.m, mk.n);
    run<2>("v_min3_f32", d, mk.m, mk.n);
    run<3>("v_fma_f32", d, mk.m, mk.n);
    run<4>("v_bfe_u32", d, mk.m, mk.n);
    run<5>("v_cmp_lt_f32", d, mk.m, mk.n);
  }
  return 0;
}
