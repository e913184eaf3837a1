// Follow-up to exec_mask_test.hip: is the 4-5 x cost of a wave64 select-class instruction with <= 8 active lanes a
// property of the instruction (then a SIMD that mixes sparse and full waves pays the average), or does it only show
// when EVERY wave on the SIMD is sparse?  Waves alternate between a full mask and a sparse one.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long sparse_mask, int sparse_every) {
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
  float c = 1.0001f + threadIdx.x;
  const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const unsigned long long mask_v = (sparse_every > 0 && wave % sparse_every == 0) ? sparse_mask : ~0ull;
  const unsigned long long mask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mask_v >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)mask_v);
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=&s"(saved) : "s"(mask));
  for (int it = 0; it < iters; it++) {
    REP16(asm volatile("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));)
  }
  asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void run(const char* name, float* d, unsigned long long m, int every) {
  const int iters = 1000, grid = 256 * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, m, every);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, m, every);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = (double)grid * 4 * iters * 16 * 8;
  const double simd_cycles = ms * 1e-3 * 2.4e9 * 256 * 4;
  printf("v_min3_f32  %-44s %.3f ms  %.2f cycles per wave64 instruction per SIMD\n", name, ms, simd_cycles / wave_insts);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run("all waves full", d, ~0ull, 0);
  run("all waves 4 lanes", d, 0xfull, 1);
  run("every 2nd wave 4 lanes, the others full", d, 0xfull, 2);
  run("every 4th wave 4 lanes, the others full", d, 0xfull, 4);
  run("every 8th wave 4 lanes, the others full", d, 0xfull, 8);
  run("every 2nd wave 12 lanes, the others full", d, 0xfffull, 2);
  return 0;
}
