// Issue rate of packed FP32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) against the scalar forms on gfx950:
// the same number of floating-point operations as independent dependency chains, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8];
  v2f p[4];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 4; i++) p[i] = v2f{a[2 * i], a[2 * i + 1]};
  const float c = 1.0001f;
  const v2f pc = {c, c};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      } else if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
      } else if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* d) {
  const int iters = 2000, grid = 256 * 8;
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
  const double flop_lane = (double)grid * 256 * iters * 16 * 8 * ((MODE == 2 || MODE == 3) ? 2 : 1);
  printf("%-14s %.3f ms  %.1f TFLOP/s\n", name, ms, flop_lane / ms * 1e-9);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("v_add_f32", d);
  run<1>("v_pk_add_f32", d);
  run<4>("v_mul_f32", d);
  run<5>("v_pk_mul_f32", d);
  run<2>("v_fma_f32", d);
  run<3>("v_pk_fma_f32", d);
  return 0;
}
