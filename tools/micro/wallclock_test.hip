// micro-benchmark: rate of wall_clock64() against HIP event time
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  out[0] = wall_clock64() - t0;
}
int main() {
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("hipDeviceAttributeWallClockRate = %d kHz\n", rate);
  unsigned long long* d;
  hipMalloc(&d, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (unsigned long long t : {1000ull, 10000ull, 100000ull}) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, t, d);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, t, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("spin %llu ticks: %.2f us by events -> %.1f ticks/us\n", t, ms * 1000, t / (ms * 1000));
  }
  return 0;
}
