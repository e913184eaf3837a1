// micro-benchmark: does a 1-workgroup kernel's duration depend on its dynamic LDS request / block size?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(unsigned* out) {
  extern __shared__ unsigned sm[];
  if (threadIdx.x == 0) out[0] = sm[0];
}
__global__ __launch_bounds__(1024) void k_touch(unsigned* out, unsigned words) {
  extern __shared__ unsigned sm[];
  for (unsigned i = threadIdx.x; i < words; i += blockDim.x) sm[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sm[words - 1];
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64);
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned sizes[] = {0, 16 * 1024, 64 * 1024, 65 * 1024, 128 * 1024, 156 * 1024};
  for (unsigned threads : {64u, 1024u})
    for (unsigned s : sizes) {
      for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(threads), s, 0, d);
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int w = 0; w < 200; w++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(threads), s, 0, d);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("empty threads %4u lds %6u B: %.2f us per launch (back to back)\n", threads, s, ms * 1000 / 200);
    }
  for (unsigned s : {64u * 1024, 156u * 1024}) {
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int w = 0; w < 200; w++) hipLaunchKernelGGL(k_touch, dim3(1), dim3(1024), s, 0, d, s / 4);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("touch threads 1024 lds %6u B: %.2f us per launch\n", s, ms * 1000 / 200);
  }
  return 0;
}
