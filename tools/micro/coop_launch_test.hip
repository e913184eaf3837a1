// What does hipLaunchCooperativeKernel cost against a plain launch for a small kernel in a chain of dependent launches
// on one stream (the population kernel's device-scope barriers assume co-resident workgroups, which only a
// cooperative launch guarantees)?  32 workgroups of 256 threads, as k_population at 8 192 particles; alternating with a
// plain kernel as in a frame.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_small(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }

int main() {
  float* d;
  (void)hipMalloc(&d, 32 * 256 * sizeof(float));
  (void)hipMemset(d, 0, 32 * 256 * sizeof(float));
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int N = 2000;
  void* args[] = {&d};
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      (void)hipEventRecord(e0, s);
      for (int i = 0; i < N; i++) {
        if (mode == 0 || (mode == 2 && (i & 1)))
          hipLaunchKernelGGL(k_small, dim3(32), dim3(256), 0, s, d);
        else if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_small), dim3(32), dim3(256), args, 0, s) != hipSuccess) {
          printf("cooperative launch failed\n");
          return 1;
        }
      }
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1)
        printf("%-46s %.2f us per launch\n", mode == 0 ? "plain launches" : mode == 1 ? "cooperative launches" : "alternating cooperative / plain",
               ms * 1e3 / N);
    }
  }
  return 0;
}
