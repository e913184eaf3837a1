// Cost of a chain of short dependent kernels: plain stream launches against the same chain captured in a hipGraph.
// (decides whether the 15 launches of a frame are worth capturing)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_small(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ void k_one_wg(float* p, int iters) {  // one workgroup, ~tens of us: like the octree build
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; i++) v = v * 1.0001f + 1.0f;
  p[threadIdx.x] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  float* d;
  const int n = 8192;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0, n * sizeof(float)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int chain = 15, reps = 400;
  for (int grid : {1, 32, 256}) {
    auto enqueue = [&]() {
      for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d, n);
    };
    for (int i = 0; i < 20; i++) enqueue();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) enqueue();
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("grid %3d stream launches: %.2f us per kernel\n", grid, us / (reps * chain));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("grid %3d graph launches:  %.2f us per kernel\n", grid, us / (reps * chain));
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  return 0;
}
