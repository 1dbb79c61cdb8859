// What does ONE dependent kernel cost in a HIP graph chain on this device?  60-node chains of (a) empty kernels, (b) kernels that
// read and write 2 MB (so that every boundary has dirty L2 lines to write back), 1 / 256 / 1024 workgroups; graph replay vs
// plain stream launches.
//   hipcc --offload-arch=gfx950 -O3 tools/chain_floor_lab.hip -o tools/bin/chain_floor_lab && tools/bin/chain_floor_lab
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void touch_k(const float* __restrict__ a, float* __restrict__ b, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i] + 1.f;
}
int main() {
  const int N = 60, n = 512 * 1024;
  float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMemset(a, 0, n * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++)
    for (int wgs : {1, 256, 1024}) {
      auto chain = [&]() {
        for (int i = 0; i < N; i++) {
          if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(wgs), dim3(256), 0, st, a);
          else hipLaunchKernelGGL(touch_k, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
        }
      };
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int w = 0; w < 3; w++) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int r = 0; r < 20; r++) hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipStreamSynchronize(st);
      float msg; hipEventElapsedTime(&msg, e0, e1);
      chain(); hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int r = 0; r < 20; r++) chain();
      hipEventRecord(e1, st); hipStreamSynchronize(st);
      float mss; hipEventElapsedTime(&mss, e0, e1);
      printf("%-28s %4d workgroups: %.2f us per node in a graph, %.2f us per launch on a stream\n",
             mode ? "2 MB read + 2 MB write" : "empty kernel", wgs, msg * 1000 / 20 / N, mss * 1000 / 20 / N);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  return 0;
}
