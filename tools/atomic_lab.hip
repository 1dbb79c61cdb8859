// How long do the fp32 atomic accumulations of a hidden-split fused MLP take?  312 workgroups (39 row tiles x 8 hidden chunks),
// each adds a 64 x 512 fp32 tile into its row tile of X (8 workgroups per tile), vs. plain stores of the same tile.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_lab.hip -o tools/bin/atomic_lab && tools/bin/atomic_lab
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void acc(float* X, int tiles, int splits) {
  const int t = blockIdx.x % tiles, j = blockIdx.x / tiles;
  float* x = X + (size_t)t * 64 * 512;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // MFMA-like ownership: wave -> 64 columns, lane -> (row r16 + 16 m, 4 consecutive columns)
  for (int m = 0; m < 4; m++)
    for (int n = 0; n < 4; n++) {
      const int row = m * 16 + (lane & 15), col = wave * 64 + n * 16 + (lane >> 4) * 4;
      float* p = x + row * 512 + col;
      for (int r = 0; r < 4; r++) {
        const float v = (float)(j + r) * 0.001f;
        if (MODE == 0) __hip_atomic_fetch_add(p + r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) unsafeAtomicAdd(p + r, v);
        else p[r] = v;
      }
    }
}
int main() {
  const int tiles = 39, splits = 8;
  float* X; hipMalloc(&X, (size_t)tiles * 64 * 512 * 4); hipMemset(X, 0, (size_t)tiles * 64 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"atomic add (agent scope)", "unsafeAtomicAdd", "plain store"};
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; it++) {
        if (mode == 0) hipLaunchKernelGGL(acc<0>, dim3(tiles * splits), dim3(512), 0, 0, X, tiles, splits);
        else if (mode == 1) hipLaunchKernelGGL(acc<1>, dim3(tiles * splits), dim3(512), 0, 0, X, tiles, splits);
        else hipLaunchKernelGGL(acc<2>, dim3(tiles * splits), dim3(512), 0, 0, X, tiles, splits);
      }
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-26s %.2f us per launch (%d workgroups, %.1f MB)\n", names[mode], best * 100, tiles * splits, tiles * splits * 64 * 512 * 4 / 1e6);
  }
  return 0;
}
