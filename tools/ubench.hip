// Micro-benchmarks on gfx950: shader clock under light load, s_barrier round trip, glds issue cost, MFMA issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void clk_kernel(long long* out, int spin) {
  long long c0 = clock64(), w0 = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}
__global__ void barrier_kernel(long long* out, int n) {
  long long c0 = clock64();
  for (int i = 0; i < n; i++) __builtin_amdgcn_s_barrier();
  long long c1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
}
__global__ void mfma_kernel(long long* out, int n, float* sink) {
  f32x4 acc[8];
  for (int j = 0; j < 8; j++) acc[j] = (f32x4){0, 0, 0, 0};
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e); }
  long long c0 = clock64();
  for (int i = 0; i < n; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  long long c1 = clock64();
  float s = 0; for (int j = 0; j < 8; j++) s += acc[j][0];
  if (s == 1234.5f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
}
// n rounds of `per` glds pieces (1 KiB each per wave) from an L2-resident buffer, then wait
__global__ void glds_kernel(long long* out, const char* src, int n, int per) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* s = src + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  long long c0 = clock64();
  for (int i = 0; i < n; i++) {
    for (int r = 0; r < per; r++)
      __builtin_amdgcn_global_load_lds((const void*)(s + (size_t)((i * per + r) & 63) * 65536),
          (__attribute__((address_space(3))) void*)(lds + (r * (blockDim.x >> 6) + wave) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long long c1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
  (void)lane;
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[8]; float* sink; hipMalloc(&sink, 64);
  char* buf; hipMalloc(&buf, (size_t)64 << 20); hipMemset(buf, 1, (size_t)64 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  for (int grid : {1, 256, 1024}) {
    hipLaunchKernelGGL(clk_kernel, dim3(grid), dim3(256), 0, 0, d, 200000);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("clock: grid %4d  clock64 ticks %lld  wall(100MHz) ticks %lld -> clock64 runs at %.0f MHz\n", grid, h[0], h[1], 100.0 * h[0] / h[1]);
  }
  for (int th : {256, 512, 1024}) {
    hipLaunchKernelGGL(barrier_kernel, dim3(256), dim3(th), 0, 0, d, 10000);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("s_barrier: %4d threads: %.1f ticks per barrier\n", th, h[0] / 10000.0);
  }
  for (int th : {256, 512}) {
    hipEventRecord(e0); hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(th), 0, 0, d, 20000, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma 16x16x32 bf16: %d threads: %.2f ticks per MFMA per wave; kernel %.1f us -> %.0f TF chip\n", th, h[0] / 160000.0, ms * 1e3,
           256.0 * (th / 64) * 160000 * 16384 / (ms * 1e-3) / 1e12);
  }
  for (int th : {256, 512})
    for (int per : {1, 3, 6}) {
      hipLaunchKernelGGL(glds_kernel, dim3(256), dim3(th), 65536, 0, d, buf, 2000, per);
      hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
      printf("glds: %d threads, %d pieces/wave/round (wait each round): %.0f ticks per round = %.1f B/tick/CU\n", th, per, h[0] / 2000.0,
             (double)per * th * 16 / (h[0] / 2000.0));
    }
  return 0;
}
