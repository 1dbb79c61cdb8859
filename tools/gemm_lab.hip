// Stand-alone lab for the bf16 glds GEMM (igemm2.hip compiled with its lab switches): per-shape timing of tile
// configurations and of the kernel without its compute / staging loads / stores, each checked against a naive kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DAVLEN_G2_LAB tools/gemm_lab.hip -o build/gemm_lab && build/gemm_lab [abl]
#include "../avlen_amd/csrc/igemm2.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <array>

__global__ void naive_kernel(const bf16* A, const bf16* B, const float* bias, float* C, int M, int N, int K) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; k++) s += (float)A[(long)m * K + k] * (float)B[(long)n * K + k];
  C[(long)m * N + n] = s + bias[n];
}
__global__ void diff_kernel(const float* ref, const bf16* c, long n, float* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = fabsf(ref[i] - (float)c[i]);
  atomicMax((int*)out, __float_as_int(d));
  atomicMax((int*)out + 1, __float_as_int(fabsf(ref[i])));
}

static float time_us(int M, int N, int K, const void* A, const void* B, void* C16, const float* bias, void* ws, size_t wsb, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) avlen_gemm_bf16(A, K, B, K, nullptr, N, C16, N, bias, nullptr, 0, M, N, K, 0, ws, wsb, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; i++) avlen_gemm_bf16(A, K, B, K, nullptr, N, C16, N, bias, nullptr, 0, M, N, K, 0, ws, wsb, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

template <int BM, int BN, int WM, int WN, int NS, int NTH>
static void occ(const char* name) {
  size_t lds = (size_t)NS * (BM * 128 + (BN * 8 >= NTH ? BN * 128 : NTH * 16));
  hipFuncSetAttribute(reinterpret_cast<const void*>(&g2_kernel<BM, BN, WM, WN, NS, NTH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nb = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&g2_kernel<BM, BN, WM, WN, NS, NTH, false>), NTH, lds);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&g2_kernel<BM, BN, WM, WN, NS, NTH, false>));
  printf("occupancy %s: %d blocks/CU (dyn LDS %zu B, %d VGPR+AGPR regs, static LDS %zu)\n", name, nb, lds, fa.numRegs, fa.sharedSizeBytes);
}

int main(int argc, char** argv) {
  const bool ablate = argc > 1 && !strcmp(argv[1], "abl");
  if (argc > 1 && !strcmp(argv[1], "occ")) {
    occ<64, 128, 2, 4, 2, 512>("64x128 ns2 t512"); occ<64, 128, 2, 4, 4, 512>("64x128 ns4 t512");
    occ<128, 128, 2, 4, 2, 512>("128x128 ns2 t512"); occ<256, 128, 4, 2, 3, 512>("256x128 ns3 t512");
    return 0;
  }
  const bool clip = argc > 1 && !strcmp(argv[1], "clip");
  const int shapes_all[][3] = {{64, 512, 512}, {64, 2048, 512}, {64, 512, 2048}, {2464, 512, 512}, {2464, 512, 2048}, {2464, 2048, 512}, {2464, 1536, 512},
                           {9664, 256, 256}, {9664, 768, 256}, {4800, 64, 8192}, {8192, 8192, 1024}, {8192, 8192, 8192}};
  const int shapes_clip[][3] = {{2464, 512, 512}, {2464, 512, 2048}, {2464, 2048, 512}, {2464, 1536, 512}};
  std::vector<std::array<int, 3>> shapes;
  if (clip) for (auto& s_ : shapes_clip) shapes.push_back({s_[0], s_[1], s_[2]});
  else for (auto& s_ : shapes_all) shapes.push_back({s_[0], s_[1], s_[2]});
  size_t maxA = (size_t)9664 * 8192, maxB = (size_t)8192 * 8192, maxC = (size_t)8192 * 8192;
  std::vector<unsigned short> h(maxB);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff) + ((s >> 31) << 15)); }
  void *A, *B, *C, *ws; float *bias, *ref, *err;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C, maxC * 2); hipMalloc(&bias, 8192 * 4);
  hipMalloc(&ref, (size_t)9664 * 2048 * 4); hipMalloc(&err, 8);
  size_t wsb = (size_t)256 << 20; hipMalloc(&ws, wsb);
  hipMemcpy(A, h.data(), maxB * 2, hipMemcpyHostToDevice);
  hipMemcpy((char*)A + maxB * 2, h.data(), (maxA - maxB) * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data() + 777, (maxB - 777) * 2, hipMemcpyHostToDevice);
  std::vector<float> hb(8192); for (int i = 0; i < 8192; i++) hb[i] = 0.01f * (i % 97);
  hipMemcpy(bias, hb.data(), 8192 * 4, hipMemcpyHostToDevice);
  // bm, bn, threads, stages, splitk
  const int cfgs_all[][5] = {{0, 0, 0, 0, 0}, {64, 128, 256, 4, 0}, {64, 128, 512, 2, 0}, {64, 128, 512, 3, 0}, {64, 128, 512, 4, 0},
                         {128, 128, 256, 2, 0}, {128, 128, 512, 2, 0}, {128, 128, 512, 3, 0}, {128, 128, 512, 4, 0},
                         {256, 128, 512, 2, 0}, {256, 128, 512, 3, 0}, {128, 64, 512, 3, 0}, {64, 128, 512, 3, 2}, {64, 64, 256, 0, 0}};
  const int cfgs_abl[][5] = {{64, 128, 512, 2, 0}, {64, 128, 512, 4, 0}, {128, 128, 512, 2, 0}, {256, 128, 512, 3, 0}};
  const int cfgs_clip[][5] = {{0, 0, 0, 0, 0}, {64, 128, 512, 2, 0}, {64, 256, 512, 2, 0}, {64, 256, 512, 3, 0}, {128, 256, 512, 2, 0},
                              {128, 256, 512, 3, 0}, {128, 128, 256, 2, 0}, {64, 64, 256, 0, 0}};
  const int (*cfgs)[5] = ablate ? cfgs_abl : clip ? cfgs_clip : cfgs_all;
  const int ncfg = ablate ? 4 : clip ? 8 : 14;
  for (auto& sh : shapes) {
    int M = sh[0], N = sh[1], K = sh[2];
    bool big = (double)M * N * K > 1e11;
    bool check = (size_t)M * N <= (size_t)9664 * 2048;
    if (check) hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, (const bf16*)A, (const bf16*)B, bias, ref, M, N, K);
    printf("== M %d N %d K %d\n", M, N, K);
    for (int ci = 0; ci < ncfg; ci++) {
      const int* c = cfgs[ci];
      if (c[1] == 64 && N > 64 && N % 64) continue;
      for (int i = 0; i < 5; i++) g_lab_cfg[i] = c[i];
      const int abl[] = {0, 1, 2, 4, 3, 7};
      printf("  cfg %3d x %3d  t%3d ns%d sk%d :", c[0], c[1], c[2], c[3], c[4]);
      for (int a : abl) {
        if (a && !ablate) break;
        g_lab_ablate = a;
        hipMemset(C, 0, (size_t)M * N * 2);
        float t = time_us(M, N, K, A, B, C, bias, ws, wsb, big ? 5 : 40);
        if (a == 0) {
          printf(" %8.1f us %7.1f TF", t, 2.0 * M * N * K / t / 1e6);
          if (check) {
            hipMemset(err, 0, 8);
            long n = (long)M * N;
            hipLaunchKernelGGL(diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ref, (const bf16*)C, n, err);
            float he[2]; hipMemcpy(he, err, 8, hipMemcpyDeviceToHost);
            printf("  err %.1e%s", he[0] / he[1], he[0] / he[1] > 6e-3 ? " WRONG" : "");
          }
        } else printf("  abl%d %8.1f", a, t);
      }
      printf("\n"); fflush(stdout);
    }
  }
  return 0;
}
