// Latency of the product's bf16 GEMM kernel as a NODE OF A DEPENDENT GRAPH CHAIN (no host launch cost in the number): the four CLIP
// shapes at the live row count of a benched step (M = 2464) and at M = 320, with the ablation switches of igemm2.hip (1: no
// fragment reads / MFMA, 2: no staging loads, 4: no C stores).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DAVLEN_G2_LAB tools/gemm_graph_lab.hip -o tools/bin/gemm_graph_lab
#include "../avlen_amd/csrc/igemm2.hip"
#include <stdio.h>
#include <vector>
#include <stdlib.h>

int main(int argc, char** argv) {
  const int max_abl = argc > 1 ? atoi(argv[1]) : 0;
  const int ns_override = argc > 2 ? atoi(argv[2]) : 0;      // stage count for every shape (0: the dispatcher's choice)
  const int nth_override = argc > 3 ? atoi(argv[3]) : 512;   // 256: the 4-wave instance (32x64 wave tiles)
  const int split_override = argc > 4 ? atoi(argv[4]) : 0;   // split-K factor (GEMM + reduce kernel per node pair)
  g_lab_cfg[4] = split_override;
  if (ns_override) { g_lab_cfg[0] = 64; g_lab_cfg[1] = 128; g_lab_cfg[2] = nth_override; g_lab_cfg[3] = ns_override; }
  const int shapes[][3] = {{2464, 1536, 512}, {2464, 512, 512}, {2464, 2048, 512}, {2464, 512, 2048},
                           {320, 1536, 512}, {320, 512, 512}, {320, 2048, 512}, {320, 512, 2048}};
  const char* names[4] = {"in_proj", "out_proj", "c_fc", "c_proj"};
  const size_t maxA = (size_t)2464 * 2048, maxB = (size_t)2048 * 2048, maxC = (size_t)2464 * 2048;
  std::vector<unsigned short> h(maxA > maxB ? maxA : maxB);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0xff));       // bf16 near 0.0078..0.0156
  void *A, *B, *C16, *ws; float *C32, *bias;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C16, maxC * 2); hipMalloc(&C32, maxC * 4); hipMalloc(&bias, 4096 * 4);
  const size_t wsb = 64 << 20; hipMalloc(&ws, wsb);
  hipMemcpy(A, h.data(), maxA * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), maxB * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, 4096 * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int NODES = 24;
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    const int which = N == 1536 ? 0 : (N == 512 && K == 512) ? 1 : N == 2048 ? 2 : 3;
    printf("%-8s M=%4d N=%4d K=%4d:", names[which], M, N, K); fflush(stdout);
    for (int abl : {0, 1, 2, 3, 7}) {
      if (abl > max_abl) break;
      g_lab_ablate = abl;
      hipGraph_t g; hipGraphExec_t ge;
      avlen_gemm_bf16(A, K, B, K, nullptr, N, C16, N, bias, nullptr, 0, M, N, K, 0, ws, wsb, st);      // attribute / first-use set-up outside capture
      hipStreamSynchronize(st);
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < NODES; i++) avlen_gemm_bf16(A, K, B, K, nullptr, N, C16, N, bias, nullptr, 0, M, N, K, 0, ws, wsb, st);
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int w = 0; w < 2; w++) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int r = 0; r < 10; r++) hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("  abl %d: %5.2f us", abl, ms * 1000 / 10 / NODES); fflush(stdout);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    printf("\n");
  }
  return 0;
}
