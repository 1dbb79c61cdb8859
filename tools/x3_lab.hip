// Stand-alone lab for csrc/tower_x3.hip's rest_x3_kernel (layers 2-4 of a tower in compensated bf16): launches the kernel on
// random data (B images x G towers) and prints the launch time and the mean duration of each phase.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off -DAVLEN_X3_PROF=0 tools/x3_lab.hip -o tools/bin/x3_lab
//   tools/bin/x3_lab [B=64] [G=6]
#include "../avlen_amd/csrc/tower_x3.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int avlen_zero_bytes(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? 0 : 2; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, G = argc > 2 ? atoi(argv[2]) : 6;
  RestArgs a = {};
  srand(3);
  auto dev_bf = [&](size_t n, float scale) {
    std::vector<unsigned short> h(n);
    for (auto& v : h) { float f = ((rand() % 2001) / 1000.f - 1.f) * scale; union { float ff; unsigned uu; } cv; cv.ff = f; v = (unsigned short)(cv.uu >> 16); }
    void* d; hipMalloc(&d, n * 2); hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return (bf16*)d;
  };
  auto dev_f = [&](size_t n, float lo, float hi) {
    std::vector<float> h(n);
    for (auto& v : h) v = lo + (hi - lo) * (rand() % 10001) / 10000.f;
    void* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return (float*)d;
  };
  static const size_t WS[15] = {32 * 16, 32 * 144, 32 * 288, 32 * 288, 32 * 288, 64 * 32, 64 * 288, 64 * 576, 64 * 576, 64 * 576,
                                128 * 64, 128 * 576, 128 * 1152, 128 * 1152, 128 * 1152};
  a.y_lo = (long)B * 8192;
  for (int g = 0; g < G; g++) {
    RestTower& t = a.t[g];
    t.x = dev_f((size_t)B * 4096 * 16, -1.f, 1.f); t.r = dev_f((size_t)B * 4096 * 16, 0.f, 1.f);
    std::vector<float> st((size_t)B * NBAND * 32);
    for (int b = 0; b < B * NBAND; b++) for (int c = 0; c < 16; c++) { st[b * 32 + c] = 0.f; st[b * 32 + 16 + c] = 4096.f / 3 / NBAND; }
    float* dst; hipMalloc((void**)&dst, st.size() * 4); hipMemcpy(dst, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    t.xst = dst; t.xg = dev_f(16, 1.f, 1.f); t.xb = dev_f(16, 0.f, 0.f);
    for (int i = 0; i < 15; i++) {
      const int nch = i < 5 ? 32 : i < 10 ? 64 : 128;
      t.wh[i] = dev_bf(WS[i], 0.05f); t.wl[i] = dev_bf(WS[i], 0.0002f);
      t.g[i] = dev_f(nch, 1.f, 1.f); t.b[i] = dev_f(nch, 0.f, 0.f);
    }
    void* y; hipMalloc(&y, (size_t)2 * B * 8192 * 2); t.y = (bf16*)y;
  }
  long long* prof; CK(hipMalloc((void**)&prof, (size_t)B * G * 32 * 8)); CK(hipMemset(prof, 0, (size_t)B * G * 32 * 8));
  a.prof = prof;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rest_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, REST_LDS));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(rest_x3_kernel, dim3(B, G), dim3(RTH), REST_LDS, 0, a);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int it = 0; it < 10; it++) hipLaunchKernelGGL(rest_x3_kernel, dim3(B, G), dim3(RTH), REST_LDS, 0, a);
  hipEventRecord(e1); CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("rest_x3_kernel: %d images x %d towers: %.1f us per launch\n", B, G, ms * 100.f);
  std::vector<long long> hp((size_t)B * G * 32);
  CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
  static const char* NAME[17] = {"setup", "entry2 load h0", "entry2 mma h0", "entry2 load h1", "entry2 mma h1", "entry2 stats+apply",
                                 "conv32 #1", "conv32 #2", "conv32 #3", "layer3 entry", "conv64 #1", "conv64 #2", "conv64 #3", "layer4 entry",
                                 "conv128 #1", "conv128 #2", "conv128 #3"};
  double tot = 0;
  for (int k = 1; k <= 16; k++) {
    double s = 0; int n = 0;
    for (int w = 0; w < B * G; w++) if (hp[w * 32 + k] && hp[w * 32 + k - 1]) { s += (double)(hp[w * 32 + k] - hp[w * 32 + k - 1]); n++; }
    if (n) { printf("  %-20s %8.0f ticks (100 MHz) = %6.2f us\n", NAME[k], s / n, s / n / 100.0); tot += s / n; }
  }
  printf("  total %.2f us per workgroup\n", tot / 100.0);
  return 0;
}
