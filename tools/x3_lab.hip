// Stand-alone lab for csrc/tower_x3.hip (a GroupNorm ResNet-18 tower in compensated bf16): runs the persistent work-queue launch
// (stem + layer 1 items, layer 2-4 items) on random data (B images x G towers) and prints the launch time and the mean duration of
// each phase of an item.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off -DAVLEN_X3_PROF=0 tools/x3_lab.hip -o tools/bin/x3_lab
//   tools/bin/x3_lab [B=64] [G=6]
#include "../avlen_amd/csrc/tower_x3.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int avlen_zero_bytes(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? 0 : 2; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void report(const char* kernel, const long long* hp, int n_wg, const char* const* name, int n_phase) {
  double tot = 0;
  for (int k = 1; k <= n_phase; k++) {
    double s = 0; int n = 0;
    for (int w = 0; w < n_wg; w++) if (hp[w * 32 + k] && hp[w * 32 + k - 1]) { s += (double)(hp[w * 32 + k] - hp[w * 32 + k - 1]); n++; }
    if (n) { printf("  %-22s %8.0f cycles\n", name[k], s / n); tot += s / n; }
  }
  printf("  %s: total %.0f cycles per item\n", kernel, tot);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, G = argc > 2 ? atoi(argv[2]) : 6;
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
  auto dev_u8 = [&](size_t n) {
    std::vector<unsigned char> h(n);
    for (auto& v : h) v = (unsigned char)(rand() & 255);
    void* d; hipMalloc(&d, n); hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice); return d;
  };
  const int n = B * G;
  long long* prof; CK(hipMalloc((void**)&prof, (size_t)2 * n * 32 * 8));
  std::vector<long long> hp((size_t)2 * n * 32);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  TowerArgs a = {};
  a.B = B; a.n = n; a.prof = prof;
  const size_t qb = tower_x3_queue_bytes(n);
  CK(hipMalloc((void**)&a.q, qb));
  a.l1.S = 128;
  a.rest.y_lo = (long)B * 8192;
  static const size_t WS[15] = {32 * 16, 32 * 144, 32 * 288, 32 * 288, 32 * 288, 64 * 32, 64 * 288, 64 * 576, 64 * 576, 64 * 576,
                                128 * 64, 128 * 576, 128 * 1152, 128 * 1152, 128 * 1152};
  for (int g = 0; g < G; g++) {
    L1Tower& t = a.l1.t[g];
    t.C = g & 1 ? 1 : 3; t.u8 = g & 1 ? 0 : 1; t.div = g & 1 ? 1.f : 255.f;           // rgb: uint8 / 255, depth: fp32 (the step's sensors)
    t.img = g & 1 ? (void*)dev_f((size_t)B * 128 * 128, 0.f, 1.f) : dev_u8((size_t)B * 128 * 128 * 3);
    t.wh[0] = dev_bf(16 * 392, 0.1f); t.wl[0] = dev_bf(16 * 392, 0.0004f);
    for (int i = 1; i < 5; i++) { t.wh[i] = dev_bf(16 * 144, 0.1f); t.wl[i] = dev_bf(16 * 144, 0.0004f); }
    for (int i = 0; i < 5; i++) { t.g[i] = dev_f(16, 1.f, 1.f); t.b[i] = dev_f(16, 0.f, 0.f); }
    void* p; hipMalloc(&p, (size_t)B * 2 * APLANE * 2); t.a0 = (bf16*)p; hipMalloc(&p, (size_t)B * 2 * APLANE * 2); t.a1 = (bf16*)p;
    RestTower& r = a.rest.t[g];
    r.a = t.a0;
    for (int i = 0; i < 15; i++) {
      const int nch = i < 5 ? 32 : i < 10 ? 64 : 128;
      r.wh[i] = dev_bf(WS[i], 0.05f); r.wl[i] = dev_bf(WS[i], 0.0002f);
      r.g[i] = dev_f(nch, 1.f, 1.f); r.b[i] = dev_f(nch, 0.f, 0.f);
    }
    void* y; hipMalloc(&y, (size_t)2 * B * 8192 * 2); r.y = (bf16*)y;
  }
  CK(hipMemset(prof, 0, (size_t)2 * n * 32 * 8));
  for (int it = 0; it < 3; it++) if (launch_tower_x3(a, qb, 0) != 0) { printf("launch failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int it = 0; it < 10; it++) launch_tower_x3(a, qb, 0);
  hipEventRecord(e1); CK(hipDeviceSynchronize());
  hipEventElapsedTime(&ms, e0, e1);
  unsigned hq[4]; CK(hipMemcpy(hq, a.q, 16, hipMemcpyDeviceToHost));
  printf("tower_x3_kernel: %d images x %d towers: %.1f us per launch (queue head %u, give-up word %u)\n", B, G, ms * 100.f, hq[0], hq[1]);
  CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
  static const char* N1[25] = {"setup", "stem fill", "stem mma", "stem stats", "stem apply",
                               "c1 mma h0", "c1 load h1", "c1 mma h1", "c1 stats", "c1 apply", "c2 mma h0", "c2 load h1", "c2 mma h1", "c2 stats", "c2 apply+res",
                               "c3 mma h0", "c3 load h1", "c3 mma h1", "c3 stats", "c3 apply", "c4 mma h0", "c4 load h1", "c4 mma h1", "c4 stats", "c4 apply+res"};
  report("stem + layer 1", hp.data(), n, N1, 24);
  static const char* N2[17] = {"setup", "entry2 load h0", "entry2 mma h0", "entry2 load h1", "entry2 mma h1", "entry2 stats+apply",
                               "conv32 #1", "conv32 #2", "conv32 #3", "layer3 entry", "conv64 #1", "conv64 #2", "conv64 #3", "layer4 entry",
                               "conv128 #1", "conv128 #2", "conv128 #3"};
  report("layers 2-4", hp.data() + (size_t)n * 32, n, N2, 16);
  // timeline from the chip-wide 100 MHz clock: when items of each class started and ended, relative to the first start
  long long t0 = 0;
  for (int w = 0; w < 2 * n; w++) if (hp[w * 32 + 29] && (!t0 || hp[w * 32 + 29] < t0)) t0 = hp[w * 32 + 29];
  auto us = [&](long long v) { return (double)(v - t0) / 100.0; };
  for (int cls = 0; cls < 2; cls++) {
    for (int blk = 0; blk < (n + 127) / 128; blk++) {
      double s_lo = 1e30, s_hi = 0, e_lo = 1e30, e_hi = 0, wait = 0; int cnt = 0;
      for (int i = blk * 128; i < n && i < blk * 128 + 128; i++) {
        const long long* r = &hp[(size_t)(cls * n + i) * 32];
        if (!r[29] || !r[31]) continue;
        const double a0 = us(r[29]), a1 = us(r[31]);
        s_lo = a0 < s_lo ? a0 : s_lo; s_hi = a0 > s_hi ? a0 : s_hi; e_lo = a1 < e_lo ? a1 : e_lo; e_hi = a1 > e_hi ? a1 : e_hi;
        if (cls) wait += (double)(r[30] - r[29]) / 100.0;
        cnt++;
      }
      printf("  %s items %4d..%4d: start %6.1f..%6.1f us, end %6.1f..%6.1f us%s", cls ? "layers 2-4 " : "stem+layer1", blk * 128, blk * 128 + cnt - 1,
             s_lo, s_hi, e_lo, e_hi, cls ? "" : "\n");
      if (cls) printf(", mean flag wait %.1f us\n", wait / (cnt ? cnt : 1));
    }
  }
  return 0;
}
