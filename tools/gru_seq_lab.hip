// Lab for the resident GRU sequence kernels (csrc/train_gru.hip): T steps of N rows, time per step and per-phase wall-clock totals
// (hand-off wait + staging | products + reductions | gate arithmetic + stores | closing barrier) of the mean and the slowest workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off -DAVLEN_SEQ_PROF tools/gru_seq_lab.hip -Lavlen_amd/lib -lavlen_hip -Wl,-rpath,$PWD/avlen_amd/lib -o tools/bin/gru_seq_lab
#include "../avlen_amd/csrc/train_gru.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 150, N = argc > 2 ? atoi(argv[2]) : 8, H = 512, K = 3 * H;
  auto dev = [&](size_t n, float lo, float hi) { std::vector<float> h(n); for (auto& v : h) v = lo + (hi - lo) * (rand() % 10001) / 10000.f;
    float* d; hipMalloc((void**)&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; };
  const size_t R = (size_t)T * N;
  float *whh = dev((size_t)K * H, -0.04f, 0.04f), *bhh = dev(K, -0.1f, 0.1f), *gi = dev(R * K, -1.f, 1.f), *h0 = dev((size_t)N * H, -0.5f, 0.5f);
  float *masks = dev(R, 1.f, 1.f), *out = dev(R * H, 0, 0), *hm = dev(R * H, 0, 0), *gh = dev(R * K, 0, 0), *dout = dev(R * H, -0.1f, 0.1f);
  float *dgi = dev(R * K, 0, 0), *dgh = dev(R * K, 0, 0), *whhT = dev((size_t)K * H, -0.04f, 0.04f);
  unsigned* err; CK(hipMalloc((void**)&err, 256)); CK(hipMemset(err, 0, 256));
  long long* prof; CK(hipMalloc((void**)&prof, 64 * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_seq_prof), &prof, sizeof(prof)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static const char* NAME[4] = {"hand-off wait + staging", "products + reductions", "gate arithmetic + stores", "closing barrier"};
  for (int pass = 0; pass < 2; pass++) {
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
      CK(hipMemset(pass ? dgh : out, 0xff, pass ? R * K * 4 : R * H * 4));
      CK(hipMemset(prof, 0, 64 * 8 * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      if (!pass) launch_gru_seq_fwd(whh, bhh, gi, h0, masks, out, hm, gh, T, N, H, err, 0);
      else launch_gru_seq_bwd(whhT, gi, gh, hm, dout, masks, dgi, dgh, T, N, H, err, 0);
      CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned he; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
    std::vector<long long> hp(64 * 8); CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
    printf("%s: T=%d N=%d: %.2f us per step%s\n", pass ? "gru_seq_bwd_kernel" : "gru_seq_fwd_kernel", T, N, best * 1000.f / T, he ? " ** hand-off timed out **" : "");
    for (int k = 0; k < 4; k++) {
      double s = 0, mx = 0; for (int w = 0; w < H / 8; w++) { s += hp[w * 8 + k]; if (hp[w * 8 + k] > mx) mx = hp[w * 8 + k]; }
      printf("  %-26s mean %.2f us  max %.2f us per step\n", NAME[k], s / (H / 8) / T / 100.0, mx / T / 100.0);
    }
  }
  return 0;
}
