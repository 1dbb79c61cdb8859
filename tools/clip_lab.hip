// Stand-alone lab for csrc/clip_tower.hip: the one-launch CLIP text tower on random weights / tokens (B dialogs with EOT positions
// spread over 2 .. 72), launch time and the per-phase cycle totals of the slowest and the mean workgroup.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off -DAVLEN_CT_PROF tools/clip_lab.hip -o tools/bin/clip_lab
//   tools/bin/clip_lab [B=64] [fixed_len=0] [co-runner workgroups=0] [co-runner mode: 1 spin, 2 stream memory]
#include "../avlen_amd/csrc/clip_tower.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
int avlen_zero_bytes(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? 0 : 2; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// a co-runner on another stream: `n` workgroups that hold a whole CU each (158 KB of LDS) for ~`us` microseconds, either spinning
// (mode 1) or streaming a private 4 MB window of memory (mode 2: L2 / Infinity Cache traffic like the visual towers' scratch)
__global__ __launch_bounds__(512) void corunner_kernel(float4* buf, long per_wg, long long ticks, int mode) {
  extern __shared__ char sm[];
  float4* p = buf + (long)blockIdx.x * per_wg;
  const long long t0 = wall_clock64();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  long i = threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
    if (mode == 2) {
#pragma unroll
      for (int k = 0; k < 8; k++) { const float4 v = p[i]; acc.x += v.x; p[i] = make_float4(v.x + 1.f, v.y, v.z, v.w); i += 512; if (i >= per_wg) i = threadIdx.x; }
    } else __builtin_amdgcn_s_sleep(64);
  }
  if (acc.x == 12345.678f) sm[0] = 1;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, fixed = argc > 2 ? atoi(argv[2]) : 0, layers = 12, ctx = 77, vocab = 49408;
  srand(5);
  auto dev_f = [&](size_t n, float lo, float hi) {
    std::vector<float> h(n);
    for (auto& v : h) v = lo + (hi - lo) * (rand() % 10001) / 10000.f;
    void* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return (float*)d;
  };
  ClipArgs a = {};
  std::vector<long long> tok((size_t)B * ctx, 0);
  for (int b = 0; b < B; b++) {
    const int ln = fixed ? fixed : 2 + (b * 71) / (B > 1 ? B - 1 : 1);
    for (int k = 0; k < ln; k++) tok[(size_t)b * ctx + k] = 1 + rand() % 40000;
    tok[(size_t)b * ctx] = 49406; tok[(size_t)b * ctx + ln] = 49407;
  }
  void* dt; CK(hipMalloc(&dt, tok.size() * 8)); CK(hipMemcpy(dt, tok.data(), tok.size() * 8, hipMemcpyHostToDevice));
  a.tokens = (const int64_t*)dt; a.tok_emb = dev_f((size_t)vocab * 512, -0.05f, 0.05f); a.pos_emb = dev_f((size_t)ctx * 512, -0.02f, 0.02f);
  a.ctx = ctx; a.vocab = vocab; a.layers = layers; a.frags_per_wave = clip_frags_per_wave(layers);
  const size_t wb = (size_t)16 * a.frags_per_wave * 1024;
  void* ws; CK(hipMalloc(&ws, wb));
  { std::vector<unsigned short> h(wb / 2); for (auto& v : h) { _Float16 f = (_Float16)(((rand() % 2001) / 1000.f - 1.f) * 0.03f); v = __builtin_bit_cast(unsigned short, f); }
    CK(hipMemcpy(ws, h.data(), wb, hipMemcpyHostToDevice)); }
  a.wstream = (const uint4*)ws;
  for (int l = 0; l < layers; l++)
    a.L[l] = ClipLayerP{dev_f(512, 1.f, 1.f), dev_f(512, 0.f, 0.f), dev_f(512, 1.f, 1.f), dev_f(512, 0.f, 0.f), dev_f(1536, -0.01f, 0.01f),
                        dev_f(512, -0.01f, 0.01f), dev_f(2048, -0.01f, 0.01f), dev_f(512, -0.01f, 0.01f)};
  void* E; CK(hipMalloc(&E, (size_t)B * 512 * 4)); a.E = (float*)E;
  long long* prof; CK(hipMalloc((void**)&prof, (size_t)4 * B * 8 * 8)); CK(hipMemset(prof, 0, (size_t)4 * B * 64)); a.prof = prof;
  void* xw; CK(hipMalloc(&xw, avlen_clip_tower_stream_ws_bytes(B))); a.flags = (unsigned*)xw; a.xflags = a.flags + 2 * B; a.xchg = (char*)xw + 16384; a.B = B;
  a.xslots = a.xchg + (size_t)B * 2 * CT_SLOTS * CT_SLOT;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&clip_tower_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int co_n = argc > 3 ? atoi(argv[3]) : 0, co_mode = argc > 4 ? atoi(argv[4]) : 1;
  hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const long per_wg = (4l << 20) / 16;
  float4* cobuf = nullptr;
  if (co_n) { CK(hipMalloc((void**)&cobuf, (size_t)co_n * per_wg * 16)); CK(hipMemset(cobuf, 0, (size_t)co_n * per_wg * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&corunner_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024)); }
  // each timed iteration: co-runner first (it takes its CUs), then the tower; the next iteration waits for both
  hipEvent_t ec; hipEventCreateWithFlags(&ec, hipEventDisableTiming);
  auto run = [&]() {
    if (co_n) hipLaunchKernelGGL(corunner_kernel, dim3(co_n), dim3(512), 158 * 1024, s2, cobuf, per_wg, 70000ll, co_mode);   // ~700 us at 100 MHz
    hipMemsetAsync(xw, 0, 16384, s1);
    hipLaunchKernelGGL(clip_tower_kernel<true>, dim3(((2 * B + 7) / 8) * 16), dim3(CT_TH), CT_LDS, s1, a);
    if (co_n) { hipEventRecord(ec, s2); hipStreamWaitEvent(s1, ec, 0); hipEventRecord(ec, s1); hipStreamWaitEvent(s2, ec, 0); }
  };
  for (int it = 0; it < 2; it++) run();
  CK(hipDeviceSynchronize());
  hipEventRecord(e0, s1);
  for (int it = 0; it < 5; it++) run();
  hipEventRecord(e1, s1); CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("clip_tower_kernel: %d dialogs: %.1f us per launch\n", B, ms * 200.f);
  std::vector<long long> hp((size_t)4 * B * 8); CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
  std::vector<float> he((size_t)B * 512); CK(hipMemcpy(he.data(), E, he.size() * 4, hipMemcpyDeviceToHost));
  double cs = 0; for (float v : he) cs += v; printf("  output checksum %.6f (the random inputs differ from process to process: rand() is shared with the runtime)\n", cs);
  {   // run-to-run determinism inside this process: one more launch, bitwise comparison
    run(); CK(hipDeviceSynchronize());
    std::vector<float> he2((size_t)B * 512); CK(hipMemcpy(he2.data(), E, he2.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < he.size(); i++) diff += memcmp(&he[i], &he2[i], 4) != 0;
    double amax = 0; for (float v : he) amax = fabs(v) > amax ? fabs(v) : amax;
    printf("  relaunch: %zu of %zu output words differ; max |output| %.3g\n", diff, he.size(), amax);
  }
  static const char* NAME[8] = {"ln1", "in_proj", "attention", "out_proj", "ln2", "c_fc + gelu", "c_proj", "bias / loop"};
  double tot_mean = 0, tot_max = 0;
  for (int k = 0; k < 8; k++) {
    double s = 0, mx = 0, s2 = 0; int n2 = 0;
    for (int id = 0; id < 4 * B; id++) {
      const int p = (id >> 4) * 8 + (id & 7);
      if (p < B) { s += (double)hp[id * 8 + k] / 2; if ((double)hp[id * 8 + k] > mx) mx = (double)hp[id * 8 + k]; }
      else if (p < 2 * B && hp[id * 8 + 1]) { s2 += (double)hp[id * 8 + k]; n2++; }
    }
    printf("  %-12s mean %9.0f  max %9.0f cycles per layer | second halves (%d): mean %9.0f\n", NAME[k], s / B / layers, mx / layers, n2, n2 ? s2 / n2 / layers : 0.0);
    tot_mean += s / B / layers; tot_max += mx / layers;
  }
  printf("  per layer: mean %.0f, sum of maxima %.0f cycles\n", tot_mean, tot_max);
  return 0;
}
