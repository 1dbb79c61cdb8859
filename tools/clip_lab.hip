// Stand-alone lab for csrc/clip_tower.hip: the one-launch CLIP text tower on random weights / tokens (B dialogs with EOT positions
// spread over 2 .. 72), launch time and the per-phase cycle totals of the slowest and the mean workgroup.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off -DAVLEN_CT_PROF tools/clip_lab.hip -o tools/bin/clip_lab
//   (without -DAVLEN_CT_PROF: the product's code, time only -- the phase counters cost the 4-way kernel a factor of two)
//   tools/bin/clip_lab [B=64] [fixed_len=0] [co-runner workgroups=0] [co-runner mode: 1 spin, 2 stream memory] [split4 workgroup limit]
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

// CLIP_LAB_LDS=<word>: before every tower launch every CU's LDS is filled with this word -- a result that depends on it reads LDS
// the launch never wrote
__global__ __launch_bounds__(512) void lds_fill_kernel(unsigned word, unsigned* sink) {
  extern __shared__ char sm[];
  unsigned* w = reinterpret_cast<unsigned*>(sm);
  for (int i = threadIdx.x; i < 158 * 256; i += 512) w[i] = word;
  __syncthreads();
  __builtin_amdgcn_s_sleep(100);
  if (w[(threadIdx.x * 7) % (158 * 256)] == 0x12345u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, fixed = argc > 2 ? atoi(argv[2]) : 0, layers = 12, ctx = 77, vocab = 49408;
  // own generator: glibc's rand() state is shared with the HIP runtime's threads (the inputs differed from process to process)
  unsigned long long rng = 5;
  auto rnd = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
  auto dev_f = [&](size_t n, float lo, float hi) {
    std::vector<float> h(n);
    for (auto& v : h) v = lo + (hi - lo) * (rnd() % 10001) / 10000.f;
    void* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return (float*)d;
  };
  std::vector<long long> tok((size_t)B * ctx, 0);
  for (int b = 0; b < B; b++) {
    const int ln = fixed ? fixed : 2 + (b * 71) / (B > 1 ? B - 1 : 1);
    for (int k = 0; k < ln; k++) tok[(size_t)b * ctx + k] = 1 + rnd() % 40000;
    tok[(size_t)b * ctx] = 49406; tok[(size_t)b * ctx + ln] = 49407;
  }
  void* dt; CK(hipMalloc(&dt, tok.size() * 8)); CK(hipMemcpy(dt, tok.data(), tok.size() * 8, hipMemcpyHostToDevice));
  avlen_clip_text P = {};
  P.width = 512; P.heads = 8; P.layers = layers; P.ctx = ctx; P.vocab = vocab; P.out_dim = 512;
  P.tok_emb = dev_f((size_t)vocab * 512, -0.05f, 0.05f); P.pos_emb = dev_f((size_t)ctx * 512, -0.02f, 0.02f);
  for (int l = 0; l < layers; l++) {
    avlen_clip_block& b = P.block[l];
    b.ln1.g = dev_f(512, 1.f, 1.f); b.ln1.b = dev_f(512, 0.f, 0.f); b.ln2.g = dev_f(512, 1.f, 1.f); b.ln2.b = dev_f(512, 0.f, 0.f);
    auto lin = [&](avlen_linear& L, int o, int in) { L.w = dev_f((size_t)o * in, -0.03f, 0.03f); L.b = dev_f(o, -0.01f, 0.01f); L.out_f = o; L.in_f = in; };
    lin(b.attn.in_proj, 1536, 512); lin(b.attn.out_proj, 512, 512); lin(b.fc, 2048, 512); lin(b.proj, 512, 2048);
  }
  void* ws; CK(hipMalloc(&ws, avlen_clip_stream_bytes(&P)));
  P.wstream = ws;
  if (avlen_clip_pack_stream(&P, ws, 1, 0) != 0) { printf("pack failed\n"); return 1; }
  void* E; CK(hipMalloc(&E, (size_t)B * 512 * 4));
  const int grid_max = ((2 * B + 7) / 8) * 16 + 256;
  long long* prof; CK(hipMalloc((void**)&prof, (size_t)grid_max * 64)); CK(hipMemset(prof, 0, (size_t)grid_max * 64));
#ifdef AVLEN_CT_PROF
  g_ct_prof = prof;
#endif
  const size_t xb = avlen_clip_tower_stream_ws_bytes(B);
  void* xw; CK(hipMalloc(&xw, xb));
  if (getenv("CLIP_LAB_FILL")) { const int f = atoi(getenv("CLIP_LAB_FILL")); CK(hipMemset(xw, f, xb)); CK(hipMemset(E, f, (size_t)B * 512 * 4)); }
  if (argc > 5) avlen_set_clip_tower_split4_wgs(atoi(argv[5]));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int co_n = argc > 3 ? atoi(argv[3]) : 0, co_mode = argc > 4 ? atoi(argv[4]) : 1;
  hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const long per_wg = (4l << 20) / 16;
  float4* cobuf = nullptr;
  if (co_n) { CK(hipMalloc((void**)&cobuf, (size_t)co_n * per_wg * 16)); CK(hipMemset(cobuf, 0, (size_t)co_n * per_wg * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&corunner_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024)); }
  hipEvent_t ec; hipEventCreateWithFlags(&ec, hipEventDisableTiming);
  const bool fill_lds = getenv("CLIP_LAB_LDS") != nullptr;
  const unsigned lds_word = fill_lds ? (unsigned)strtoul(getenv("CLIP_LAB_LDS"), nullptr, 0) : 0u;
  unsigned* sink; CK(hipMalloc((void**)&sink, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
  auto run = [&]() {
    if (fill_lds) hipLaunchKernelGGL(lds_fill_kernel, dim3(1024), dim3(512), 158 * 1024, s1, lds_word, sink);
    if (co_n) hipLaunchKernelGGL(corunner_kernel, dim3(co_n), dim3(512), 158 * 1024, s2, cobuf, per_wg, 70000ll, co_mode);   // ~700 us at 100 MHz
    if (avlen_clip_tower_stream_fwd(&P, (const int64_t*)dt, (float*)E, B, 1, xw, xb, s1, nullptr) != 0) printf("launch failed\n");
    if (co_n) { hipEventRecord(ec, s2); hipStreamWaitEvent(s1, ec, 0); hipEventRecord(ec, s1); hipStreamWaitEvent(s2, ec, 0); }
  };
  for (int it = 0; it < 2; it++) run();
  CK(hipDeviceSynchronize());
  CK(hipMemset(prof, 0, (size_t)grid_max * 64));
  hipEventRecord(e0, s1);
  for (int it = 0; it < 5; it++) run();
  hipEventRecord(e1, s1); CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("text tower (work list + both launches): %d dialogs: %.1f us per call\n", B, ms * 200.f);
  std::vector<long long> hp((size_t)grid_max * 8); CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
  std::vector<float> he((size_t)B * 512); CK(hipMemcpy(he.data(), E, he.size() * 4, hipMemcpyDeviceToHost));
  {   // run-to-run determinism inside this process: one more launch, bitwise comparison
    run(); CK(hipDeviceSynchronize());
    std::vector<float> he2((size_t)B * 512); CK(hipMemcpy(he2.data(), E, he2.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < he.size(); i++) diff += memcmp(&he[i], &he2[i], 4) != 0;
    double amax = 0; for (float v : he) amax = fabs(v) > amax ? fabs(v) : amax;
    unsigned long long ck = 0; for (size_t i = 0; i < he.size(); i++) { unsigned u; memcpy(&u, &he[i], 4); ck = ck * 1099511628211ull + u; }
    printf("  relaunch: %zu of %zu output words differ; max |output| %.6g; checksum %016llx\n", diff, he.size(), amax, ck);
    if (getenv("CLIP_LAB_ROWS")) for (int b = 0; b < B; b++) { unsigned long long c = 0; for (int k = 0; k < 512; k++) { unsigned u; memcpy(&u, &he[(size_t)b * 512 + k], 4); c = c * 1099511628211ull + u; } printf("   row %d %016llx\n", b, c); }
    if (diff) { printf("  rows that differ:"); for (int b = 0; b < B; b++) if (memcmp(&he[(size_t)b * 512], &he2[(size_t)b * 512], 2048)) printf(" %d", b); printf("\n"); }
  }
  // phase totals summed over the 5 timed launches (the table was zeroed before them)
  static const char* NAME[8] = {"ln1", "in_proj", "attention", "out_proj", "ln2 (+ exchange 1)", "c_fc + gelu", "c_proj", "bias / loop (+ exchange 2)"};
  int active = 0;
  for (int id = 0; id < grid_max; id++) active += hp[(size_t)id * 8 + 1] != 0;
  printf("  %d workgroups ran\n", active);
  double tot_mean = 0, tot_max = 0;
  for (int k = 0; k < 8; k++) {
    double s = 0, mx = 0;
    for (int id = 0; id < grid_max; id++) if (hp[(size_t)id * 8 + 1]) { s += (double)hp[(size_t)id * 8 + k]; if ((double)hp[(size_t)id * 8 + k] > mx) mx = (double)hp[(size_t)id * 8 + k]; }
    printf("  %-28s mean %9.0f  max %9.0f cycles per layer\n", NAME[k], active ? s / active / layers / 5 : 0.0, mx / layers / 5);
    tot_mean += active ? s / active / layers / 5 : 0.0; tot_max += mx / layers / 5;
  }
  printf("  per layer: mean %.0f, sum of maxima %.0f cycles\n", tot_mean, tot_max);
  return 0;
}
