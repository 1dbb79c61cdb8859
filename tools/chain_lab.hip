// Stand-alone lab for the fused row-batch chain (chain.hip with per-step time stamps): the pi_q program on random weights.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DAVLEN_CHAIN_LAB tools/chain_lab.hip -o build/chain_lab && build/chain_lab
#include "../avlen_amd/csrc/chain.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  const int x3 = argc > 1 ? atoi(argv[1]) : 0;
  const int B = 64, d = 256, NL = 18;
  std::vector<unsigned short> h((size_t)NL * d * 512);
  unsigned s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3a00 + ((s >> 16) & 0x1ff) + ((s >> 31) << 15)); }
  char* W; hipMalloc(&W, h.size() * 2); hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  float *bias, *goal, *out; hipMalloc(&bias, 4096); hipMemset(bias, 0, 4096); hipMalloc(&goal, B * d * 4); hipMemset(goal, 0, B * d * 4);
  hipMalloc(&out, B * d * 4);
  char* X; hipMalloc(&X, B * 320 * 2); hipMemcpy(X, h.data(), B * 320 * 2, hipMemcpyHostToDevice);
  char* big; hipMalloc(&big, (size_t)512 << 20);
  avlen_chain p; p.n = 0;
  auto add = [&](int kind, int k, int ld, int act, int res, int buf, int ob, const void* p0, const void* p1) {
    p.op[p.n++] = avlen_chain_op{kind, k, ld, 0, act, res, buf, ob, 1, 0, 0.f, 0, p0, p1, nullptr};
  };
  int wi = 0;
  auto lin = [&](int k, int act, int res, int buf, int ob) {
    add(AVLEN_CH_LINEAR, k, k, act, res, buf, ob, W + (size_t)(wi++) * d * 512 * 2, bias);
    p.op[p.n - 1].p2 = W + (size_t)(wi - 1) * d * 512 * 2 + d * 256 * 2;          // x3: the "low plane" (any bf16 data does for timing)
  };
  auto ln = [&](int ob) { add(AVLEN_CH_LAYERNORM, 0, 0, 0, 0, 0, ob, bias, bias); };
  add(AVLEN_CH_LOAD_X16, 320, 320, 0, 0, 0, 0, X, x3 ? X : nullptr);
  lin(320, 1, 0, 0, 1); lin(256, 0, 0, 1, 0); add(AVLEN_CH_SAVE, 0, 0, 0, 0, 0, 0, nullptr, nullptr);
  lin(256, 0, 0, 0, 1); lin(256, 0, 1, 1, 0); ln(0); add(AVLEN_CH_SAVE, 0, 0, 0, 0, 0, 0, nullptr, nullptr);
  lin(256, 1, 0, 0, 1); lin(256, 0, 1, 1, 0); ln(0); ln(0);
  lin(256, 0, 0, 0, 1); add(AVLEN_CH_SAVE, 0, 0, 0, 1, 0, 0, nullptr, nullptr);
  add(AVLEN_CH_LOAD_CUR, 0, d, 0, 0, 0, 0, goal, nullptr); add(AVLEN_CH_SAVE, 0, 0, 0, 0, 0, 0, nullptr, nullptr);
  lin(256, 0, 0, 0, 1); lin(256, 0, 1, 1, 0); ln(0); add(AVLEN_CH_SAVE, 0, 0, 0, 0, 0, 0, nullptr, nullptr);
  add(AVLEN_CH_RECALL, 0, 0, 0, 1, 0, 1, nullptr, nullptr);
  lin(256, 0, 1, 1, 0); ln(0); add(AVLEN_CH_SAVE, 0, 0, 0, 0, 0, 0, nullptr, nullptr);
  lin(256, 1, 0, 0, 1); lin(256, 0, 1, 1, 0); ln(0); ln(0);
  add(AVLEN_CH_STORE, 0, d, 0, 0, 0, 0, out, nullptr);
  const char* names[] = {"?", "LOAD_X16", "LOAD_CUR", "LINEAR", "LAYERNORM", "SAVE", "STORE", "RECALL"};
  for (int cold = 0; cold < 2; cold++) {
    for (int rep = 0; rep < 3; rep++) {
      if (cold) hipMemsetAsync(big, rep, (size_t)512 << 20, 0);          // evict L2 / Infinity Cache
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      int rc = avlen_chain_run(&p, B, 0, x3);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long st[64]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chain_stamps), sizeof(st));
      printf("%s rep %d rc %d: kernel %.1f us; in-kernel %lld ticks\n", cold ? "cold" : "warm", rep, rc, ms * 1e3, st[p.n] - st[0]);
      if (rep == 2) for (int i = 0; i < p.n; i++) printf("   step %2d %-10s k=%3d  %6lld ticks\n", i, names[p.op[i].kind], p.op[i].k, st[i + 1] - st[i]);
    }
  }
  return 0;
}
