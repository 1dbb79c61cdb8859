// Stand-alone lab for the fused tower tail (tower_tail.hip with phase time stamps): random weights / activations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DAVLEN_TT_LAB tools/tower_lab.hip -o build/tower_lab && build/tower_lab
#include "../avlen_amd/csrc/tower_tail.hip"
#include <stdio.h>
#include <string.h>
#include <vector>

int main() {
  const int B = 384;
  std::vector<unsigned short> h((size_t)B * 32 * 32 * 32);
  unsigned s = 7;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff) + ((s >> 31) << 15)); }
  void *x, *y, *w; float* gb;
  hipMalloc(&x, h.size() * 2); hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&y, (size_t)B * 64 * 128 * 2);
  hipMalloc(&w, (size_t)4 << 20); hipMemcpy(w, h.data(), (size_t)4 << 20, hipMemcpyHostToDevice);
  std::vector<float> ones(256, 1.f); hipMalloc(&gb, 1024); hipMemcpy(gb, ones.data(), 1024, hipMemcpyHostToDevice);
  avlen_resnet18 net; memset(&net, 0, sizeof(net));
  char* wp = (char*)w;
  auto conv = [&](avlen_conv& c, int cin, int cout, int k, int stride) {
    c.cin = cin; c.cin16 = cin; c.cout = cout; c.kh = c.kw = k; c.stride = stride; c.pad = k / 2; c.w16 = wp; wp += (size_t)cout * k * k * cin * 2;
  };
  for (int l = 0; l < 2; l++) {
    const int cin = l ? 64 : 32, co = l ? 128 : 64;
    avlen_resblock& b0 = net.block[4 + 2 * l]; avlen_resblock& b1 = net.block[5 + 2 * l];
    conv(b0.conv1, cin, co, 3, 2); conv(b0.conv2, co, co, 3, 1); conv(b0.down, cin, co, 1, 2); b0.has_down = 1;
    conv(b1.conv1, co, co, 3, 1); conv(b1.conv2, co, co, 3, 1);
    for (avlen_affine* a : {&b0.bn1, &b0.bn2, &b0.bnd, &b1.bn1, &b1.bn2}) { a->g = gb; a->b = gb; }
  }
  const avlen_resnet18* nets[1] = {&net}; const void* X[1] = {x}; void* Y[1] = {y};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    int zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_tt_n), &zero, 4);
    hipEventRecord(e0, 0);
    int rc = avlen_tower_tail_bf16(nets, X, Y, 1, B, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long st[64]; int n; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tt_stamps), sizeof(st)); hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tt_n), 4);
    printf("rep %d rc %d: %d images, kernel %.1f us; block 0: %lld ticks in %d phases\n", rep, rc, B, ms * 1e3, st[n - 1] - st[0], n);
    if (rep == 2) {
      const char* names[] = {"prime", "conv1 s2", "down 1x1", "gn apply", "conv2", "gn+res apply", "conv1'", "gn apply'", "conv2'", "gn+res apply'"};
      for (int i = 1; i < n; i++) printf("   %2d %-14s %7lld ticks\n", i, i <= 20 ? names[(i - 1) % 10 == 0 && i > 1 ? 0 : (i - 1) % 10] : "?", st[i] - st[i - 1]);
    }
  }
  return 0;
}
