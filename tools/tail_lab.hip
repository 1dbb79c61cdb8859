// Stand-alone lab for csrc/tower_tail.hip (layers 3-4 of a tower): launches the kernel on random data (B images x G towers),
// prints the launch time, the mean duration of each phase (-DAVLEN_TAIL_PROF=<stamping thread>) and the difference to a CPU
// restatement of the same arithmetic for tower 0, image 0.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off [-DAVLEN_TAIL_PROF=0] tools/tail_lab.hip -o tools/bin/tail_lab
//   tools/bin/tail_lab [B=64] [G=6]
#include "../avlen_amd/csrc/tower_tail.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

static float bf_round(float f) { unsigned u; memcpy(&u, &f, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; memcpy(&f, &u, 4); return f; }
static float bf_to_f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f_to_bf(float f) { f = bf_round(f); unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

static void conv_gn_cpu(const std::vector<float>& in, int H, int cin, int cout, int ks, int stride, const std::vector<unsigned short>& wt,
                        const std::vector<float>& g, const std::vector<float>& b, const std::vector<float>* resid, bool relu,
                        std::vector<float>& o) {
  const int pad = ks / 2, K = ks * ks * cin, OH = (H + 2 * pad - ks) / stride + 1, npx = OH * OH, cpg = cout / 16;
  std::vector<double> raw((size_t)npx * cout);
  for (int y = 0; y < OH; y++) for (int xx = 0; xx < OH; xx++) for (int co = 0; co < cout; co++) {
    double a = 0;
    for (int ky = 0; ky < ks; ky++) for (int kx = 0; kx < ks; kx++) {
      const int iy = y * stride + ky - pad, ix = xx * stride + kx - pad;
      if (iy < 0 || iy >= H || ix < 0 || ix >= H) continue;
      for (int c = 0; c < cin; c++) a += (double)bf_to_f(wt[(size_t)co * K + (ky * ks + kx) * cin + c]) * in[((size_t)iy * H + ix) * cin + c];
    }
    raw[((size_t)y * OH + xx) * cout + co] = a;
  }
  o.assign((size_t)npx * cout, 0.f);
  for (int grp = 0; grp < 16; grp++) {
    double s1 = 0, s2 = 0;
    for (int p = 0; p < npx; p++) for (int j = 0; j < cpg; j++) { const double v = raw[(size_t)p * cout + grp * cpg + j]; s1 += v; s2 += v * v; }
    const double n = (double)npx * cpg, mean = s1 / n, var = s2 / n - mean * mean;
    for (int j = 0; j < cpg; j++) {
      const int co = grp * cpg + j;
      const float sc = g[co] * (float)(1.0 / sqrt(var + 1e-5)), sh = b[co] - (float)mean * sc;
      for (int p = 0; p < npx; p++) {
        float v = bf_round((float)raw[(size_t)p * cout + co]) * sc + sh;
        if (resid) v += (*resid)[(size_t)p * cout + co];
        o[(size_t)p * cout + co] = bf_round(relu ? (v > 0.f ? v : 0.f) : v);
      }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static const size_t WSIZE[10] = {64 * 32, 64 * 288, 64 * 576, 64 * 576, 64 * 576, 128 * 64, 128 * 576, 128 * 1152, 128 * 1152, 128 * 1152};

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, G = argc > 2 ? atoi(argv[2]) : 6;
  TailArgs a = {};
  srand(3);
  std::vector<unsigned short> hx0;                       // tower 0's input, kept for the CPU check
  std::vector<unsigned short> hw[10]; std::vector<float> hg[10], hb[10];
  auto rnd = []() { return (rand() % 2001) / 1000.f - 1.f; };
  for (int g = 0; g < G; g++) {
    TailTower& t = a.t[g];
    std::vector<unsigned short> hx((size_t)B * 1024 * 32);
    for (auto& v : hx) v = f_to_bf(fabsf(rnd()) * 1.5f);                    // post-ReLU activations
    void* dx; hipMalloc(&dx, hx.size() * 2); hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); t.x = (const bf16*)dx;
    if (g == 0) hx0.assign(hx.begin(), hx.begin() + 1024 * 32);
    for (int i = 0; i < 10; i++) {
      std::vector<unsigned short> w(WSIZE[i]);
      const float scale = (i % 5 == 0) ? 0.25f : (i < 5 ? 0.06f : 0.04f);
      for (auto& v : w) v = f_to_bf(scale * rnd());
      const int nch = i < 5 ? 64 : 128;
      std::vector<float> gm(nch), bt(nch);
      for (auto& v : gm) v = 1.f + 0.2f * rnd();
      for (auto& v : bt) v = 0.2f * rnd();
      void *dw, *dg, *db;
      {                                                   // fragment order (avlen_conv::w16f)
        const int cout = nch, K = (int)(WSIZE[i] / cout), ks = K / 32;
        std::vector<unsigned short> wf(w.size());
        for (int tt = 0; tt < cout / 16; tt++) for (int ii = 0; ii < ks; ii++) for (int ln = 0; ln < 64; ln++) for (int e = 0; e < 8; e++)
          wf[(((size_t)tt * ks + ii) * 64 + ln) * 8 + e] = w[(size_t)(tt * 16 + (ln & 15)) * K + ii * 32 + 8 * (ln >> 4) + e];
        hipMalloc(&dw, wf.size() * 2); hipMemcpy(dw, wf.data(), wf.size() * 2, hipMemcpyHostToDevice);
      }
      hipMalloc(&dg, nch * 4); hipMemcpy(dg, gm.data(), nch * 4, hipMemcpyHostToDevice);
      hipMalloc(&db, nch * 4); hipMemcpy(db, bt.data(), nch * 4, hipMemcpyHostToDevice);
      t.w[i] = (const bf16*)dw; t.g[i] = (const float*)dg; t.b[i] = (const float*)db;
      if (g == 0) { hw[i] = w; hg[i] = gm; hb[i] = bt; }
    }
    void* y; hipMalloc(&y, (size_t)B * 64 * 128 * 2); t.y = (bf16*)y;
  }
  long long* prof; CK(hipMalloc(&prof, (size_t)B * G * 32 * 8));
  a.prof = prof;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(tower_tail_kernel, dim3(B, G), dim3(TTH), TAIL_LDS, 0, a, B);
  CK(hipDeviceSynchronize());
  const int IT = 20;
  hipEventRecord(e0);
  for (int it = 0; it < IT; it++) hipLaunchKernelGGL(tower_tail_kernel, dim3(B, G), dim3(TTH), TAIL_LDS, 0, a, B);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("B=%d G=%d: %.1f us per launch (%d workgroups)\n", B, G, ms * 1000 / IT, B * G);
#ifdef AVLEN_TAIL_PROF
  {
    std::vector<long long> h((size_t)B * G * 32);
    hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
    const int NP = 12;
    const char* names[NP] = {"load image", "l3 downsample", "l3 conv s2", "l3 conv2", "l3 conv3", "l3 conv4", "l4 downsample", "l4 conv s2",
                             "l4 conv2", "l4 conv3", "l4 conv4", "store"};
    double tot = 0; std::vector<double> ph(NP, 0.0);
    for (int w = 0; w < B * G; w++) for (int k = 0; k < NP; k++) ph[k] += (double)(h[w * 32 + k + 1] - h[w * 32 + k]);
    for (int k = 0; k < NP; k++) tot += ph[k];
    for (int k = 0; k < NP; k++) printf("  %-14s %6.1f %%  (%.0f ticks)\n", names[k], 100 * ph[k] / tot, ph[k] / (B * G));
    printf("  workgroup total %.0f ticks\n", tot / (B * G));
  }
#endif
  {
    std::vector<float> x(hx0.size());
    for (size_t i = 0; i < x.size(); i++) x[i] = bf_to_f(hx0[i]);
    std::vector<float> d, e1v, e2, e3, e4, d4, f1, f2, f3, f4;
    conv_gn_cpu(x, 32, 32, 64, 1, 2, hw[0], hg[0], hb[0], nullptr, false, d);
    conv_gn_cpu(x, 32, 32, 64, 3, 2, hw[1], hg[1], hb[1], nullptr, true, e1v);
    conv_gn_cpu(e1v, 16, 64, 64, 3, 1, hw[2], hg[2], hb[2], &d, true, e2);
    conv_gn_cpu(e2, 16, 64, 64, 3, 1, hw[3], hg[3], hb[3], nullptr, true, e3);
    conv_gn_cpu(e3, 16, 64, 64, 3, 1, hw[4], hg[4], hb[4], &e2, true, e4);
    conv_gn_cpu(e4, 16, 64, 128, 1, 2, hw[5], hg[5], hb[5], nullptr, false, d4);
    conv_gn_cpu(e4, 16, 64, 128, 3, 2, hw[6], hg[6], hb[6], nullptr, true, f1);
    conv_gn_cpu(f1, 8, 128, 128, 3, 1, hw[7], hg[7], hb[7], &d4, true, f2);
    conv_gn_cpu(f2, 8, 128, 128, 3, 1, hw[8], hg[8], hb[8], nullptr, true, f3);
    conv_gn_cpu(f3, 8, 128, 128, 3, 1, hw[9], hg[9], hb[9], &f2, true, f4);
    std::vector<unsigned short> yy((size_t)64 * 128); hipMemcpy(yy.data(), a.t[0].y, yy.size() * 2, hipMemcpyDeviceToHost);
    double mx = 0, sm = 0, rf = 0; int bad = 0;
    for (size_t i = 0; i < f4.size(); i++) {
      const double dd = fabs((double)bf_to_f(yy[i]) - f4[i]); mx = dd > mx ? dd : mx; sm += dd; rf += fabs(f4[i]); bad += dd > 0.05 + 0.02 * fabs(f4[i]);
    }
    printf("  vs CPU restatement: max |d| %.4f, mean |d| %.5f (mean |ref| %.4f), %d of %zu beyond 0.05 + 2 %%\n", mx, sm / f4.size(), rf / f4.size(), bad, f4.size());
  }
  return 0;
}
