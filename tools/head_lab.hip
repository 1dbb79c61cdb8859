// Stand-alone lab for csrc/tower_head.hip: launches the kernel on random data (B images x G towers), prints the launch time and
// the mean duration of each phase from wave-0 timestamps (s_memtime).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off [-DAVLEN_HEAD_PROF=<stamping thread: 0, 448>] tools/head_lab.hip -o tools/bin/head_lab
//   tools/bin/head_lab [B=64] [G=6] [S=128] [u8=0]
#include "../avlen_amd/csrc/tower_head.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

__global__ void dpp_selftest(float* out) { out[threadIdx.x] = row16_sum((float)threadIdx.x); }

static float bf_round(float f) { unsigned u; memcpy(&u, &f, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; memcpy(&f, &u, 4); return f; }
static float bf_to_f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

// CPU restatement of the kernel's arithmetic for one image (double accumulation; the comparison allows for bf16 roundings that flip)
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

static void cpu_head(const std::vector<float>& img, int S, int C, float div, const std::vector<unsigned short>* w, const std::vector<float>* gm,
                     const std::vector<float>* bt, std::vector<float>& out) {
  const int k = S / 64;
  std::vector<float> x((size_t)64 * 64 * 8, 0.f);
  for (int oy = 0; oy < 64; oy++) for (int ox = 0; ox < 64; ox++) for (int c = 0; c < C; c++) {
    float s = 0.f;
    for (int dy = 0; dy < k; dy++) for (int dx = 0; dx < k; dx++) s += img[((size_t)(oy * k + dy) * S + ox * k + dx) * C + c] / div;
    x[((size_t)oy * 64 + ox) * 8 + c] = bf_round(s * (1.f / (k * k)));
  }
  std::vector<float> a0, a1, a2, a3, a4, d, e1, e2, e3, e4;
  conv_gn_cpu(x, 64, 8, 16, 7, 1, w[0], gm[0], bt[0], nullptr, true, a0);
  conv_gn_cpu(a0, 64, 16, 16, 3, 1, w[1], gm[1], bt[1], nullptr, true, a1);
  conv_gn_cpu(a1, 64, 16, 16, 3, 1, w[2], gm[2], bt[2], &a0, true, a2);
  conv_gn_cpu(a2, 64, 16, 16, 3, 1, w[3], gm[3], bt[3], nullptr, true, a3);
  conv_gn_cpu(a3, 64, 16, 16, 3, 1, w[4], gm[4], bt[4], &a2, true, a4);
  conv_gn_cpu(a4, 64, 16, 32, 1, 2, w[5], gm[5], bt[5], nullptr, false, d);          // downsample + norm (no ReLU)
  conv_gn_cpu(a4, 64, 16, 32, 3, 2, w[6], gm[6], bt[6], nullptr, true, e1);
  conv_gn_cpu(e1, 32, 32, 32, 3, 1, w[7], gm[7], bt[7], &d, true, e2);
  conv_gn_cpu(e2, 32, 32, 32, 3, 1, w[8], gm[8], bt[8], nullptr, true, e3);
  conv_gn_cpu(e3, 32, 32, 32, 3, 1, w[9], gm[9], bt[9], &e2, true, e4);
  out = e4;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static const size_t WSIZE[10] = {16 * 392, 16 * 144, 16 * 144, 16 * 144, 16 * 144, 32 * 16, 32 * 144, 32 * 288, 32 * 288, 32 * 288};

static const bf16* wrow[8][3];      // [cout][K] copies of the fragment-order weights (CPU check)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, G = argc > 2 ? atoi(argv[2]) : 6, S = argc > 3 ? atoi(argv[3]) : 128;
  const int u8 = argc > 4 ? atoi(argv[4]) : 0;
  HeadArgs a = {};
  a.S = S; a.row_index = nullptr;
  srand(1);
  auto dev = [&](size_t bytes, bool rnd, float scale) {
    void* p; hipMalloc(&p, bytes);
    std::vector<unsigned short> h(bytes / 2);
    for (auto& v : h) { float f = rnd ? scale * ((rand() % 2001) / 1000.f - 1.f) : 0.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
    return p;
  };
  auto devf = [&](size_t n, float base, float scale) {
    void* p; hipMalloc(&p, n * 4);
    std::vector<float> h(n);
    for (auto& v : h) v = base + scale * ((rand() % 2001) / 1000.f - 1.f);
    hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return (float*)p;
  };
  for (int g = 0; g < G; g++) {
    HeadTower& t = a.t[g];
    t.C = (g & 1) ? 1 : 3; t.u8 = u8 && t.C == 3; t.div = t.C == 3 ? 255.f : 1.f;
    const size_t n = (size_t)B * S * S * t.C;
    if (t.u8) { void* p; hipMalloc(&p, n); hipMemset(p, 77, n); t.img = p; }
    else t.img = devf(n, t.C == 3 ? 128.f : 0.5f, t.C == 3 ? 100.f : 0.4f);
    for (int i = 0; i < 10; i++) {
      t.w[i] = (const bf16*)dev(WSIZE[i] * 2, true, i == 0 ? 0.1f : i == 5 ? 0.3f : i < 7 ? 0.12f : 0.08f);
      if (i >= 7) {                                       // the 32 -> 32 convs are read in fragment order (avlen_conv::w16f)
        std::vector<unsigned short> w(WSIZE[i]), wf(WSIZE[i]);
        hipMemcpy(w.data(), t.w[i], w.size() * 2, hipMemcpyDeviceToHost);
        for (int tt = 0; tt < 2; tt++) for (int ii = 0; ii < 9; ii++) for (int ln = 0; ln < 64; ln++) for (int e = 0; e < 8; e++)
          wf[(((size_t)tt * 9 + ii) * 64 + ln) * 8 + e] = w[(size_t)(tt * 16 + (ln & 15)) * 288 + ii * 32 + 8 * (ln >> 4) + e];
        void* d; hipMalloc(&d, wf.size() * 2); hipMemcpy(d, wf.data(), wf.size() * 2, hipMemcpyHostToDevice);
        wrow[g][i - 7] = t.w[i]; t.w[i] = (const bf16*)d;
      }
      const int nch = i < 5 ? 16 : 32;
      t.g[i] = devf(nch, 1.f, 0.2f); t.b[i] = devf(nch, 0.f, 0.2f);
    }
    void* y; hipMalloc(&y, (size_t)B * 1024 * 32 * 2); t.y = (bf16*)y;
  }
  {
    float* d; hipMalloc(&d, 64 * 4); float h[64];
    hipLaunchKernelGGL(dpp_selftest, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; l++) { const int r = l / 16; ok = ok && h[l] == (float)(16 * (16 * r) + 120); }
    printf("row16_sum self-test: %s (%g %g %g %g)\n", ok ? "ok" : "WRONG", h[0], h[16], h[37], h[63]);
  }
  long long* prof; CK(hipMalloc(&prof, (size_t)B * G * 32 * 8));
  a.prof = prof;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HEAD_LDS));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(tower_head_kernel, dim3(B, G), dim3(HTH), HEAD_LDS, 0, a, B);
  CK(hipDeviceSynchronize());
  const int IT = 20;
  hipEventRecord(e0);
  for (int it = 0; it < IT; it++) hipLaunchKernelGGL(tower_head_kernel, dim3(B, G), dim3(HTH), HEAD_LDS, 0, a, B);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("B=%d G=%d S=%d u8=%d: %.1f us per launch (%d workgroups)\n", B, G, S, u8, ms * 1000 / IT, B * G);
#ifdef AVLEN_HEAD_PROF
  std::vector<long long> h((size_t)B * G * 32);
  hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
  const int NP = 19;
  const char* names[NP] = {"preprocess", "stem conv", "stem stats", "a0 write", "conv1", "stats1", "apply1+conv2", "stats2", "apply2+conv3",
                           "stats3", "apply3+conv4", "stats4", "apply4", "l2 downsample", "l2 conv s2", "l2 conv2", "l2 conv3", "l2 conv4", "store"};
  double tot = 0; std::vector<double> ph(NP, 0.0);
  for (int w = 0; w < B * G; w++)
    for (int k = 0; k < NP; k++) ph[k] += (double)(h[w * 32 + k + 1] - h[w * 32 + k]);
  for (int k = 0; k < NP; k++) tot += ph[k];
  for (int k = 0; k < NP; k++) printf("  %-14s %6.1f %%  (%.0f ticks)\n", names[k], 100 * ph[k] / tot, ph[k] / (B * G));
  printf("  workgroup total %.0f ticks\n", tot / (B * G));
#endif
  // tower 0, image 0 against the CPU restatement
  if (!a.t[0].u8) {
    const HeadTower& t = a.t[0];
    std::vector<float> img((size_t)S * S * t.C); hipMemcpy(img.data(), t.img, img.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned short> w[10]; std::vector<float> gm[10], bt[10];
    for (int i = 0; i < 10; i++) {
      w[i].resize(WSIZE[i]); hipMemcpy(w[i].data(), i >= 7 ? wrow[0][i - 7] : t.w[i], w[i].size() * 2, hipMemcpyDeviceToHost);
      const int nch = i < 5 ? 16 : 32;
      gm[i].resize(nch); bt[i].resize(nch);
      hipMemcpy(gm[i].data(), t.g[i], nch * 4, hipMemcpyDeviceToHost); hipMemcpy(bt[i].data(), t.b[i], nch * 4, hipMemcpyDeviceToHost);
    }
    std::vector<float> ref; cpu_head(img, S, t.C, t.div, w, gm, bt, ref);
    std::vector<unsigned short> yy((size_t)1024 * 32); hipMemcpy(yy.data(), t.y, yy.size() * 2, hipMemcpyDeviceToHost);
    double mx = 0, sm = 0, rf = 0; int bad = 0;
    for (size_t i = 0; i < ref.size(); i++) { const double d = fabs((double)bf_to_f(yy[i]) - ref[i]); mx = d > mx ? d : mx; sm += d; rf += fabs(ref[i]); bad += d > 0.05; }
    printf("  vs CPU restatement: max |d| %.4f, mean |d| %.5f (mean |ref| %.4f), %d of %zu beyond 0.05\n", mx, sm / ref.size(), rf / ref.size(), bad, ref.size());
  }
  // sanity: output finite and non-trivial
  std::vector<unsigned short> y((size_t)1024 * 32);
  hipMemcpy(y.data(), a.t[0].y, y.size() * 2, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : y) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); sum += f; }
  printf("  mean of tower 0 image 0 output: %.5f\n", sum / y.size());
  return 0;
}
