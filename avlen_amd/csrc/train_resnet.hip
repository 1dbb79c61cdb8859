// Training path of the GroupNorm ResNet-18 (`CustomResNet`, smt_resnet.py:56-149; blocks :37-53): forward with every
// activation and GroupNorm statistic kept, and the backward into canonical-layout gradient tensors.  Two users:
//   * BeliefPredictor's online regression (`train_belief_predictor`, ppo_trainer.py:959-1030; predictor = custom_resnet18 at
//     the 65x26 spectrogram, fc 4608 -> 2), which the interactive trainer runs after every PPO update (ddppo_trainer.py:977-978);
//   * the visual towers of pi_l under `PPO.update_dialog` (ppo.py:99-154), whose loss back-propagates into the encoders.
// Generic in the input extent (H, W, C): the spatial sizes follow from the strides.
//
// Convolutions: forward on the implicit-GEMM kernel of the fp32-staged path (avlen_conv2d_nhwc); backward as
// dW = dY^T im2col(X), dX = col2im(dY W) on the training GEMMs (train_kernels.h), in sample chunks so that the im2col buffer
// stays bounded (the chunks accumulate into the same gradient).  GroupNorm forward / backward: one block per sample
// (train_kernels.h), the ReLU masks are applied inside the GroupNorm backward that follows them.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include <math.h>

#define TRY(x) do { int _rc = (x); if (_rc != AVLEN_OK) return _rc; } while (0)
#include "train_kernels.h"

namespace {

constexpr size_t GEMM_SCRATCH = 96u << 20;
constexpr size_t COLS_MAX = (size_t)192 << 20;       // floats: im2col chunk buffer (768 MB)
inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

struct Geo { int h[9], w[9], c[9]; };                // [0] = after the stem, [i+1] = after block i
Geo geometry(const avlen_resnet18* n, int H, int W) {
  Geo g;
  const avlen_conv& k = n->conv1;
  g.h[0] = (H + 2 * k.pad - k.kh) / k.stride + 1; g.w[0] = (W + 2 * k.pad - k.kw) / k.stride + 1; g.c[0] = k.cout;
  for (int i = 0; i < 8; i++) {
    const avlen_conv& a = n->block[i].conv1;
    g.h[i + 1] = (g.h[i] + 2 * a.pad - a.kh) / a.stride + 1;
    g.w[i + 1] = (g.w[i] + 2 * a.pad - a.kw) / a.stride + 1;
    g.c[i + 1] = a.cout;
  }
  return g;
}

struct BlkWs { float *c1, *s1, *t1, *c2, *s2, *cd, *sd, *nd, *out; };
struct Ws {
  float *c0, *s0, *a0;
  BlkWs b[8];
  float *dA, *dB, *dT, *cols, *gpack;
  void* gws; void* xs; size_t xs_bytes;
};

bool layout(WsBump& w, Ws& s, const avlen_resnet18* n, long B, int H, int W, int prec) {
  const Geo g = geometry(n, H, W);
  const size_t a0 = (size_t)B * g.h[0] * g.w[0] * g.c[0];
  s.c0 = w.take<float>(a0); s.s0 = w.take<float>((size_t)B * 16 * 2); s.a0 = w.take<float>(a0);
  size_t amax = a0, gmax = (size_t)n->fc.out_f * n->fc.in_f;
  gmax = zmax(gmax, (size_t)n->conv1.cout * n->conv1.kh * n->conv1.kw * n->conv1.cin);
  for (int i = 0; i < 8; i++) {
    const size_t e = (size_t)B * g.h[i + 1] * g.w[i + 1] * g.c[i + 1];
    BlkWs& b = s.b[i];
    b.c1 = w.take<float>(e); b.s1 = w.take<float>((size_t)B * 32); b.t1 = w.take<float>(e);
    b.c2 = w.take<float>(e); b.s2 = w.take<float>((size_t)B * 32);
    if (n->block[i].has_down) { b.cd = w.take<float>(e); b.sd = w.take<float>((size_t)B * 32); b.nd = w.take<float>(e); }
    else { b.cd = b.sd = b.nd = nullptr; }
    b.out = w.take<float>(e);
    amax = zmax(amax, e);
    gmax = zmax(gmax, (size_t)g.c[i + 1] * 9 * g.c[i + 1]);
  }
  s.dA = w.take<float>(amax); s.dB = w.take<float>(amax); s.dT = w.take<float>(amax);
  s.cols = w.take<float>(COLS_MAX);
  s.gpack = w.take<float>(gmax);
  s.gws = w.take<char>(GEMM_SCRATCH);
  s.xs = nullptr; s.xs_bytes = 0;
  if (prec == AVLEN_PREC_BF16) {
    s.xs_bytes = (COLS_MAX + COLS_MAX / 2 + ((size_t)16 << 20)) * 2;
    s.xs = w.take<char>(s.xs_bytes);
  }
  return w.ok();
}

int gn_fwd(hipStream_t st, const float* x, const avlen_affine& a, const float* res, float* y, float* stats, long B, int HW, int C,
           int relu) {
  if (256 % C || C > 128) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(gn_train_fwd_kernel, dim3((unsigned)B), dim3(256), 0, st, x, a.g, a.b, res, y, stats, HW, C, 16, relu, 1e-5f);
  return avlen_launch_status();
}
int gn_bwd(hipStream_t st, const float* dy, const float* relu_y, const float* x, const float* stats, const avlen_affine& a,
           const avlen_affine& ga, float* dx, long B, int HW, int C) {
  hipLaunchKernelGGL(gn_train_bwd_kernel, dim3((unsigned)B), dim3(256), 0, st, dy, relu_y, x, stats, a.g, dx, ga.g, ga.b, HW, C, 16);
  return avlen_launch_status();
}

// conv backward in sample chunks.  x (B,H,W,Cin) input, dy (B,OH,OW,Cout); gw: canonical OIHW gradient (accumulated);
// dx (optional): input gradient, written (accumulate = 0) or added to (1); relu_in (optional): mask by (relu_in > 0)
int conv_bwd(const avlen_ctx& c, Ws& s, const avlen_conv& k, float* gw, const float* x, const float* dy, float* dx,
             const float* relu_in, int accumulate, long B, int H, int W, int OH, int OW) {
  const int Kc = k.kh * k.kw * k.cin;
  const long per = (long)OH * OW * Kc;
  long chunk = (long)(COLS_MAX / (size_t)per);
  if (chunk < 1) return AVLEN_ERR_WS;
  if (chunk > B) chunk = B;
  TRY(avlen_zero_bytes(s.gpack, (size_t)k.cout * Kc * 4, c.st));
  avlen_linear G{s.gpack, nullptr, k.cout, Kc, nullptr, 0};
  avlen_linear Wl{k.w, nullptr, k.cout, Kc, nullptr, 0};
  const bool pointwise = k.kh == 1 && k.kw == 1 && k.stride == 1 && k.pad == 0;
  for (long b0 = 0; b0 < B; b0 += chunk) {
    const long nb = (B - b0 < chunk) ? B - b0 : chunk;
    const long M = nb * OH * OW;
    if (M > 0x7fffffffL) return AVLEN_ERR_ARG;
    const float* xc = x + b0 * H * W * k.cin;
    const float* dyc = dy + b0 * OH * OW * k.cout;
    int rc = pointwise ? AVLEN_NOT_BIG : avlen_i_conv_dw16(c, G, dyc, k.cout, xc, nb, H, W, k.cin, OH, OW, k.kh, k.kw, k.stride, k.pad);
    if (rc == AVLEN_NOT_BIG) {
      const float* cols = xc;
      if (!pointwise) { TRY(im2col(c.st, xc, s.cols, nb, H, W, k.cin, OH, OW, k.kh, k.kw, k.stride, k.pad)); cols = s.cols; }
      TRY(avlen_i_linear_dw(c, G, dyc, k.cout, cols, Kc, (int)M));
    } else TRY(rc);
    if (dx) {
      float* dxc = dx + b0 * H * W * k.cin;
      TRY(avlen_i_linear_dx(c, Wl, dyc, k.cout, s.cols, Kc, (int)M, nullptr, 0));
      TRY(col2im(c.st, s.cols, relu_in ? relu_in + b0 * H * W * k.cin : nullptr, dxc, nb, H, W, k.cin, OH, OW, k.kh, k.kw, k.stride,
                 k.pad, accumulate));
    }
  }
  const long nw = (long)k.cout * Kc;
  hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c.st, s.gpack, gw, k.cout, k.cin, k.kh,
                     k.kw);
  return avlen_launch_status();
}

// d += (y > 0) ? g : 0
__global__ void add_masked_kernel(float* __restrict__ d, const float* __restrict__ g, const float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && y[i] > 0.f) d[i] += g[i];
}

// train_belief_predictor's loss (ppo_trainer.py:1000-1008, 1017-1022): mask = (sum of the row's spectrogram != 0);
// loss = mean over all R x 2 elements of (mask*pred - mask*gt')^2 with gt' = (gt[1], -gt[0]);  d_pred = 2 mask (pred - gt') / (2R);
// acc[0] += loss, acc[1] += #rows with mask and round(pred) close to gt' in both coordinates, acc[2] += #rows with mask
__global__ __launch_bounds__(256) void belief_reg_loss_kernel(const float* __restrict__ preds, const float* __restrict__ spec,
                                                              long spec_elems, const float* __restrict__ gts, int ld_gt,
                                                              float* __restrict__ d_preds, float* __restrict__ acc, int R) {
  __shared__ float sh[16];
  const int r = blockIdx.x;
  const float* sp = spec + (long)r * spec_elems;
  float s = 0.f;
  for (long i = threadIdx.x; i < spec_elems; i += 256) s += sp[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float m = s != 0.f ? 1.f : 0.f;
    const float g0 = gts[(long)r * ld_gt + 1], g1 = -gts[(long)r * ld_gt];
    const float p0 = preds[r * 2], p1 = preds[r * 2 + 1];
    const float e0 = m * p0 - m * g0, e1 = m * p1 - m * g1;
    const float inv = 1.f / (2.f * (float)R);
    d_preds[r * 2] = 2.f * m * e0 * inv; d_preds[r * 2 + 1] = 2.f * m * e1 * inv;
    atomicAdd(&acc[0], (e0 * e0 + e1 * e1) * inv);
    auto close = [](float a, float b) { return fabsf(a - b) <= 1e-8f + 1e-5f * fabsf(b); };      // torch.isclose defaults
    if (m != 0.f) {
      atomicAdd(&acc[2], 1.f);
      if (close(rintf(p0), g0) && close(rintf(p1), g1)) atomicAdd(&acc[1], 1.f);
    }
  }
}

}  // namespace

extern "C" int avlen_belief_regression_loss(const float* preds, const float* spec, long spec_elems, const float* gts, int ld_gt,
                                            float* d_preds, float* acc, int R, hipStream_t st) {
  if (!preds || !spec || !gts || !d_preds || !acc || R <= 0) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(belief_reg_loss_kernel, dim3(R), dim3(256), 0, st, preds, spec, spec_elems, gts, ld_gt, d_preds, acc, R);
  return avlen_launch_status();
}

extern "C" size_t avlen_resnet18_train_workspace_bytes(const avlen_resnet18* net, int B, int H, int W, int prec) {
  WsBump w(nullptr, 0);
  Ws s;
  layout(w, s, net, B, H, W, prec);
  return w.off + 4096;
}

extern "C" int avlen_resnet18_train_fwd(const avlen_resnet18* n, const float* x, int B, int H, int W, float* out, int ld_out, int prec,
                                        void* ws, size_t ws_bytes, hipStream_t st) {
  if (!n || !x || B <= 0) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes);
  Ws s;
  if (!ws || !layout(w, s, n, B, H, W, prec)) return AVLEN_ERR_WS;
  const Geo g = geometry(n, H, W);
  if (n->fc.in_f != g.h[8] * g.w[8] * g.c[8]) return AVLEN_ERR_ARG;
  avlen_ctx c{st, prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  const avlen_conv& k0 = n->conv1;
  TRY(avlen_conv2d_nhwc(x, k0.w, nullptr, nullptr, s.c0, B, H, W, k0.cin, k0.cout, k0.kh, k0.kw, k0.stride, k0.pad, 0, prec, st));
  TRY(gn_fwd(st, s.c0, n->bn1, nullptr, s.a0, s.s0, B, g.h[0] * g.w[0], g.c[0], 1));
  const float* cur = s.a0;
  for (int i = 0; i < 8; i++) {
    const avlen_resblock& k = n->block[i];
    BlkWs& b = s.b[i];
    const int Hi = g.h[i], Wi = g.w[i], Ci = g.c[i], Ho = g.h[i + 1], Wo = g.w[i + 1], Co = g.c[i + 1];
    TRY(avlen_conv2d_nhwc(cur, k.conv1.w, nullptr, nullptr, b.c1, B, Hi, Wi, Ci, Co, 3, 3, k.conv1.stride, 1, 0, prec, st));
    TRY(gn_fwd(st, b.c1, k.bn1, nullptr, b.t1, b.s1, B, Ho * Wo, Co, 1));
    TRY(avlen_conv2d_nhwc(b.t1, k.conv2.w, nullptr, nullptr, b.c2, B, Ho, Wo, Co, Co, 3, 3, 1, 1, 0, prec, st));
    const float* idt = cur;
    if (k.has_down) {
      TRY(avlen_conv2d_nhwc(cur, k.down.w, nullptr, nullptr, b.cd, B, Hi, Wi, Ci, Co, 1, 1, k.down.stride, 0, 0, prec, st));
      TRY(gn_fwd(st, b.cd, k.bnd, nullptr, b.nd, b.sd, B, Ho * Wo, Co, 0));
      idt = b.nd;
    }
    TRY(gn_fwd(st, b.c2, k.bn2, idt, b.out, b.s2, B, Ho * Wo, Co, 1));
    cur = b.out;
  }
  return avlen_i_linear(c, n->fc, cur, n->fc.in_f, out, ld_out, B, 0, nullptr, 0);
}

extern "C" int avlen_resnet18_train_bwd(const avlen_resnet18* n, const avlen_resnet18* gr, const float* x, const float* d_out,
                                        int ld_dout, int B, int H, int W, float* d_x, int prec, void* ws, size_t ws_bytes,
                                        hipStream_t st) {
  if (!n || !gr || !x || !d_out || B <= 0) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes);
  Ws s;
  if (!ws || !layout(w, s, n, B, H, W, prec)) return AVLEN_ERR_WS;
  const Geo g = geometry(n, H, W);
  avlen_ctx c{st, prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  // ---- fc
  const int O = n->fc.out_f, K = n->fc.in_f;
  float* dcur = s.dA; float* dnext = s.dB;
  {
    TRY(avlen_zero_bytes(s.gpack, (size_t)O * K * 4, st));
    avlen_linear G = n->fc; G.w = s.gpack; G.b = nullptr; G.w16 = nullptr;
    TRY(avlen_i_linear_dw(c, G, d_out, ld_dout, s.b[7].out, K, B));
    const long nw = (long)O * K;
    hipLaunchKernelGGL(unpack_fc_grad_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, s.gpack, gr->fc.w, O, g.c[8],
                       g.h[8] * g.w[8]);
    TRY(avlen_i_colsum_acc(c, d_out, ld_dout, gr->fc.b, B, O));
    TRY(avlen_i_linear_dx(c, n->fc, d_out, ld_dout, dcur, K, B, nullptr, 0));
  }
  // ---- blocks, last to first.  dcur = gradient w.r.t. the block's (post-ReLU) output
  for (int i = 7; i >= 0; i--) {
    const avlen_resblock& k = n->block[i];
    const avlen_resblock& gk = gr->block[i];
    BlkWs& b = s.b[i];
    const float* in = i ? s.b[i - 1].out : s.a0;
    const int Hi = g.h[i], Wi = g.w[i], Ho = g.h[i + 1], Wo = g.w[i + 1], Co = g.c[i + 1];
    const long HWo = (long)Ho * Wo;
    // main branch: out = relu(GN2(c2) + idt)
    TRY(gn_bwd(st, dcur, b.out, b.c2, b.s2, k.bn2, gk.bn2, s.dT, B, (int)HWo, Co));                         // d c2
    TRY(conv_bwd(c, s, k.conv2, gk.conv2.w, b.t1, s.dT, s.dT, nullptr, 0, B, Ho, Wo, Ho, Wo));              // d t1 (in place: see below)
    TRY(gn_bwd(st, s.dT, b.t1, b.c1, b.s1, k.bn1, gk.bn1, s.dT, B, (int)HWo, Co));                          // d c1 (ReLU mask of t1 inside)
    TRY(conv_bwd(c, s, k.conv1, gk.conv1.w, in, s.dT, dnext, nullptr, 0, B, Hi, Wi, Ho, Wo));               // d in  (=)
    // identity branch
    if (k.has_down) {
      TRY(gn_bwd(st, dcur, b.out, b.cd, b.sd, k.bnd, gk.bnd, s.dT, B, (int)HWo, Co));                       // d cd
      TRY(conv_bwd(c, s, k.down, gk.down.w, in, s.dT, dnext, nullptr, 1, B, Hi, Wi, Ho, Wo));               // d in (+=)
    } else {
      const long ne = (long)B * HWo * Co;
      hipLaunchKernelGGL(add_masked_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, dnext, dcur, b.out, ne);
    }
    float* t = dcur; dcur = dnext; dnext = t;
  }
  // ---- stem: a0 = relu(GN(c0)), c0 = conv1(x)
  TRY(gn_bwd(st, dcur, s.a0, s.c0, s.s0, n->bn1, gr->bn1, s.dT, B, g.h[0] * g.w[0], g.c[0]));
  TRY(conv_bwd(c, s, n->conv1, gr->conv1.w, x, s.dT, d_x, nullptr, 0, B, H, W, g.h[0], g.w[0]));
  return avlen_launch_status();
}
