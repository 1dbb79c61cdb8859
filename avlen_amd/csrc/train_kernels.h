// Kernels shared by the training paths (train_gru.hip: GRU baseline; train_resnet.hip: GroupNorm ResNet-18 towers / belief
// predictor).  Header-only, internal linkage.  NHWC activations, fp32.
//
// Convolution backward = two GEMMs around im2col / col2im:
//   dW_packed[O][KH][KW][I] += dY^T * im2col(X)        dX = col2im(dY * W_packed)
// im2col writes the [kh][kw][c] K-order of the packed weights; without padding a (kw, c) run is contiguous in NHWC and is copied in
// 16- or 8-byte pieces; with padding every (kh, kw) tap is tested.  col2im is a GATHER (one thread per input element sums the taps
// that touched it): no atomics, deterministic; optionally masks with the ReLU of the layer below and / or accumulates.
#pragma once
#include "common.h"

namespace {

template <int V>
__global__ void im2col_kernel(const float* __restrict__ X, float* __restrict__ cols, long total, int H, int W, int C, int OH, int OW,
                              int KH, int KW, int s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int run = KW * C / V;                    // V-float pieces per (m, kh)
  const int j = (int)(i % run);
  long r = i / run;
  const int kh = (int)(r % KH); r /= KH;         // r = m
  const int ow = (int)(r % OW); long q = r / OW;
  const int oh = (int)(q % OH); const long b = q / OH;
  const float* src = X + (((b * H + (long)oh * s + kh) * W + (long)ow * s) * C) + (long)j * V;
  float* dst = cols + (r * KH + kh) * (long)(KW * C) + (long)j * V;
  if (V == 4) *(float4*)dst = *(const float4*)src;
  else if (V == 2) *(float2*)dst = *(const float2*)src;
  else dst[0] = src[0];
}
// padded variant: one thread per V channels of one tap
template <int V>
__global__ void im2col_pad_kernel(const float* __restrict__ X, float* __restrict__ cols, long total, int H, int W, int C, int OH,
                                  int OW, int KH, int KW, int s, int pad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cv = C / V;
  const int c = (int)(i % cv) * V;
  long r = i / cv;
  const int kw = (int)(r % KW); r /= KW;
  const int kh = (int)(r % KH); r /= KH;         // r = m
  const int ow = (int)(r % OW); long q = r / OW;
  const int oh = (int)(q % OH); const long b = q / OH;
  const int ih = oh * s + kh - pad, iw = ow * s + kw - pad;
  const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
  float* dst = cols + ((r * KH + kh) * KW + kw) * (long)C + c;
  const float* src = X + ((b * H + ih) * W + iw) * C + c;
  if (V == 4) *(float4*)dst = ok ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
  else dst[0] = ok ? src[0] : 0.f;
}
inline int im2col(hipStream_t st, const float* X, float* cols, long B, int H, int W, int C, int OH, int OW, int KH, int KW, int s,
                  int pad) {
  if (pad) {
    const int V = (C % 4 == 0 && (((uintptr_t)X | (uintptr_t)cols) & 15) == 0) ? 4 : 1;
    const long total = B * OH * OW * KH * KW * (long)(C / V);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (V == 4) hipLaunchKernelGGL(im2col_pad_kernel<4>, dim3(blocks), dim3(256), 0, st, X, cols, total, H, W, C, OH, OW, KH, KW, s, pad);
    else hipLaunchKernelGGL(im2col_pad_kernel<1>, dim3(blocks), dim3(256), 0, st, X, cols, total, H, W, C, OH, OW, KH, KW, s, pad);
    return avlen_launch_status();
  }
  // a (kw, c) run starts at float offset ((b*H + ih)*W + ow*s)*C: V-float pieces need W*C, s*C and KW*C to be multiples of V
  auto ok = [&](int v) { return (KW * C) % v == 0 && (W * C) % v == 0 && (s * C) % v == 0 &&
                                (((uintptr_t)X | (uintptr_t)cols) & (size_t)(4 * v - 1)) == 0; };
  const int V = ok(4) ? 4 : ok(2) ? 2 : 1;
  const long total = B * OH * OW * KH * (long)(KW * C / V);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (V == 4) hipLaunchKernelGGL(im2col_kernel<4>, dim3(blocks), dim3(256), 0, st, X, cols, total, H, W, C, OH, OW, KH, KW, s);
  else if (V == 2) hipLaunchKernelGGL(im2col_kernel<2>, dim3(blocks), dim3(256), 0, st, X, cols, total, H, W, C, OH, OW, KH, KW, s);
  else hipLaunchKernelGGL(im2col_kernel<1>, dim3(blocks), dim3(256), 0, st, X, cols, total, H, W, C, OH, OW, KH, KW, s);
  return avlen_launch_status();
}

// dX[b][h][w][c] (=|+=) relu'(act) * sum over taps (kh, kw) with (h + pad - kh) % s == 0, (w + pad - kw) % s == 0, in range, of
// dcols[(b, (h+pad-kh)/s, (w+pad-kw)/s)][kh][kw][c].  act (optional) = the post-ReLU activation that was this conv's input.
template <int V>
__global__ void col2im_kernel(const float* __restrict__ dcols, const float* __restrict__ act, float* __restrict__ dX, long total,
                              int H, int W, int C, int OH, int OW, int KH, int KW, int s, int pad, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cv = C / V;
  const int c = (int)(i % cv) * V;
  long r = i / cv;
  const int w = (int)(r % W); r /= W;
  const int h = (int)(r % H); const long b = r / H;
  float acc[V];
#pragma unroll
  for (int v = 0; v < V; v++) acc[v] = 0.f;
  const long K = (long)KH * KW * C;
  const int hp = h + pad, wp = w + pad;
  for (int kh = hp % s; kh < KH; kh += s) {
    if (hp - kh < 0) break;
    const int oh = (hp - kh) / s;
    if (oh >= OH) continue;
    for (int kw = wp % s; kw < KW; kw += s) {
      if (wp - kw < 0) break;
      const int ow = (wp - kw) / s;
      if (ow >= OW) continue;
      const float* p = dcols + ((b * OH + oh) * OW + ow) * K + ((long)kh * KW + kw) * C + c;
      if (V == 4) { const float4 t = *(const float4*)p; acc[0] += t.x; acc[1 % V] += t.y; acc[2 % V] += t.z; acc[3 % V] += t.w; }
      else acc[0] += p[0];
    }
  }
  const long o = ((b * H + h) * W + w) * C + c;
#pragma unroll
  for (int v = 0; v < V; v++) {
    float g = (!act || act[o + v] > 0.f) ? acc[v] : 0.f;
    dX[o + v] = accumulate ? dX[o + v] + g : g;
  }
}
inline int col2im(hipStream_t st, const float* dcols, const float* act, float* dX, long B, int H, int W, int C, int OH, int OW, int KH,
                  int KW, int s, int pad, int accumulate) {
  const bool v4 = C % 4 == 0;
  const long total = B * H * W * (C / (v4 ? 4 : 1));
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (v4) hipLaunchKernelGGL(col2im_kernel<4>, dim3(blocks), dim3(256), 0, st, dcols, act, dX, total, H, W, C, OH, OW, KH, KW, s, pad, accumulate);
  else hipLaunchKernelGGL(col2im_kernel<1>, dim3(blocks), dim3(256), 0, st, dcols, act, dX, total, H, W, C, OH, OW, KH, KW, s, pad, accumulate);
  return avlen_launch_status();
}
inline int col2im_relu(hipStream_t st, const float* dcols, const float* act, float* dX, long B, int H, int W, int C, int OH, int OW,
                       int KH, int KW, int s, int pad) {
  return col2im(st, dcols, act, dX, B, H, W, C, OH, OW, KH, KW, s, pad, 0);
}

// dst[i] = (y[i] > 0) ? src[i] : 0 over rows of different strides (a Linear+ReLU output that lives inside wider rows)
__global__ void relu_mask_rows_kernel(const float* __restrict__ src, int lds, const float* __restrict__ y, int ldy, float* __restrict__ dst,
                                      int ldd, long rows, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols; const int c = (int)(i % cols);
  dst[r * ldd + c] = y[r * ldy + c] > 0.f ? src[r * lds + c] : 0.f;
}

// packed gradient layouts -> canonical parameter layouts (accumulating)
__global__ void unpack_conv_grad_kernel(const float* __restrict__ gp, float* __restrict__ g, int O, int I, int KH, int KW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // canonical index ((o*I + c)*KH + kh)*KW + kw
  if (i >= (long)O * I * KH * KW) return;
  const int kw = (int)(i % KW); long r = i / KW;
  const int kh = (int)(r % KH); r /= KH;
  const int c = (int)(r % I); const int o = (int)(r / I);
  g[i] += gp[(((long)o * KH + kh) * KW + kw) * I + c];
}
__global__ void unpack_fc_grad_kernel(const float* __restrict__ gp, float* __restrict__ g, int O, int C, int HW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // canonical index (o*C + c)*HW + p
  if (i >= (long)O * C * HW) return;
  const int p = (int)(i % HW); long r = i / HW;
  const int c = (int)(r % C); const long o = r / C;
  g[i] += gp[(o * HW + p) * C + c];
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm(G) over NHWC (B, HW, C), training form: statistics kept for the backward.
// One 256-thread block per sample; 256 % C == 0 so a thread always meets the same channel (partials live in registers).
// ---------------------------------------------------------------------------------------------------------------
// stats[b][g] = (mean, rstd) ;  y = relu?( (x - mean) * rstd * gamma + beta (+ residual) )
__global__ __launch_bounds__(256) void gn_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ residual,
                                                           float* __restrict__ y, float* __restrict__ stats, int HW, int C, int G,
                                                           int relu, float eps) {
  __shared__ float s1[256], mu[128], rs[128];
  const int b = blockIdx.x, t = threadIdx.x;
  const long n = (long)HW * C;
  const float* xb = x + (long)b * n;
  const int cg = C / G, c = t % C, g = c / cg;
  const float cnt = (float)HW * cg;
  // two passes (mean, then centred sum of squares): E[x^2] - mean^2 in fp32 loses the variance of groups whose mean dwarfs
  // their spread, and the backward divides by that variance (1 % gradient errors through the towers with the fixture weights)
  float a = 0.f;
  for (long i = t; i < n; i += 256) a += xb[i];
  s1[t] = a;
  __syncthreads();
  if (t < G) {                                     // channels t*cg .. +cg; thread slots with the same channel: k % C == c
    float sa = 0.f;
    for (int cc = t * cg; cc < (t + 1) * cg; cc++)
      for (int k = cc; k < 256; k += C) sa += s1[k];
    mu[t] = sa / cnt;
  }
  __syncthreads();
  const float m = mu[g];
  float q = 0.f;
  for (long i = t; i < n; i += 256) { const float v = xb[i] - m; q += v * v; }
  s1[t] = q;
  __syncthreads();
  if (t < G) {
    float sq = 0.f;
    for (int cc = t * cg; cc < (t + 1) * cg; cc++)
      for (int k = cc; k < 256; k += C) sq += s1[k];
    const float r = rsqrtf(sq / cnt + eps);
    rs[t] = r;
    stats[((long)b * G + t) * 2] = mu[t]; stats[((long)b * G + t) * 2 + 1] = r;
  }
  __syncthreads();
  const float sc = rs[g] * gamma[c], sh = beta[c] - m * sc;
  float* yb = y + (long)b * n;
  const float* rb = residual ? residual + (long)b * n : nullptr;
  for (long i = t; i < n; i += 256) {
    float v = xb[i] * sc + sh;
    if (rb) v += rb[i];
    yb[i] = relu ? fmaxf(v, 0.f) : v;
  }
}
// dy_eff = dy * (relu_y > 0 if relu_y)   ;   dx = rstd * (dy_eff*gamma - m1 - xhat * m2),  m1 = mean_g(dy_eff*gamma),
// m2 = mean_g(dy_eff*gamma*xhat);  dgamma[c] += sum dy_eff*xhat, dbeta[c] += sum dy_eff (atomics over samples).  dx may alias dy.
__global__ __launch_bounds__(256) void gn_train_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ relu_y,
                                                           const float* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int HW, int C,
                                                           int G) {
  __shared__ float s1[256], s2[256], m1[128], m2[128];
  const int b = blockIdx.x, t = threadIdx.x;
  const long n = (long)HW * C;
  const float* xb = x + (long)b * n; const float* db = dy + (long)b * n;
  const float* yb = relu_y ? relu_y + (long)b * n : nullptr;
  const int cg = C / G, c = t % C, g = c / cg;
  const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
  float a = 0.f, q = 0.f;                          // sum dy_eff, sum dy_eff * xhat  (this thread's channel)
  for (long i = t; i < n; i += 256) {
    float d = db[i];
    if (yb && !(yb[i] > 0.f)) d = 0.f;
    a += d; q += d * (xb[i] - mean) * rstd;
  }
  s1[t] = a; s2[t] = q;
  __syncthreads();
  if (t < C) {
    float sa = 0.f, sq = 0.f;
    for (int k = t; k < 256; k += C) { sa += s1[k]; sq += s2[k]; }
    if (dbeta) atomicAdd(&dbeta[t], sa);
    if (dgamma) atomicAdd(&dgamma[t], sq);
    s1[t] = sa * gamma[t]; s2[t] = sq * gamma[t];  // slots 0..C-1 now hold the per-channel sums scaled by gamma
  }
  __syncthreads();
  if (t < G) {
    float sa = 0.f, sq = 0.f;
    for (int k = t * cg; k < (t + 1) * cg; k++) { sa += s1[k]; sq += s2[k]; }
    const float cnt = (float)HW * cg;
    m1[t] = sa / cnt; m2[t] = sq / cnt;
  }
  __syncthreads();
  const float gm = gamma[c], a1 = m1[g], a2 = m2[g];
  float* ob = dx + (long)b * n;
  for (long i = t; i < n; i += 256) {
    float d = db[i];
    if (yb && !(yb[i] > 0.f)) d = 0.f;
    const float xh = (xb[i] - mean) * rstd;
    ob[i] = rstd * (d * gm - a1 - xh * a2);
  }
}

}  // namespace
