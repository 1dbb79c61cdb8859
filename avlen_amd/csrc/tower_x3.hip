// CustomResNet tower (smt_resnet.py:56-149; preprocessing smt_cnn.py:83-93) in COMPENSATED bf16 ("bf16x3", AVLEN_PREC_BF16X3):
// every MFMA operand is a pair hi = bf16(x), lo = bf16(x - hi) and every product is three bf16 MFMAs
// (W_hi X_lo + W_lo X_hi + W_hi X_hi) into one fp32 accumulator -- fp32-grade results (~2^-17 per product) on the matrix cores.
// Raw conv outputs, GroupNorm statistics, the normalisation and the residual stream are fp32.
//
// Pairs double the activation footprint: the 64x64x16 stage (136 KiB as a bf16 LDS image in tower_head.hip) no longer fits a CU's
// LDS, so the tower runs as
//   stem_x3   : preprocessing + 7x7 stem, one workgroup per 8-row band          -> raw0 (fp32, HBM) + statistics
//   c16_x3    : a 3x3 16 -> 16 conv over 8-row bands; the PRODUCER's GroupNorm + ReLU (+ residual) is applied while the halo is
//               staged, so no normalised 64x64x16 activation is ever written      -> raw_k (fp32, HBM) + statistics   (x 4)
//   rest_x3   : layers 2-4 (15 convs, 15 GroupNorms) in ONE launch, one workgroup per image: the stride-2 entry of layer 2 reads
//               the layer-1 output in two half-image passes; from there on the activation is a pair of zero-framed bf16 images
//               in LDS (34x34x32, 18x18x64, 10x10x128), raw outputs and the residual stay in registers as fp32.
// Convs run TAP-OUTER: the weight fragments of one tap (hi + lo, a few KiB per wave) are fetched from L2 one tap ahead and the
// activation fragments are re-read from LDS per tap -- with three MFMAs per fragment pair the matrix pipe, not LDS, is the bound
// (the bf16 kernels are input-row stationary with register-resident weights: 2 x 144 registers would not fit here).
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include "tower_util.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f2;

__device__ __forceinline__ f32x4 mma(const bf16x8& w, const bf16x8& x, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
// acc += W X with W = wh + wl, X = xh + xl (the wl * xl term, 2^-18 relative, is dropped); small terms first
__device__ __forceinline__ f32x4 mma3(const bf16x8& wh, const bf16x8& wl, const bf16x8& xh, const bf16x8& xl, f32x4 c) {
  c = mma(wh, xl, c);
  c = mma(wl, xh, c);
  return mma(wh, xh, c);
}
// four fp32 values -> packed hi and lo bf16 quadruples (8 bytes each)
__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo) {
  const unsigned h0 = pack2((f32x2){v[0], v[1]}), h1 = pack2((f32x2){v[2], v[3]});
  const f32x2 a = {__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
  const f32x2 b = {__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
  hi = make_uint2(h0, h1);
  lo = make_uint2(pack2((f32x2){v[0] - a[0], v[1] - a[1]}), pack2((f32x2){v[2] - b[0], v[3] - b[1]}));
}
__device__ __forceinline__ bf16x8 zero_frag() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; e++) z[e] = (bf16)0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------------------------------
// GroupNorm(16) scale / shift of a channel from the producer's (sum, sum of squares): the arithmetic of tower_head.hip
// (moments combined in double, 1 / sqrtf(var + eps) in fp32)
__device__ __forceinline__ void gn_coef(double sum, double sq, double inv_n, float gamma, float beta, float& sc, float& sh) {
  const double mean = sum * inv_n;
  double var = sq * inv_n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
  sc = gamma * rstd;
  sh = beta - (float)mean * sc;
}

// ========================================================================================================================
// 64 x 64 x 16 stage: band kernels (256 threads, 8 output rows per workgroup)
// ========================================================================================================================
constexpr int BAND = 8, NBAND = 64 / BAND;
// GroupNorm statistics of a 64 x 64 x 16 tensor: every band workgroup WRITES its (sum, sum of squares) per channel -- layout
// [image][band][2][16] -- and the consumer adds the 8 partials in band order: no atomics, bit-reproducible
__device__ __forceinline__ void band_sums(const float* __restrict__ st, int c, double& sum, double& sq) {
  sum = 0.0; sq = 0.0;
#pragma unroll
  for (int k = 0; k < NBAND; k++) { sum += (double)st[k * 32 + c]; sq += (double)st[k * 32 + 16 + c]; }
}
constexpr int HROWS = BAND + 2, HCOLS = 66;                 // 3x3 halo of a band
constexpr int HPLANE = HROWS * HCOLS * 32;                  // bytes of one plane (hi or lo) of the 16-channel halo
__device__ __forceinline__ int h16(int y, int p, int chunk) { return (y * HCOLS + p) * 32 + ((chunk ^ ((p >> 3) & 1)) << 4); }

struct C16Tower {
  const float* x; const float* xst; const float* xg; const float* xb;         // producer's raw output + its GroupNorm
  const float* r; const float* rst; const float* rg; const float* rb;         // optional residual: raw (rst != null -> relu(GN(r))) or materialised
  float* a_out;                                                                 // optional: the staged activation, fp32 (interior rows)
  const bf16* wh; const bf16* wl;                                               // [16][9][16] hi / lo
  float* y; float* yst;                                                         // raw output (B, 64, 64, 16) fp32, statistics [B][2][16]
};
struct C16Args { C16Tower t[8]; };

// in = relu(GN(x) [+ res]) staged as hi / lo halo planes; 3x3 conv 16 -> 16; raw fp32 out + per-(sample, channel) sums
__global__ __launch_bounds__(256) void c16_x3_kernel(C16Args args) {
  __shared__ __attribute__((aligned(16))) char halo[2 * HPLANE];
  __shared__ float s_coef[4][16];                       // scale, shift of x; scale, shift of r
  __shared__ float bst[4][2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int y0 = blockIdx.x * BAND, b = blockIdx.y;
  const C16Tower& t = args.t[blockIdx.z];
  // weight fragments: k-step s = taps 2s, 2s+1 x 16 channels; lane (r16, q): tap 2s + (q >> 1), channels 8 (q & 1) ..
  bf16x8 wh[5], wl[5];
#pragma unroll
  for (int s = 0; s < 5; s++) {
    const int k = 32 * s + 8 * q;
    if (k < 144) {
      wh[s] = *reinterpret_cast<const bf16x8*>(t.wh + (long)r16 * 144 + k);
      wl[s] = *reinterpret_cast<const bf16x8*>(t.wl + (long)r16 * 144 + k);
    } else { wh[s] = zero_frag(); wl[s] = zero_frag(); }
  }
  if (tid < 16) {
    double sum, sq;
    band_sums(t.xst + (long)b * NBAND * 32, tid, sum, sq);
    gn_coef(sum, sq, 1.0 / 4096.0, t.xg[tid], t.xb[tid], s_coef[0][tid], s_coef[1][tid]);
  } else if (tid < 32 && t.r && t.rst) {
    const int c = tid - 16;
    double sum, sq;
    band_sums(t.rst + (long)b * NBAND * 32, c, sum, sq);
    gn_coef(sum, sq, 1.0 / 4096.0, t.rg[c], t.rb[c], s_coef[2][c], s_coef[3][c]);
  }
  __syncthreads();
  // ---- halo staging: 10 rows x 64 pixels x 2 chunks of 8 channels; a thread keeps the same 8 channels on every item
  {
    const int c0 = (tid & 1) * 8;
    float sc[8], sh[8], rsc[8], rsh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { sc[i] = s_coef[0][c0 + i]; sh[i] = s_coef[1][c0 + i]; rsc[i] = s_coef[2][c0 + i]; rsh[i] = s_coef[3][c0 + i]; }
    const bool has_r = t.r != nullptr, r_raw = t.rst != nullptr;
    const float* __restrict__ x = t.x + (long)b * 4096 * 16;
    const float* __restrict__ r = has_r ? t.r + (long)b * 4096 * 16 : nullptr;
    float* __restrict__ ao = t.a_out ? t.a_out + (long)b * 4096 * 16 : nullptr;
#pragma unroll
    for (int i = tid; i < HROWS * 128; i += 256) {          // 5 rounds: every load of the halo in flight together
      const int hr = i >> 7, px = (i >> 1) & 63, ch = i & 1;
      const int iy = y0 - 1 + hr;
      uint2 h0 = make_uint2(0u, 0u), h1 = h0, l0 = h0, l1 = h0;
      if (iy >= 0 && iy < 64) {
        const long off = ((long)iy * 64 + px) * 16 + c0;
        const float4 a = *reinterpret_cast<const float4*>(x + off), c = *reinterpret_cast<const float4*>(x + off + 4);
        float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = v[e] * sc[e] + sh[e];
        if (has_r) {
          const float4 ra = *reinterpret_cast<const float4*>(r + off), rc = *reinterpret_cast<const float4*>(r + off + 4);
          float rv[8] = {ra.x, ra.y, ra.z, ra.w, rc.x, rc.y, rc.z, rc.w};
          if (r_raw) {
#pragma unroll
            for (int e = 0; e < 8; e++) rv[e] = fmaxf(rv[e] * rsc[e] + rsh[e], 0.f);
          }
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] += rv[e];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
        if (ao && hr >= 1 && hr <= BAND) {
          *reinterpret_cast<float4*>(ao + off) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(ao + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
        split4(va, h0, l0); split4(vb, h1, l1);
      }
      const int ad = h16(hr, px + 1, ch);
      *reinterpret_cast<uint4*>(halo + ad) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(halo + HPLANE + ad) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    for (int i = tid; i < HROWS * 4; i += 256) {          // zero columns 0 and 65 (2 chunks each), both planes
      const int hr = i >> 2, side = (i >> 1) & 1, ch = i & 1;
      const int ad = h16(hr, side ? 65 : 0, ch);
      *reinterpret_cast<uint4*>(halo + ad) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(halo + HPLANE + ad) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  // ---- conv: wave owns rows 2 wave, 2 wave + 1 of the band x 4 column tiles
  int rd[5];
#pragma unroll
  for (int s = 0; s < 5; s++) {
    int tap = 2 * s + (q >> 1);
    if (tap > 8) tap = 8;                                // zero weights there
    const int ky = tap / 3, kx = tap - ky * 3;
    rd[s] = h16(2 * wave + ky, r16 + kx, q & 1);
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float* __restrict__ y = t.y + (long)b * 4096 * 16;
#pragma unroll
  for (int rr = 0; rr < 2; rr++)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; s++) {
        const int off = rd[s] + (rr * HCOLS + mt * 16) * 32;
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(halo + off), xl = *reinterpret_cast<const bf16x8*>(halo + HPLANE + off);
        acc = mma3(wh[s], wl[s], xh, xl, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) { s1[r] += acc[r]; s2[r] = __builtin_fmaf(acc[r], acc[r], s2[r]); }
      const int oy = y0 + 2 * wave + rr, ox = mt * 16 + r16;
      *reinterpret_cast<float4*>(y + ((long)oy * 64 + ox) * 16 + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  // ---- statistics: 16-lane row sums, the block's 4 waves in LDS, one atomic per (statistic, channel) and block
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float a = row16_sum(s1[r]), c = row16_sum(s2[r]);
    if (r16 == 0) { bst[wave][0][q * 4 + r] = a; bst[wave][1][q * 4 + r] = c; }
  }
  __syncthreads();
  if (tid < 32) {
    const int which = tid >> 4, ch = tid & 15;
    const float v = (bst[0][which][ch] + bst[1][which][ch]) + (bst[2][which][ch] + bst[3][which][ch]);
    t.yst[(((long)b * NBAND + blockIdx.x) * 2 + which) * 16 + ch] = v;
  }
}

// ---- stem: preprocessing (x / div, k x k mean) + 7x7 conv, C (<= 4) -> 16 channels
constexpr int SROWS = BAND + 6, SCOLS = 70;
constexpr int SPLANE = (SROWS * SCOLS + 2) * 8;            // 4 channels x bf16 per pixel (+ 2: the zero tap of the last tile reads past the row)
struct StemTower { const void* img; int u8; int C; float div; const bf16* wh; const bf16* wl; float* y; float* yst; };
struct StemArgs { StemTower t[8]; const int* row_index; int S; };

// The common sensor shapes (128 x 128 -> 64 x 64, rgb or depth): the K * C inputs of one source row of an output pixel are contiguous
// and even in number -> vector loads, all of a pixel's loads in flight together.  Same arithmetic as the generic loop: each element
// divided by `div`, summed in (dy, dx) order, scaled by 1 / K^2.
template <int K, int C, typename T>
__device__ __forceinline__ void stem_fill(const T* __restrict__ img, float div, float inv, char* halo, int y0, int tid) {
  constexpr int S = 64 * K, E = K * C;
  static_assert(E % 2 == 0 && C <= 4, "vector loads need an even span");
  typedef __attribute__((ext_vector_type(2))) T T2;
#pragma unroll 2
  for (int i = tid; i < SROWS * SCOLS + 2; i += 256) {
    const int hr = i / SCOLS, col = i - hr * SCOLS;
    const int oy = y0 - 3 + hr, ox = col - 3;
    uint2 h = make_uint2(0u, 0u), l = h;
    if (hr < SROWS && oy >= 0 && oy < 64 && ox >= 0 && ox < 64) {
      const T* p = img + ((long)oy * K * S + ox * K) * C;
      T2 v[K][E / 2];
#pragma unroll
      for (int dy = 0; dy < K; dy++)
#pragma unroll
        for (int j = 0; j < E / 2; j++) v[dy][j] = *reinterpret_cast<const T2*>(p + (long)dy * S * C + 2 * j);
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C; c++) {
        float sm = 0.f;
#pragma unroll
        for (int dy = 0; dy < K; dy++)
#pragma unroll
          for (int dx = 0; dx < K; dx++) { const int e = dx * C + c; sm += (float)v[dy][e >> 1][e & 1] / div; }
        o[c] = sm * inv;
      }
      split4(o, h, l);
    }
    *reinterpret_cast<uint2*>(halo + i * 8) = h;
    *reinterpret_cast<uint2*>(halo + SPLANE + i * 8) = l;
  }
}

__global__ __launch_bounds__(256, 2) void stem_x3_kernel(StemArgs args) {
  __shared__ __attribute__((aligned(16))) char halo[2 * SPLANE];
  __shared__ float bst[4][2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int y0 = blockIdx.x * BAND, b = blockIdx.y;
  const StemTower& t = args.t[blockIdx.z];
  // weights [16][49][8] (channels 4..7 zero): k-step = kernel row ky, lane (r16, q) takes taps kx = 2 q, 2 q + 1 (4 channels each)
  bf16x8 wh[7], wl[7];
#pragma unroll
  for (int ky = 0; ky < 7; ky++) {
    const int kx = 2 * q;
    const long o0 = (long)r16 * 392 + (ky * 7 + kx) * 8;
    const uint2 a0 = *reinterpret_cast<const uint2*>(t.wh + o0), b0 = *reinterpret_cast<const uint2*>(t.wl + o0);
    uint2 a1 = make_uint2(0u, 0u), b1 = a1;
    if (kx + 1 < 7) { a1 = *reinterpret_cast<const uint2*>(t.wh + o0 + 8); b1 = *reinterpret_cast<const uint2*>(t.wl + o0 + 8); }
    wh[ky] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
    wl[ky] = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
  }
  {
    const int S = args.S, k = S / 64, C = t.C;
    const long bs = args.row_index ? args.row_index[b] : b;
    const float div = t.div, inv = 1.f / (float)(k * k);
    if (k == 2 && C == 3 && t.u8) stem_fill<2, 3, unsigned char>((const unsigned char*)t.img + bs * S * S * 3, div, inv, halo, y0, tid);
    else if (k == 2 && C == 3) stem_fill<2, 3, float>((const float*)t.img + bs * S * S * 3, div, inv, halo, y0, tid);
    else if (k == 2 && C == 1 && !t.u8) stem_fill<2, 1, float>((const float*)t.img + bs * S * S, div, inv, halo, y0, tid);
    else
    for (int i = tid; i < SROWS * SCOLS + 2; i += 256) {
      const int hr = i / SCOLS, col = i - hr * SCOLS;
      const int oy = y0 - 3 + hr, ox = col - 3;
      uint2 h = make_uint2(0u, 0u), l = h;
      if (hr < SROWS && oy >= 0 && oy < 64 && ox >= 0 && ox < 64) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        const long base = ((bs * S + (long)oy * k) * S + (long)ox * k) * C;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if (c >= C) break;
          float s = 0.f;
          for (int dy = 0; dy < k; dy++)
            for (int dx = 0; dx < k; dx++) {
              const long idx = base + ((long)dy * S + dx) * C + c;
              s += (t.u8 ? (float)((const unsigned char*)t.img)[idx] : ((const float*)t.img)[idx]) / div;
            }
          o[c] = s * inv;
        }
        split4(o, h, l);
      }
      *reinterpret_cast<uint2*>(halo + i * 8) = h;
      *reinterpret_cast<uint2*>(halo + SPLANE + i * 8) = l;
    }
  }
  __syncthreads();
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float* __restrict__ y = t.y + (long)b * 4096 * 16;
  const int base = ((2 * wave) * SCOLS + r16 + 2 * q) * 8;
#pragma unroll 1
  for (int tile = 0; tile < 8; tile++) {                  // not unrolled: 8 x 28 hoisted LDS reads would spill
      const int rr = tile >> 2, mt = tile & 3;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 7; ky++) {
        const int off = base + ((rr + ky) * SCOLS + mt * 16) * 8;
        const uint2 x0 = *reinterpret_cast<const uint2*>(halo + off), x1 = *reinterpret_cast<const uint2*>(halo + off + 8);
        const uint2 l0 = *reinterpret_cast<const uint2*>(halo + SPLANE + off), l1 = *reinterpret_cast<const uint2*>(halo + SPLANE + off + 8);
        acc = mma3(wh[ky], wl[ky], __builtin_bit_cast(bf16x8, make_uint4(x0.x, x0.y, x1.x, x1.y)),
                   __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y)), acc);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) { s1[r] += acc[r]; s2[r] = __builtin_fmaf(acc[r], acc[r], s2[r]); }
      const int oy = y0 + 2 * wave + rr, ox = mt * 16 + r16;
      *reinterpret_cast<float4*>(y + ((long)oy * 64 + ox) * 16 + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float a = row16_sum(s1[r]), c = row16_sum(s2[r]);
    if (r16 == 0) { bst[wave][0][q * 4 + r] = a; bst[wave][1][q * 4 + r] = c; }
  }
  __syncthreads();
  if (tid < 32) {
    const int which = tid >> 4, ch = tid & 15;
    const float v = (bst[0][which][ch] + bst[1][which][ch]) + (bst[2][which][ch] + bst[3][which][ch]);
    t.yst[(((long)b * NBAND + blockIdx.x) * 2 + which) * 16 + ch] = v;
  }
}

// ========================================================================================================================
// layers 2-4: one workgroup (512 threads) per image
// ========================================================================================================================
constexpr int RTH = 512;
#ifndef AVLEN_X3_STAGGER
#define AVLEN_X3_STAGGER 8       // s_sleep units (64 cycles) by which waves 4-7 enter a conv's tap loop late: their LDS read bursts then
#endif                           // fall under the MFMA segments of their SIMD partners (waves 0-3) instead of colliding with them
constexpr int R32 = 34, R64 = 18, R128 = 10;
constexpr int PLANE = R32 * R32 * 64;                       // 73984 B: one plane of the largest frame; the later frames reuse the space
constexpr int HALF16 = 34 * HCOLS * 32;                     // 71808 B: one plane of a half image of the 16-channel stage
static_assert(HALF16 <= PLANE && R64 * R64 * 128 <= PLANE && R128 * R128 * 256 <= PLANE, "frames share one LDS region");
constexpr int XPART_OFF = 2 * PLANE;                        // statistics partials: [8 waves][16 slots][2] fp32
constexpr int XCOEF_OFF = XPART_OFF + 8 * 16 * 2 * 4;       // scale[128], shift[128]
constexpr int XGB_OFF = XCOEF_OFF + 1024;                   // gamma | beta of the 15 GroupNorms: 5 x 64, 5 x 128, 5 x 256 fp32
constexpr int REST_LDS = XGB_OFF + (5 * 64 + 5 * 128 + 5 * 256) * 4;
static_assert(REST_LDS <= 160 * 1024, "tower x3 LDS budget");

__device__ __forceinline__ int a32(int y, int p, int chunk) { return (y * R32 + p) * 64 + ((chunk ^ ((p >> 1) & 3)) << 4); }
__device__ __forceinline__ int a64(int y, int p, int chunk) { return (y * R64 + p) * 128 + ((chunk ^ (p & 7)) << 4); }
// 128 channels: a pixel is one 256-byte bank row; a ds_read_b128 lane group holds 8 lanes of each of two lane quarters (chunks
// c, c + 1) on image rows y, y + 1 -- the row parity moves the slot into the other half, so the 16 lanes hit 16 different slots
__device__ __forceinline__ int a128(int y, int p, int chunk) { return (y * R128 + p) * 256 + ((chunk ^ (p & 7) ^ ((y & 1) << 3)) << 4); }

// GroupNorm index per stage: 0 downsample, 1 block 0 conv1 (stride 2), 2 block 0 conv2, 3 block 1 conv1, 4 block 1 conv2
struct RestTower {
  const float* x; const float* xst; const float* xg; const float* xb;        // layer-1 block 1 conv2: raw output + its GroupNorm
  const float* r;                                                             // block input of that block (materialised fp32): the residual
  const bf16* wh[15]; const bf16* wl[15];                                     // [stage * 5 + conv]
  const float* g[15]; const float* b[15];
  bf16* y;                                                                    // layer-4 output NHWC (B, 8, 8, 128) as a bf16 pair
};
struct RestArgs { RestTower t[8]; long y_lo; long long* prof; };              // y_lo: elements from the hi plane to the lo plane
#ifdef AVLEN_X3_PROF            // tools/x3_lab.hip: phase timestamps of one thread of every workgroup
#define X3_STAMP(k) do { if (args.prof && tid == AVLEN_X3_PROF) args.prof[(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define X3_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void stat_pair(const f32x4& v, float& g0, float& g1, float& h0, float& h1) {     // 32 channels: 2 per group
  g0 += v[0] + v[1]; g1 += v[2] + v[3];
  h0 = __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], h0)); h1 = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], h1));
}
__device__ __forceinline__ void stat_quad(const f32x4& v, float& s1, float& s2) {                            // 64 / 128 channels
  s1 += (v[0] + v[1]) + (v[2] + v[3]);
  s2 = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], s2))));
}

// 32 channels: s1 / s2 = the lane's four partial sums, groups (r >> 1) * 8 + q * 2 + (r & 1) (r >> 1 = cout tile)
__device__ __forceinline__ void finish32(float (&s1)[4], float (&s2)[4], char* lds, int gb, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + XGB_OFF) + gb;
  const float* beta = gamma + 32;
  float* part = reinterpret_cast<float*>(lds + XPART_OFF);
  float* coef = reinterpret_cast<float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float a = row16_sum(s1[r]), c = row16_sum(s2[r]);
    if (r16 == 0) *reinterpret_cast<float2*>(&part[(wave * 16 + (r >> 1) * 8 + q * 2 + (r & 1)) * 2]) = make_float2(a, c);
  }
  lds_barrier();
  if (tid < 32) {
    const int g = tid >> 1;
    double sum = 0.0, sq = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) { const float2 v = *reinterpret_cast<const float2*>(&part[(w * 16 + g) * 2]); sum += v.x; sq += v.y; }
    gn_coef(sum, sq, 1.0 / 2048.0, gamma[tid], beta[tid], coef[tid], coef[128 + tid]);
  }
  lds_barrier();
}
// 64 channels: the lane's sums are those of group (cout tile) * 4 + q over the wave's 8 rows (other half: wave + 4);
// 128 channels: of half a group, (cout tile = wave) * 2 + (q >> 1)
template <int NCH>
__device__ __forceinline__ void finish_q(float s1, float s2, char* lds, int gb, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + XGB_OFF) + gb;
  const float* beta = gamma + NCH;
  float2* part = reinterpret_cast<float2*>(lds + XPART_OFF);
  float* coef = reinterpret_cast<float*>(lds + XCOEF_OFF);
  const float a = row16_sum(s1), c = row16_sum(s2);
  if (r16 == 0) part[wave * 4 + q] = make_float2(a, c);
  lds_barrier();
  if (tid < NCH) {
    float2 u, v;
    if (NCH == 64) { const int g = tid >> 2, ct = g >> 2, qq = g & 3; u = part[ct * 4 + qq]; v = part[(ct + 4) * 4 + qq]; }
    else { const int w = tid >> 4, qq = (tid >> 2) & 2; u = part[w * 4 + qq]; v = part[w * 4 + qq + 1]; }
    gn_coef((double)u.x + (double)v.x, (double)u.y + (double)v.y, NCH == 64 ? 1.0 / 1024.0 : 1.0 / 512.0, gamma[tid], beta[tid],
            coef[tid], coef[128 + tid]);
  }
  lds_barrier();
}

// y = [relu](raw * scale + shift [+ res]) of the lane's four channels c0 .. c0 + 3 -> hi / lo frames at `ad` (8 bytes each);
// MODE 0: relu, no residual; 1: + res, relu, result becomes the new residual; 2: no relu, result -> res only (downsample branch)
template <int MODE>
__device__ __forceinline__ void apply4(const f32x4& raw, f32x4& res, const float* coef, int c0, char* lds, int ad) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    v[r] = raw[r] * coef[c0 + r] + coef[128 + c0 + r];
    if (MODE == 1) v[r] += res[r];
    if (MODE != 2) v[r] = fmaxf(v[r], 0.f);
  }
  if (MODE != 0) res = (f32x4){v[0], v[1], v[2], v[3]};
  if (MODE != 2) {
    uint2 h, l;
    split4(v, h, l);
    *reinterpret_cast<uint2*>(lds + ad) = h;
    *reinterpret_cast<uint2*>(lds + PLANE + ad) = l;
  }
}

// fragment-order weights (avlen_conv::w16f): [cout tile][k-step][lane][8]
__device__ __forceinline__ bf16x8 wfrag(const bf16* __restrict__ w, int ct, int ksteps, int i, int lane) {
  return *reinterpret_cast<const bf16x8*>(w + ((long)(ct * ksteps + i) * 64 + lane) * 8);
}

// The stride-1 convs below share one schedule.  Per tap: the weight fragments of the NEXT tap are requested first (two register
// sets, the tap loop unrolled by two: no copies, no wait at the loop end), then the tap's tiles are processed in groups with the
// LDS reads of group g + 1 in flight under the MFMAs of group g (sched_barrier pins "reads issued, then MFMAs": left alone the
// scheduler issued two reads, waited, one MFMA).  Waves 4-7 enter the tap loop half a tap late (AVLEN_X3_STAGGER): their read
// bursts fall under the MFMA segments of their SIMD partners instead of colliding with them.
#define X3_PIN() __builtin_amdgcn_sched_barrier(0)

// ---- 32 channels @ 32 x 32: wave owns rows oy(rr) = 16 (rr >> 1) + 2 wave + (rr & 1), both column tiles, both cout tiles
__device__ __forceinline__ int row32(int wave, int rr) { return 16 * (rr >> 1) + 2 * wave + (rr & 1); }
template <int MODE>
__device__ __forceinline__ void conv32(const bf16* __restrict__ wh, const bf16* __restrict__ wl, int gb, char* lds, f32x4 (&res)[16],
                                       int tid, int wave, int lane, int r16, int q) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto loadw = [&](bf16x8 (&D)[4], int tap) {             // [hi ct0, hi ct1, lo ct0, lo ct1]
    D[0] = wfrag(wh, 0, 9, tap, lane); D[1] = wfrag(wh, 1, 9, tap, lane);
    D[2] = wfrag(wl, 0, 9, tap, lane); D[3] = wfrag(wl, 1, 9, tap, lane);
  };
  // group g = tile (rr = g >> 1, pt = g & 1): 2 reads, 6 MFMAs (64 + 64 accumulator / residual registers leave room for no more)
  auto body = [&](const bf16x8 (&W)[4], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    int ad[8];
#pragma unroll
    for (int g = 0; g < 8; g++) ad[g] = a32(row32(wave, g >> 1) + ky, (g & 1) * 16 + r16 + kx, q);
    bf16x8 F[2][2];                                       // [set][hi, lo]
    auto rd = [&](int g, bf16x8 (&D)[2]) {
      D[0] = *reinterpret_cast<const bf16x8*>(lds + ad[g]); D[1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad[g]);
    };
    auto mm = [&](int g, const bf16x8 (&D)[2]) {
      acc[2 * g] = mma3(W[0], W[2], D[0], D[1], acc[2 * g]);
      acc[2 * g + 1] = mma3(W[1], W[3], D[0], D[1], acc[2 * g + 1]);
    };
    rd(0, F[0]);
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g + 1 < 8) rd(g + 1, F[(g + 1) & 1]);
      X3_PIN();
      mm(g, F[g & 1]);
    }
  };
  bf16x8 W0[4], W1[4];
  loadw(W0, 0);
  if (AVLEN_X3_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(AVLEN_X3_STAGGER);
#pragma unroll 1
  for (int t2 = 0; t2 < 8; t2 += 2) {
    loadw(W1, t2 + 1); X3_PIN();
    body(W0, t2);
    loadw(W0, t2 + 2); X3_PIN();
    body(W1, t2 + 1);
  }
  body(W0, 8);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; i++) { const int ct = i & 1; stat_pair(acc[i], s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]); }
  finish32(s1, s2, lds, gb, tid, wave, r16, q);
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int rr = 0; rr < 4; rr++)
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        const int ti = (rr * 2 + pt) * 2 + ct;
        apply4<MODE>(acc[ti], res[ti], coef, ct * 16 + q * 4, lds, a32(row32(wave, rr) + 1, pt * 16 + r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
      }
  lds_barrier();
}

// ---- 64 channels @ 16 x 16: wave owns cout tile wave & 3 and rows 8 (wave >> 2) .. + 7
template <int MODE>
__device__ __forceinline__ void conv64(const bf16* __restrict__ wh, const bf16* __restrict__ wl, int gb, char* lds, f32x4 (&res)[8],
                                       int tid, int wave, int lane, int r16, int q) {
  const int ct = wave & 3, half = wave >> 2;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto loadw = [&](bf16x8 (&D)[4], int tap) {             // [hi hf0, hi hf1, lo hf0, lo hf1]
    D[0] = wfrag(wh, ct, 18, tap * 2, lane); D[1] = wfrag(wh, ct, 18, tap * 2 + 1, lane);
    D[2] = wfrag(wl, ct, 18, tap * 2, lane); D[3] = wfrag(wl, ct, 18, tap * 2 + 1, lane);
  };
  // group g = rows 2 g, 2 g + 1 (8 reads, 12 MFMAs)
  auto body = [&](const bf16x8 (&W)[4], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int ad0 = a64(half * 8 + ky, r16 + kx, q), ad1 = a64(half * 8 + ky, r16 + kx, 4 + q);      // + rr rows: the swizzle does not depend on the row
    bf16x8 F[2][8];                                       // [set][row j: hf0 hi, hf0 lo, hf1 hi, hf1 lo]
    auto rd = [&](int g, bf16x8 (&D)[8]) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int o = (2 * g + j) * (R64 * 128);
        D[4 * j] = *reinterpret_cast<const bf16x8*>(lds + ad0 + o); D[4 * j + 1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad0 + o);
        D[4 * j + 2] = *reinterpret_cast<const bf16x8*>(lds + ad1 + o); D[4 * j + 3] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad1 + o);
      }
    };
    auto mm = [&](int g, const bf16x8 (&D)[8]) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        acc[2 * g + j] = mma3(W[0], W[2], D[4 * j], D[4 * j + 1], acc[2 * g + j]);
        acc[2 * g + j] = mma3(W[1], W[3], D[4 * j + 2], D[4 * j + 3], acc[2 * g + j]);
      }
    };
    rd(0, F[0]);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g + 1 < 4) rd(g + 1, F[(g + 1) & 1]);
      X3_PIN();
      mm(g, F[g & 1]);
    }
  };
  bf16x8 W0[4], W1[4];
  loadw(W0, 0);
  if (AVLEN_X3_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(AVLEN_X3_STAGGER);
#pragma unroll 1
  for (int t2 = 0; t2 < 8; t2 += 2) {
    loadw(W1, t2 + 1); X3_PIN();
    body(W0, t2);
    loadw(W0, t2 + 2); X3_PIN();
    body(W1, t2 + 1);
  }
  body(W0, 8);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int rr = 0; rr < 8; rr++) stat_quad(acc[rr], s1, s2);
  finish_q<64>(s1, s2, lds, gb, tid, wave, r16, q);
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int rr = 0; rr < 8; rr++)
    apply4<MODE>(acc[rr], res[rr], coef, ct * 16 + q * 4, lds, a64(half * 8 + rr + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
  lds_barrier();
}

// ---- 128 channels @ 8 x 8: wave owns cout tile `wave` and the four column tiles (rows 2 pt, 2 pt + 1; lane: row r16 >> 3, pixel r16 & 7)
template <int MODE>
__device__ __forceinline__ void conv128(const bf16* __restrict__ wh, const bf16* __restrict__ wl, int gb, char* lds, f32x4 (&res)[4],
                                        int tid, int wave, int lane, int r16, int q) {
  const int ct = wave, ly = r16 >> 3, lx = r16 & 7;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto loadw = [&](bf16x8 (&D)[8], int tap) {             // [hi j0..3, lo j0..3]
#pragma unroll
    for (int j = 0; j < 4; j++) { D[j] = wfrag(wh, ct, 36, tap * 4 + j, lane); D[4 + j] = wfrag(wl, ct, 36, tap * 4 + j, lane); }
  };
  // group g = column tile pt = g (8 reads, 12 MFMAs); + 2 rows per tile: the row parity (the swizzle's top bit) does not change
  auto body = [&](const bf16x8 (&W)[8], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    int ad[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ad[j] = a128(ly + ky, lx + kx, 4 * j + q);
    bf16x8 F[2][8];                                       // [set][j hi x4, j lo x4]
    auto rd = [&](int g, bf16x8 (&D)[8]) {
      const int o = 2 * g * (R128 * 256);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        D[j] = *reinterpret_cast<const bf16x8*>(lds + ad[j] + o); D[4 + j] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad[j] + o);
      }
    };
    auto mm = [&](int g, const bf16x8 (&D)[8]) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc[g] = mma3(W[j], W[4 + j], D[j], D[4 + j], acc[g]);
    };
    rd(0, F[0]);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g + 1 < 4) rd(g + 1, F[(g + 1) & 1]);
      X3_PIN();
      mm(g, F[g & 1]);
    }
  };
  bf16x8 W0[8], W1[8];
  loadw(W0, 0);
  if (AVLEN_X3_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(AVLEN_X3_STAGGER);
#pragma unroll 1
  for (int t2 = 0; t2 < 8; t2 += 2) {
    loadw(W1, t2 + 1); X3_PIN();
    body(W0, t2);
    loadw(W0, t2 + 2); X3_PIN();
    body(W1, t2 + 1);
  }
  body(W0, 8);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int pt = 0; pt < 4; pt++) stat_quad(acc[pt], s1, s2);
  finish_q<128>(s1, s2, lds, gb, tid, wave, r16, q);
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int pt = 0; pt < 4; pt++)
    apply4<MODE>(acc[pt], res[pt], coef, ct * 16 + q * 4, lds, a128(2 * pt + ly + 1, lx + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
  lds_barrier();
}

template <int CH>                                          // zero the one-pixel frame of the CH-channel image (both planes)
__device__ __forceinline__ void zero_frame(char* lds, int tid) {
  constexpr int R = CH == 32 ? R32 : CH == 64 ? R64 : R128, PB = CH * 2, CPP = PB / 16;     // chunks per pixel
  constexpr int NROW = 2 * R * CPP, NCOL = (R - 2) * 2 * CPP;
  for (int i = tid; i < NROW + NCOL; i += RTH) {
    int off;
    if (i < NROW) off = ((i / (R * CPP)) * (R - 1) * R) * PB + (i % (R * CPP)) * 16;
    else { const int j = i - NROW, row = 1 + j / (2 * CPP), rem = j % (2 * CPP); off = (row * R + (rem / CPP) * (R - 1)) * PB + (rem % CPP) * 16; }
    *reinterpret_cast<uint4*>(lds + off) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(lds + PLANE + off) = make_uint4(0u, 0u, 0u, 0u);
  }
}

__global__ __launch_bounds__(RTH) void rest_x3_kernel(RestArgs args) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const RestTower& t = args.t[blockIdx.y];
  const int img = blockIdx.x;
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
  // ---- GroupNorm affine parameters -> LDS; the scale / shift of the layer-1 output's GroupNorm -> coef (input transform)
  for (int i = tid; i < 5 * 64 + 5 * 128 + 5 * 256; i += RTH) {
    int n, j, nch;
    if (i < 320) { n = i >> 6; j = i & 63; nch = 32; }
    else if (i < 960) { n = 5 + ((i - 320) >> 7); j = (i - 320) & 127; nch = 64; }
    else { n = 10 + ((i - 960) >> 8); j = (i - 960) & 255; nch = 128; }
    reinterpret_cast<float*>(lds + XGB_OFF)[i] = j < nch ? t.g[n][j] : t.b[n][j - nch];
  }
  if (tid < 16) {
    float sc, sh;
    double sum, sq;
    band_sums(t.xst + (long)img * NBAND * 32, tid, sum, sq);
    gn_coef(sum, sq, 1.0 / 4096.0, t.xg[tid], t.xb[tid], sc, sh);
    reinterpret_cast<float*>(lds + XCOEF_OFF)[tid] = sc; reinterpret_cast<float*>(lds + XCOEF_OFF)[128 + tid] = sh;
  }
  lds_barrier();
  X3_STAMP(0);

  // =========================================================== layer 2 entry ===========================================================
  // in = relu(GN(x) + r) (64 x 64 x 16) read in two half-image passes; 3x3 stride-2 conv 16 -> 32 and the 1x1 stride-2 downsample.
  // Wave owns output rows 16 h + 2 wave, + 1 of half h (the row mapping of conv32), both column tiles, both cout tiles.
  f32x4 raw2[16], res2[16];
  {
    const bf16* __restrict__ wAh = t.wh[1]; const bf16* __restrict__ wAl = t.wl[1];          // [32][9][16]
    const bf16* __restrict__ wDh = t.wh[0]; const bf16* __restrict__ wDl = t.wl[0];          // [32][16]
    // weight fragments are fetched per k-step (2 taps x 16 channels) inside each half: [hi ct0, hi ct1, lo ct0, lo ct1]
    auto loadA = [&](bf16x8 (&W)[4], int s) {
      const int k = 32 * s + 8 * q;
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        W[ct] = k < 144 ? *reinterpret_cast<const bf16x8*>(wAh + (long)(ct * 16 + r16) * 144 + k) : zero_frag();
        W[2 + ct] = k < 144 ? *reinterpret_cast<const bf16x8*>(wAl + (long)(ct * 16 + r16) * 144 + k) : zero_frag();
      }
    };
    bf16x8 WD[4];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      WD[ct] = (q >> 1) == 0 ? *reinterpret_cast<const bf16x8*>(wDh + (long)(ct * 16 + r16) * 16 + 8 * (q & 1)) : zero_frag();
      WD[2 + ct] = (q >> 1) == 0 ? *reinterpret_cast<const bf16x8*>(wDl + (long)(ct * 16 + r16) * 16 + 8 * (q & 1)) : zero_frag();
    }
    const int c0 = (tid & 1) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { sc[i] = coef[c0 + i]; sh[i] = coef[128 + c0 + i]; }
    const float* __restrict__ x = t.x + (long)img * 4096 * 16;
    const float* __restrict__ r = t.r + (long)img * 4096 * 16;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      // frame row fr <-> image row 32 h - 1 + fr, fr = 0 .. 32; frame column = image column + 1
#pragma unroll 3
      for (int i = tid; i < 33 * 128; i += RTH) {
        const int fr = i >> 7, px = (i >> 1) & 63, ch = i & 1;
        const int iy = 32 * h - 1 + fr;
        uint2 h0 = make_uint2(0u, 0u), h1 = h0, l0 = h0, l1 = h0;
        if (iy >= 0) {
          const long off = ((long)iy * 64 + px) * 16 + c0;
          const float4 a = *reinterpret_cast<const float4*>(x + off), c = *reinterpret_cast<const float4*>(x + off + 4);
          const float4 ra = *reinterpret_cast<const float4*>(r + off), rc = *reinterpret_cast<const float4*>(r + off + 4);
          float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
          const float rv[8] = {ra.x, ra.y, ra.z, ra.w, rc.x, rc.y, rc.z, rc.w};
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e] * sc[e] + sh[e] + rv[e], 0.f);
          const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
          split4(va, h0, l0); split4(vb, h1, l1);
        }
        const int ad = h16(fr, px + 1, ch);
        *reinterpret_cast<uint4*>(lds + ad) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(lds + PLANE + ad) = make_uint4(l0.x, l0.y, l1.x, l1.y);
      }
      for (int i = tid; i < 33 * 2; i += RTH) {           // frame column 0 (image column -1); column 65 is never read by a stride-2 tap
        const int ad = h16(i >> 1, 0, i & 1);
        *reinterpret_cast<uint4*>(lds + ad) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(lds + PLANE + ad) = make_uint4(0u, 0u, 0u, 0u);
      }
      lds_barrier();
      X3_STAMP(1 + 2 * h);
      const int rdD = h16(4 * wave + 1, 2 * r16 + 1, q & 1);
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          const int ti = ((2 * h + j) * 2 + pt) * 2, o = (2 * j * HCOLS + 32 * pt) * 32;
          raw2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f}; raw2[ti + 1] = raw2[ti];
          const bf16x8 dh = *reinterpret_cast<const bf16x8*>(lds + rdD + o), dl = *reinterpret_cast<const bf16x8*>(lds + PLANE + rdD + o);
          res2[ti] = mma3(WD[0], WD[2], dh, dl, (f32x4){0.f, 0.f, 0.f, 0.f});
          res2[ti + 1] = mma3(WD[1], WD[3], dh, dl, (f32x4){0.f, 0.f, 0.f, 0.f});
        }
      bf16x8 WA[2][4];
      loadA(WA[0], 0);
#pragma unroll
      for (int s = 0; s < 5; s++) {
        if (s + 1 < 5) loadA(WA[(s + 1) & 1], s + 1);
        int tap = 2 * s + (q >> 1);
        if (tap > 8) tap = 8;                             // zero weights there
        const int ky = tap / 3, kx = tap - ky * 3;
        const int rdA = h16(4 * wave + ky, 2 * r16 + kx, q & 1);            // output row 16 h + 2 wave + j reads frame rows 2 (2 wave + j) + ky
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int pt = 0; pt < 2; pt++) {
            const int ti = ((2 * h + j) * 2 + pt) * 2, o = (2 * j * HCOLS + 32 * pt) * 32;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(lds + rdA + o), xl = *reinterpret_cast<const bf16x8*>(lds + PLANE + rdA + o);
            raw2[ti] = mma3(WA[s & 1][0], WA[s & 1][2], xh, xl, raw2[ti]);
            raw2[ti + 1] = mma3(WA[s & 1][1], WA[s & 1][3], xh, xl, raw2[ti + 1]);
          }
      }
      lds_barrier();                                      // every wave is done with this half before it is overwritten
      X3_STAMP(2 + 2 * h);
    }
  }
  {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; i++) { const int ct = i & 1; stat_pair(res2[i], s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]); }
    finish32(s1, s2, lds, 0, tid, wave, r16, q);                          // downsample norm (no ReLU) -> residual
#pragma unroll
    for (int i = 0; i < 16; i++) { f32x4 dummy = res2[i]; apply4<2>(dummy, res2[i], coef, (i & 1) * 16 + q * 4, lds, 0); }
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; i++) { const int ct = i & 1; stat_pair(raw2[i], s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]); }
    finish32(s1, s2, lds, 64, tid, wave, r16, q);                         // block 0 bn1 + ReLU -> the 32-channel frame
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int pt = 0; pt < 2; pt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          f32x4 dummy;
          apply4<0>(raw2[(rr * 2 + pt) * 2 + ct], dummy, coef, ct * 16 + q * 4, lds,
                    a32(row32(wave, rr) + 1, pt * 16 + r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
        }
    zero_frame<32>(lds, tid);
  }
  lds_barrier();
  X3_STAMP(5);
  conv32<1>(t.wh[2], t.wl[2], 128, lds, res2, tid, wave, lane, r16, q);                // block 0 conv2 + skip
  X3_STAMP(6);
  conv32<0>(t.wh[3], t.wl[3], 192, lds, res2, tid, wave, lane, r16, q);                // block 1 conv1
  X3_STAMP(7);
  conv32<1>(t.wh[4], t.wl[4], 256, lds, res2, tid, wave, lane, r16, q);                // block 1 conv2 + identity
  X3_STAMP(8);

  // =========================================================== layer 3 ===========================================================
  f32x4 res3[8];
  {
    const int ct = wave & 3, half = wave >> 2;
    f32x4 raw3[8];
    const bf16x8 wDh = wfrag(t.wh[5], ct, 1, 0, lane), wDl = wfrag(t.wl[5], ct, 1, 0, lane);
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      const int ad = a32(2 * (half * 8 + rr) + 1, 2 * r16 + 1, q);
      res3[rr] = mma3(wDh, wDl, *reinterpret_cast<const bf16x8*>(lds + ad), *reinterpret_cast<const bf16x8*>(lds + PLANE + ad),
                      (f32x4){0.f, 0.f, 0.f, 0.f});
      raw3[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto body3 = [&](const bf16x8& Wh, const bf16x8& Wl, int tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int ad = a32(2 * (half * 8) + ky, 2 * r16 + kx, q);               // + 2 rr rows: the swizzle does not depend on the row
      bf16x8 F[8][2];
#pragma unroll
      for (int rr = 0; rr < 8; rr++) {
        F[rr][0] = *reinterpret_cast<const bf16x8*>(lds + ad + 2 * rr * (R32 * 64));
        F[rr][1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad + 2 * rr * (R32 * 64));
      }
      X3_PIN();
#pragma unroll
      for (int rr = 0; rr < 8; rr++) raw3[rr] = mma3(Wh, Wl, F[rr][0], F[rr][1], raw3[rr]);
    };
    bf16x8 W0h = wfrag(t.wh[6], ct, 9, 0, lane), W0l = wfrag(t.wl[6], ct, 9, 0, lane), W1h, W1l;
#pragma unroll 1
    for (int t2 = 0; t2 < 8; t2 += 2) {
      W1h = wfrag(t.wh[6], ct, 9, t2 + 1, lane); W1l = wfrag(t.wl[6], ct, 9, t2 + 1, lane); X3_PIN();
      body3(W0h, W0l, t2);
      W0h = wfrag(t.wh[6], ct, 9, t2 + 2, lane); W0l = wfrag(t.wl[6], ct, 9, t2 + 2, lane); X3_PIN();
      body3(W1h, W1l, t2 + 1);
    }
    body3(W0h, W0l, 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) stat_quad(res3[rr], s1, s2);
    finish_q<64>(s1, s2, lds, 320, tid, wave, r16, q);
#pragma unroll
    for (int rr = 0; rr < 8; rr++) { f32x4 dummy = res3[rr]; apply4<2>(dummy, res3[rr], coef, ct * 16 + q * 4, lds, 0); }
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) stat_quad(raw3[rr], s1, s2);
    finish_q<64>(s1, s2, lds, 320 + 128, tid, wave, r16, q);             // the barrier inside: every wave has left the 32-channel frame
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      f32x4 dummy;
      apply4<0>(raw3[rr], dummy, coef, ct * 16 + q * 4, lds, a64(half * 8 + rr + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
    }
    zero_frame<64>(lds, tid);
  }
  lds_barrier();
  X3_STAMP(9);
  conv64<1>(t.wh[7], t.wl[7], 320 + 256, lds, res3, tid, wave, lane, r16, q);
  X3_STAMP(10);
  conv64<0>(t.wh[8], t.wl[8], 320 + 384, lds, res3, tid, wave, lane, r16, q);
  X3_STAMP(11);
  conv64<1>(t.wh[9], t.wl[9], 320 + 512, lds, res3, tid, wave, lane, r16, q);
  X3_STAMP(12);

  // =========================================================== layer 4 ===========================================================
  f32x4 res4[4];
  {
    const int ct = wave, ly = r16 >> 3, lx = r16 & 7;
    f32x4 raw4[4];
    bf16x8 wD[4];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) { wD[hf] = wfrag(t.wh[10], ct, 2, hf, lane); wD[2 + hf] = wfrag(t.wl[10], ct, 2, hf, lane); }
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int ad = a64(2 * (2 * pt + ly) + 1, 2 * lx + 1, 4 * hf + q);
        v = mma3(wD[hf], wD[2 + hf], *reinterpret_cast<const bf16x8*>(lds + ad), *reinterpret_cast<const bf16x8*>(lds + PLANE + ad), v);
      }
      res4[pt] = v;
      raw4[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto loadw = [&](bf16x8 (&D)[4], int tap) {
#pragma unroll
      for (int hf = 0; hf < 2; hf++) { D[hf] = wfrag(t.wh[11], ct, 18, tap * 2 + hf, lane); D[2 + hf] = wfrag(t.wl[11], ct, 18, tap * 2 + hf, lane); }
    };
    auto body4 = [&](const bf16x8 (&W)[4], int tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int ad0 = a64(2 * ly + ky, 2 * lx + kx, q), ad1 = a64(2 * ly + ky, 2 * lx + kx, 4 + q);
      bf16x8 F[4][4];
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        const int o = 4 * pt * (R64 * 128);
        F[pt][0] = *reinterpret_cast<const bf16x8*>(lds + ad0 + o); F[pt][1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad0 + o);
        F[pt][2] = *reinterpret_cast<const bf16x8*>(lds + ad1 + o); F[pt][3] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad1 + o);
      }
      X3_PIN();
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        raw4[pt] = mma3(W[0], W[2], F[pt][0], F[pt][1], raw4[pt]);
        raw4[pt] = mma3(W[1], W[3], F[pt][2], F[pt][3], raw4[pt]);
      }
    };
    bf16x8 W0[4], W1[4];
    loadw(W0, 0);
#pragma unroll 1
    for (int t2 = 0; t2 < 8; t2 += 2) {
      loadw(W1, t2 + 1); X3_PIN();
      body4(W0, t2);
      loadw(W0, t2 + 2); X3_PIN();
      body4(W1, t2 + 1);
    }
    body4(W0, 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) stat_quad(res4[pt], s1, s2);
    finish_q<128>(s1, s2, lds, 960, tid, wave, r16, q);
#pragma unroll
    for (int pt = 0; pt < 4; pt++) { f32x4 dummy = res4[pt]; apply4<2>(dummy, res4[pt], coef, ct * 16 + q * 4, lds, 0); }
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) stat_quad(raw4[pt], s1, s2);
    finish_q<128>(s1, s2, lds, 960 + 256, tid, wave, r16, q);
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      f32x4 dummy;
      apply4<0>(raw4[pt], dummy, coef, ct * 16 + q * 4, lds, a128(2 * pt + ly + 1, lx + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
    }
    zero_frame<128>(lds, tid);
  }
  lds_barrier();
  X3_STAMP(13);
  conv128<1>(t.wh[12], t.wl[12], 960 + 512, lds, res4, tid, wave, lane, r16, q);
  X3_STAMP(14);
  conv128<0>(t.wh[13], t.wl[13], 960 + 768, lds, res4, tid, wave, lane, r16, q);
  X3_STAMP(15);
  conv128<1>(t.wh[14], t.wl[14], 960 + 1024, lds, res4, tid, wave, lane, r16, q);
  X3_STAMP(16);
  // ---- layer-4 output (post ReLU) = the residual registers of the last block: NHWC fp32 (8, 8, 128)
  {
    const int ly = r16 >> 3, lx = r16 & 7;
    bf16* __restrict__ y = t.y + (long)img * 64 * 128;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      const float v[4] = {res4[pt][0], res4[pt][1], res4[pt][2], res4[pt][3]};
      uint2 hh, ll;
      split4(v, hh, ll);
      const long o = ((long)(2 * pt + ly) * 8 + lx) * 128 + wave * 16 + q * 4;
      *reinterpret_cast<uint2*>(y + o) = hh;
      *reinterpret_cast<uint2*>(y + args.y_lo + o) = ll;
    }
  }
}

}  // namespace

bool avlen_tower_x3_supported(const avlen_resnet18* n, int S, int C) {
  if (!n || S % 64 || S < 64 || C < 1 || C > 4) return false;
  const avlen_conv& k = n->conv1;
  if (!k.w16 || !k.w16lo || k.cin16 != 8 || k.cout != 16 || k.kh != 7 || k.kw != 7 || k.stride != 1 || k.pad != 3) return false;
  auto conv3 = [](const avlen_conv& c, int cin, int cout, int stride, bool frag) {
    return c.w16 && c.w16lo && (!frag || (c.w16f && c.w16flo)) && c.cin16 == cin && c.cout == cout && c.kh == 3 && c.kw == 3 &&
           c.stride == stride && c.pad == 1;
  };
  auto down = [](const avlen_conv& d, int cin, int cout, bool frag) {
    return d.w16 && d.w16lo && (!frag || (d.w16f && d.w16flo)) && d.cin16 == cin && d.cout == cout && d.kh == 1 && d.kw == 1 &&
           d.stride == 2 && d.pad == 0;
  };
  for (int i = 0; i < 2; i++)
    if (n->block[i].has_down || !conv3(n->block[i].conv1, 16, 16, 1, false) || !conv3(n->block[i].conv2, 16, 16, 1, false)) return false;
  int cin = 16;
  for (int l = 0; l < 3; l++) {
    const int co = 32 << l;
    const avlen_resblock& b0 = n->block[2 + 2 * l]; const avlen_resblock& b1 = n->block[3 + 2 * l];
    if (!b0.has_down || b1.has_down || !down(b0.down, cin, co, l > 0) || !conv3(b0.conv1, cin, co, 2, l > 0) ||
        !conv3(b0.conv2, co, co, 1, true) || !conv3(b1.conv1, co, co, 1, true) || !conv3(b1.conv2, co, co, 1, true)) return false;
    cin = co;
  }
  return true;
}

// scratch per tower: 4 fp32 tensors of the 64 x 64 x 16 stage, 5 statistics blocks, the layer-4 output
size_t avlen_tower_x3_workspace_bytes(int groups, int B) {
  const size_t act = (size_t)B * 4096 * 16 * sizeof(float);
  return (size_t)groups * (4 * (act + 256) + 5 * ((size_t)B * NBAND * 32 * sizeof(float) + 256) + (size_t)B * 8192 * sizeof(float) + 256) + 4096;
}

// Y[g] = layer-4 output NHWC (B, 8, 8, 128) of tower g as a compensated bf16 pair (hi plane, lo plane B * 8192 elements behind;
// the caller applies fc); imgs[g]: B (or, with row_index, more) images
int avlen_tower_x3_fwd(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                       const float* divisors, const int* row_index, void* const* Y, int groups, int B, int S, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
  if (groups < 1 || groups > 8 || B <= 0 || ws_bytes < avlen_tower_x3_workspace_bytes(groups, B)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes);
  const size_t act = (size_t)B * 4096 * 16;
  float* raw[8][3]; float* a2[8]; float* st[8];
  float* st_all = w.take<float>((size_t)groups * 5 * B * NBAND * 32);
  for (int g = 0; g < groups; g++) {
    for (int i = 0; i < 3; i++) raw[g][i] = w.take<float>(act);
    a2[g] = w.take<float>(act);
    st[g] = st_all + (size_t)g * 5 * B * NBAND * 32;
    if (!avlen_tower_x3_supported(nets[g], S, channels[g])) return AVLEN_ERR_ARG;
  }
  const size_t sb = (size_t)B * NBAND * 32;               // one statistics block (written whole by its producer: no zeroing)
  dim3 grid(64 / BAND, B, groups);
  {
    StemArgs a = {};
    a.row_index = row_index; a.S = S;
    for (int g = 0; g < groups; g++) {
      const avlen_resnet18* n = nets[g];
      a.t[g] = StemTower{imgs[g], img_u8 ? img_u8[g] : 0, channels[g], divisors[g], (const bf16*)n->conv1.w16, (const bf16*)n->conv1.w16lo,
                         raw[g][0], st[g]};
    }
    hipLaunchKernelGGL(stem_x3_kernel, grid, dim3(256), 0, stream, a);
  }
  // raw0 = stem; a0 = relu(GN0(raw0)).  block 0: raw1 = conv1(a0), raw2 = conv2(relu(GN1(raw1))), a2 = relu(GN2(raw2) + a0);
  // block 1: raw3 = conv1(a2) [a2 materialised], raw4 = conv2(relu(GN3(raw3))), layer-1 output = relu(GN4(raw4) + a2)
  for (int i = 0; i < 4; i++) {
    C16Args a = {};
    for (int g = 0; g < groups; g++) {
      const avlen_resnet18* n = nets[g];
      C16Tower& t = a.t[g];
      // buffers: raw[0] = raw0 (kept until a2 exists), raw[1] / raw[2] ping-pong
      if (i == 0) { t.x = raw[g][0]; t.xst = st[g]; t.xg = n->bn1.g; t.xb = n->bn1.b; t.y = raw[g][1]; t.yst = st[g] + sb;
                    t.wh = (const bf16*)n->block[0].conv1.w16; t.wl = (const bf16*)n->block[0].conv1.w16lo; }
      if (i == 1) { t.x = raw[g][1]; t.xst = st[g] + sb; t.xg = n->block[0].bn1.g; t.xb = n->block[0].bn1.b; t.y = raw[g][2]; t.yst = st[g] + 2 * sb;
                    t.wh = (const bf16*)n->block[0].conv2.w16; t.wl = (const bf16*)n->block[0].conv2.w16lo; }
      if (i == 2) { t.x = raw[g][2]; t.xst = st[g] + 2 * sb; t.xg = n->block[0].bn2.g; t.xb = n->block[0].bn2.b;
                    t.r = raw[g][0]; t.rst = st[g]; t.rg = n->bn1.g; t.rb = n->bn1.b; t.a_out = a2[g];
                    t.y = raw[g][1]; t.yst = st[g] + 3 * sb;
                    t.wh = (const bf16*)n->block[1].conv1.w16; t.wl = (const bf16*)n->block[1].conv1.w16lo; }
      if (i == 3) { t.x = raw[g][1]; t.xst = st[g] + 3 * sb; t.xg = n->block[1].bn1.g; t.xb = n->block[1].bn1.b; t.y = raw[g][2]; t.yst = st[g] + 4 * sb;
                    t.wh = (const bf16*)n->block[1].conv2.w16; t.wl = (const bf16*)n->block[1].conv2.w16lo; }
    }
    hipLaunchKernelGGL(c16_x3_kernel, grid, dim3(256), 0, stream, a);
  }
  {
    RestArgs a = {};
    a.y_lo = (long)B * 8192;
    for (int g = 0; g < groups; g++) {
      const avlen_resnet18* n = nets[g];
      RestTower& t = a.t[g];
      t.x = raw[g][2]; t.xst = st[g] + 4 * sb; t.xg = n->block[1].bn2.g; t.xb = n->block[1].bn2.b; t.r = a2[g]; t.y = (bf16*)Y[g];
      for (int l = 0; l < 3; l++) {
        const avlen_resblock& b0 = n->block[2 + 2 * l]; const avlen_resblock& b1 = n->block[3 + 2 * l];
        const int o = 5 * l;
        const bool frag = l > 0;
        t.wh[o] = (const bf16*)(frag ? b0.down.w16f : b0.down.w16); t.wl[o] = (const bf16*)(frag ? b0.down.w16flo : b0.down.w16lo);
        t.wh[o + 1] = (const bf16*)(frag ? b0.conv1.w16f : b0.conv1.w16); t.wl[o + 1] = (const bf16*)(frag ? b0.conv1.w16flo : b0.conv1.w16lo);
        t.wh[o + 2] = (const bf16*)b0.conv2.w16f; t.wl[o + 2] = (const bf16*)b0.conv2.w16flo;
        t.wh[o + 3] = (const bf16*)b1.conv1.w16f; t.wl[o + 3] = (const bf16*)b1.conv1.w16flo;
        t.wh[o + 4] = (const bf16*)b1.conv2.w16f; t.wl[o + 4] = (const bf16*)b1.conv2.w16flo;
        t.g[o] = b0.bnd.g; t.b[o] = b0.bnd.b; t.g[o + 1] = b0.bn1.g; t.b[o + 1] = b0.bn1.b; t.g[o + 2] = b0.bn2.g; t.b[o + 2] = b0.bn2.b;
        t.g[o + 3] = b1.bn1.g; t.b[o + 3] = b1.bn1.b; t.g[o + 4] = b1.bn2.g; t.b[o + 4] = b1.bn2.b;
      }
    }
    static unsigned long long attr_done = 0;
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&rest_x3_kernel), REST_LDS, &attr_done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(rest_x3_kernel, dim3(B, groups), dim3(RTH), REST_LDS, stream, a);
  }
  return avlen_launch_status();
}
