// Stem + layers 1 and 2 of the CustomResNet tower (smt_resnet.py:132-141: conv 7x7 -> GroupNorm -> ReLU -> two basic blocks
// of 16 channels at 64x64 -> two basic blocks of 32 channels at 32x32, the first with stride 2 and a 1x1 stride-2 downsample
// + GroupNorm on the skip) -- with the sensor preprocessing (x / divisor, k x k block mean: smt_cnn.py:83-93) in front -- as ONE
// launch, one 512-thread workgroup per image:
//  * the activation of the current stage is ONE zero-framed bf16 image in LDS (64x64x16: 66 x 66 x 32 B = 136 KiB; 32x32x32:
//    34 x 34 x 64 B in the same space), so that the taps need no bounds tests;
//  * the basic blocks' residual and every raw conv output live in REGISTERS as packed bf16 pairs (layer 1: a wave owns 8 image
//    rows = 32 MFMA tiles, 64 + 64 registers) -- the second and third live tensor of a block never need LDS;
//  * the convs run INPUT-ROW STATIONARY: the kx-shifted B fragments of a frame row are read from LDS once and feed the output
//    rows row - ky (with 16 output channels a fragment read otherwise feeds a single MFMA and LDS bandwidth is the bound);
//  * the conv weights are MFMA fragments in registers, fetched one conv ahead (barriers order LDS traffic only);
//  * GroupNorm statistics come from the kernel's own fp32 accumulators: packed per-lane sums, 16-lane DPP row sums, a fixed-order
//    combine of the 8 waves in double (deterministic: no atomics); normalise / residual / ReLU on packed pairs.
// Replaces preprocess + 10 conv + 4 GroupNorm-apply launches that moved the activation through HBM ~20 times: the image is read
// once and the 32x32x32 layer-2 output (64 KiB) written once.
//
// Arithmetic: bf16 operands, fp32 MFMA accumulation (K order: frame row, then kx, then channel), raw conv outputs rounded to
// bf16 before the normalisation, statistics from the fp32 accumulators, var = E[x^2] - mean^2 in double, rstd = 1 / sqrt(var + eps)
// in fp32, y = relu(x * (gamma * rstd) + (beta - mean * gamma * rstd) [+ residual]), rounded to bf16.
//
// LDS layout of the 16-channel image: [66 rows][66 pixels][32 B]; the two 16-byte chunks of a pixel are swapped when
// (pixel >> 3) & 1 so that the 16 pixels of an MFMA row tile read conflict-free.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

#include "tower_util.h"

namespace {

constexpr int HTH = 512, HNW = 8;
constexpr int ROWP = 66;                          // pixels per LDS row at 16 channels (1 + 64 + 1); 66 rows (1 + 64 + 1)
constexpr int IMG_BYTES = 66 * ROWP * 32;         // 139392
constexpr int ROWP0 = 70;                         // stem input: 70 rows x 70 pixels (3 + 64 + 3), 8 B per pixel (4 channels)
constexpr int PART_OFF = IMG_BYTES;               // [8 waves][16 channels][2] fp32
constexpr int COEF_OFF = PART_OFF + HNW * 16 * 2 * 4; // scale[32], shift[32]
constexpr int GB_OFF = COEF_OFF + 2 * 32 * 4;        // gamma, beta of the ten GroupNorms: [5][2][16] then [5][2][32] fp32
constexpr int HEAD_LDS = GB_OFF + (5 * 32 + 5 * 64) * 4;
constexpr int ROWQ = 34;                          // layer 2: 34 rows x 34 pixels (1 + 32 + 1) x 64 B (32 channels)
static_assert(ROWQ * ROWQ * 64 <= IMG_BYTES, "the 32-channel frame reuses the 16-channel frame's space");
static_assert((70 * ROWP0 + 2) * 8 <= IMG_BYTES && HEAD_LDS <= 160 * 1024, "tower head LDS budget");

// w / g / b: 0 stem, 1..4 layer 1 (block 0 conv1, conv2, block 1 conv1, conv2), 5 block 2 downsample, 6 / 7 block 2 conv1 /
// conv2, 8 / 9 block 3 conv1 / conv2; w[7..9] are the fragment-order copies avlen_conv::w16f, the others [cout][kh][kw][cin16]
struct HeadTower { const void* img; int u8; int C; float div; const bf16* w[10]; const float* g[10]; const float* b[10]; bf16* y; };
struct HeadArgs { HeadTower t[8]; const int* row_index; int S; long long* prof; };
#ifdef AVLEN_HEAD_PROF          // tools/head_lab.hip: phase timestamps of every workgroup's wave 0
#define HEAD_STAMP(k) do { if (args.prof && tid == AVLEN_HEAD_PROF) args.prof[(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define HEAD_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int a16(int y, int p, int chunk) { return (y * ROWP + p) * 32 + ((chunk ^ ((p >> 3) & 1)) << 4); }

// 32-channel frame: 64 B per pixel = four 16-byte chunks, rotated by (pixel >> 1) & 3 so that 8 neighbouring pixels' reads of
// the same chunk fall on 8 different 16-byte bank groups
__device__ __forceinline__ int a32(int y, int p, int chunk) { return (y * ROWQ + p) * 64 + ((chunk ^ ((p >> 1) & 3)) << 4); }

// The common sensor shapes (128 x 128 -> 64 x 64, rgb or depth): a lane's K * C inputs of one source row are contiguous and
// even in number -> 8-byte (fp32) / 2-byte (uint8) vector loads, all of a pixel's loads in flight together.  Same arithmetic
// as the generic loop: each element divided by `div`, summed in (dy, dx) order, scaled by 1 / K^2.
template <int K, int C, typename T>
__device__ __forceinline__ void preprocess_tile(const T* __restrict__ img, float div, float inv, char* lds, int tid) {
  constexpr int S = 64 * K, E = K * C;            // elements per source row and output pixel
  static_assert(E % 2 == 0 && C <= 4, "vector loads need an even span");
  typedef __attribute__((ext_vector_type(2))) T T2;
#pragma unroll 2
  for (int i = tid; i < 4096; i += HTH) {
    const int oy = i >> 6, ox = i & 63;
    const T* p = img + ((long)oy * K * S + ox * K) * C;
    T2 v[K][E / 2];
#pragma unroll
    for (int dy = 0; dy < K; dy++)
#pragma unroll
      for (int j = 0; j < E / 2; j++) v[dy][j] = *reinterpret_cast<const T2*>(p + (long)dy * S * C + 2 * j);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; c++) {
      float s = 0.f;
#pragma unroll
      for (int dy = 0; dy < K; dy++)
#pragma unroll
        for (int dx = 0; dx < K; dx++) { const int e = dx * C + c; s += (float)v[dy][e >> 1][e & 1] / div; }
      o[c] = s * inv;
    }
    const P4 pk = pack4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint2*>(lds + ((oy + 3) * ROWP0 + ox + 3) * 8) = make_uint2(pk.lo, pk.hi);
  }
}

// per-wave statistics -> block statistics -> scale / shift per channel.  GroupNorm(16): NCH = 16 -> one channel per group
// (4096 values), NCH = 32 -> two channels per group (2048 values).  s1 / s2 hold this lane's four partial sums: NCH = 16 the
// channels q * 4 + r; NCH = 32 the groups (r >> 1) * 8 + q * 2 + (r & 1) (r >> 1 = cout tile, r & 1 = channel pair).
// Fixed summation order (lanes by DPP butterflies, waves 0..7 in sequence): deterministic.  The moments are combined in double
// (E[x^2] - mean^2 cancels), the reciprocal square root is taken in fp32 (correctly rounded sqrt and division).
template <int NCH>
__device__ __forceinline__ void finish_stats(float (&s1)[4], float (&s2)[4], char* lds, int gb, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + GB_OFF) + gb;
  const float* beta = gamma + NCH;
  float* part = reinterpret_cast<float*>(lds + PART_OFF);
  float* coef = reinterpret_cast<float*>(lds + COEF_OFF);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float a = row16_sum(s1[r]), c = row16_sum(s2[r]);
    const int slot = NCH == 16 ? q * 4 + r : (r >> 1) * 8 + q * 2 + (r & 1);
    if (r16 == 0) *reinterpret_cast<float2*>(&part[(wave * 16 + slot) * 2]) = make_float2(a, c);
  }
  lds_barrier();                                // every wave has also finished reading the image of this conv
  if (tid < NCH) {
    const int g = NCH == 16 ? tid : tid >> 1;
    double sum = 0.0, sq = 0.0;
#pragma unroll
    for (int w = 0; w < HNW; w++) { const float2 v = *reinterpret_cast<const float2*>(&part[(w * 16 + g) * 2]); sum += v.x; sq += v.y; }
    constexpr double inv_n = NCH == 16 ? 1.0 / 4096.0 : 1.0 / 2048.0;
    const double mean = sum * inv_n;
    double var = sq * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
    const float sc = gamma[tid] * rstd;
    coef[tid] = sc; coef[32 + tid] = beta[tid] - (float)mean * sc;
  }
  lds_barrier();
}

// One 3x3 stride-1 conv 32 -> 32 over the 32 x 32 frame + GroupNorm (+ residual) + ReLU, input-row stationary: a wave owns 4
// output rows x 2 column tiles x 2 cout tiles.  Per column tile: the three kx-shifted fragments of frame row f (lane (x, q):
// channels 8 q .. 8 q + 7 of pixel x + kx) are read once and feed the output rows f - ky through the 18 weight fragments.
__device__ __forceinline__ void load_w32(bf16x8 (&W)[9][2], const bf16* __restrict__ wt, int r16, int q) {
#pragma unroll
  for (int tap = 0; tap < 9; tap++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++) W[tap][ct] = *reinterpret_cast<const bf16x8*>(wt + ((long)(ct * 9 + tap) * 64 + q * 16 + r16) * 8);   // w16f
}
// W: this conv's weight fragments (already loaded or in flight); next_wt: the following 32-channel conv's weights, fetched into
// W as soon as the MFMAs are done, so that the L2 round trip runs under the statistics / apply phases (nullptr: none)
template <bool SECOND>
__device__ __forceinline__ void conv32_gn(bf16x8 (&W)[9][2], const bf16* __restrict__ next_wt, int gb, char* lds, P4 (&rawp)[32],
                                          P4 (&res)[32], int tid, int wave, int r16, int q) {
  const float* coef = reinterpret_cast<const float*>(lds + COEF_OFF);
  int rd[3];
#pragma unroll
  for (int kx = 0; kx < 3; kx++) rd[kx] = a32(wave * 4, r16 + kx, q);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    f32x4 acc[4][2];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) { acc[rr][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rr][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int f = 0; f < 6; f++) {
      bf16x8 F[3];
#pragma unroll
      for (int kx = 0; kx < 3; kx++) F[kx] = *reinterpret_cast<const bf16x8*>(lds + rd[kx] + (f * ROWQ + pt * 16) * 64);
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int rr = f - ky;
        if (rr >= 0 && rr < 4) {
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            acc[rr][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[ky * 3 + kx][0], F[kx], acc[rr][0], 0, 0, 0);
            acc[rr][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[ky * 3 + kx][1], F[kx], acc[rr][1], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        const f32x4 v = acc[rr][ct];
        stat32(v, s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]);
        rawp[(rr * 2 + pt) * 2 + ct] = pack4(v[0], v[1], v[2], v[3]);
      }
    if (pt == 1 && next_wt) load_w32(W, next_wt, r16, q);
  }
  finish_stats<32>(s1, s2, lds, gb, tid, wave, r16, q);
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    const int c0 = ct * 16 + q * 4;
    const f32x2 sc0 = {coef[c0], coef[c0 + 1]}, sc1 = {coef[c0 + 2], coef[c0 + 3]};
    const f32x2 sh0 = {coef[32 + c0], coef[32 + c0 + 1]}, sh1 = {coef[32 + c0 + 2], coef[32 + c0 + 3]};
    const int wr = a32(wave * 4 + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8;
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        const int ti = (rr * 2 + pt) * 2 + ct;
        f32x2 v0 = unlo(rawp[ti]) * sc0 + sh0, v1 = unhi(rawp[ti]) * sc1 + sh1;
        if (SECOND) { v0 += unlo(res[ti]); v1 += unhi(res[ti]); }
        const P4 o = {relu_pk(pack2(v0)), relu_pk(pack2(v1))};
        if (SECOND) res[ti] = o;
        *reinterpret_cast<uint2*>(lds + wr + (rr * ROWQ + pt * 16) * 64) = make_uint2(o.lo, o.hi);
      }
  }
  lds_barrier();
}

__global__ __launch_bounds__(HTH) void tower_head_kernel(HeadArgs args, int B) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const HeadTower& t = args.t[blockIdx.y];
  const int b = blockIdx.x;
  const float* coef = reinterpret_cast<const float*>(lds + COEF_OFF);
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; e++) zero8[e] = (bf16)0.f;

  HEAD_STAMP(0);
  // the stem's weight fragments: fetched first, the L2 round trip runs under the preprocessing
  bf16x8 wf[7];
  {
    const bf16* __restrict__ wt = t.w[0];         // [16][49][8] (the packing of the launch-per-layer path); channels 4..7 are zero
#pragma unroll
    for (int ky = 0; ky < 7; ky++) {
      const int kx = 2 * q;
      const uint2 w0 = *reinterpret_cast<const uint2*>(wt + (long)r16 * 392 + (ky * 7 + kx) * 8);
      const uint2 w1 = kx + 1 < 7 ? *reinterpret_cast<const uint2*>(wt + (long)r16 * 392 + (ky * 7 + kx + 1) * 8) : make_uint2(0u, 0u);
      wf[ky] = __builtin_bit_cast(bf16x8, make_uint4(w0.x, w0.y, w1.x, w1.y));
    }
  }
  if (tid < 480) {                                // GroupNorm affine parameters -> LDS (read inside the statistics' critical section)
    const bool l1 = tid < 160;
    const int n = l1 ? tid >> 5 : 5 + ((tid - 160) >> 6), j = l1 ? tid & 31 : (tid - 160) & 63, nch = l1 ? 16 : 32;
    reinterpret_cast<float*>(lds + GB_OFF)[tid] = j < nch ? t.g[n][j] : t.b[n][j - nch];
  }
  // ---- sensor preprocessing straight into the stem's LDS image: (x / div, k x k mean) -> 8 channels (>= C: zero), 3-pixel zero frame
  {
    const int S = args.S, k = S / 64, C = t.C;
    const long bs = args.row_index ? args.row_index[b] : b;
    const float div = t.div, inv = 1.f / (float)(k * k);
    if (k == 2 && C == 3 && !t.u8) preprocess_tile<2, 3, float>((const float*)t.img + bs * S * S * 3, div, inv, lds, tid);
    else if (k == 2 && C == 3) preprocess_tile<2, 3, unsigned char>((const unsigned char*)t.img + bs * S * S * 3, div, inv, lds, tid);
    else if (k == 2 && C == 1 && !t.u8) preprocess_tile<2, 1, float>((const float*)t.img + bs * S * S, div, inv, lds, tid);
    else
      for (int i = tid; i < 4096; i += HTH) {
        const int oy = i >> 6, ox = i & 63;
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
        const P4 pk = pack4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint2*>(lds + ((oy + 3) * ROWP0 + ox + 3) * 8) = make_uint2(pk.lo, pk.hi);
      }
    for (int i = tid; i < 70 * 70 + 2; i += HTH) {                       // + 2: the zero tap of the last row's last tile reads past the frame
      const int row = i / 70, col = i - row * 70;
      if (row < 3 || row >= 67 || col < 3 || col >= 67) *reinterpret_cast<uint2*>(lds + i * 8) = make_uint2(0u, 0u);
    }
  }
  lds_barrier();
  HEAD_STAMP(1);

  P4 rawp[32], res[32];
  float s1[4], s2[4];

  // ---- stem: 7x7, 4 (padded) -> 16 channels, INPUT-ROW STATIONARY.  A k-step of 32 = one kernel row (7 taps + a zero tap) x 4
  // channels, so the B fragment of frame row f -- lane (x, q): pixels x + 2 q, x + 2 q + 1, 16 contiguous bytes -- serves the 7
  // output rows f - ky with the weights of kernel row ky: each fragment is read from LDS once and used by up to 7 MFMAs (LDS
  // bandwidth, not MFMA, bounded the output-stationary form).  Addresses = per-lane base + compile-time offset.
  {
    const int st_base = ((wave * 8) * ROWP0 + r16 + 2 * q) * 8;
    f32x2 sA = {0.f, 0.f}, sB = {0.f, 0.f}, qA = {0.f, 0.f}, qB = {0.f, 0.f};       // sums / sums of squares of channels (0, 1), (2, 3)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      f32x4 acc[8];
#pragma unroll
      for (int rr = 0; rr < 8; rr++) acc[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int fr = 0; fr < 14; fr++) {
        const uint2 x0 = *reinterpret_cast<const uint2*>(lds + st_base + (fr * ROWP0 + mt * 16) * 8);
        const uint2 x1 = *reinterpret_cast<const uint2*>(lds + st_base + (fr * ROWP0 + mt * 16) * 8 + 8);
        const bf16x8 xf = __builtin_bit_cast(bf16x8, make_uint4(x0.x, x0.y, x1.x, x1.y));
#pragma unroll
        for (int ky = 0; ky < 7; ky++) {
          const int rr = fr - ky;
          if (rr >= 0 && rr < 8) acc[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky], xf, acc[rr], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rr = 0; rr < 8; rr++) {
        stat16(acc[rr], sA, sB, qA, qB);
        rawp[rr * 4 + mt] = pack4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
      }
    }
    s1[0] = sA[0]; s1[1] = sA[1]; s1[2] = sB[0]; s1[3] = sB[1]; s2[0] = qA[0]; s2[1] = qA[1]; s2[2] = qB[0]; s2[3] = qB[1];
  }
  HEAD_STAMP(2);
  finish_stats<16>(s1, s2, lds, 0, tid, wave, r16, q);
  HEAD_STAMP(3);
  // pixel (y, x) of the 16-channel image lives at frame (y + 1, x + 1); this lane's store slot for tile (rr, mt) = wr_base + const
  const int wr_base = a16(wave * 8 + 1, r16 + 1, q >> 1) + (q & 1) * 8;
  {
    // a0 = relu(GN(raw)) -> LDS (16-channel layout) and the residual registers; the one-pixel frame zeroed
    const f32x2 sc0 = {coef[q * 4], coef[q * 4 + 1]}, sc1 = {coef[q * 4 + 2], coef[q * 4 + 3]};
    const f32x2 sh0 = {coef[32 + q * 4], coef[32 + q * 4 + 1]}, sh1 = {coef[32 + q * 4 + 2], coef[32 + q * 4 + 3]};
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        const P4 o = {relu_pk(pack2(unlo(rawp[rr * 4 + mt]) * sc0 + sh0)), relu_pk(pack2(unhi(rawp[rr * 4 + mt]) * sc1 + sh1))};
        res[rr * 4 + mt] = o;
        *reinterpret_cast<uint2*>(lds + wr_base + (rr * ROWP + mt * 16) * 32) = make_uint2(o.lo, o.hi);
      }
    }
    for (int i = tid; i < 2 * 132 + 64 * 4; i += HTH) {
      int off;
      if (i < 264) off = ((i / 132) * 65 * ROWP) * 32 + (i % 132) * 16;                    // frame rows 0 and 65
      else { const int j = i - 264, row = 1 + (j >> 2); off = (row * ROWP + ((j >> 1) & 1) * 65) * 32 + (j & 1) * 16; }
      *reinterpret_cast<bf16x8*>(lds + off) = zero8;
    }
  }
  lds_barrier();
  HEAD_STAMP(4);

  // ---- layer 1: four 3x3 convs, 16 -> 16, input-row stationary as well.  k-steps of 32 = 2 taps x 16 channels:
  //   F(f) = frame row f, taps kx 0 / 1 (lane quarter q: tap q >> 1, channel half q & 1): used by output rows f - ky, ky = 0..2
  //   G(f) = frame rows f / f + 1, tap kx 2: (ky 0, ky 1) of output row f and (ky 2, zero) of output row f - 2
  // -> 5 MFMAs per output tile as before, but 2 fragment reads per (frame row, column tile) instead of 5 per output tile.
  const int f_base = a16(wave * 8, r16 + (q >> 1), q & 1);
  const int g_base = a16(wave * 8 + (q >> 1), r16 + 2, q & 1);
  const int g_last = a16(wave * 8, r16 + 2, q & 1);          // frame row 9 of the wave: its second half would leave the frame (weights zero)
  bf16x8 wF[3], wG01, wG2;                        // [16][9][16] weights as fragments, fetched one conv ahead
  auto load_w16 = [&](const bf16* __restrict__ w) {
    const bf16* wt = w + (long)r16 * 144 + (q & 1) * 8;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) wF[ky] = *reinterpret_cast<const bf16x8*>(wt + (ky * 3 + (q >> 1)) * 16);
    wG01 = *reinterpret_cast<const bf16x8*>(wt + ((q >> 1) * 3 + 2) * 16);
    wG2 = (q >> 1) == 0 ? *reinterpret_cast<const bf16x8*>(wt + 8 * 16) : zero8;
  };
  load_w16(t.w[1]);
  for (int blk = 0; blk < 2; blk++)
#pragma unroll
  for (int cj = 0; cj < 2; cj++) {
    const int ci = blk * 2 + cj;
    const bool second = cj == 1;                  // conv2 of a basic block: + residual, result becomes the next residual
    f32x2 sA = {0.f, 0.f}, sB = {0.f, 0.f}, qA = {0.f, 0.f}, qB = {0.f, 0.f};       // sums / sums of squares of channels (0, 1), (2, 3)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      f32x4 acc[8];
#pragma unroll
      for (int rr = 0; rr < 8; rr++) acc[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int fr = 0; fr < 10; fr++) {
        const bf16x8 F = *reinterpret_cast<const bf16x8*>(lds + f_base + (fr * ROWP + mt * 16) * 32);
        const bf16x8 G = *reinterpret_cast<const bf16x8*>(lds + (fr == 9 ? g_last : g_base) + (fr * ROWP + mt * 16) * 32);
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const int rr = fr - ky;
          if (rr >= 0 && rr < 8) acc[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wF[ky], F, acc[rr], 0, 0, 0);
        }
        if (fr < 8) acc[fr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wG01, G, acc[fr], 0, 0, 0);
        if (fr >= 2) acc[fr - 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wG2, G, acc[fr - 2], 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 8; rr++) {
        stat16(acc[rr], sA, sB, qA, qB);
        rawp[rr * 4 + mt] = pack4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
      }
    }
    s1[0] = sA[0]; s1[1] = sA[1]; s1[2] = sB[0]; s1[3] = sB[1]; s2[0] = qA[0]; s2[1] = qA[1]; s2[2] = qB[0]; s2[3] = qB[1];
    if (ci < 3) load_w16(t.w[2 + ci]);
    HEAD_STAMP(5 + 2 * ci);
    finish_stats<16>(s1, s2, lds, (1 + ci) * 32, tid, wave, r16, q);
    HEAD_STAMP(6 + 2 * ci);
    const f32x2 sc0 = {coef[q * 4], coef[q * 4 + 1]}, sc1 = {coef[q * 4 + 2], coef[q * 4 + 3]};
    const f32x2 sh0 = {coef[32 + q * 4], coef[32 + q * 4 + 1]}, sh1 = {coef[32 + q * 4 + 2], coef[32 + q * 4 + 3]};
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        f32x2 v0 = unlo(rawp[rr * 4 + mt]) * sc0 + sh0, v1 = unhi(rawp[rr * 4 + mt]) * sc1 + sh1;
        if (second) { v0 += unlo(res[rr * 4 + mt]); v1 += unhi(res[rr * 4 + mt]); }
        const P4 o = {relu_pk(pack2(v0)), relu_pk(pack2(v1))};
        if (second) res[rr * 4 + mt] = o;
        *reinterpret_cast<uint2*>(lds + wr_base + (rr * ROWP + mt * 16) * 32) = make_uint2(o.lo, o.hi);
      }
    }
    lds_barrier();
  }
  HEAD_STAMP(13);
  // ---- layer 2, block 0: 1x1 stride-2 downsample + GroupNorm of the skip -> residual registers (no ReLU).  Output pixel (oy, ox)
  // reads the layer-1 frame at (2 oy + 1, 2 ox + 1); K = 16 channels: the upper half of the k-step carries zero weights.
  {
    const bf16* __restrict__ wt = t.w[5];         // [32][16]
    bf16x8 wD[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) wD[ct] = (q >> 1) == 0 ? *reinterpret_cast<const bf16x8*>(wt + (long)(ct * 16 + r16) * 16 + 8 * (q & 1)) : zero8;
    const int rdD = a16(2 * (wave * 4) + 1, 2 * r16 + 1, q & 1);
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + rdD + (2 * rr * ROWP + 32 * pt) * 32);
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const f32x4 v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wD[ct], xf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          stat32(v, s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]);
          res[(rr * 2 + pt) * 2 + ct] = pack4(v[0], v[1], v[2], v[3]);
        }
      }
    finish_stats<32>(s1, s2, lds, 160, tid, wave, r16, q);
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const int c0 = ct * 16 + q * 4;
      const f32x2 sc0 = {coef[c0], coef[c0 + 1]}, sc1 = {coef[c0 + 2], coef[c0 + 3]};
      const f32x2 sh0 = {coef[32 + c0], coef[32 + c0 + 1]}, sh1 = {coef[32 + c0 + 2], coef[32 + c0 + 3]};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int ti = i * 2 + ct;
        res[ti] = P4{pack2(unlo(res[ti]) * sc0 + sh0), pack2(unhi(res[ti]) * sc1 + sh1)};
      }
    }
  }
  HEAD_STAMP(14);
  bf16x8 W32[9][2];                               // weight fragments of the 32 -> 32 convs: fetched one conv ahead
  // ---- block 0 conv1: 3x3 stride 2, 16 -> 32 (k-steps of 2 taps x 16 channels as in layer 1; the fragment of a column tile serves
  // both cout tiles), GroupNorm + ReLU -> the 32-channel frame, which takes the 16-channel frame's place once every wave is done
  {
    const bf16* __restrict__ wt = t.w[6];         // [32][9][16]
    bf16x8 wA[5][2];
    int rdA[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
      const int k = 32 * s + 8 * q;
#pragma unroll
      for (int ct = 0; ct < 2; ct++) wA[s][ct] = k < 144 ? *reinterpret_cast<const bf16x8*>(wt + (long)(ct * 16 + r16) * 144 + k) : zero8;
      int tap = 2 * s + (q >> 1);
      if (tap > 8) tap = 8;
      const int ky = tap / 3, kx = tap - ky * 3;
      rdA[s] = a16(2 * (wave * 4) + ky, 2 * r16 + kx, q & 1);            // frame (2 oy + ky, 2 ox + kx)
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 5; s++) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + rdA[s] + (2 * rr * ROWP + 32 * pt) * 32);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[s][0], xf, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[s][1], xf, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const f32x4 v = acc[ct];
          stat32(v, s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]);
          rawp[(rr * 2 + pt) * 2 + ct] = pack4(v[0], v[1], v[2], v[3]);
        }
      }
    load_w32(W32, t.w[7], r16, q);
    finish_stats<32>(s1, s2, lds, 160 + 64, tid, wave, r16, q);
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const int c0 = ct * 16 + q * 4;
      const f32x2 sc0 = {coef[c0], coef[c0 + 1]}, sc1 = {coef[c0 + 2], coef[c0 + 3]};
      const f32x2 sh0 = {coef[32 + c0], coef[32 + c0 + 1]}, sh1 = {coef[32 + c0 + 2], coef[32 + c0 + 3]};
      const int wr = a32(wave * 4 + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8;
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          const int ti = (rr * 2 + pt) * 2 + ct;
          const P4 o = {relu_pk(pack2(unlo(rawp[ti]) * sc0 + sh0)), relu_pk(pack2(unhi(rawp[ti]) * sc1 + sh1))};
          *reinterpret_cast<uint2*>(lds + wr + (rr * ROWQ + pt * 16) * 64) = make_uint2(o.lo, o.hi);
        }
    }
    // the one-pixel zero frame of the 32-channel image: rows 0 and 33 (34 px x 4 chunks each), columns 0 and 33 of rows 1..32
    for (int i = tid; i < 2 * 136 + 32 * 8; i += HTH) {
      int off;
      if (i < 272) off = ((i / 136) * 33 * ROWQ) * 64 + (i % 136) * 16;
      else { const int j = i - 272, row = 1 + (j >> 3); off = (row * ROWQ + ((j >> 2) & 1) * 33) * 64 + (j & 3) * 16; }
      *reinterpret_cast<bf16x8*>(lds + off) = zero8;
    }
  }
  lds_barrier();
  HEAD_STAMP(15);
  conv32_gn<true>(W32, t.w[8], 160 + 128, lds, rawp, res, tid, wave, r16, q);          // block 0 conv2 + skip
  HEAD_STAMP(16);
  conv32_gn<false>(W32, t.w[9], 160 + 192, lds, rawp, res, tid, wave, r16, q);         // block 1 conv1
  HEAD_STAMP(17);
  conv32_gn<true>(W32, nullptr, 160 + 256, lds, rawp, res, tid, wave, r16, q);         // block 1 conv2 + identity
  HEAD_STAMP(18);
  // ---- layer-2 output, NHWC bf16 (32 x 32 x 32): the frame's interior, 16 B per lane, consecutive lanes consecutive addresses
  {
    uint4* __restrict__ yo = reinterpret_cast<uint4*>(t.y + (long)b * 1024 * 32);
#pragma unroll 4
    for (int i = tid; i < 4096; i += HTH) {
      const int px = i >> 2, y = px >> 5, x = px & 31;
      yo[i] = *reinterpret_cast<const uint4*>(lds + a32(y + 1, x + 1, i & 3));
    }
  }
  HEAD_STAMP(19);
  (void)B;
}

}  // namespace

bool avlen_tower_head_supported(const avlen_resnet18* n, int S, int C) {
  if (!n || S % 64 || S < 64 || C < 1 || C > 4) return false;
  const avlen_conv& k = n->conv1;
  if (!k.w16 || k.cin16 != 8 || k.cout != 16 || k.kh != 7 || k.kw != 7 || k.stride != 1 || k.pad != 3) return false;
  auto conv3 = [](const avlen_conv& c, int cin, int cout, int stride) {
    return c.w16 && c.cin16 == cin && c.cout == cout && c.kh == 3 && c.kw == 3 && c.stride == stride && c.pad == 1;
  };
  for (int i = 0; i < 2; i++) {
    const avlen_resblock& bl = n->block[i];
    if (bl.has_down || !conv3(bl.conv1, 16, 16, 1) || !conv3(bl.conv2, 16, 16, 1)) return false;
  }
  const avlen_resblock& b2 = n->block[2];
  const avlen_resblock& b3 = n->block[3];
  if (!b2.has_down || !conv3(b2.conv1, 16, 32, 2) || !conv3(b2.conv2, 32, 32, 1) || !b2.conv2.w16f) return false;
  const avlen_conv& d = b2.down;
  if (!d.w16 || d.cin16 != 16 || d.cout != 32 || d.kh != 1 || d.kw != 1 || d.stride != 2 || d.pad != 0) return false;
  return !b3.has_down && conv3(b3.conv1, 32, 32, 1) && conv3(b3.conv2, 32, 32, 1) && b3.conv1.w16f && b3.conv2.w16f;
}

// Y[g] = layer-2 output (post-ReLU) NHWC bf16 (B, 32, 32, 32) of tower g; imgs[g] (B or more images of S x S x C, fp32 or uint8)
int avlen_tower_head_bf16(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                          const float* divisors, const int* row_index, void* const* Y, int groups, int B, int S,
                          hipStream_t stream) {
  if (groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  HeadArgs a = {};
  a.row_index = row_index; a.S = S;
  for (int g = 0; g < groups; g++) {
    const avlen_resnet18* n = nets[g];
    if (!avlen_tower_head_supported(n, S, channels[g])) return AVLEN_ERR_ARG;
    HeadTower& t = a.t[g];
    t.img = imgs[g]; t.u8 = img_u8 ? img_u8[g] : 0; t.C = channels[g]; t.div = divisors[g]; t.y = (bf16*)Y[g];
    t.w[0] = (const bf16*)n->conv1.w16; t.g[0] = n->bn1.g; t.b[0] = n->bn1.b;
    for (int i = 0; i < 2; i++) {
      t.w[1 + 2 * i] = (const bf16*)n->block[i].conv1.w16; t.g[1 + 2 * i] = n->block[i].bn1.g; t.b[1 + 2 * i] = n->block[i].bn1.b;
      t.w[2 + 2 * i] = (const bf16*)n->block[i].conv2.w16; t.g[2 + 2 * i] = n->block[i].bn2.g; t.b[2 + 2 * i] = n->block[i].bn2.b;
    }
    const avlen_resblock& b2 = n->block[2];
    const avlen_resblock& b3 = n->block[3];
    t.w[5] = (const bf16*)b2.down.w16; t.g[5] = b2.bnd.g; t.b[5] = b2.bnd.b;
    t.w[6] = (const bf16*)b2.conv1.w16; t.g[6] = b2.bn1.g; t.b[6] = b2.bn1.b;
    t.w[7] = (const bf16*)b2.conv2.w16f; t.g[7] = b2.bn2.g; t.b[7] = b2.bn2.b;
    t.w[8] = (const bf16*)b3.conv1.w16f; t.g[8] = b3.bn1.g; t.b[8] = b3.bn1.b;
    t.w[9] = (const bf16*)b3.conv2.w16f; t.g[9] = b3.bn2.g; t.b[9] = b3.bn2.b;
  }
  static unsigned long long attr_done = 0;
  if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&tower_head_kernel), HEAD_LDS, &attr_done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  hipLaunchKernelGGL(tower_head_kernel, dim3(B, groups), dim3(HTH), HEAD_LDS, stream, a, B);
  return avlen_launch_status();
}
