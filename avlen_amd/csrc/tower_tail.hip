// Layers 3 and 4 of the CustomResNet tower (smt_resnet.py:37-53, 100-131: per layer a stride-2 basic block with a 1x1 stride-2
// downsample + GroupNorm on the skip, then a stride-1 basic block; 32 -> 64 channels at 16x16, 64 -> 128 at 8x8) as ONE launch:
// one workgroup per image, the activation of the current stage as a zero-framed bf16 image in LDS (34x34x32 input, 18x18x64,
// 10x10x128), every raw conv output and the residual in registers, the conv weights as MFMA fragments in REGISTERS -- a wave
// multiplies only its own 16 output channels, so a weight is fetched from L2 exactly once per workgroup and never passes
// through LDS -- read from the fragment-order copy avlen_conv::w16f (1 KiB contiguous per wave and k-step; from the [cout][K]
// layout a load touched 16 half-used lines and the 288 KiB of a layer-4 conv took 7 us per workgroup) one conv ahead, and the
// GroupNorm statistics from the fp32 accumulators (fixed order, no atomics).
//
// Work split (8 waves): layer 3 -- wave w owns cout tile w & 3 and output rows 8 (w >> 2) .. + 7 (an MFMA column tile = one
// 16-pixel row); layer 4 -- wave w owns cout tile w and all four column tiles (two 8-pixel rows each).  The stride-1 convs
// run input-row stationary: the kx-shifted fragments of a frame row are read once and feed the output rows row - ky.
// GroupNorm(16): 64 channels -> a lane's four accumulator channels are exactly one group; 128 channels -> half a group.
//
// Same arithmetic as the launch-per-layer path: bf16 operands, fp32 MFMA accumulation over [ky][kx][c], raw conv outputs
// rounded to bf16 before the normalisation, statistics from the fp32 accumulators, var = E[x^2] - mean^2 in double.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include "tower_util.h"

namespace {

constexpr int TTH = 512;
constexpr int R32 = 34, R64 = 18, R128 = 10;      // frame edge (pixels) of the 32- / 64- / 128-channel images
constexpr int FRAME_BYTES = R32 * R32 * 64;       // 73984: the largest frame; the later ones reuse its space
constexpr int TPART_OFF = FRAME_BYTES;            // [8 waves][4 lane quarters][2] fp32 (+ pad)
constexpr int TCOEF_OFF = TPART_OFF + 1024;       // scale[128], shift[128]
constexpr int TGB_OFF = TCOEF_OFF + 1024;         // gamma | beta of the ten GroupNorms: 5 x (64 + 64), 5 x (128 + 128) fp32
constexpr int TAIL_LDS = TGB_OFF + (5 * 128 + 5 * 256) * 4;
static_assert(R64 * R64 * 128 <= FRAME_BYTES && R128 * R128 * 256 <= FRAME_BYTES && TAIL_LDS <= 160 * 1024, "tower tail LDS budget");

// w / g / b per layer (l = 0: layer 3, l = 1: layer 4), index 5 l + {0 downsample, 1 block 0 conv1 (stride 2), 2 block 0 conv2,
// 3 block 1 conv1, 4 block 1 conv2}
struct TailTower { const bf16* x; bf16* y; const bf16* w[10]; const float* g[10]; const float* b[10]; };
struct TailArgs { TailTower t[8]; long long* prof; };
#ifdef AVLEN_TAIL_PROF          // tools/tail_lab.hip: phase timestamps of one wave of every workgroup
#define TAIL_STAMP(k) do { if (args.prof && tid == AVLEN_TAIL_PROF) args.prof[(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif

// frames: [row][pixel][C * 2 bytes], 16-byte chunks rotated by the pixel index so that neighbouring pixels' reads of one chunk
// fall on different bank groups; + 16 pixels (a column tile) leaves the rotation unchanged
__device__ __forceinline__ int a32(int y, int p, int chunk) { return (y * R32 + p) * 64 + ((chunk ^ ((p >> 1) & 3)) << 4); }
__device__ __forceinline__ int a64(int y, int p, int chunk) { return (y * R64 + p) * 128 + ((chunk ^ (p & 7)) << 4); }
__device__ __forceinline__ int a128(int y, int p, int chunk) { return (y * R128 + p) * 256 + ((chunk ^ (p & 7)) << 4); }

// per-lane sums -> scale / shift of NCH channels.  NCH = 64: the lane's sums are those of group (cout tile) * 4 + q over the
// wave's 8 rows (the group's other half: wave + 4); NCH = 128: of half a group, (cout tile = wave) * 2 + (q >> 1).
template <int NCH>
__device__ __forceinline__ void finish_stats_t(float s1, float s2, char* lds, int gb, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + TGB_OFF) + gb;
  const float* beta = gamma + NCH;
  float2* part = reinterpret_cast<float2*>(lds + TPART_OFF);
  float* coef = reinterpret_cast<float*>(lds + TCOEF_OFF);
  const float a = row16_sum(s1), c = row16_sum(s2);
  if (r16 == 0) part[wave * 4 + q] = make_float2(a, c);
  lds_barrier();                                  // every wave has also finished reading the image of this conv
  if (tid < NCH) {
    float2 u, v;
    if (NCH == 64) { const int g = tid >> 2, ct = g >> 2, qq = g & 3; u = part[ct * 4 + qq]; v = part[(ct + 4) * 4 + qq]; }
    else { const int w = tid >> 4, qq = (tid >> 2) & 2; u = part[w * 4 + qq]; v = part[w * 4 + qq + 1]; }
    constexpr double inv_n = NCH == 64 ? 1.0 / 1024.0 : 1.0 / 512.0;
    const double mean = ((double)u.x + (double)v.x) * inv_n;
    double var = ((double)u.y + (double)v.y) * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
    const float sc = gamma[tid] * rstd;
    coef[tid] = sc; coef[128 + tid] = beta[tid] - (float)mean * sc;
  }
  lds_barrier();
}

__device__ __forceinline__ void stat1(const f32x4& v, float& s1, float& s2) {
  s1 += (v[0] + v[1]) + (v[2] + v[3]);
  s2 = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], s2))));
}

// y = [relu](raw * scale + shift [+ res]) for NT packed tiles of the lane's four channels c0 .. c0 + 3 -> LDS (8 bytes at
// wr + tile * tile_stride) and, with SECOND, the residual registers
template <int NT, bool SECOND>
__device__ __forceinline__ void apply_tiles(const P4 (&rawp)[NT], P4 (&res)[NT], const float* coef, int c0, char* lds, int wr, int tile_stride) {
  const f32x2 sc0 = {coef[c0], coef[c0 + 1]}, sc1 = {coef[c0 + 2], coef[c0 + 3]};
  const f32x2 sh0 = {coef[128 + c0], coef[128 + c0 + 1]}, sh1 = {coef[128 + c0 + 2], coef[128 + c0 + 3]};
#pragma unroll
  for (int i = 0; i < NT; i++) {
    f32x2 v0 = unlo(rawp[i]) * sc0 + sh0, v1 = unhi(rawp[i]) * sc1 + sh1;
    if (SECOND) { v0 += unlo(res[i]); v1 += unhi(res[i]); }
    const P4 o = {relu_pk(pack2(v0)), relu_pk(pack2(v1))};
    if (SECOND) res[i] = o;
    *reinterpret_cast<uint2*>(lds + wr + i * tile_stride) = make_uint2(o.lo, o.hi);
  }
}

// ---- layer 3: 3x3 stride-1 conv 64 -> 64 over the 16x16 frame.  W[(tap) * 2 + half]: the lane's fragment of its cout tile.
__device__ __forceinline__ void load_w64(bf16x8 (&W)[18], const bf16* __restrict__ wt, int ct, int r16, int q) {
#pragma unroll
  for (int i = 0; i < 18; i++) W[i] = *reinterpret_cast<const bf16x8*>(wt + ((long)(ct * 18 + i) * 64 + q * 16 + r16) * 8);
}
template <bool SECOND>
__device__ __forceinline__ void conv64_gn(bf16x8 (&W)[18], const bf16* __restrict__ next_wt, int gb, char* lds, P4 (&rawp)[8], P4 (&res)[8],
                                          int tid, int wave, int r16, int q) {
  const int ct = wave & 3, half = wave >> 2;
  int rd[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; kx++)
#pragma unroll
    for (int hf = 0; hf < 2; hf++) rd[kx][hf] = a64(half * 8, r16 + kx, 4 * hf + q);
  f32x4 acc[8];
#pragma unroll
  for (int rr = 0; rr < 8; rr++) acc[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 10; f++) {
    bf16x8 F[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; kx++)
#pragma unroll
      for (int hf = 0; hf < 2; hf++) F[kx][hf] = *reinterpret_cast<const bf16x8*>(lds + rd[kx][hf] + f * (R64 * 128));
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int rr = f - ky;
      if (rr >= 0 && rr < 8) {
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int hf = 0; hf < 2; hf++)
            acc[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[(ky * 3 + kx) * 2 + hf], F[kx][hf], acc[rr], 0, 0, 0);
      }
    }
  }
  if (next_wt) load_w64(W, next_wt, ct, r16, q);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int rr = 0; rr < 8; rr++) { stat1(acc[rr], s1, s2); rawp[rr] = pack4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]); }
  finish_stats_t<64>(s1, s2, lds, gb, tid, wave, r16, q);
  apply_tiles<8, SECOND>(rawp, res, reinterpret_cast<const float*>(lds + TCOEF_OFF), ct * 16 + q * 4, lds,
                         a64(half * 8 + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8, R64 * 128);
  lds_barrier();
}

// ---- layer 4: 3x3 stride-1 conv 128 -> 128 over the 8x8 frame.  Column tile pt = output rows 2 pt, 2 pt + 1 (lane: row
// r16 >> 3, pixel r16 & 7).  The fragment of frame rows (f, f + 1) serves every (pt, ky) with 2 pt + ky = f.
__device__ __forceinline__ void load_w128(bf16x8 (&W)[36], const bf16* __restrict__ wt, int ct, int r16, int q) {
#pragma unroll
  for (int i = 0; i < 36; i++) W[i] = *reinterpret_cast<const bf16x8*>(wt + ((long)(ct * 36 + i) * 64 + q * 16 + r16) * 8);
}
template <bool SECOND>
__device__ __forceinline__ void conv128_gn(bf16x8 (&W)[36], const bf16* __restrict__ next_wt, int gb, char* lds, P4 (&rawp)[4], P4 (&res)[4],
                                           int tid, int wave, int r16, int q) {
  const int ct = wave, ly = r16 >> 3, lx = r16 & 7;
  int rd[3][2];                                   // chunk 4 j + q: j even / odd (j >> 1 adds 8 chunks = 128 B)
#pragma unroll
  for (int kx = 0; kx < 3; kx++)
#pragma unroll
    for (int jo = 0; jo < 2; jo++) rd[kx][jo] = a128(ly, lx + kx, 4 * jo + q);
  f32x4 acc[4];
#pragma unroll
  for (int pt = 0; pt < 4; pt++) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 9; f++) {
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      bf16x8 F[4];
#pragma unroll
      for (int j = 0; j < 4; j++) F[j] = *reinterpret_cast<const bf16x8*>(lds + rd[kx][j & 1] + (j >> 1) * 128 + f * (R128 * 256));
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int pt2 = f - ky;                   // = 2 pt
        if (pt2 >= 0 && pt2 < 8 && (pt2 & 1) == 0) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[pt2 >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[(ky * 3 + kx) * 4 + j], F[j], acc[pt2 >> 1], 0, 0, 0);
        }
      }
    }
  }
  if (next_wt) load_w128(W, next_wt, ct, r16, q);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int pt = 0; pt < 4; pt++) { stat1(acc[pt], s1, s2); rawp[pt] = pack4(acc[pt][0], acc[pt][1], acc[pt][2], acc[pt][3]); }
  finish_stats_t<128>(s1, s2, lds, gb, tid, wave, r16, q);
  apply_tiles<4, SECOND>(rawp, res, reinterpret_cast<const float*>(lds + TCOEF_OFF), ct * 16 + q * 4, lds,
                         a128(ly + 1, lx + 1, ct * 2 + (q >> 1)) + (q & 1) * 8, 2 * R128 * 256);
  lds_barrier();
}

__global__ __launch_bounds__(TTH) void tower_tail_kernel(TailArgs args, int Bn) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const TailTower& t = args.t[blockIdx.y];
  const int img = blockIdx.x;
  const float* coef = reinterpret_cast<const float*>(lds + TCOEF_OFF);
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; e++) zero8[e] = (bf16)0.f;
  TAIL_STAMP(0);

  // =========================================================== layer 3 ===========================================================
  const int ct3 = wave & 3, half = wave >> 2;
  // weights of the downsample (K = 32: one k-step) and of the stride-2 conv (K = 288: one tap per k-step): first, under the image load
  bf16x8 wD3 = *reinterpret_cast<const bf16x8*>(t.w[0] + ((long)ct3 * 64 + lane) * 8);
  bf16x8 wA3[9];
#pragma unroll
  for (int tap = 0; tap < 9; tap++) wA3[tap] = *reinterpret_cast<const bf16x8*>(t.w[1] + ((long)(ct3 * 9 + tap) * 64 + lane) * 8);
  // GroupNorm affine parameters -> LDS; the layer-2 output (32 x 32 x 32, NHWC) -> the zero-framed 34 x 34 image
  for (int i = tid; i < 5 * 128 + 5 * 256; i += TTH) {
    const bool l3 = i < 640;
    const int n = l3 ? i >> 7 : 5 + ((i - 640) >> 8), j = l3 ? i & 127 : (i - 640) & 255, nch = l3 ? 64 : 128;
    reinterpret_cast<float*>(lds + TGB_OFF)[i] = j < nch ? t.g[n][j] : t.b[n][j - nch];
  }
  {
    const uint4* __restrict__ x = reinterpret_cast<const uint4*>(t.x + (long)img * 1024 * 32);
#pragma unroll 4
    for (int i = tid; i < 4096; i += TTH) {
      const int px = i >> 2, y = px >> 5, xx = px & 31;
      *reinterpret_cast<uint4*>(lds + a32(y + 1, xx + 1, i & 3)) = x[i];
    }
    for (int i = tid; i < 2 * 136 + 32 * 8; i += TTH) {
      int off;
      if (i < 272) off = ((i / 136) * 33 * R32) * 64 + (i % 136) * 16;
      else { const int j = i - 272, row = 1 + (j >> 3); off = (row * R32 + ((j >> 2) & 1) * 33) * 64 + (j & 3) * 16; }
      *reinterpret_cast<bf16x8*>(lds + off) = zero8;
    }
  }
  lds_barrier();
  TAIL_STAMP(1);

  P4 rawp3[8], res3[8];
  // ---- downsample 1x1 stride 2 + GroupNorm (no ReLU) -> residual registers.  Output (oy, ox) reads frame (2 oy + 1, 2 ox + 1).
  {
    const int rdD = a32(2 * (half * 8) + 1, 2 * r16 + 1, q);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + rdD + rr * (2 * R32 * 64));
      const f32x4 v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wD3, xf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      stat1(v, s1, s2);
      res3[rr] = pack4(v[0], v[1], v[2], v[3]);
    }
    finish_stats_t<64>(s1, s2, lds, 0, tid, wave, r16, q);
    const int c0 = ct3 * 16 + q * 4;
    const f32x2 sc0 = {coef[c0], coef[c0 + 1]}, sc1 = {coef[c0 + 2], coef[c0 + 3]};
    const f32x2 sh0 = {coef[128 + c0], coef[128 + c0 + 1]}, sh1 = {coef[128 + c0 + 2], coef[128 + c0 + 3]};
#pragma unroll
    for (int rr = 0; rr < 8; rr++) res3[rr] = P4{pack2(unlo(res3[rr]) * sc0 + sh0), pack2(unhi(res3[rr]) * sc1 + sh1)};
  }
  TAIL_STAMP(2);
  bf16x8 W64[18];
  // ---- block 0 conv1: 3x3 stride 2, 32 -> 64: frame (2 oy + ky, 2 ox + kx), chunk q; GroupNorm + ReLU -> the 64-channel frame
  {
    int rdA[3];
#pragma unroll
    for (int kx = 0; kx < 3; kx++) rdA[kx] = a32(2 * (half * 8), 2 * r16 + kx, q);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + rdA[kx] + (2 * rr + ky) * (R32 * 64));
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA3[ky * 3 + kx], xf, acc, 0, 0, 0);
        }
      stat1(acc, s1, s2);
      rawp3[rr] = pack4(acc[0], acc[1], acc[2], acc[3]);
    }
    load_w64(W64, t.w[2], ct3, r16, q);
    finish_stats_t<64>(s1, s2, lds, 128, tid, wave, r16, q);
    apply_tiles<8, false>(rawp3, res3, coef, ct3 * 16 + q * 4, lds, a64(half * 8 + 1, r16 + 1, ct3 * 2 + (q >> 1)) + (q & 1) * 8, R64 * 128);
    // zero frame of the 64-channel image: rows 0 and 17 (18 px x 8 chunks each), columns 0 and 17 of rows 1..16
    for (int i = tid; i < 2 * 144 + 16 * 16; i += TTH) {
      int off;
      if (i < 288) off = ((i / 144) * 17 * R64) * 128 + (i % 144) * 16;
      else { const int j = i - 288, row = 1 + (j >> 4); off = (row * R64 + ((j >> 3) & 1) * 17) * 128 + (j & 7) * 16; }
      *reinterpret_cast<bf16x8*>(lds + off) = zero8;
    }
  }
  lds_barrier();
  TAIL_STAMP(3);
  conv64_gn<true>(W64, t.w[3], 256, lds, rawp3, res3, tid, wave, r16, q);           // block 0 conv2 + skip
  TAIL_STAMP(4);
  conv64_gn<false>(W64, t.w[4], 384, lds, rawp3, res3, tid, wave, r16, q);          // block 1 conv1
  TAIL_STAMP(5);
  conv64_gn<true>(W64, nullptr, 512, lds, rawp3, res3, tid, wave, r16, q);          // block 1 conv2 + identity
  TAIL_STAMP(6);

  // =========================================================== layer 4 ===========================================================
  const int ct4 = wave, ly = r16 >> 3, lx = r16 & 7;
  P4 rawp4[4], res4[4];
  bf16x8 W128[36];
  // ---- downsample 1x1 stride 2, 64 -> 128 (K = 64: two k-steps) + GroupNorm -> residual registers
  {
    bf16x8 wD[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) wD[hf] = *reinterpret_cast<const bf16x8*>(t.w[5] + ((long)(ct4 * 2 + hf) * 64 + lane) * 8);
    int rdD[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) rdD[hf] = a64(2 * ly + 1, 2 * lx + 1, 4 * hf + q);
    // the stride-2 conv's weights (K = 576: 18 k-steps) into the first half of W128, under the downsample
#pragma unroll
    for (int i = 0; i < 18; i++) W128[i] = *reinterpret_cast<const bf16x8*>(t.w[6] + ((long)(ct4 * 18 + i) * 64 + lane) * 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < 2; hf++)
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wD[hf], *reinterpret_cast<const bf16x8*>(lds + rdD[hf] + pt * (4 * R64 * 128)), v, 0, 0, 0);
      stat1(v, s1, s2);
      res4[pt] = pack4(v[0], v[1], v[2], v[3]);
    }
    finish_stats_t<128>(s1, s2, lds, 640, tid, wave, r16, q);
    const int c0 = ct4 * 16 + q * 4;
    const f32x2 sc0 = {coef[c0], coef[c0 + 1]}, sc1 = {coef[c0 + 2], coef[c0 + 3]};
    const f32x2 sh0 = {coef[128 + c0], coef[128 + c0 + 1]}, sh1 = {coef[128 + c0 + 2], coef[128 + c0 + 3]};
#pragma unroll
    for (int pt = 0; pt < 4; pt++) res4[pt] = P4{pack2(unlo(res4[pt]) * sc0 + sh0), pack2(unhi(res4[pt]) * sc1 + sh1)};
  }
  TAIL_STAMP(7);
  // ---- block 0 conv1: 3x3 stride 2, 64 -> 128: frame (2 oy + ky, 2 ox + kx), chunks 4 hf + q
  {
    int rdA[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; kx++)
#pragma unroll
      for (int hf = 0; hf < 2; hf++) rdA[kx][hf] = a64(2 * ly, 2 * lx + kx, 4 * hf + q);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + rdA[kx][hf] + (4 * pt + ky) * (R64 * 128));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W128[(ky * 3 + kx) * 2 + hf], xf, acc, 0, 0, 0);
          }
      stat1(acc, s1, s2);
      rawp4[pt] = pack4(acc[0], acc[1], acc[2], acc[3]);
    }
    load_w128(W128, t.w[7], ct4, r16, q);
    finish_stats_t<128>(s1, s2, lds, 640 + 256, tid, wave, r16, q);
    apply_tiles<4, false>(rawp4, res4, coef, ct4 * 16 + q * 4, lds, a128(ly + 1, lx + 1, ct4 * 2 + (q >> 1)) + (q & 1) * 8, 2 * R128 * 256);
    // zero frame of the 128-channel image: rows 0 and 9 (10 px x 16 chunks each), columns 0 and 9 of rows 1..8
    for (int i = tid; i < 2 * 160 + 8 * 32; i += TTH) {
      int off;
      if (i < 320) off = ((i / 160) * 9 * R128) * 256 + (i % 160) * 16;
      else { const int j = i - 320, row = 1 + (j >> 5); off = (row * R128 + ((j >> 4) & 1) * 9) * 256 + (j & 15) * 16; }
      *reinterpret_cast<bf16x8*>(lds + off) = zero8;
    }
  }
  lds_barrier();
  TAIL_STAMP(8);
  conv128_gn<true>(W128, t.w[8], 640 + 512, lds, rawp4, res4, tid, wave, r16, q);    // block 0 conv2 + skip
  TAIL_STAMP(9);
  conv128_gn<false>(W128, t.w[9], 640 + 768, lds, rawp4, res4, tid, wave, r16, q);   // block 1 conv1
  TAIL_STAMP(10);
  conv128_gn<true>(W128, nullptr, 640 + 1024, lds, rawp4, res4, tid, wave, r16, q);  // block 1 conv2 + identity
  TAIL_STAMP(11);
  // ---- layer-4 output, NHWC bf16 (8 x 8 x 128): the frame's interior
  {
    uint4* __restrict__ yo = reinterpret_cast<uint4*>(t.y + (long)img * 64 * 128);
    for (int i = tid; i < 1024; i += TTH) {
      const int px = i >> 4, y = px >> 3, x = px & 7;
      yo[i] = *reinterpret_cast<const uint4*>(lds + a128(y + 1, x + 1, i & 15));
    }
  }
  TAIL_STAMP(12);
  (void)Bn;
}

}  // namespace

// Layers 3 + 4 of `groups` (<= 8) towers: X[g] = layer-2 output NHWC bf16 (B, 32, 32, 32); Y[g] = layer-4 output NHWC bf16
// (B, 8, 8, 128).  Conv weights: the fragment-order copies avlen_conv::w16f; every tower must have the 32 -> 64 -> 128
// channel plan.
int avlen_tower_tail_bf16(const avlen_resnet18* const* nets, const void* const* X, void* const* Y, int groups, int B,
                          hipStream_t stream) {
  if (groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  TailArgs a = {};
  for (int g = 0; g < groups; g++) {
    const avlen_resnet18* n = nets[g];
    TailTower& t = a.t[g];
    t.x = (const bf16*)X[g]; t.y = (bf16*)Y[g];
    for (int l = 0; l < 2; l++) {                          // l = 0: layer 3 (blocks 4, 5); l = 1: layer 4 (blocks 6, 7)
      const avlen_resblock& b0 = n->block[4 + 2 * l]; const avlen_resblock& b1 = n->block[5 + 2 * l];
      const int cin = l == 0 ? 32 : 64, co = l == 0 ? 64 : 128;
      auto conv3 = [](const avlen_conv& c, int ci, int cout, int stride) {
        return c.w16 && c.w16f && c.cin16 == ci && c.cout == cout && c.kh == 3 && c.kw == 3 && c.stride == stride && c.pad == 1;
      };
      const avlen_conv& d = b0.down;
      if (!b0.has_down || b1.has_down || !conv3(b0.conv1, cin, co, 2) || !conv3(b0.conv2, co, co, 1) || !conv3(b1.conv1, co, co, 1) ||
          !conv3(b1.conv2, co, co, 1) || !d.w16 || !d.w16f || d.cin16 != cin || d.cout != co || d.kh != 1 || d.kw != 1 || d.stride != 2 || d.pad != 0)
        return AVLEN_ERR_ARG;
      const int o = 5 * l;
      t.w[o + 0] = (const bf16*)d.w16f; t.g[o + 0] = b0.bnd.g; t.b[o + 0] = b0.bnd.b;
      t.w[o + 1] = (const bf16*)b0.conv1.w16f; t.g[o + 1] = b0.bn1.g; t.b[o + 1] = b0.bn1.b;
      t.w[o + 2] = (const bf16*)b0.conv2.w16f; t.g[o + 2] = b0.bn2.g; t.b[o + 2] = b0.bn2.b;
      t.w[o + 3] = (const bf16*)b1.conv1.w16f; t.g[o + 3] = b1.bn1.g; t.b[o + 3] = b1.bn1.b;
      t.w[o + 4] = (const bf16*)b1.conv2.w16f; t.g[o + 4] = b1.bn2.g; t.b[o + 4] = b1.bn2.b;
    }
  }
  static unsigned long long attr_done = 0;
  if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&tower_tail_kernel), TAIL_LDS, &attr_done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  hipLaunchKernelGGL(tower_tail_kernel, dim3(B, groups), dim3(TTH), TAIL_LDS, stream, a, B);
  return avlen_launch_status();
}
