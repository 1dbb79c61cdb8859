// Direct 3x3 / stride-1 / pad-1 convolution for the small-channel ResNet stages (C = 16 @64x64, C = 32 @32x32), bf16 in/out.
//
// Why not the implicit GEMM (igemm2.hip) here: with N = 16 or 32 output channels every 128-pixel tile re-gathers its nine
// taps from L2 (15x read amplification incl. K padding), and the L2->LDS path, not HBM or MFMA, bounds the kernel
// (1.7 TB/s algorithmic on MI355X).  This kernel stages an (8+2) x (W+2) x C halo of ONE image in LDS once
// (global_load_lds, 1 KiB per wave-instruction, rows are contiguous in NHWC), keeps the whole 3x3xCxC weight block in
// registers as MFMA fragments, and walks the taps by shifting the LDS read address: each input element is read from
// global memory ~1.25x.  The MFMA is issued "transposed" (weights as the A operand), so a lane ends up with 4 consecutive
// output channels of one pixel and a wave's store covers 16 pixels x 32 B contiguously.  GroupNorm statistics
// (per sample / channel sum and sum of squares of the fp32 accumulators) are reduced in registers, 4 shuffles, and one
// atomic per channel per wave.  Grouped (blockIdx.z = tower), like the rest of the fast path.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_dc[256];     // 1 KiB of zeros: halo rows outside the image

namespace {

struct DcGroups { const bf16* x[8]; const bf16* w[8]; bf16* y[8]; float* stats[8];
                  // NORM: the input is a RAW conv output; GroupNorm(16) + ReLU of it is applied while the halo is staged
                  const float* nstats[8]; const float* ngamma[8]; const float* nbeta[8]; };

template <int C, int CO, int W, int KSZ, bool NORM>      // C input channels (power of two >= 8), CO output channels, image W x W, KSZ x KSZ taps
__global__ __launch_bounds__(256) void dconv3x3_kernel(DcGroups gg, int B) {
  constexpr int TR = 8;                       // output rows per block
  constexpr int PAD = KSZ / 2;
  constexpr int PB = C * 2;                   // bytes per input pixel
  constexpr int ROWB = (W + 2 * PAD) * PB;    // bytes per halo row
  constexpr int NT = CO / 16;                 // output-channel tiles
  constexpr int KTOT = KSZ * KSZ * C;         // reduction length
  constexpr int KSTEPS = (KTOT + 31) / 32;    // MFMA k-steps (a partial tail step multiplies zero weights)
  constexpr int MT_ROW = W / 16;              // 16-pixel tiles per image row
  constexpr int ROWS_PER_WAVE = TR / 4;
  constexpr int HR = TR + 2 * PAD;            // halo rows
  __shared__ __attribute__((aligned(16))) char halo[HR * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int y0 = blockIdx.x * TR, b = blockIdx.y;
  const bf16* __restrict__ x = gg.x[blockIdx.z] + (long)b * W * W * C;
  const bf16* __restrict__ wt = gg.w[blockIdx.z];
  bf16* __restrict__ y = gg.y[blockIdx.z] + (long)b * W * W * CO;
  float* __restrict__ stats = gg.stats[blockIdx.z];

  // ---- weights -> MFMA fragments in registers: lane (r16, q) holds W[cout = j*16 + r16][k = 32 s + 8 q .. +7]
  bf16x8 wf[KSTEPS][NT];
#pragma unroll
  for (int s = 0; s < KSTEPS; s++)
#pragma unroll
    for (int j = 0; j < NT; j++) {
      int k = 32 * s + 8 * q;
      if (k < KTOT) wf[s][j] = *reinterpret_cast<const bf16x8*>(wt + (long)(j * 16 + r16) * KTOT + k);
      else { bf16x8 z; for (int e = 0; e < 8; e++) z[e] = (bf16)0.f; wf[s][j] = z; }
    }

  // ---- halo: rows y0-1 .. y0+TR, interior columns by global_load_lds (1 KiB pieces), border columns zeroed
  constexpr int PIECES_PER_ROW = (W * PB) / 1024;                                 // C=16,W=64: 2 ; C=32,W=32: 2 ; C=8,W=64: 1
  static_assert(PIECES_PER_ROW >= 1 && (W * PB) % 1024 == 0, "halo rows are staged in 1 KiB wave-instructions");
  if constexpr (!NORM) {
    constexpr int NPIECES = HR * PIECES_PER_ROW;
    for (int pc = wave; pc < NPIECES; pc += 4) {
      int hr = pc / PIECES_PER_ROW, part = pc % PIECES_PER_ROW;
      int iy = y0 - PAD + hr;
      const char* src = (iy >= 0 && iy < W) ? (const char*)(x + (long)iy * W * C) + part * 1024 + lane * 16
                                            : (const char*)g_zero_page_dc + lane * 16;
      __builtin_amdgcn_global_load_lds((const void*)src,
          (__attribute__((address_space(3))) void*)(halo + hr * ROWB + PAD * PB + part * 1024), 16, 0, 0);
    }
  } else {
    // Fused GroupNorm(16) + ReLU of the producer: x is the producer's RAW output, its per-(sample, channel) sums are in
    // nstats.  Scale / shift per channel once per block, then every 16-byte chunk (8 channels of one pixel) goes
    // global -> registers -> a*x+b, max 0 -> bf16 -> LDS; rows outside the image stay exactly zero (the conv pads the
    // NORMALISED activation).  Saves the separate apply pass: one read + one write of the activation.
    __shared__ float s_scale[C], s_shift[C];
    constexpr int CPG = C / 16;                                                   // channels per group
    if (tid < 16) {
      const float* st = gg.nstats[blockIdx.z] + (long)b * 2 * C;
      double sum = 0.0, sq = 0.0;
      for (int c = tid * CPG; c < (tid + 1) * CPG; c++) { sum += st[c]; sq += st[C + c]; }
      double n = (double)(W * W) * CPG, mean = sum / n, var = sq / n - mean * mean;
      if (var < 0.0) var = 0.0;
      float rstd = (float)(1.0 / sqrt(var + 1e-5));
      for (int c = tid * CPG; c < (tid + 1) * CPG; c++) {
        float sc = gg.ngamma[blockIdx.z][c] * rstd;
        s_scale[c] = sc; s_shift[c] = gg.nbeta[blockIdx.z][c] - (float)mean * sc;
      }
    }
    __syncthreads();
    constexpr int CH_ROW = W * PB / 16;                                           // 16-byte chunks per image row
    constexpr int CH_PIX = PB / 16;                                               // chunks per pixel
    static_assert((256 % CH_PIX) == 0, "a thread keeps the same 8 channels on every chunk it stages");
    const int c0 = (tid % CH_PIX) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { sc[i] = s_scale[c0 + i]; sh[i] = s_shift[c0 + i]; }
    for (int i = tid; i < HR * CH_ROW; i += 256) {
      const int hr = i / CH_ROW, ch = i - hr * CH_ROW;
      const int iy = y0 - PAD + hr;
      bf16x8 o;
      if (iy >= 0 && iy < W) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>((const char*)(x + (long)iy * W * C) + ch * 16);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (bf16)fmaxf((float)v[e] * sc[e] + sh[e], 0.f);
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(halo + hr * ROWB + PAD * PB + ch * 16) = o;
    }
  }
  // left / right padding columns (PAD pixels each side)
  constexpr int PADCH = PAD * PB / 16;                                            // 16-byte chunks per side per row
  for (int i = tid; i < HR * 2 * PADCH; i += 256) {
    int hr = i / (2 * PADCH), rem = i % (2 * PADCH);
    int side = rem / PADCH, ch = rem % PADCH;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(halo + hr * ROWB + (side ? (W + PAD) * PB : 0) + ch * 16) = z;
  }
  __syncthreads();                            // waits vmcnt(0): the LDS-DMA pieces have landed

  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[j][r] = 0.f; s2[j][r] = 0.f; }

#pragma unroll
  for (int rr = 0; rr < ROWS_PER_WAVE; rr++) {
    const int ry = wave * ROWS_PER_WAVE + rr;             // output row within the tile
#pragma unroll
    for (int mt = 0; mt < MT_ROW; mt++) {
      f32x4 acc[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KSTEPS; s++) {
        int k = 32 * s + 8 * q;                            // this lane's 8 reduction elements: one tap, 8 channels
        int tap = k / C, ci = k % C;
        if (tap > KSZ * KSZ - 1) { tap = KSZ * KSZ - 1; }  // zero weights there; any finite halo data will do
        int ky = tap / KSZ, kx = tap - ky * KSZ;
        const char* ap = halo + (ry + ky) * ROWB + (mt * 16 + r16 + kx) * PB + ci * 2;
        bf16x8 xf = *reinterpret_cast<const bf16x8*>(ap);
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], xf, acc[j], 0, 0, 0);
      }
      // D[cout = q*4 + r][pixel = r16]
      const int px = mt * 16 + r16, oy = y0 + ry;
#pragma unroll
      for (int j = 0; j < NT; j++) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float v = acc[j][r];
          s1[j][r] += v; s2[j][r] += v * v;
          o[r] = (bf16)v;
        }
        *reinterpret_cast<bf16x4*>(y + ((long)oy * W + px) * CO + j * 16 + q * 4) = o;
      }
    }
  }
  // ---- GroupNorm statistics: reduce over the 16 pixel lanes, then over the block's 4 waves in LDS, then ONE atomic
  // wave-instruction per block (2*CO consecutive floats of stats[b]).  Per-wave atomics (128 four-lane instructions per
  // block) cost 7 of the kernel's 26 us.
  if (stats) {
    __shared__ float bst[4][2][CO];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float a = s1[j][r], c = s2[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (r16 == 0) {
          int ch = j * 16 + q * 4 + r;
          bst[wave][0][ch] = a; bst[wave][1][ch] = c;
        }
      }
    __syncthreads();
    if (tid < 2 * CO) {
      const int which = tid / CO, ch = tid % CO;
      const float v = (bst[0][which][ch] + bst[1][which][ch]) + (bst[2][which][ch] + bst[3][which][ch]);
      atomicAdd(&stats[((long)b * 2 + which) * CO + ch], v);
    }
  }
}

}  // namespace

// X: NHWC bf16 (B, W, W, Cin); Y: (B, W, W, Cout) bf16; Wp: bf16 [Cout][K][K][Cin]; stats (optional, pre-zeroed):
// [B][2][Cout] fp32.  Stride 1, "same" padding.
int avlen_dconv_bf16_grouped(const void* const* X, const void* const* Wp, void* const* Y16, float* const* gn_stats,
                             int groups, int B, int W, int Cin, int Cout, int K, hipStream_t stream,
                             const float* const* in_stats, const float* const* in_gamma, const float* const* in_beta) {
  if (groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  DcGroups gg = {};
  const bool norm = in_stats != nullptr;
  for (int g = 0; g < groups; g++) {
    gg.x[g] = (const bf16*)X[g]; gg.w[g] = (const bf16*)Wp[g]; gg.y[g] = (bf16*)Y16[g];
    gg.stats[g] = gn_stats ? gn_stats[g] : nullptr;
    if (norm) { gg.nstats[g] = in_stats[g]; gg.ngamma[g] = in_gamma[g]; gg.nbeta[g] = in_beta[g]; }
  }
  dim3 grid(W / 8, B, groups), block(256);
  if (Cin == 16 && Cout == 16 && W == 64 && K == 3) {
    if (norm) hipLaunchKernelGGL((dconv3x3_kernel<16, 16, 64, 3, true>), grid, block, 0, stream, gg, B);
    else hipLaunchKernelGGL((dconv3x3_kernel<16, 16, 64, 3, false>), grid, block, 0, stream, gg, B);
  } else if (Cin == 32 && Cout == 32 && W == 32 && K == 3) {
    if (norm) hipLaunchKernelGGL((dconv3x3_kernel<32, 32, 32, 3, true>), grid, block, 0, stream, gg, B);
    else hipLaunchKernelGGL((dconv3x3_kernel<32, 32, 32, 3, false>), grid, block, 0, stream, gg, B);
  } else if (Cin == 8 && Cout == 16 && W == 64 && K == 7 && !norm)
    hipLaunchKernelGGL((dconv3x3_kernel<8, 16, 64, 7, false>), grid, block, 0, stream, gg, B);
  else
    return AVLEN_ERR_ARG;
  return avlen_launch_status();
}

bool avlen_dconv_supported(int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  if (KH != KW || stride != 1 || pad != KH / 2) return false;
  return (KH == 3 && Cin == 16 && Cout == 16 && W == 64) || (KH == 3 && Cin == 32 && Cout == 32 && W == 32) ||
         (KH == 7 && Cin == 8 && Cout == 16 && W == 64);
}

extern "C" int avlen_conv_direct_bf16(const void* X, const void* Wp, void* Y16, float* gn_stats, int B, int W, int Cin,
                                      int Cout, int K, hipStream_t stream) {
  if (!avlen_dconv_supported(W, Cin, Cout, K, K, 1, K / 2)) return AVLEN_ERR_ARG;
  return avlen_dconv_bf16_grouped(&X, &Wp, &Y16, gn_stats ? &gn_stats : nullptr, 1, B, W, Cin, Cout, K, stream, nullptr, nullptr,
                                  nullptr);
}
