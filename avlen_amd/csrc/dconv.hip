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

struct DcGroups { const bf16* x[8]; const bf16* w[8]; bf16* y[8]; float* stats[8]; };

template <int C, int W>      // C channels in = out, image W x W
__global__ __launch_bounds__(256) void dconv3x3_kernel(DcGroups gg, int B) {
  constexpr int TR = 8;                       // output rows per block
  constexpr int PB = C * 2;                   // bytes per pixel
  constexpr int ROWB = (W + 2) * PB;          // bytes per halo row
  constexpr int NT = C / 16;                  // output-channel tiles
  constexpr int KS = 9 * C / 32;              // MFMA k-steps (C=16: 4.5 -> 5, the tail half is zero weights)
  constexpr int KSTEPS = (9 * C + 31) / 32;
  constexpr int MT_ROW = W / 16;              // 16-pixel tiles per image row
  constexpr int ROWS_PER_WAVE = TR / 4;
  (void)KS;
  __shared__ __attribute__((aligned(16))) char halo[(TR + 2) * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int y0 = blockIdx.x * TR, b = blockIdx.y;
  const bf16* __restrict__ x = gg.x[blockIdx.z] + (long)b * W * W * C;
  const bf16* __restrict__ wt = gg.w[blockIdx.z];
  bf16* __restrict__ y = gg.y[blockIdx.z] + (long)b * W * W * C;
  float* __restrict__ stats = gg.stats[blockIdx.z];

  // ---- weights -> MFMA fragments in registers: lane (r16, q) holds W[cout = j*16 + r16][k = 32 s + 8 q .. +7]
  bf16x8 wf[KSTEPS][NT];
#pragma unroll
  for (int s = 0; s < KSTEPS; s++)
#pragma unroll
    for (int j = 0; j < NT; j++) {
      int k = 32 * s + 8 * q;
      if (k < 9 * C) wf[s][j] = *reinterpret_cast<const bf16x8*>(wt + (long)(j * 16 + r16) * 9 * C + k);
      else { bf16x8 z; for (int e = 0; e < 8; e++) z[e] = (bf16)0.f; wf[s][j] = z; }
    }

  // ---- halo: rows y0-1 .. y0+TR, interior columns by global_load_lds (1 KiB pieces), border columns zeroed
  constexpr int PIECES_PER_ROW = (W * PB) / 1024 > 0 ? (W * PB) / 1024 : 1;      // C=16,W=64: 2 ; C=32,W=32: 2
  constexpr int PIECE_B = (W * PB) / PIECES_PER_ROW;                              // bytes per piece (1024)
  static_assert(PIECE_B == 1024, "halo rows are staged in 1 KiB wave-instructions");
  constexpr int NPIECES = (TR + 2) * PIECES_PER_ROW;                              // 20
  for (int pc = wave; pc < NPIECES; pc += 4) {
    int hr = pc / PIECES_PER_ROW, part = pc % PIECES_PER_ROW;
    int iy = y0 - 1 + hr;
    const char* src = (iy >= 0 && iy < W) ? (const char*)(x + (long)iy * W * C) + part * 1024 + lane * 16
                                          : (const char*)g_zero_page_dc + lane * 16;
    __builtin_amdgcn_global_load_lds((const void*)src,
        (__attribute__((address_space(3))) void*)(halo + hr * ROWB + PB + part * 1024), 16, 0, 0);
  }
  // left / right padding columns
  for (int i = tid; i < (TR + 2) * 2 * (PB / 16); i += 256) {
    int hr = i / (2 * (PB / 16)), rem = i % (2 * (PB / 16));
    int side = rem / (PB / 16), ch = rem % (PB / 16);
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(halo + hr * ROWB + (side ? (W + 1) * PB : 0) + ch * 16) = z;
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
        if (tap > 8) { tap = 8; }                          // zero weights there; any finite halo data will do
        int ky = tap / 3, kx = tap - ky * 3;
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
        *reinterpret_cast<bf16x4*>(y + ((long)oy * W + px) * C + j * 16 + q * 4) = o;
      }
    }
  }
  // ---- GroupNorm statistics: reduce over the 16 pixel lanes, one atomic per channel per wave
  if (stats) {
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float a = s1[j][r], c = s2[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (r16 == 0) {
          int ch = j * 16 + q * 4 + r;
          atomicAdd(&stats[((long)b * 2) * C + ch], a);
          atomicAdd(&stats[((long)b * 2 + 1) * C + ch], c);
        }
      }
  }
}

}  // namespace

// X, Y: NHWC bf16 (B, W, W, C); Wp: bf16 [C][3][3][C]; stats (optional, pre-zeroed): [B][2][C] fp32.
int avlen_dconv3x3_bf16_grouped(const void* const* X, const void* const* Wp, void* const* Y16, float* const* gn_stats,
                                int groups, int B, int W, int C, hipStream_t stream) {
  if (groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  DcGroups gg = {};
  for (int g = 0; g < groups; g++) {
    gg.x[g] = (const bf16*)X[g]; gg.w[g] = (const bf16*)Wp[g]; gg.y[g] = (bf16*)Y16[g];
    gg.stats[g] = gn_stats ? gn_stats[g] : nullptr;
  }
  if (C == 16 && W == 64)
    hipLaunchKernelGGL((dconv3x3_kernel<16, 64>), dim3(W / 8, B, groups), dim3(256), 0, stream, gg, B);
  else if (C == 32 && W == 32)
    hipLaunchKernelGGL((dconv3x3_kernel<32, 32>), dim3(W / 8, B, groups), dim3(256), 0, stream, gg, B);
  else
    return AVLEN_ERR_ARG;
  return avlen_launch_status();
}

bool avlen_dconv3x3_supported(int W, int C, int KH, int KW, int stride, int pad) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && ((C == 16 && W == 64) || (C == 32 && W == 32));
}
