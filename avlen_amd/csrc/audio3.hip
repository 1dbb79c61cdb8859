// The three convolutions of the AudioCNN (ss_baselines/av_nav/models/audio_cnn.py: Conv 8x8 s4 (2 -> 32) + ReLU, Conv 4x4 s2 (32 -> 64)
// + ReLU, Conv 3x3 s1 (64 -> 64), flattened NHWC for the Linear) as ONE launch, one workgroup (8 waves) per (encoder, spectrogram):
// the 16-bit activations of a spectrogram never leave the CU's LDS.
//
// Why: as a cast + three implicit-GEMM launches the branch costs 66 us of the rollout step's critical path (7 + 24.7 + 18.7 + 15.2 us
// at 3 x 64 spectrograms of 257 x 101 x 2) for 53 MFLOP per spectrogram: every launch re-reads its input through L2 with the
// patch overlap (4x, 4x, 9x) and pays its own ramp and tail, and nothing can overlap the visual towers' persistent launch.
//   conv 1: the fp32 spectrogram is staged in BANDS (the input rows of 16 output rows, converted to 16-bit, 16-byte aligned rows) in
//           the LDS region conv 2's output takes later; K = 8 x 8 x 2 = 128 in (ky, kx, c) order, so an MFMA k-step is two kernel
//           rows of 16 contiguous values: one 16-byte LDS read per fragment.  Weights (8 KB) live in registers.
//   conv 2: input = conv 1's output [63][24][32] in LDS, one k-step per tap; a wave holds TWO cout tiles' weights (128 registers), so
//           an activation fragment feeds two MFMAs.
//   conv 3: input [30][11][64] in LDS, two k-steps per tap, again two cout tiles per wave (144 registers); the result goes to global
//           memory as the row [oh][ow][64] the Linear's GEMM reads.
// The products are transposed (A = weights, B = pixels): a lane ends up with 4 consecutive channels of one pixel = one 8-byte store.
// Pixel rows in LDS are swizzled per 16-byte chunk (chunk ^ f(x)) so that the 16 pixels of a fragment spread over the banks although
// their stride is 64 / 128 bytes.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int A3_TH = 512, A3_BAND = 8, A3_MAXG = 8, A3_NLD = 12;     // A3_NLD: float2 loads per thread and band

struct A3Args {
  const float* x; const int* row_index; int B, H, W;
  int oh1, ow1, oh2, ow2, oh3, ow3;
  int row_bytes;                                            // staged input row (W * 2 values of 2 bytes, rounded up to 16)
  int a2_off;                                               // LDS offset of conv 2's output (conv 1's band lives there before)
  const void* w1[A3_MAXG]; const void* w2[A3_MAXG]; const void* w3[A3_MAXG];
  const float* b1[A3_MAXG]; const float* b2[A3_MAXG]; const float* b3[A3_MAXG];
  void* out[A3_MAXG];                                       // [B][oh3 * ow3 * 64] 16-bit
};

template <bool F16> __device__ __forceinline__ f32x4 a3_mma(const uint4& w, const uint4& x, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w), __builtin_bit_cast(h16x8, x), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ unsigned short a3_cvt(float v) {
  if constexpr (F16) { const _Float16 h = (_Float16)v; return __builtin_bit_cast(unsigned short, h); }
  else { const bf16 h = (bf16)v; return __builtin_bit_cast(unsigned short, h); }
}
template <bool F16> __device__ __forceinline__ uint2 a3_pack4(float a, float b, float c, float d) {
  return make_uint2((unsigned)a3_cvt<F16>(a) | ((unsigned)a3_cvt<F16>(b) << 16), (unsigned)a3_cvt<F16>(c) | ((unsigned)a3_cvt<F16>(d) << 16));
}

template <bool F16>
__global__ __launch_bounds__(A3_TH) void audio3_kernel(A3Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, g = blockIdx.y;
  const long bs = a.row_index ? a.row_index[b] : b;
  char* a1 = lds; char* a2 = lds + a.a2_off;
  const int ow1 = a.ow1, ow2 = a.ow2, ow3 = a.ow3;
  // conv 2's weights are requested first: they arrive under conv 1 (the registers are free until then)
  const int cp = wave & 1, grp = wave >> 1;                 // convs 2 / 3: cout tiles 2 cp, 2 cp + 1; pixel tiles grp, grp + 4, ...
  uint4 wf2[2][16];                                         // (half of them: all 128 registers beside conv 1's own spill)
#pragma unroll
  for (int tap = 0; tap < 16; tap++)
    wf2[0][tap] = *reinterpret_cast<const uint4*>((const char*)a.w2[g] + ((long)((2 * cp) * 16 + r16) * 512 + 32 * tap + 8 * q) * 2);
  // ================= conv 1: bands of A3_BAND output rows, double-buffered =================
  {
    uint4 wf[2][4];
    float bias[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
        wf[nt][kk] = *reinterpret_cast<const uint4*>((const char*)a.w1[g] + ((long)(nt * 16 + r16) * 128 + 32 * kk + 8 * q) * 2);
#pragma unroll
      for (int r = 0; r < 4; r++) bias[nt][r] = a.b1[g][nt * 16 + 4 * q + r];
    }
    const float* img = a.x + bs * a.H * a.W * 2;
    const int band_bytes = (4 * A3_BAND + 4) * a.row_bytes;
    float2 st[A3_NLD];
    auto fetch = [&](int oy0) {                             // the band's input rows -> registers
      const int nb = min(A3_BAND, a.oh1 - oy0), n = (4 * nb + 4) * a.W;
#pragma unroll
      for (int j = 0; j < A3_NLD; j++) {
        const int i = tid + A3_TH * j;
        if (i < n) st[j] = *reinterpret_cast<const float2*>(img + ((long)4 * oy0 * a.W + i) * 2);
      }
    };
    auto stash = [&](int oy0, char* band) {                 // registers -> the 16-bit band image
      const int nb = min(A3_BAND, a.oh1 - oy0), n = (4 * nb + 4) * a.W;
#pragma unroll
      for (int j = 0; j < A3_NLD; j++) {
        const int i = tid + A3_TH * j;
        if (i < n) {
          const int r = i / a.W, xx = i - r * a.W;
          *reinterpret_cast<unsigned*>(band + r * a.row_bytes + xx * 4) = (unsigned)a3_cvt<F16>(st[j].x) | ((unsigned)a3_cvt<F16>(st[j].y) << 16);
        }
      }
    };
    fetch(0);
    stash(0, a2);
    __syncthreads();
    int buf = 0;
    for (int oy0 = 0; oy0 < a.oh1; oy0 += A3_BAND, buf ^= 1) {
      const int nb = min(A3_BAND, a.oh1 - oy0);
      const bool more = oy0 + A3_BAND < a.oh1;
      if (more) fetch(oy0 + A3_BAND);                       // in flight under this band's MFMAs
      const char* band = a2 + buf * band_bytes;
      const int npix = nb * ow1;
      for (int t = wave; t * 16 < npix; t += A3_TH / 64) {
        const int p = t * 16 + r16, pc = p < npix ? p : npix - 1;
        const int oyl = pc / ow1, ox = pc - oyl * ow1;
        const char* src = band + (4 * oyl + (q >> 1)) * a.row_bytes + 16 * ox + 16 * (q & 1);
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const uint4 xf = *reinterpret_cast<const uint4*>(src + 2 * kk * a.row_bytes);
          acc[0] = a3_mma<F16>(wf[0][kk], xf, acc[0]);
          acc[1] = a3_mma<F16>(wf[1][kk], xf, acc[1]);
        }
        if (p < npix) {
          char* dst = a1 + (long)((oy0 + oyl) * ow1 + ox) * 64 + (q & 1) * 8;
          const int sw = (ox >> 1) & 3;
#pragma unroll
          for (int nt = 0; nt < 2; nt++) {
            const uint2 o = a3_pack4<F16>(fmaxf(acc[nt][0] + bias[nt][0], 0.f), fmaxf(acc[nt][1] + bias[nt][1], 0.f),
                                          fmaxf(acc[nt][2] + bias[nt][2], 0.f), fmaxf(acc[nt][3] + bias[nt][3], 0.f));
            *reinterpret_cast<uint2*>(dst + (((2 * nt + (q >> 1)) ^ sw) << 4)) = o;
          }
        }
      }
      if (more) stash(oy0 + A3_BAND, a2 + (buf ^ 1) * band_bytes);     // (that buffer's band was finished before the last barrier)
      __syncthreads();
    }
  }
  // ================= conv 2: [oh1][ow1][32] -> [oh2][ow2][64], 16 taps =================
#pragma unroll
  for (int tap = 0; tap < 16; tap++)
    wf2[1][tap] = *reinterpret_cast<const uint4*>((const char*)a.w2[g] + ((long)((2 * cp + 1) * 16 + r16) * 512 + 32 * tap + 8 * q) * 2);
  {
    uint4 (&wf)[2][16] = wf2;
    float bias[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) bias[nt][r] = a.b2[g][(2 * cp + nt) * 16 + 4 * q + r];
    const int npix = a.oh2 * ow2;
    for (int t = grp; t * 16 < npix; t += 4) {
      const int p = t * 16 + r16, pc = p < npix ? p : npix - 1;
      const int oy = pc / ow2, ox = pc - oy * ow2;
      f32x4 acc[2];
      acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
      for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
          const int x = 2 * ox + kx;
          const uint4 xf = *reinterpret_cast<const uint4*>(a1 + (long)((2 * oy + ky) * ow1 + x) * 64 + ((q ^ ((x >> 1) & 3)) << 4));
          acc[0] = a3_mma<F16>(wf[0][ky * 4 + kx], xf, acc[0]);
          acc[1] = a3_mma<F16>(wf[1][ky * 4 + kx], xf, acc[1]);
        }
      if (p < npix) {
        char* dst = a2 + (long)p * 128 + (q & 1) * 8;
        const int sw = ox & 7;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
          const uint2 o = a3_pack4<F16>(fmaxf(acc[nt][0] + bias[nt][0], 0.f), fmaxf(acc[nt][1] + bias[nt][1], 0.f),
                                        fmaxf(acc[nt][2] + bias[nt][2], 0.f), fmaxf(acc[nt][3] + bias[nt][3], 0.f));
          *reinterpret_cast<uint2*>(dst + (((2 * (2 * cp + nt) + (q >> 1)) ^ sw) << 4)) = o;
        }
      }
    }
  }
  // ================= conv 3: [oh2][ow2][64] -> [oh3][ow3][64] (no ReLU), 9 taps x 2 k-steps =================
  // one cout tile per wave here (72 weight registers: with two, 144, the fragment reads of a tile could not be hoisted over its MFMAs)
  {
    const int ct = wave & 3, grp3 = wave >> 2;              // pixel tiles grp3, grp3 + 2, ...
    uint4 wf[18];
    float bias[4];
#pragma unroll
    for (int ks = 0; ks < 18; ks++)
      wf[ks] = *reinterpret_cast<const uint4*>((const char*)a.w3[g] + ((long)(ct * 16 + r16) * 576 + 32 * ks + 8 * q) * 2);
#pragma unroll
    for (int r = 0; r < 4; r++) bias[r] = a.b3[g][ct * 16 + 4 * q + r];
    __syncthreads();                                        // conv 2's output is complete
    const int npix = a.oh3 * ow3;
    for (int t = grp3; t * 16 < npix; t += 2) {
      const int p = t * 16 + r16, pc = p < npix ? p : npix - 1;
      const int oy = pc / ow3, ox = pc - oy * ow3;
      f32x4 acc[2];                                         // two accumulation chains (even / odd k-steps), added at the end
      acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const int x = ox + kx;
          const char* px = a2 + (long)((oy + ky) * ow2 + x) * 128;
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            const uint4 xf = *reinterpret_cast<const uint4*>(px + (((4 * hh + q) ^ (x & 7)) << 4));
            acc[hh] = a3_mma<F16>(wf[(ky * 3 + kx) * 2 + hh], xf, acc[hh]);
          }
        }
      if (p < npix) {
        const uint2 o = a3_pack4<F16>((acc[0][0] + acc[1][0]) + bias[0], (acc[0][1] + acc[1][1]) + bias[1], (acc[0][2] + acc[1][2]) + bias[2],
                                      (acc[0][3] + acc[1][3]) + bias[3]);
        *reinterpret_cast<uint2*>((char*)a.out[g] + (((long)b * npix + p) * 64 + ct * 16 + 4 * q) * 2) = o;
      }
    }
  }
}

}  // namespace

// The geometry the fused kernel covers (the AudioCNN of audio_cnn.py on a 2-channel spectrogram whose activations fit one CU's LDS).
bool avlen_i_audio3_ok(const avlen_cnn3* n, int H, int W) {
  const avlen_conv &c1 = n->conv[0], &c2 = n->conv[1], &c3 = n->conv[2];
  if (!(c1.cin == 2 && c1.cout == 32 && c1.kh == 8 && c1.kw == 8 && c1.stride == 4 && c1.pad == 0 && c1.w16c)) return false;
  if (!(c2.cin == 32 && c2.cin16 == 32 && c2.cout == 64 && c2.kh == 4 && c2.kw == 4 && c2.stride == 2 && c2.pad == 0 && c2.w16)) return false;
  if (!(c3.cin == 64 && c3.cin16 == 64 && c3.cout == 64 && c3.kh == 3 && c3.kw == 3 && c3.stride == 1 && c3.pad == 0 && c3.w16)) return false;
  const int oh1 = (H - 8) / 4 + 1, ow1 = (W - 8) / 4 + 1, oh2 = (oh1 - 4) / 2 + 1, ow2 = (ow1 - 4) / 2 + 1, oh3 = oh2 - 2, ow3 = ow2 - 2;
  if (H < 8 || W < 8 || oh1 < 4 || ow1 < 4 || oh3 < 1 || ow3 < 1) return false;
  const int row_bytes = (W * 4 + 15) & ~15;
  const long a1 = (long)oh1 * ow1 * 64, a2 = (long)oh2 * ow2 * 128, band = 2L * (4 * A3_BAND + 4) * row_bytes;
  if ((4 * A3_BAND + 4) * W > A3_TH * A3_NLD) return false;
  return a1 + (a2 > band ? a2 : band) <= 160 * 1024 - 512;
}

// x [B][H][W][2] fp32 (spectrogram b = row row_index[b] when given); nets[g]: `groups` AudioCNNs on the same input; outs[g]: the conv
// stack's output [B][oh3 * ow3 * 64] in the nets' 16-bit format (bias added, no ReLU: what the Linear's GEMM reads).
int avlen_i_audio3_fwd(const avlen_cnn3* const* nets, const float* x, const int* row_index, int groups, int B, int H, int W, void* const* outs,
                       hipStream_t st) {
  if (!nets || !x || !outs || groups < 1 || groups > A3_MAXG || B <= 0) return AVLEN_ERR_ARG;
  for (int g = 0; g < groups; g++)
    if (!avlen_i_audio3_ok(nets[g], H, W) || nets[g]->half_fmt != nets[0]->half_fmt) return AVLEN_ERR_ARG;
  A3Args a = {};
  a.x = x; a.row_index = row_index; a.B = B; a.H = H; a.W = W;
  a.oh1 = (H - 8) / 4 + 1; a.ow1 = (W - 8) / 4 + 1; a.oh2 = (a.oh1 - 4) / 2 + 1; a.ow2 = (a.ow1 - 4) / 2 + 1; a.oh3 = a.oh2 - 2; a.ow3 = a.ow2 - 2;
  a.row_bytes = (W * 4 + 15) & ~15;
  a.a2_off = (int)(((long)a.oh1 * a.ow1 * 64 + 255) & ~255L);
  const long a2 = (long)a.oh2 * a.ow2 * 128, band = 2L * (4 * A3_BAND + 4) * a.row_bytes;
  const int lds = a.a2_off + (int)(a2 > band ? a2 : band);
  for (int g = 0; g < groups; g++) {
    a.w1[g] = nets[g]->conv[0].w16c; a.w2[g] = nets[g]->conv[1].w16; a.w3[g] = nets[g]->conv[2].w16;
    a.b1[g] = nets[g]->conv[0].b; a.b2[g] = nets[g]->conv[1].b; a.b3[g] = nets[g]->conv[2].b;
    a.out[g] = outs[g];
    if (!a.b1[g] || !a.b2[g] || !a.b3[g]) return AVLEN_ERR_ARG;
  }
  static unsigned long long done16 = 0, doneb = 0;
  if (nets[0]->half_fmt == 1) {
    if (avlen_set_dyn_lds((const void*)audio3_kernel<true>, 160 * 1024, &done16) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(audio3_kernel<true>, dim3(B, groups), dim3(A3_TH), lds, st, a);
  } else {
    if (avlen_set_dyn_lds((const void*)audio3_kernel<false>, 160 * 1024, &doneb) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(audio3_kernel<false>, dim3(B, groups), dim3(A3_TH), lds, st, a);
  }
  return avlen_launch_status();
}
