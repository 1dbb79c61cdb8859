// Fused "row-batch chain" kernel: a whole sequence of d=256 Linear / ReLU / residual / LayerNorm steps on a small batch
// (the single-target-token decoder, the collapsed `pretraining` encoder, the per-step tails of pi_q / pi_g / pi_l) in ONE
// launch instead of ~20 launches of 5-12 us each.
//
// One block = 16 batch rows, 16/J waves (J = 16-feature tiles per wave; J = 1 -> 1024 threads, used when the batch is a
// few blocks only and the chain is pure latency; J = 4 -> 256 threads).  The activation lives in registers between steps
// (fp32, layout of the transposed MFMA result: lane (c = lane&15, q = lane>>4) of wave w holds batch row c, features
// 16(J w + j) + 4q + r, j in 0..J-1, r in 0..3) and
// as a bf16 copy in LDS (the B operand of the next step: X[row][k]).  Weights stream from L2 straight into registers as
// the A operand (row = output feature), 16 B per lane per fragment; there is no LDS staging of weights and no inter-block
// communication.  LayerNorm reduces inside the lane, across q by 2 shuffles and across the waves through 2 KB of LDS.
// The residual operand comes from a register save slot.  The step list is a small program in the kernel arguments.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int D = 256;          // feature width of every step's output
constexpr int KMAX = 512;       // widest input (dialog fusion: [state | text] = 512)
constexpr int XLD = KMAX + 8;   // LDS row stride (elements) of the bf16 activation image

template <int J>
__global__ __launch_bounds__(1024 / J) void chain_kernel(avlen_chain prog, int B) {
  constexpr int NW = 16 / J, NT = NW * 64;
  __shared__ __attribute__((aligned(16))) bf16 xs[3][16 * XLD];
  __shared__ float red[NW][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
  const int row = blockIdx.x * 16 + c;                 // this lane's batch row
  const bool rok = row < B;
  const int n0 = wave * 16 * J;
  float cur[J][4], sav[J][4];
#pragma unroll
  for (int j = 0; j < J; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) { cur[j][r] = 0.f; sav[j][r] = 0.f; }

  auto publish = [&](int buf) {                        // cur -> bf16 image xs[buf][row][feature]
#pragma unroll
    for (int j = 0; j < J; j++) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] = (bf16)cur[j][r];
      *reinterpret_cast<bf16x4*>(&xs[buf][c * XLD + n0 + j * 16 + q * 4]) = o;
    }
  };

  for (int s = 0; s < prog.n; s++) {
    const avlen_chain_op op = prog.op[s];
    switch (op.kind) {
      case AVLEN_CH_LOAD_X16: {                        // bf16 global rows [B][ld] -> xs[buf][.][0:K)
        __syncthreads();
        const bf16* src = (const bf16*)op.p0;
        for (int i = tid; i < 16 * (op.k / 8); i += NT) {
          int rr = i / (op.k / 8), ch = i % (op.k / 8);
          int gr = blockIdx.x * 16 + rr;
          bf16x8 v;
          if (gr < B) v = *reinterpret_cast<const bf16x8*>(src + (long)gr * op.ld + ch * 8);
          else for (int e = 0; e < 8; e++) v[e] = (bf16)0.f;
          *reinterpret_cast<bf16x8*>(&xs[op.buf][rr * XLD + ch * 8]) = v;
        }
        __syncthreads();
        break;
      }
      case AVLEN_CH_LOAD_CUR: {                        // fp32 global [B][ld] (256 features) -> cur (+ bf16 image)
        const float* src = (const float*)op.p0;
#pragma unroll
        for (int j = 0; j < J; j++) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rok) v = *reinterpret_cast<const float4*>(src + (long)row * op.ld + n0 + j * 16 + q * 4);
          cur[j][0] = v.x; cur[j][1] = v.y; cur[j][2] = v.z; cur[j][3] = v.w;
        }
        __syncthreads();
        publish(op.buf);
        __syncthreads();
        break;
      }
      case AVLEN_CH_LINEAR: {                          // cur = act(W x + b) [+ sav[slot]]; x = xs[buf][.][0:K)
        const bf16* W = (const bf16*)op.p0;            // [256][ld] bf16, row = output feature
        const float* bias = (const float*)op.p1;
        f32x4 acc[J];
#pragma unroll
        for (int j = 0; j < J; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nks = op.k / 32;
        const bf16* wrow[J];
#pragma unroll
        for (int j = 0; j < J; j++) wrow[j] = W + (long)(n0 + j * 16 + c) * op.ld + q * 8;
        const bf16* xrow = &xs[op.buf][c * XLD + q * 8];
        for (int k0 = 0; k0 < nks; k0 += 8) {          // 8 k-steps of weight fragments in flight, then the MFMAs
          bf16x8 wf[8][J];
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (k0 + u < nks) {
#pragma unroll
              for (int j = 0; j < J; j++) wf[u][j] = *reinterpret_cast<const bf16x8*>(wrow[j] + (k0 + u) * 32);
            }
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (k0 + u < nks) {
              bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + (k0 + u) * 32);
#pragma unroll
              for (int j = 0; j < J; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][j], xf, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < J; j++) {
          float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n0 + j * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float v = acc[j][r] + b4[r];
            if (op.act == AVLEN_ACT_RELU) v = fmaxf(v, 0.f);
            if (op.res) v += sav[j][r];
            cur[j][r] = v;
          }
        }
        __syncthreads();                               // every wave has finished reading xs[buf]
        publish(op.out_buf);
        __syncthreads();
        break;
      }
      case AVLEN_CH_LAYERNORM: {                       // cur = LN(cur) * g + b over the 256 features of each row
        const float* g = (const float*)op.p0; const float* bb = (const float*)op.p1;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < J; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) { s1 += cur[j][r]; s2 += cur[j][r] * cur[j][r]; }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (q == 0) { red[wave][c][0] = s1; red[wave][c][1] = s2; }
        __syncthreads();
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) { t1 += red[w][c][0]; t2 += red[w][c][1]; }
        const float mean = t1 * (1.f / D);
        const float var = fmaxf(t2 * (1.f / D) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
        for (int j = 0; j < J; j++) {
          float4 gv = *reinterpret_cast<const float4*>(g + n0 + j * 16 + q * 4);
          float4 bv = *reinterpret_cast<const float4*>(bb + n0 + j * 16 + q * 4);
          float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int r = 0; r < 4; r++) cur[j][r] = (cur[j][r] - mean) * rstd * g4[r] + b4[r];
        }
        publish(op.out_buf);                           // xs was last read before the barrier above
        __syncthreads();
        break;
      }
      case AVLEN_CH_SAVE: {
#pragma unroll
        for (int j = 0; j < J; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) sav[j][r] = cur[j][r];
        break;
      }
      case AVLEN_CH_STORE: {                           // cur -> fp32 global [B][ld] and/or bf16 global [B][ld2]
        float* dst = (float*)op.p0; bf16* dst16 = (bf16*)op.p1;
        if (rok) {
#pragma unroll
          for (int j = 0; j < J; j++) {
            if (dst) *reinterpret_cast<float4*>(dst + (long)row * op.ld + n0 + j * 16 + q * 4) =
                make_float4(cur[j][0], cur[j][1], cur[j][2], cur[j][3]);
            if (dst16) {
              bf16x4 o;
#pragma unroll
              for (int r = 0; r < 4; r++) o[r] = (bf16)cur[j][r];
              *reinterpret_cast<bf16x4*>(dst16 + (long)row * op.ld2 + n0 + j * 16 + q * 4) = o;
            }
          }
        }
        break;
      }
      default: break;
    }
  }
}

}  // namespace

int avlen_chain_run(const avlen_chain* prog, int B, hipStream_t stream) {
  if (!prog || prog->n < 1 || prog->n > AVLEN_CHAIN_MAX_OPS || B <= 0) return AVLEN_ERR_ARG;
  for (int i = 0; i < prog->n; i++) {
    const avlen_chain_op& o = prog->op[i];
    if ((o.kind == AVLEN_CH_LINEAR || o.kind == AVLEN_CH_LOAD_X16) && (o.k % 32 || o.k > KMAX || o.ld % 8)) return AVLEN_ERR_ARG;
    if (o.buf < 0 || o.buf > 2 || o.out_buf < 0 || o.out_buf > 2) return AVLEN_ERR_ARG;
  }
  const int blocks = ceil_div(B, 16);
  if (blocks <= 64) hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(1024), 0, stream, *prog, B);
  else hipLaunchKernelGGL(chain_kernel<4>, dim3(blocks), dim3(256), 0, stream, *prog, B);
  return avlen_launch_status();
}
