// The decoder's cross attention of the scene-memory transformer -- ONE query per sample over the S = M + 1 encoded memory rows
// (ss_baselines/savi/models/smt_state_encoder.py:152-166: nn.Transformer with a single target token) -- in "memory space", for the
// training path at scale (2nd-stage update: 2,400 samples x 301 rows of d = 256 per minibatch).
//
// With one query per (sample, head) the K / V projections of the memory can be absorbed into the query side:
//   score[b,h,key] = scale * q_h . (W_k[h] mem[key] + b_k[h])  =  scale * (q_h W_k[h]) . mem[key] + const(b, h)      (A = q_h W_k[h]: 1 x d)
//   out[b,h,:]     = sum_key p[key] (W_v[h] mem[key] + b_v[h]) =  W_v[h] (sum_key p[key] mem[key]) + b_v[h]            (m = sum p mem: 1 x d)
// (the constant drops out of the softmax).  So neither K | V = mem W_kv^T (a 722 k x 512 x 256 product, compensated: 0.94 ms), nor
// their gradient rows (1.5 GB fp32), nor that product's backward (cast + dW + dX: 1.1 ms) exist: the forward reads the memory rows
// once for the scores and once for m, the backward once for dp = dm . mem and once for dA and d mem:
//   dm = dout_h W_v[h];  dp[key] = dm . mem[key];  g = scale * p (dp - sum p dp);  dA = sum_key g[key] mem[key];
//   d mem[key] = sum_h p[h,key] dm_h + g[h,key] A_h;   dW_v[h] += dout_h^T m_h;  dq_h = dA W_k[h]^T;  dW_k[h] += q_h^T dA;  db_v += dout
//   (db_k = 0: the scores' shift invariance -- the reference's autograd leaves rounding noise there).
// All of it fp32 on the VALU (the memory rows are the 16-bit planes the encoder's last LayerNorm left: hi (+ lo, compensated mode) are
// added back to the fp32 value): the kernels are bound by reading the rows (2 x 0.37 GB per pass and direction) and writing d mem.
// d = 256, 8 heads of 32, S <= 320.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include "tower_util.h"

namespace {

constexpr int XD = 256, XH = 8, XS = 320, XTH = 512, XW = XTH / 64;

// columns 4 lane .. 4 lane + 3 of one memory row: hi plane (+ lo plane `lo` elements behind it)
__device__ __forceinline__ void x_row(const bf16* p, long lo, float v[4]) {
  const uint2 h = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(h.x << 16); v[1] = __uint_as_float(h.x & 0xffff0000u);
  v[2] = __uint_as_float(h.y << 16); v[3] = __uint_as_float(h.y & 0xffff0000u);
  if (lo) {
    const uint2 l = *reinterpret_cast<const uint2*>(p + lo);
    v[0] += __uint_as_float(l.x << 16); v[1] += __uint_as_float(l.x & 0xffff0000u);
    v[2] += __uint_as_float(l.y << 16); v[3] += __uint_as_float(l.y & 0xffff0000u);
  }
}
// sc[key][h][row of 16 lanes] = partial dot of vec[h] with memory row `key` (the wave's keys: wave, wave + 8, ...)
// four rows of the wave's key sequence (k0, k0 + 8, k0 + 16, k0 + 24; clamped: the extra loads are discarded)
__device__ __forceinline__ void x_load4(const bf16* rows, long lo, int S, int k0, int lane, float (&v)[4][4]) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int key = k0 + u * XW < S ? k0 + u * XW : (k0 < S ? k0 : S - 1);
    x_row(rows + (long)key * XD + 4 * lane, lo, v[u]);
  }
}
// The loops over a sample's rows run the NEXT four loads under the arithmetic of the current four (two register sets, the loop
// unrolled by two): a block has 8 waves and its CU two or three blocks -- without it every group of rows paid a full memory latency.
#define X_ROWS_LOOP(BODY)                                                                             \
  {                                                                                                   \
    float va_[4][4], vb_[4][4];                                                                       \
    x_load4(rows, lo, S, wave, lane, va_);                                                            \
    for (int k0 = wave; k0 < S; k0 += 8 * XW) {                                                       \
      if (k0 + 4 * XW < S) x_load4(rows, lo, S, k0 + 4 * XW, lane, vb_);                              \
      BODY(va_, k0);                                                                                  \
      if (k0 + 4 * XW < S) {                                                                          \
        if (k0 + 8 * XW < S) x_load4(rows, lo, S, k0 + 8 * XW, lane, va_);                            \
        BODY(vb_, k0 + 4 * XW);                                                                       \
      }                                                                                               \
    }                                                                                                 \
  }
// sc[key][h][row of 16 lanes] = partial dot of vec[h] with memory row `key` (the wave's keys: wave, wave + 8, ...)
__device__ __forceinline__ void x_dots(const bf16* rows, long lo, int S, const float (&vec)[XH][4], float (*sc)[XH][4], int wave, int lane) {
  auto body = [&](const float (&v)[4][4], int k0) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int key = k0 + u * XW;
      if (key < S) {                                          // (uniform per wave)
        float part[XH];
#pragma unroll
        for (int h = 0; h < XH; h++) {
          part[h] = vec[h][0] * v[u][0] + vec[h][1] * v[u][1] + vec[h][2] * v[u][2] + vec[h][3] * v[u][3];
          part[h] = row16_sum(part[h]);
        }
        if ((lane & 15) == 0) {
#pragma unroll
          for (int h = 0; h < XH; h++) sc[key][h][lane >> 4] = part[h];
        }
      }
    }
  };
  X_ROWS_LOOP(body)
}

// forward: A [B][H][d] -> P [B][H][S] (softmax over the valid keys), Mo [B][H][d] = sum_key p mem[key]
__global__ __launch_bounds__(XTH) void cross1_fwd_kernel(const float* __restrict__ A, const bf16* __restrict__ MEM16, long lo,
                                                         const float* __restrict__ maskx, float* __restrict__ P, float* __restrict__ Mo,
                                                         int S, float scale) {
  __shared__ float sc[XS][XH][4];
  __shared__ float pk[XS][XH];
  __shared__ float macc[XH][XD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x;
  const bf16* rows = MEM16 + b * S * XD;
  float vec[XH][4];
#pragma unroll
  for (int h = 0; h < XH; h++) *reinterpret_cast<float4*>(vec[h]) = *reinterpret_cast<const float4*>(A + (b * XH + h) * XD + 4 * lane);
  for (int i = tid; i < XH * XD; i += XTH) (&macc[0][0])[i] = 0.f;
  x_dots(rows, lo, S, vec, sc, wave, lane);
  __syncthreads();
  {                                                           // wave = head: masked softmax over the keys
    const int h = wave;
    float s[XS / 64];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < XS / 64; i++) {
      const int key = lane + 64 * i;
      s[i] = -INFINITY;
      if (key < S && maskx[b * S + key] != 0.f) s[i] = scale * ((sc[key][h][0] + sc[key][h][1]) + (sc[key][h][2] + sc[key][h][3]));
      mx = fmaxf(mx, s[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < XS / 64; i++) { s[i] = s[i] == -INFINITY ? 0.f : __expf(s[i] - mx); sum += s[i]; }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
    for (int i = 0; i < XS / 64; i++) {
      const int key = lane + 64 * i;
      if (key < S) { const float p = s[i] * inv; pk[key][h] = p; P[(b * XH + h) * S + key] = p; }
    }
  }
  __syncthreads();
  float acc[XH][4];
#pragma unroll
  for (int h = 0; h < XH; h++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[h][e] = 0.f;
  auto mix = [&](const float (&v)[4][4], int k0) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int key = k0 + u * XW;
      if (key < S) {
        const float4 p0 = *reinterpret_cast<const float4*>(&pk[key][0]), p1 = *reinterpret_cast<const float4*>(&pk[key][4]);
        const float ph[XH] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
        for (int h = 0; h < XH; h++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[h][e] = __builtin_fmaf(ph[h], v[u][e], acc[h][e]);
      }
    }
  };
  X_ROWS_LOOP(mix)
#pragma unroll
  for (int h = 0; h < XH; h++)
#pragma unroll
    for (int e = 0; e < 4; e++) atomicAdd(&macc[h][4 * lane + e], acc[h][e]);
  __syncthreads();
  for (int i = tid; i < XH * XD; i += XTH) Mo[b * XH * XD + i] = (&macc[0][0])[i];
}

// backward: P, DM (= dout_h W_v[h]) [B][H][d], A [B][H][d] -> dA [B][H][d], dMEM [B * S][d] (overwritten)
__global__ __launch_bounds__(XTH) void cross1_bwd_kernel(const float* __restrict__ P, const float* __restrict__ DM, const float* __restrict__ A,
                                                         const bf16* __restrict__ MEM16, long lo, float* __restrict__ dA,
                                                         float* __restrict__ dMEM, int S, float scale) {
  __shared__ float sc[XS][XH][4];
  __shared__ float pk[XS][XH], gk[XS][XH];
  __shared__ float macc[XH][XD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x;
  const bf16* rows = MEM16 + b * S * XD;
  float dm[XH][4];
#pragma unroll
  for (int h = 0; h < XH; h++) *reinterpret_cast<float4*>(dm[h]) = *reinterpret_cast<const float4*>(DM + (b * XH + h) * XD + 4 * lane);
  for (int i = tid; i < XH * XD; i += XTH) (&macc[0][0])[i] = 0.f;
  x_dots(rows, lo, S, dm, sc, wave, lane);
  __syncthreads();
  {                                                           // wave = head: g = scale * p (dp - sum p dp)
    const int h = wave;
    float p[XS / 64], dp[XS / 64];
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < XS / 64; i++) {
      const int key = lane + 64 * i;
      p[i] = 0.f; dp[i] = 0.f;
      if (key < S) {
        p[i] = P[(b * XH + h) * S + key];
        dp[i] = (sc[key][h][0] + sc[key][h][1]) + (sc[key][h][2] + sc[key][h][3]);
      }
      dl += p[i] * dp[i];
    }
    dl = wave_sum(dl);
#pragma unroll
    for (int i = 0; i < XS / 64; i++) {
      const int key = lane + 64 * i;
      if (key < S) { pk[key][h] = p[i]; gk[key][h] = scale * p[i] * (dp[i] - dl); }
    }
  }
  __syncthreads();
  float av[XH][4], acc[XH][4];
#pragma unroll
  for (int h = 0; h < XH; h++) {
    *reinterpret_cast<float4*>(av[h]) = *reinterpret_cast<const float4*>(A + (b * XH + h) * XD + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; e++) acc[h][e] = 0.f;
  }
  auto mix = [&](const float (&v)[4][4], int k0) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int key = k0 + u * XW;
      if (key < S) {
        const float4 p0 = *reinterpret_cast<const float4*>(&pk[key][0]), p1 = *reinterpret_cast<const float4*>(&pk[key][4]);
        const float4 g0 = *reinterpret_cast<const float4*>(&gk[key][0]), g1 = *reinterpret_cast<const float4*>(&gk[key][4]);
        const float ph[XH] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const float gh[XH] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < XH; h++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            acc[h][e] = __builtin_fmaf(gh[h], v[u][e], acc[h][e]);
            o[e] = __builtin_fmaf(ph[h], dm[h][e], __builtin_fmaf(gh[h], av[h][e], o[e]));
          }
        *reinterpret_cast<float4*>(dMEM + (b * S + key) * XD + 4 * lane) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  };
  X_ROWS_LOOP(mix)
#pragma unroll
  for (int h = 0; h < XH; h++)
#pragma unroll
    for (int e = 0; e < 4; e++) atomicAdd(&macc[h][4 * lane + e], acc[h][e]);
  __syncthreads();
  for (int i = tid; i < XH * XD; i += XTH) dA[b * XH * XD + i] = (&macc[0][0])[i];
}

// ---- per-head products against the rows W[h * 32 + j][0 .. d) of a [d][d] projection slice (row stride ldw) ----
// expand: out[b][h][c] = sum_j X[b][h * 32 + j] W[h * 32 + j][c]                 (X [B][ldx])
__global__ __launch_bounds__(256) void hw_expand_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                        float* __restrict__ out, int B) {
  __shared__ float xs[8][XD];
  const int c = threadIdx.x, b0 = blockIdx.x * 8;
  for (int i = threadIdx.x; i < 8 * XD; i += 256) { const int r = i >> 8, k = i & 255; xs[r][k] = b0 + r < B ? X[(long)(b0 + r) * ldx + k] : 0.f; }
  __syncthreads();
#pragma unroll 1
  for (int h = 0; h < XH; h++) {
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
      const float w = W[(long)(h * 32 + j) * ldw + c];
#pragma unroll
      for (int r = 0; r < 8; r++) acc[r] = __builtin_fmaf(xs[r][h * 32 + j], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
      if (b0 + r < B) out[((long)(b0 + r) * XH + h) * XD + c] = acc[r];
  }
}
// reduce: Y[b][h * 32 + j] = sum_c Z[b][h][c] W[h * 32 + j][c] (+ bias[h * 32 + j])        (Y [B][ldy]); one block per (8 rows, head)
__global__ __launch_bounds__(256) void hw_reduce_kernel(const float* __restrict__ Z, const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int ldy, int B) {
  __shared__ float zs[8][XD + 1], ws[32][XD + 1];
  const int h = blockIdx.y, b0 = blockIdx.x * 8, t = threadIdx.x;
  for (int i = t; i < 8 * XD; i += 256) { const int r = i >> 8, k = i & 255; zs[r][k] = b0 + r < B ? Z[((long)(b0 + r) * XH + h) * XD + k] : 0.f; }
  for (int i = t; i < 32 * XD; i += 256) { const int j = i >> 8, k = i & 255; ws[j][k] = W[(long)(h * 32 + j) * ldw + k]; }
  __syncthreads();
  const int r = t >> 5, j = t & 31;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (int k = 0; k < XD; k += 4) {
    a0 = __builtin_fmaf(zs[r][k], ws[j][k], a0); a1 = __builtin_fmaf(zs[r][k + 1], ws[j][k + 1], a1);
    a2 = __builtin_fmaf(zs[r][k + 2], ws[j][k + 2], a2); a3 = __builtin_fmaf(zs[r][k + 3], ws[j][k + 3], a3);
  }
  if (b0 + r < B) Y[(long)(b0 + r) * ldy + h * 32 + j] = (a0 + a1) + (a2 + a3) + (bias ? bias[h * 32 + j] : 0.f);
}
// dW[h * 32 + j][c] += sum_b X[b][h * 32 + j] Z[b][h][c]                           one block per (head, 8 rows j, chunk of samples)
__global__ __launch_bounds__(256) void hw_dw_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ Z, float* __restrict__ dW,
                                                    int ldw, int B, int per) {
  const int h = blockIdx.x >> 2, j0 = (blockIdx.x & 3) * 8, c = threadIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  float acc[8];
#pragma unroll
  for (int jj = 0; jj < 8; jj++) acc[jj] = 0.f;
#pragma unroll 4
  for (int b = b0; b < b1; b++) {
    const float z = Z[((long)b * XH + h) * XD + c];
    const float4 x0 = *reinterpret_cast<const float4*>(X + (long)b * ldx + h * 32 + j0), x1 = *reinterpret_cast<const float4*>(X + (long)b * ldx + h * 32 + j0 + 4);
    acc[0] = __builtin_fmaf(x0.x, z, acc[0]); acc[1] = __builtin_fmaf(x0.y, z, acc[1]); acc[2] = __builtin_fmaf(x0.z, z, acc[2]);
    acc[3] = __builtin_fmaf(x0.w, z, acc[3]); acc[4] = __builtin_fmaf(x1.x, z, acc[4]); acc[5] = __builtin_fmaf(x1.y, z, acc[5]);
    acc[6] = __builtin_fmaf(x1.z, z, acc[6]); acc[7] = __builtin_fmaf(x1.w, z, acc[7]);
  }
#pragma unroll
  for (int jj = 0; jj < 8; jj++) atomicAdd(&dW[(long)(h * 32 + j0 + jj) * ldw + c], acc[jj]);
}

bool x_ok(int d, int H, int S) { return d == XD && H == XH && S >= 1 && S <= XS; }

}  // namespace

bool avlen_i_cross1_ok(int d, int H, int S) { return x_ok(d, H, S); }

// A [B][H][d] = per-head q W_k[h]  (q [B][ldq], Wk = rows of the [d][d] K slice, row stride ldw)
int avlen_i_cross1_expand(const float* X, int ldx, const float* W, int ldw, float* out, int B, hipStream_t st) {
  if (!X || !W || !out || B <= 0 || (ldx & 3)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(hw_expand_kernel, dim3((B + 7) / 8), dim3(256), 0, st, X, ldx, W, ldw, out, B);
  return avlen_launch_status();
}
int avlen_i_cross1_reduce(const float* Z, const float* W, int ldw, const float* bias, float* Y, int ldy, int B, hipStream_t st) {
  if (!Z || !W || !Y || B <= 0) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(hw_reduce_kernel, dim3((B + 7) / 8, XH), dim3(256), 0, st, Z, W, ldw, bias, Y, ldy, B);
  return avlen_launch_status();
}
int avlen_i_cross1_dw(const float* X, int ldx, const float* Z, float* dW, int ldw, int B, hipStream_t st) {
  if (!X || !Z || !dW || B <= 0 || (ldx & 3) || ((uintptr_t)X & 15)) return AVLEN_ERR_ARG;
  const int chunks = B >= 512 ? 16 : 1, per = (B + chunks - 1) / chunks;
  hipLaunchKernelGGL(hw_dw_kernel, dim3(XH * 4, chunks), dim3(256), 0, st, X, ldx, Z, dW, ldw, B, per);
  return avlen_launch_status();
}
// MEM16: [B * S][256] bf16 (lo plane `lo` elements behind, 0 = none); maskx [B][S] (1 = valid)
int avlen_i_cross1_fwd(const float* A, const void* MEM16, long lo, const float* maskx, float* P, float* Mo, int B, int S, float scale,
                       hipStream_t st) {
  if (!A || !MEM16 || !maskx || !P || !Mo || B <= 0 || S <= 0 || S > XS || ((uintptr_t)MEM16 & 7) || (lo & 3)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(cross1_fwd_kernel, dim3(B), dim3(XTH), 0, st, A, (const bf16*)MEM16, lo, maskx, P, Mo, S, scale);
  return avlen_launch_status();
}
int avlen_i_cross1_bwd(const float* P, const float* DM, const float* A, const void* MEM16, long lo, float* dA, float* dMEM, int B, int S,
                       float scale, hipStream_t st) {
  if (!P || !DM || !A || !MEM16 || !dA || !dMEM || B <= 0 || S <= 0 || S > XS || ((uintptr_t)dMEM & 15)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(cross1_bwd_kernel, dim3(B), dim3(XTH), 0, st, P, DM, A, (const bf16*)MEM16, lo, dA, dMEM, S, scale);
  return avlen_launch_status();
}

// ---- C ABI (tests/test_gpu_primitives.py drives the five steps against torch autograd of the K | V formulation) ----
extern "C" int avlen_cross1_expand(const float* X, int ldx, const float* W, int ldw, float* out, int B, hipStream_t st) {
  return avlen_i_cross1_expand(X, ldx, W, ldw, out, B, st);
}
extern "C" int avlen_cross1_reduce(const float* Z, const float* W, int ldw, const float* bias, float* Y, int ldy, int B, hipStream_t st) {
  return avlen_i_cross1_reduce(Z, W, ldw, bias, Y, ldy, B, st);
}
extern "C" int avlen_cross1_dw(const float* X, int ldx, const float* Z, float* dW, int ldw, int B, hipStream_t st) {
  return avlen_i_cross1_dw(X, ldx, Z, dW, ldw, B, st);
}
extern "C" int avlen_cross1_fwd(const float* A, const void* MEM16, long lo, const float* maskx, float* P, float* Mo, int B, int S, float scale,
                                hipStream_t st) {
  return avlen_i_cross1_fwd(A, MEM16, lo, maskx, P, Mo, B, S, scale, st);
}
extern "C" int avlen_cross1_bwd(const float* P, const float* DM, const float* A, const void* MEM16, long lo, float* dA, float* dMEM, int B,
                                int S, float scale, hipStream_t st) {
  return avlen_i_cross1_bwd(P, DM, A, MEM16, lo, dA, dMEM, B, S, scale, st);
}
