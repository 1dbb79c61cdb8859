// GroupNorm (NHWC, fused ReLU / residual) and LayerNorm forward/backward for gfx950.
// Both are HBM/L2-bound streaming kernels: 16-byte loads, wave(64)-level reductions, no re-reads
// from HBM (a sample's activations -- <= 256 KB -- stay in the XCD's L2 between the two passes).
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;

// GroupNorm apply for the bf16 fast path: x = raw fp32 conv output, statistics (per sample/channel sum, sumsq) come
// from the conv epilogue; y (bf16) = [relu](xhat*g + b [+ res (bf16)]).  8 channels (16 B out) per thread-iteration.
// RN: the residual is itself a RAW conv output whose own GroupNorm (statistics rstats, affine rgamma / rbeta, optional ReLU)
// is applied on the fly and rounded to bf16 exactly as a separate apply pass would have stored it -- the stem's bn1 and the
// downsample branch's norm never make their own trip through HBM.
struct GnGroups { const void* x[8]; const float* stats[8]; const float* gamma[8]; const float* beta[8];
                  const __bf16* res[8]; __bf16* y[8];
                  const float* rstats[8]; const float* rgamma[8]; const float* rbeta[8]; };

template <bool RAW16, bool RN>      // RAW16: the raw conv output is stored in bf16 (statistics were taken from the fp32 accumulators)
__global__ __launch_bounds__(256) void gn_apply_bf16_kernel(GnGroups gg, int HW, int C, int G, int splits, int relu,
                                                            float eps, int rrelu) {
  const float* __restrict__ x = (const float*)gg.x[blockIdx.y]; const float* __restrict__ stats = gg.stats[blockIdx.y];
  const __bf16* __restrict__ x16 = (const __bf16*)gg.x[blockIdx.y];
  const float* __restrict__ gamma = gg.gamma[blockIdx.y]; const float* __restrict__ beta = gg.beta[blockIdx.y];
  const __bf16* __restrict__ res = gg.res[blockIdx.y]; __bf16* __restrict__ y = gg.y[blockIdx.y];
  __shared__ float s_scale[128], s_shift[128], s_rscale[RN ? 128 : 1], s_rshift[RN ? 128 : 1];
  const int b = blockIdx.x / splits, sp = blockIdx.x % splits, tid = threadIdx.x;
  const int cg = C / G;
  if (RN && tid >= 64 && tid < 64 + G) {
    const int t = tid - 64;
    const float* st = gg.rstats[blockIdx.y] + (long)b * 2 * C;
    double sum = 0.0, sq = 0.0;
    for (int c = t * cg; c < (t + 1) * cg; c++) { sum += st[c]; sq += st[C + c]; }
    double n = (double)HW * cg, mean = sum / n, var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = t * cg; c < (t + 1) * cg; c++) {
      float sc = gg.rgamma[blockIdx.y][c] * rstd;
      s_rscale[c] = sc; s_rshift[c] = gg.rbeta[blockIdx.y][c] - (float)mean * sc;
    }
  }
  if (tid < G) {
    const float* st = stats + (long)b * 2 * C;
    double sum = 0.0, sq = 0.0;
    for (int c = tid * cg; c < (tid + 1) * cg; c++) { sum += st[c]; sq += st[C + c]; }
    double n = (double)HW * cg, mean = sum / n, var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = tid * cg; c < (tid + 1) * cg; c++) {
      float sc = gamma[c] * rstd;
      s_scale[c] = sc; s_shift[c] = beta[c] - (float)mean * sc;
    }
  }
  __syncthreads();
  const long n8 = (long)HW * C / 8;
  const long per = (n8 + splits - 1) / splits;
  const long beg = sp * per, end = min(n8, beg + per);
  const long base = (long)b * HW * C;
  const int c0 = (int)(((beg + tid) * 8) % C);          // 256*8 % C == 0 -> fixed channels per thread
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { sc[i] = s_scale[c0 + i]; sh[i] = s_shift[c0 + i]; }
  float rsc[8], rsh[8];
  if constexpr (RN) {
#pragma unroll
    for (int i = 0; i < 8; i++) { rsc[i] = s_rscale[c0 + i]; rsh[i] = s_rshift[c0 + i]; }
  }
  for (long f = beg + tid; f < end; f += 256) {
    float v[8];
    if constexpr (RAW16) {
      bf16x8v xr = *reinterpret_cast<const bf16x8v*>(x16 + base + f * 8);
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = (float)xr[i];
    } else {
      const float4* xp = reinterpret_cast<const float4*>(x + base + f * 8);
      float4 a = xp[0], c = xp[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    }
    if (res) {
      bf16x8v r = *reinterpret_cast<const bf16x8v*>(res + base + f * 8);
      if constexpr (RN) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          float t = (float)r[i] * rsc[i] + rsh[i];
          r[i] = (__bf16)(rrelu ? fmaxf(t, 0.f) : t);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = v[i] * sc[i] + sh[i] + (float)r[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = v[i] * sc[i] + sh[i];
    }
    bf16x8v o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = (__bf16)(relu ? fmaxf(v[i], 0.f) : v[i]);
    *reinterpret_cast<bf16x8v*>(y + base + f * 8) = o;
  }
}

// Deterministic block reduction of per-thread 4-channel partials.  Thread u owns channels c0(u)..c0(u)+3 with
// c0(u) = (c0_t0 + 4u) % C (c0_t0 a multiple of 4); channel c is summed over its 1024/C owners in a fixed order.
__device__ __forceinline__ void reduce_channels(float4 s, float4 q, int c0_t0, int C, float* s_part /*[2][256][4]*/,
                                                float* s_sum, float* s_sq) {
  const int tid = threadIdx.x;
  float* ps = s_part + tid * 4; float* pq = s_part + 1024 + tid * 4;
  ps[0] = s.x; ps[1] = s.y; ps[2] = s.z; ps[3] = s.w;
  pq[0] = q.x; pq[1] = q.y; pq[2] = q.z; pq[3] = q.w;
  __syncthreads();
  if (tid < C) {
    const int step = C / 4, off = tid & 3;
    const int u0 = (((tid & ~3) - c0_t0 + C) % C) / 4;
    float a = 0.f, b = 0.f;
    for (int u = u0; u < 256; u += step) { a += s_part[u * 4 + off]; b += s_part[1024 + u * 4 + off]; }
    s_sum[tid] = a; s_sq[tid] = b;
  }
  __syncthreads();
}

// One block per (sample, spatial split).  C <= 128, C % 4 == 0, blockDim*4 % C == 0, so every
// thread always touches the same 4 channels and keeps 4 running sums in registers.
// Two-kernel form so that small batches still fill the chip:
//   gn_stats:  partial sum / sumsq per (sample, split, channel)  -> stats[B][S][2][C]
//   gn_apply:  combine partials -> per-group mean / rstd; y = relu(xhat*g + b + res)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                       int HW, int C, int splits) {
  __shared__ float s_sum[128], s_sq[128];
  __shared__ float s_part[2048];
  const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int tid = threadIdx.x;
  const long n4 = (long)HW * C / 4;                         // float4 per sample
  const long per = (n4 + splits - 1) / splits;
  const long beg = sp * per, end = min(n4, beg + per);
  const float4* xp = reinterpret_cast<const float4*>(x + (long)b * HW * C);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  // per*4 % C == 0 is guaranteed by the launcher (per is a multiple of C/4 * 256 / gcd) -> fixed channels
  for (long f = beg + tid; f < end; f += 256) {
    float4 v = xp[f];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  reduce_channels(s, q, (int)((beg * 4) % C), C, s_part, s_sum, s_sq);
  if (tid < C) {
    float* o = part + ((long)(b * splits + sp) * 2) * C;
    o[tid] = s_sum[tid];
    o[C + tid] = s_sq[tid];
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, float* __restrict__ y, int HW,
                                                       int C, int G, int splits, int relu, float eps) {
  __shared__ float s_scale[128], s_shift[128];
  const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int tid = threadIdx.x;
  const int cg = C / G;
  if (tid < G) {
    double sum = 0.0, sq = 0.0;
    for (int s = 0; s < splits; s++) {
      const float* o = part + ((long)(b * splits + s) * 2) * C;
      for (int c = tid * cg; c < (tid + 1) * cg; c++) { sum += o[c]; sq += o[C + c]; }
    }
    double n = (double)HW * cg;
    double mean = sum / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = tid * cg; c < (tid + 1) * cg; c++) {
      float sc = gamma[c] * rstd;
      s_scale[c] = sc;
      s_shift[c] = beta[c] - (float)mean * sc;
    }
  }
  __syncthreads();
  const long n4 = (long)HW * C / 4;
  const long per = (n4 + splits - 1) / splits;
  const long beg = sp * per, end = min(n4, beg + per);
  const long base = (long)b * HW * C;
  const float4* xp = reinterpret_cast<const float4*>(x + base);
  const float4* rp = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
  float4* yp = reinterpret_cast<float4*>(y + base);
  const int c0 = (int)(((beg + tid) * 4) % C);
  const float a0 = s_scale[c0], a1 = s_scale[c0 + 1], a2 = s_scale[c0 + 2], a3 = s_scale[c0 + 3];
  const float h0 = s_shift[c0], h1 = s_shift[c0 + 1], h2 = s_shift[c0 + 2], h3 = s_shift[c0 + 3];
  for (long f = beg + tid; f < end; f += 256) {
    float4 v = xp[f];
    v.x = v.x * a0 + h0; v.y = v.y * a1 + h1; v.z = v.z * a2 + h2; v.w = v.w * a3 + h3;
    if (rp) { float4 r = rp[f]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    yp[f] = v;
  }
}

// LayerNorm: one wave per row, d = 64 * VPL * ... handled as d/64 values per lane (d <= 1024).
template <int NV>   // NV = d / 64 values per lane, strided so loads are coalesced
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, __bf16* __restrict__ y16,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows,
                                                     float eps, const int* __restrict__ rows_dev, long y16_lo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows || (rows_dev && row >= *rows_dev)) return;
  constexpr int d = NV * 64;
  const float* xr = x + (long)row * d;
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    v[i] = xr[lane + i * 64];
    if (res) v[i] += res[(long)row * d + lane + i * 64];
    s += v[i];
  }
  float mean = wave_sum(s) * (1.f / d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) { float t = v[i] - mean; q += t * t; }
  float rstd = rsqrtf(wave_sum(q) * (1.f / d) + eps);
#pragma unroll
  for (int i = 0; i < NV; i++) {
    int c = lane + i * 64;
    float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
    if (y) y[(long)row * d + c] = o;
    if (y16) {
      const __bf16 hv = (__bf16)o;
      y16[(long)row * d + c] = hv;
      if (y16_lo) y16[y16_lo + (long)row * d + c] = (__bf16)(o - (float)hv);      // low plane of the compensated pair
    }
  }
  if (mean_o && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// Backward.  Each block handles ROWS_PER_BLOCK rows (one wave per row at a time) and accumulates
// dgamma/dbeta for its rows in registers, then one atomicAdd per column per block.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xsum,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                     int rows_per_block, __bf16* __restrict__ dx16, float* __restrict__ colsum) {
  // dx16 / colsum (optional, together): dx also as row-major bf16 rows [rows][d] -- the operand of the NEXT Linear's backward -- and
  // its column sums added to colsum[d] (that Linear's bias gradient): the cast pass over dx is skipped
  constexpr int d = NV * 64;
  __shared__ float sg[4][d], sb[4][d];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float ag[NV], ab[NV], ac[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) { ag[i] = 0.f; ab[i] = 0.f; ac[i] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  for (int row = r0 + w; row < r1; row += 4) {
    float m = mean[row], rs = rstd[row];
    float g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int c = lane + i * 64;
      float dyv = dy[(long)row * d + c];
      xh[i] = (xsum[(long)row * d + c] - m) * rs;
      g[i] = dyv * gamma[c];
      s1 += g[i]; s2 += g[i] * xh[i];
      ag[i] += dyv * xh[i]; ab[i] += dyv;
    }
    s1 = wave_sum(s1) * (1.f / d); s2 = wave_sum(s2) * (1.f / d);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const float v = rs * (g[i] - s1 - xh[i] * s2);
      dx[(long)row * d + lane + i * 64] = v;
      if (dx16) { dx16[(long)row * d + lane + i * 64] = (__bf16)v; ac[i] += v; }
    }
  }
  if (colsum) {
#pragma unroll
    for (int i = 0; i < NV; i++) sg[w][lane + i * 64] = ac[i];
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) atomicAdd(&colsum[c], sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    __syncthreads();
  }
  if (dgamma) {
#pragma unroll
    for (int i = 0; i < NV; i++) { sg[w][lane + i * 64] = ag[i]; sb[w][lane + i * 64] = ab[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
      atomicAdd(&dgamma[c], sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
      atomicAdd(&dbeta[c], sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
    }
  }
}

}  // namespace

extern "C" size_t avlen_groupnorm_workspace_bytes(int B, int C) {
  return (size_t)B * 16 * 2 * C * sizeof(float) + 256;           // up to 16 spatial splits
}

static int gn_pick_splits(int B, int HW, int C) {
  // fill >= ~512 blocks; each split must cover a multiple of 256 float4 (so thread->channel is fixed)
  long n4 = (long)HW * C / 4;
  int s = 1;
  while (s < 16 && (long)B * s < 512 && (n4 / (s * 2)) >= 256 && (n4 % ((long)s * 2 * 256)) == 0) s *= 2;
  return s;
}

// ws may be NULL only when the internal static scratch is not needed -> we always need partials, so the
// public entry point below carries its own tiny scratch inside `y`'s tail?  No: use a dedicated arg-free
// scheme: partial stats live in a caller workspace for module calls; the primitive entry uses splits=1 and
// a small static device buffer is avoided by folding stats+apply into one launch.
__global__ __launch_bounds__(256) void gn_fused_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ y, int HW, int C, int G, int relu, float eps) {
  __shared__ float s_sum[128], s_sq[128], s_scale[128], s_shift[128];
  __shared__ float s_part[2048];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long n4 = (long)HW * C / 4;
  const long base = (long)b * HW * C;
  const float4* xp = reinterpret_cast<const float4*>(x + base);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  for (long f = tid; f < n4; f += 256) {
    float4 v = xp[f];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  const int c0 = (tid * 4) % C;
  reduce_channels(s, q, 0, C, s_part, s_sum, s_sq);
  const int cg = C / G;
  if (tid < G) {
    double sum = 0.0, sq = 0.0;
    for (int c = tid * cg; c < (tid + 1) * cg; c++) { sum += s_sum[c]; sq += s_sq[c]; }
    double n = (double)HW * cg, mean = sum / n, var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = tid * cg; c < (tid + 1) * cg; c++) {
      float sc = gamma[c] * rstd;
      s_scale[c] = sc; s_shift[c] = beta[c] - (float)mean * sc;
    }
  }
  __syncthreads();
  const float4* rp = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
  float4* yp = reinterpret_cast<float4*>(y + base);
  const float a0 = s_scale[c0], a1 = s_scale[c0 + 1], a2 = s_scale[c0 + 2], a3 = s_scale[c0 + 3];
  const float h0 = s_shift[c0], h1 = s_shift[c0 + 1], h2 = s_shift[c0 + 2], h3 = s_shift[c0 + 3];
  for (long f = tid; f < n4; f += 256) {
    float4 v = xp[f];
    v.x = v.x * a0 + h0; v.y = v.y * a1 + h1; v.z = v.z * a2 + h2; v.w = v.w * a3 + h3;
    if (rp) { float4 r = rp[f]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    yp[f] = v;
  }
}

// Internal (module-level) form with spatial splits and a caller-provided partials buffer.
int avlen_groupnorm_nhwc_ws(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                            int B, int HW, int C, int G, int relu, float eps, float* part, hipStream_t stream) {
  if (C > 128 || C % 4 || 1024 % C || C % G || ((long)HW * C) % 4) return AVLEN_ERR_ARG;
  int splits = part ? gn_pick_splits(B, HW, C) : 1;
  if (splits == 1) {
    hipLaunchKernelGGL(gn_fused_kernel, dim3(B), dim3(256), 0, stream, x, gamma, beta, residual, y, HW, C, G, relu, eps);
    return avlen_launch_status();
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(B * splits), dim3(256), 0, stream, x, part, HW, C, splits);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(B * splits), dim3(256), 0, stream, x, part, gamma, beta, residual, y, HW, C,
                     G, splits, relu, eps);
  return avlen_launch_status();
}

extern "C" int avlen_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* residual,
                                    float* y, int B, int HW, int C, int G, int relu, float eps, hipStream_t stream) {
  return avlen_groupnorm_nhwc_ws(x, gamma, beta, residual, y, B, HW, C, G, relu, eps, nullptr, stream);
}

int avlen_layernorm_fwd16(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                          void* y16, float* mean, float* rstd, int rows, int d, float eps, hipStream_t stream) {
  return avlen_layernorm_fwd16_dyn(x, residual, gamma, beta, y, y16, mean, rstd, rows, nullptr, d, eps, stream);
}

int avlen_layernorm_fwd16_dyn(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                              void* y16, float* mean, float* rstd, int rows, const int* rows_dev, int d, float eps,
                              hipStream_t stream, long y16_lo) {
  if (rows <= 0) return AVLEN_ERR_ARG;
  dim3 grid(ceil_div(rows, 4)), block(256);
  __bf16* h = (__bf16*)y16;
  switch (d) {
    case 256: hipLaunchKernelGGL((ln_fwd_kernel<4>), grid, block, 0, stream, x, residual, gamma, beta, y, h, mean, rstd, rows, eps, rows_dev, y16_lo); break;
    case 512: hipLaunchKernelGGL((ln_fwd_kernel<8>), grid, block, 0, stream, x, residual, gamma, beta, y, h, mean, rstd, rows, eps, rows_dev, y16_lo); break;
    case 128: hipLaunchKernelGGL((ln_fwd_kernel<2>), grid, block, 0, stream, x, residual, gamma, beta, y, h, mean, rstd, rows, eps, rows_dev, y16_lo); break;
    case 64: hipLaunchKernelGGL((ln_fwd_kernel<1>), grid, block, 0, stream, x, residual, gamma, beta, y, h, mean, rstd, rows, eps, rows_dev, y16_lo); break;
    default: return AVLEN_ERR_ARG;
  }
  return avlen_launch_status();
}

extern "C" int avlen_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                                   float* y, float* mean, float* rstd, int rows, int d, float eps,
                                   hipStream_t stream) {
  return avlen_layernorm_fwd16(x, residual, gamma, beta, y, nullptr, mean, rstd, rows, d, eps, stream);
}

int avlen_groupnorm_apply_bf16_grouped(const void* const* x, int raw16, const float* const* stats,
                                       const float* const* gamma, const float* const* beta, const void* const* res16,
                                       void* const* y16, int groups, int B, int HW, int C, int G, int relu, float eps,
                                       hipStream_t stream, const float* const* rstats, const float* const* rgamma,
                                       const float* const* rbeta, int rrelu) {
  if (rstats && (!res16 || !raw16 || G > 64)) return AVLEN_ERR_ARG;
  if (C > 128 || C % 8 || 2048 % C || C % G || groups < 1 || groups > 8) return AVLEN_ERR_ARG;
  long n8 = (long)HW * C / 8;
  int splits = 1;
  while (splits < 16 && (long)B * splits * groups < 512 && n8 / (splits * 2) >= 256) splits *= 2;
  GnGroups gg = {};
  for (int g = 0; g < groups; g++) {
    gg.x[g] = x[g]; gg.stats[g] = stats[g]; gg.gamma[g] = gamma[g]; gg.beta[g] = beta[g];
    gg.res[g] = res16 ? (const __bf16*)res16[g] : nullptr; gg.y[g] = (__bf16*)y16[g];
    if (rstats) { gg.rstats[g] = rstats[g]; gg.rgamma[g] = rgamma[g]; gg.rbeta[g] = rbeta[g]; }
  }
  if (rstats)
    hipLaunchKernelGGL((gn_apply_bf16_kernel<true, true>), dim3(B * splits, groups), dim3(256), 0, stream, gg, HW, C, G, splits, relu, eps, rrelu);
  else if (raw16)
    hipLaunchKernelGGL((gn_apply_bf16_kernel<true, false>), dim3(B * splits, groups), dim3(256), 0, stream, gg, HW, C, G, splits, relu, eps, 0);
  else
    hipLaunchKernelGGL((gn_apply_bf16_kernel<false, false>), dim3(B * splits, groups), dim3(256), 0, stream, gg, HW, C, G, splits, relu, eps, 0);
  return avlen_launch_status();
}

int avlen_groupnorm_apply_bf16(const float* x, const float* stats, const float* gamma, const float* beta, const void* res16,
                               void* y16, int B, int HW, int C, int G, int relu, float eps, hipStream_t stream) {
  const void* xv = x;
  return avlen_groupnorm_apply_bf16_grouped(&xv, 0, &stats, &gamma, &beta, res16 ? &res16 : nullptr, &y16, 1, B, HW, C, G, relu,
                                            eps, stream, nullptr, nullptr, nullptr, 0);
}

extern "C" int avlen_layernorm_bwd(const float* dy, const float* xsum, const float* gamma, const float* mean,
                                   const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int d,
                                   hipStream_t stream) {
  return avlen_layernorm_bwd16(dy, xsum, gamma, mean, rstd, dx, dgamma, dbeta, rows, d, stream, nullptr, nullptr);
}
int avlen_layernorm_bwd16(const float* dy, const float* xsum, const float* gamma, const float* mean, const float* rstd, float* dx,
                          float* dgamma, float* dbeta, int rows, int d, hipStream_t stream, void* dx16_, float* colsum) {
  __bf16* dx16 = (__bf16*)dx16_;
  if (rows <= 0 || (dx16 != nullptr) != (colsum != nullptr)) return AVLEN_ERR_ARG;
  int rpb = rows >= 65536 ? 256 : rows >= 4096 ? 64 : 16;
  dim3 grid(ceil_div(rows, rpb)), block(256);
  switch (d) {
    case 256: hipLaunchKernelGGL((ln_bwd_kernel<4>), grid, block, 0, stream, dy, xsum, gamma, mean, rstd, dx, dgamma, dbeta, rows, rpb, dx16, colsum); break;
    case 512: hipLaunchKernelGGL((ln_bwd_kernel<8>), grid, block, 0, stream, dy, xsum, gamma, mean, rstd, dx, dgamma, dbeta, rows, rpb, dx16, colsum); break;
    default: return AVLEN_ERR_ARG;
  }
  return avlen_launch_status();
}
