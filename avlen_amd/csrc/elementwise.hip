// Small HBM-bound kernels: sensor preprocessing, feature-column assembly, weight packing, row copies.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {

// (B,S,S,C) fp32 or uint8 (RGB stays uint8 from the simulator to here) -> (x/div, kxk mean) -> (B,64,64,8) bf16,
// channels >= C zero.  One thread per output pixel.  uint8 pixels are converted to float first: bit-identical to fp32 storage.
// row_index (optional): image b of the batch is image row_index[b] of x (the PPO minibatch reads the rollout storage in place)
template <typename T>
__global__ void preprocess_bf16_kernel(const T* __restrict__ x, __bf16* __restrict__ y, int B, int S, int C, int k, float div,
                                       const int* __restrict__ row_index) {
  long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long)B * 4096) return;
  int ox = (int)(pix % 64), oy = (int)((pix / 64) % 64), b = (int)(pix / 4096);
  const long bs = row_index ? row_index[b] : b;
  const T* src = x + ((bs * S + oy * k) * S + ox * k) * C;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
  bf16x8v o;
#pragma unroll
  for (int c = 0; c < 8; c++) {
    float s = 0.f;
    if (c < C) {
      for (int dy = 0; dy < k; dy++)
        for (int dx = 0; dx < k; dx++) s += (float)src[((long)dy * S + dx) * C + c] / div;
      s /= (float)(k * k);
    }
    o[c] = (__bf16)s;
  }
  *reinterpret_cast<bf16x8v*>(y + pix * 8) = o;
}

// (B,S,S,C) -> (B,64,64,C), y = scale * mean over k x k blocks (k = S/64).  One thread per output
// pixel-channel; a wave reads k contiguous runs of k*C floats per output row => coalesced.
__global__ void preprocess_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int S, int C, int k,
                                  float scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)B * 64 * 64 * C;
  if (idx >= tot) return;
  int c = (int)(idx % C);
  long p = idx / C;
  int ox = (int)(p % 64), oy = (int)((p / 64) % 64), b = (int)(p / 4096);
  const float* src = x + (((long)b * S + oy * k) * S + ox * k) * C + c;
  float s = 0.f;
  for (int dy = 0; dy < k; dy++)
    for (int dx = 0; dx < k; dx++) s += src[((long)dy * S + dx) * C];
  // torch: (x/255) then area-mean.  Keep that order of roundings: scale each tap, then average.
  y[idx] = s * scale / (float)(k * k);
}

template <typename T>
__global__ void preprocess_exact_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int S, int C, int k,
                                        float div) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)B * 64 * 64 * C;
  if (idx >= tot) return;
  int c = (int)(idx % C);
  long p = idx / C;
  int ox = (int)(p % 64), oy = (int)((p / 64) % 64), b = (int)(p / 4096);
  const T* src = x + (((long)b * S + oy * k) * S + ox * k) * C + c;
  float s = 0.f;
  for (int dy = 0; dy < k; dy++)
    for (int dx = 0; dx < k; dx++) s += (float)src[((long)dy * S + dx) * C] / div;   // same rounding as x/255 then mean
  y[idx] = s / (float)(k * k);
}

template <typename T>
__global__ void rgbd_concat_kernel(const T* __restrict__ rgb, const float* __restrict__ depth,
                                   float* __restrict__ y, long npix) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float4 o;
  o.x = (float)rgb[i * 3] / 255.0f; o.y = (float)rgb[i * 3 + 1] / 255.0f; o.z = (float)rgb[i * 3 + 2] / 255.0f; o.w = depth[i];
  reinterpret_cast<float4*>(y)[i] = o;
}

__global__ void assemble_kernel(float* __restrict__ feats, int ldf, const float* __restrict__ aw,
                                const float* __restrict__ ab, int n_act_out, int n_act_in,
                                const int64_t* __restrict__ prev_actions, int col_action,
                                const float* __restrict__ category, int col_cat, const float* __restrict__ pose,
                                int col_pose, const float* __restrict__ extra, int n_extra, int col_extra,
                                const float* __restrict__ cbel, const float* __restrict__ lbel, float* __restrict__ goal,
                                int d_goal, int B, const float* __restrict__ vis, int ld_vis, int n_vis,
                                const float* __restrict__ aud, int ld_aud, int n_aud, int col_aud) {
  int b = blockIdx.x, t = threadIdx.x;
  if (b >= B) return;
  float* f = feats + (long)b * ldf;
  // encoder columns computed elsewhere (an EncoderGroup's shared buffers): copied here instead of by two copy_rows launches
  if (vis) for (int i = t; i < n_vis; i += blockDim.x) f[i] = vis[(long)b * ld_vis + i];
  if (aud) for (int i = t; i < n_aud; i += blockDim.x) f[col_aud + i] = aud[(long)b * ld_aud + i];
  if (aw && t < n_act_out) {
    long a = prev_actions[b];
    float v = ab[t];
    if (a >= 0 && a < n_act_in) v += aw[(long)t * n_act_in + a];
    f[col_action + t] = v;
  }
  if (category && t < 21) f[col_cat + t] = category[(long)b * 21 + t];
  if (pose && t < 4) f[col_pose + t] = pose[(long)b * 4 + t];
  if (extra) for (int i = t; i < n_extra; i += blockDim.x) f[col_extra + i] = extra[(long)b * n_extra + i];
  if (goal) {
    for (int i = t; i < d_goal; i += blockDim.x) {
      float v = 0.f;
      if (i < 21) v = cbel[(long)b * 21 + i];
      else if (i < 23) v = lbel[(long)b * 2 + (i - 21)];
      goal[(long)b * d_goal + i] = v;
    }
  }
}

__global__ void concat_rows_kernel(const float* __restrict__ a, int lda, int na, const float* __restrict__ b, int ldb,
                                   int nb, float* __restrict__ out, int ldo, int B) {
  int r = blockIdx.x;
  for (int i = threadIdx.x; i < na + nb; i += blockDim.x)
    out[(long)r * ldo + i] = i < na ? a[(long)r * lda + i] : b[(long)r * ldb + (i - na)];
}

__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ o, int O, int I, int KH, int KW) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)O * I * KH * KW;
  if (idx >= tot) return;
  int ci = (int)(idx % I);
  long r = idx / I;
  int kx = (int)(r % KW); r /= KW;
  int ky = (int)(r % KH);
  int oc = (int)(r / KH);
  o[idx] = w[(((long)oc * I + ci) * KH + ky) * KW + kx];
}

__global__ void pack_fc_kernel(const float* __restrict__ w, float* __restrict__ o, int O, int C, int HW) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)O * C * HW;
  if (idx >= tot) return;
  int c = (int)(idx % C);
  long r = idx / C;
  int p = (int)(r % HW);
  int oc = (int)(r / HW);
  o[idx] = w[((long)oc * C + c) * HW + p];
}

__global__ void copy_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows,
                                 int cols) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  int r = (int)(idx / cols), c = (int)(idx % cols);
  dst[(long)r * ldd + c] = src[(long)r * lds + c];
}

}  // namespace

static inline dim3 grid1d(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

extern "C" int avlen_preprocess_image(const void* x, int x_u8, float* y, int B, int S, int C, float divisor, hipStream_t stream) {
  if (S % 64 || B <= 0) return AVLEN_ERR_ARG;
  long tot = (long)B * 4096 * C;
  if (x_u8)
    hipLaunchKernelGGL(preprocess_exact_kernel<uint8_t>, grid1d(tot), dim3(256), 0, stream, (const uint8_t*)x, y, B, S, C, S / 64, divisor);
  else if (divisor == 1.0f)
    hipLaunchKernelGGL(preprocess_kernel, grid1d(tot), dim3(256), 0, stream, (const float*)x, y, B, S, C, S / 64, 1.0f);
  else
    hipLaunchKernelGGL(preprocess_exact_kernel<float>, grid1d(tot), dim3(256), 0, stream, (const float*)x, y, B, S, C, S / 64, divisor);
  return avlen_launch_status();
}

int avlen_preprocess_image_bf16(const void* x, int x_u8, void* y16, int B, int S, int C, float divisor, hipStream_t stream,
                                const int* row_index) {
  if (S % 64 || B <= 0 || C > 8) return AVLEN_ERR_ARG;
  long tot = (long)B * 4096;
  if (x_u8)
    hipLaunchKernelGGL(preprocess_bf16_kernel<uint8_t>, grid1d(tot), dim3(256), 0, stream, (const uint8_t*)x, (__bf16*)y16, B, S, C,
                       S / 64, divisor, row_index);
  else
    hipLaunchKernelGGL(preprocess_bf16_kernel<float>, grid1d(tot), dim3(256), 0, stream, (const float*)x, (__bf16*)y16, B, S, C, S / 64,
                       divisor, row_index);
  return avlen_launch_status();
}

extern "C" int avlen_rgbd_concat(const void* rgb, int rgb_u8, const float* depth, float* y, int B, int HW, hipStream_t stream) {
  long npix = (long)B * HW;
  if (rgb_u8) hipLaunchKernelGGL(rgbd_concat_kernel<uint8_t>, grid1d(npix), dim3(256), 0, stream, (const uint8_t*)rgb, depth, y, npix);
  else hipLaunchKernelGGL(rgbd_concat_kernel<float>, grid1d(npix), dim3(256), 0, stream, (const float*)rgb, depth, y, npix);
  return avlen_launch_status();
}

extern "C" int avlen_feature_assemble(float* feats, int ldf, const avlen_linear* act, const int64_t* prev_actions,
                                      int col_action, const float* category, int col_cat, const float* pose,
                                      int col_pose, const float* extra, int n_extra, int col_extra,
                                      const float* category_belief, const float* location_belief, float* goal,
                                      int d_goal, int B, const float* vis, int ld_vis, int n_vis, const float* aud, int ld_aud,
                                      int n_aud, int col_aud, hipStream_t stream) {
  if (B <= 0) return AVLEN_ERR_ARG;
  if (act && (act->out_f > 64 || !prev_actions)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(assemble_kernel, dim3(B), dim3(64), 0, stream, feats, ldf, act ? act->w : nullptr,
                     act ? act->b : nullptr, act ? act->out_f : 0, act ? act->in_f : 0, prev_actions, col_action,
                     category, col_cat, pose, col_pose, extra, n_extra, col_extra, category_belief, location_belief,
                     goal, d_goal, B, vis, ld_vis, n_vis, aud, ld_aud, n_aud, col_aud);
  return avlen_launch_status();
}

extern "C" int avlen_concat_rows(const float* a, int lda, int na, const float* b, int ldb, int nb, float* out, int ldo,
                                 int B, hipStream_t stream) {
  hipLaunchKernelGGL(concat_rows_kernel, dim3(B), dim3(128), 0, stream, a, lda, na, b, ldb, nb, out, ldo, B);
  return avlen_launch_status();
}

extern "C" int avlen_pack_conv_weight(const float* w, float* o, int O, int I, int KH, int KW, hipStream_t stream) {
  hipLaunchKernelGGL(pack_conv_kernel, grid1d((long)O * I * KH * KW), dim3(256), 0, stream, w, o, O, I, KH, KW);
  return avlen_launch_status();
}

extern "C" int avlen_pack_fc_after_flatten(const float* w, float* o, int O, int C, int HW, hipStream_t stream) {
  hipLaunchKernelGGL(pack_fc_kernel, grid1d((long)O * C * HW), dim3(256), 0, stream, w, o, O, C, HW);
  return avlen_launch_status();
}

// ---- batched device-to-device copies: one launch for all of a rollout step's storage writes ----
struct MultiCopy { const char* src[32]; char* dst[32]; long nbytes[32]; };
__global__ void multi_copy_kernel(MultiCopy mc) {
  const int e = blockIdx.y;
  const char* __restrict__ s = mc.src[e]; char* __restrict__ d = mc.dst[e];
  const long n = mc.nbytes[e];
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((size_t)s | (size_t)d | (size_t)n) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s); float4* d4 = reinterpret_cast<float4*>(d);
    for (long i = t0; i < (n >> 4); i += stride) d4[i] = s4[i];
  } else if ((((size_t)s | (size_t)d | (size_t)n) & 3) == 0) {
    const int* s1 = reinterpret_cast<const int*>(s); int* d1 = reinterpret_cast<int*>(d);
    for (long i = t0; i < (n >> 2); i += stride) d1[i] = s1[i];
  } else {
    for (long i = t0; i < n; i += stride) d[i] = s[i];
  }
}

extern "C" int avlen_multi_copy(const void* const* src, void* const* dst, const int64_t* nbytes, int n, hipStream_t stream) {
  if (n < 0 || (n > 0 && (!src || !dst || !nbytes))) return AVLEN_ERR_ARG;
  for (int base = 0; base < n; base += 32) {
    MultiCopy mc = {};
    const int cnt = n - base < 32 ? n - base : 32;
    long mx = 0;
    for (int i = 0; i < cnt; i++) {
      if (nbytes[base + i] < 0 || (nbytes[base + i] > 0 && (!src[base + i] || !dst[base + i]))) return AVLEN_ERR_ARG;
      mc.src[i] = (const char*)src[base + i]; mc.dst[i] = (char*)dst[base + i]; mc.nbytes[i] = nbytes[base + i];
      if (mc.nbytes[i] > mx) mx = mc.nbytes[i];
    }
    if (mx == 0) continue;
    long bx = (mx / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)bx, cnt), dim3(256), 0, stream, mc);
  }
  return avlen_launch_status();
}

// Zero fill as an ordinary kernel node.  hipMemsetAsync inside a captured graph becomes a memset node; with several graphs
// replaying concurrently on different streams those nodes were observed to clear their range at the wrong point of the
// chain (CLIP row statistics wiped after the first accumulations), so captured paths never use them.
__global__ void zero_kernel(float4* __restrict__ p, long n16, char* __restrict__ tail, int ntail) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
__global__ void zero_bytes_head_kernel(char* __restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}
int avlen_zero_bytes(void* p, size_t bytes, hipStream_t stream) {
  if (!p) return AVLEN_ERR_ARG;
  if (bytes == 0) return AVLEN_OK;
  char* q = (char*)p;
  const size_t head = (16 - ((size_t)q & 15)) & 15;             // bytes up to the first 16-byte boundary
  if (head) {
    const int n = (int)(head < bytes ? head : bytes);
    hipLaunchKernelGGL(zero_bytes_head_kernel, dim3(1), dim3(64), 0, stream, q, n);
    q += n; bytes -= n;
    if (bytes == 0) return avlen_launch_status();
  }
  const long n16 = (long)(bytes >> 4);
  long blocks = (n16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (float4*)q, n16, q + (n16 << 4), (int)(bytes & 15));
  return avlen_launch_status();
}

// L2 warm-up of weights a latency-bound kernel is about to stream (the fused row-batch chains, csrc/chain.hip: 8 .. 32 workgroups,
// 64 KiB in flight each -- 62 us with the weights in the L2s, 97-100 us from HBM, tools/chain_lab.hip; in the rollout step they
// run cold: the visual towers and the text tower have moved 3 GB through the caches since their last use).  Block b serves the XCD
// it lands on (blocks are dealt round-robin over the 8 XCDs: b % 8 labels the XCD, b / 8 the slice), so EVERY XCD's L2 ends up
// with every byte; the loads are plain (the lines stay), their values go nowhere.  Speed only: a wrong placement guess warms less.
namespace {
struct PrefetchArgs { const char* p[8]; long n16[8]; int n; };
__global__ __launch_bounds__(256) void prefetch_l2_kernel(PrefetchArgs a, unsigned* __restrict__ sink) {
  const int slices = gridDim.x >> 3, slice = blockIdx.x >> 3;
  unsigned acc = 0;
  for (int r = 0; r < a.n; r++) {
    const uint4* src = reinterpret_cast<const uint4*>(a.p[r]);
    const long per = (a.n16[r] + slices - 1) / slices, lo = per * slice, hi = lo + per < a.n16[r] ? lo + per : a.n16[r];
    long i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {              // four independent 16-byte loads in flight per lane
      const uint4 v0 = src[i], v1 = src[i + 256], v2 = src[i + 512], v3 = src[i + 768];
      acc ^= v0.x ^ v1.x ^ v2.x ^ v3.x;
    }
    for (; i < hi; i += 256) acc ^= src[i].x;
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;            // keeps the loads alive; practically never taken, harmless if it is
}
__device__ unsigned g_prefetch_sink;
}  // namespace
extern "C" int avlen_prefetch_l2(const void* const* ptrs, const int64_t* nbytes, int n, hipStream_t stream) {
  if (n < 0 || n > 8 || (n > 0 && (!ptrs || !nbytes))) return AVLEN_ERR_ARG;
  PrefetchArgs a = {};
  long total = 0;
  for (int i = 0; i < n; i++) {
    if (!ptrs[i] || nbytes[i] < 0 || ((uintptr_t)ptrs[i] & 15)) return AVLEN_ERR_ARG;
    a.p[a.n] = (const char*)ptrs[i]; a.n16[a.n] = nbytes[i] >> 4; total += a.n16[a.n]; a.n++;
  }
  if (total == 0) return AVLEN_OK;
  static unsigned* sinks[64];                              // per device, looked up once (a symbol lookup per launch is microseconds)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return AVLEN_ERR_LAUNCH;
  if (!sinks[dev] && hipGetSymbolAddress((void**)&sinks[dev], HIP_SYMBOL(g_prefetch_sink)) != hipSuccess) sinks[dev] = nullptr;
  unsigned* sink = sinks[dev];
  // 32 slices per XCD: a slice of a 5 MB set is ~10 wave-iterations of 4 KiB x 4
  hipLaunchKernelGGL(prefetch_l2_kernel, dim3(8 * 32), dim3(256), 0, stream, a, sink);
  return avlen_launch_status();
}

extern "C" int avlen_copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t stream) {
  hipLaunchKernelGGL(copy_rows_kernel, grid1d((long)rows * cols), dim3(256), 0, stream, src, lds, dst, ldd, rows, cols);
  return avlen_launch_status();
}

// dst[r][0 .. cols) = src[index[r]][0 .. cols)  (fp32 rows; cols % 4 == 0, 16-byte aligned rows)
namespace {
__global__ void gather_rows_kernel(const float* __restrict__ src, int lds, const int* __restrict__ index, float* __restrict__ dst,
                                   int ldd, long rows, int c4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4) return;
  const long r = i / c4; const int c = (int)(i - r * c4) * 4;
  *reinterpret_cast<float4*>(dst + r * ldd + c) = *reinterpret_cast<const float4*>(src + (long)index[r] * lds + c);
}
}  // namespace
extern "C" int avlen_gather_rows(const float* src, int lds, const int* index, float* dst, int ldd, int rows, int cols, hipStream_t stream) {
  if (!src || !index || !dst || rows <= 0 || cols <= 0 || (cols & 3) || (lds & 3) || (ldd & 3)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, grid1d((long)rows * (cols / 4)), dim3(256), 0, stream, src, lds, index, dst, ldd, (long)rows, cols / 4);
  return avlen_launch_status();
}

extern "C" const char* avlen_build_info(void) { return "avlen_hip gfx950 (CDNA4) " __DATE__ " " __TIME__; }
