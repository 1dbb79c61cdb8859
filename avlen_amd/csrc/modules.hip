// Module-level compositions of the avlen_hip kernels: the reference's nn.Modules on this path, each as
// ONE C entry point that enqueues its whole launch sequence on the caller's stream using caller scratch.
//   avlen_resnet18_fwd   <- SMTCNN tower        (smt_cnn.py:78-115, smt_resnet.py:132-146)
//   avlen_cnn3_fwd       <- AudioCNN/VisualCNN  (audio_cnn.py:136-151, visual_cnn.py:165-190)
//   avlen_smt_fwd/_bwd   <- SMTStateEncoder     (smt_state_encoder.py:109-276) + nn.Transformer
//   avlen_dialog_fwd     <- DialogStateEncoder  (dialog_state_encoder.py:114-155)
//   avlen_clip_text_fwd  <- CLIP.encode_text    (third party; call site policy.py:847-849)
//   avlen_gru_fwd        <- RNNStateEncoder     (av_nav/models/rnn_state_encoder.py:80-143)
#include "common.h"
#include "../../include/avlen_hip.h"
#include <math.h>
#include <algorithm>

static inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

#include "internal.h"

#define TRY(x) do { int _rc = (x); if (_rc != AVLEN_OK) return _rc; } while (0)

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr size_t GEMM_SCRATCH = 96u << 20;       // split-K / accumulate slabs shared by one module call

typedef avlen_ctx Ctx;      // internal.h (xs: operand scratch of the large-M bf16 products below)

// ---- large-M products of the TRAINING path (2nd stage: 722 k token rows per minibatch) ----
// The training forward/backward keeps fp32 activations (LayerNorm / residual / softmax math and the saved tensors of the
// backward).  The fp32-staged GEMM converts them on the fly and reaches ~95 TFLOP/s at M = 722 k, N = K = 256; casting the
// operands to bf16 once (transposed for the weight gradient, whose contraction runs over the rows) and running the
// glds / 8-wave MFMA kernel is 2x faster end to end even with the extra pass.  Same arithmetic (bf16 operands, fp32
// accumulate), different summation order.  Used when bf16 mode is on, scratch was laid out and M >= big_m().
bool attn_bwd16_on() {          // AVLEN_ATTN_BWD16=0: fp32 attention backward in bf16 mode too (A/B knob)
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_ATTN_BWD16", 1);
  return v != 0;
}
bool fused_dy_on() {             // AVLEN_FUSED_DY=0: the three separate passes over dY (A/B knob)
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_FUSED_DY", 1);
  return v != 0;
}
bool tn_dw_on() {                // AVLEN_TN_DW=0: the weight gradient through transposed operand copies (A/B knob)
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_TN_DW", 1);
  return v != 0;
}
long g_big_m = -1;
long g_mixed_rows = 65536;     // bf16x3: token rows from which avlen_smt_bwd runs its products on plain bf16 operands (0 = never)
long big_m() {
  if (g_big_m < 0) g_big_m = (long)avlen_knob("AVLEN_BIGM", 4096);
  return g_big_m;
}
inline int pad8(long x) { return (int)((x + 7) & ~7L); }

// fp32 [R][C] (row stride ld) -> bf16 [C][ldt] (transposed); columns R .. Rp-1 of every output row are zero
// lo: also write the LOW plane of the compensated pair (bf16(v - bf16(v))) `lo` elements behind the high one
__global__ void tcast_kernel(const float* __restrict__ src, int ld, bf16* __restrict__ dst, long ldt, long R, int C, long Rp, long lo) {
  __shared__ float t[32][33];
  const long r0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const long r = r0 + ty + j * 8; const int c = c0 + tx;
    t[ty + j * 8][tx] = (r < R && c < C) ? src[r * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = c0 + ty + j * 8; const long r = r0 + tx;
    if (c < C && r < Rp) {
      const float v = t[tx][ty + j * 8];
      const bf16 h = (bf16)v;
      dst[(long)c * ldt + r] = h;
      if (lo) dst[lo + (long)c * ldt + r] = (bf16)(v - (float)h);
    }
  }
}
int tcast(const Ctx& c, const float* src, int ld, bf16* dst, long ldt, long R, int C, long lo = 0) {
  const long Rp = ldt;
  hipLaunchKernelGGL(tcast_kernel, dim3((unsigned)((Rp + 31) / 32), ceil_div(C, 32)), dim3(32, 8), 0, c.st, src, ld, dst, ldt, R, C, Rp, lo);
  return avlen_launch_status();
}
// ONE pass over an upstream gradient dY [M][N] fp32 for everything a Linear's backward needs of it: the row-major 16-bit operand
// [M][Np] of dX = dY W (columns N .. Np-1 zero), the transposed operand [N][Mp] of dW = dY^T X (columns M .. Mp-1 zero) -- both with
// the low plane of the compensated pair `*_lo` elements behind the high one when NP == 2 -- and the column sums (bias gradient,
// added to `colsum`).  Replaces cast_pair + tcast + colsum_acc, i.e. three reads of dY (the 2nd-stage update's dY are 0.7 GB each).
// relu16 (optional, row stride ldr): dY is the gradient w.r.t. a ReLU's OUTPUT y and relu16 holds y as 16-bit values: elements with
// y <= 0 are taken as zero (the separate relu' pass over the fp32 gradient is skipped).
template <int NP>
__global__ __launch_bounds__(256) void dy_prep_kernel(const float* __restrict__ src, int ld, bf16* __restrict__ rows16, int Np, long rows_lo,
                                                      bf16* __restrict__ t16, long Mp, long t_lo, float* __restrict__ colsum, long M, int N, int tiles,
                                                      const bf16* __restrict__ relu16 = nullptr, int ldr = 0) {
  __shared__ float t[64][65];
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  float csum = 0.f;                                       // threads 0 .. 63: this block's column sums (ONE atomic per column per block:
  for (int it = 0; it < tiles; it++) {                 // 64-row blocks hammered the same N addresses with 3 M atomics per call)
    const long r0 = ((long)blockIdx.x * tiles + it) * 64;
    if (r0 >= Mp) break;
    if (it) __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = tid + 256 * j, row = idx >> 4, c4 = (idx & 15) * 4;
      const long r = r0 + row; const int cc = c0 + c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < M) {
        if (cc + 4 <= N) v = *reinterpret_cast<const float4*>(src + r * ld + cc);
        else { if (cc < N) v.x = src[r * ld + cc]; if (cc + 1 < N) v.y = src[r * ld + cc + 1]; if (cc + 2 < N) v.z = src[r * ld + cc + 2]; }
        if (relu16) {
          const bf16* y = relu16 + r * ldr + cc;
          if (cc < N && (float)y[0] <= 0.f) v.x = 0.f;
          if (cc + 1 < N && (float)y[1] <= 0.f) v.y = 0.f;
          if (cc + 2 < N && (float)y[2] <= 0.f) v.z = 0.f;
          if (cc + 3 < N && (float)y[3] <= 0.f) v.w = 0.f;
        }
      }
      t[row][c4] = v.x; t[row][c4 + 1] = v.y; t[row][c4 + 2] = v.z; t[row][c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++) {                          // row-major: 64 rows x 8 chunks of 8 columns
      const int idx = tid + 256 * j, row = idx >> 3, ch = (idx & 7) * 8;
      const long r = r0 + row;
      if (r < M && c0 + ch < Np) {
        bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; e++) { const float v = t[row][ch + e]; h[e] = (bf16)v; if (NP == 2) l[e] = (bf16)(v - (float)h[e]); }
        *reinterpret_cast<bf16x8*>(rows16 + r * Np + c0 + ch) = h;
        if (NP == 2) *reinterpret_cast<bf16x8*>(rows16 + rows_lo + r * Np + c0 + ch) = l;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {                          // transposed: 64 columns x 8 chunks of 8 rows (rows past M were loaded as zero)
      const int idx = tid + 256 * j, col = idx >> 3, rch = (idx & 7) * 8;
      if (t16 && c0 + col < N && r0 + rch < Mp) {
        bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; e++) { const float v = t[rch + e][col]; h[e] = (bf16)v; if (NP == 2) l[e] = (bf16)(v - (float)h[e]); }
        *reinterpret_cast<bf16x8*>(t16 + (long)(c0 + col) * Mp + r0 + rch) = h;
        if (NP == 2) *reinterpret_cast<bf16x8*>(t16 + t_lo + (long)(c0 + col) * Mp + r0 + rch) = l;
      }
    }
    if (tid < 64) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int r = 0; r < 64; r += 4) { a0 += t[r][tid]; a1 += t[r + 1][tid]; a2 += t[r + 2][tid]; a3 += t[r + 3][tid]; }
      csum += (a0 + a1) + (a2 + a3);
    }
  }
  if (colsum && tid < 64 && c0 + tid < N) atomicAdd(&colsum[c0 + tid], csum);
}
// fp32 rows -> 16-bit operand (compensated: high plane + low plane `lo` elements behind it)
int cast_pair(const Ctx& c, const float* src, int ld, bf16* dst, int ldd, long rows, int cols, long lo) {
  if (lo) return avlen_cast_pair(src, ld, dst, ldd, rows, cols, lo, c.st);
  return avlen_cast_h16(src, ld, dst, ldd, rows, cols, 0, c.st);
}
// the large-M product on the glds / MFMA kernel: plain bf16, or -- compensated mode -- three K-concatenated passes over hi / lo planes
int big_gemm(const Ctx& c, const bf16* A, int lda, long a_lo, const bf16* B, int ldb, long b_lo, float* Y, int ldy, const float* bias,
             const float* res, int ldr, int M, int N, int K, int act) {
  if (c.prec == AVLEN_PREC_BF16X3) {
    avlen_g2_opts o; o.x3 = 1; o.a_lo = a_lo * 2; o.b_lo = b_lo * 2;
    return avlen_gemm_bf16_dyn(A, lda, B, ldb, Y, ldy, nullptr, 0, bias, res, ldr, M, nullptr, N, K, act, c.gws, c.gws_bytes, c.st, &o);
  }
  return avlen_gemm_bf16(A, lda, B, ldb, Y, ldy, nullptr, 0, bias, res, ldr, M, N, K, act, c.gws, c.gws_bytes, c.st);
}
struct XsBump {        // carve 256-byte aligned pieces out of the operand scratch; ok() false -> caller falls back
  char* base; size_t off, cap; bool good = true;
  XsBump(const Ctx& c) : base((char*)c.xs), off(0), cap(c.xs_bytes) {}
  bf16* take(size_t elems) {
    size_t o = align_up(off, 256); off = o + elems * 2;
    if (off > cap) good = false;
    return (bf16*)(base + o);
  }
};
// where linear_bwd's row-major dY operand will live (its first carve): a producer may write it there directly
bf16* dy16_slot(const Ctx& c) { XsBump b(c); return b.take(8); }
bool big_path(const Ctx& c, long M) { return (c.prec == AVLEN_PREC_BF16 || c.prec == AVLEN_PREC_BF16X3) && c.xs && M >= big_m(); }

// Y[M, out_f] (ldy) = act(X[M, in_f] (ldx) * W^T + b) + res
int linear(const Ctx& c, const avlen_linear& L, const float* X, int ldx, float* Y, int ldy, int M, int act,
           const float* res, int ldr) {
  if (big_path(c, M)) {
    const int Kp = pad8(L.in_f);
    const int np = c.prec == AVLEN_PREC_BF16X3 ? 2 : 1;              // planes per operand
    XsBump b(c);
    bf16* X16 = b.take((size_t)np * M * Kp); bf16* W16 = b.take((size_t)np * L.out_f * Kp);
    if (b.good) {
      const long xl = np > 1 ? (long)M * Kp : 0, wl = np > 1 ? (long)L.out_f * Kp : 0;
      TRY(cast_pair(c, X, ldx, X16, Kp, M, L.in_f, xl));
      TRY(cast_pair(c, L.w, L.in_f, W16, Kp, L.out_f, L.in_f, wl));
      return big_gemm(c, X16, Kp, xl, W16, Kp, wl, Y, ldy, L.b, res, ldr, M, L.out_f, Kp, act);
    }
  }
  int sk = avlen_gemm_pick_splitk(M, L.out_f, L.in_f);
  if (avlen_gemm_workspace_bytes(M, L.out_f, L.in_f, sk) > c.gws_bytes) sk = 1;
  return avlen_gemm(X, ldx, 0, L.w, L.in_f, 0, Y, ldy, L.b, res, ldr, M, L.out_f, L.in_f, act, c.prec, sk, 0.f,
                    c.gws, c.gws_bytes, c.st);
}
// Y = act(X * W[r0:r0+n, :]^T + b[r0:r0+n])   (a row-slice of a packed projection, e.g. the V third of in_proj)
int linear_rows(const Ctx& c, const avlen_linear& L, int r0, int n, const float* X, int ldx, float* Y, int ldy, int M,
                int act, const float* res, int ldr) {
  avlen_linear S = L;
  S.w = L.w + (size_t)r0 * L.in_f; S.b = L.b ? L.b + r0 : nullptr; S.out_f = n;
  return linear(c, S, X, ldx, Y, ldy, M, act, res, ldr);
}
// dX[M, in_f] = dY[M, out_f] * W (+ add)     add==dX allowed (in-place accumulate)
int linear_dx(const Ctx& c, const avlen_linear& L, const float* dY, int ldy, float* dX, int ldx, int M, const float* add,
              int ldadd) {
  if (big_path(c, M)) {
    const int Np = pad8(L.out_f);
    const int np = c.prec == AVLEN_PREC_BF16X3 ? 2 : 1;
    XsBump b(c);
    bf16* dY16 = b.take((size_t)np * M * Np); bf16* WT16 = b.take((size_t)np * L.in_f * Np);
    if (b.good) {
      const long yl = np > 1 ? (long)M * Np : 0, wl = np > 1 ? (long)L.in_f * Np : 0;
      TRY(cast_pair(c, dY, ldy, dY16, Np, M, L.out_f, yl));
      TRY(tcast(c, L.w, L.in_f, WT16, Np, L.out_f, L.in_f, wl));       // W [out_f][in_f] -> W^T [in_f][Np]
      return big_gemm(c, dY16, Np, yl, WT16, Np, wl, dX, ldx, nullptr, add, ldadd, M, L.in_f, Np, 0);
    }
  }
  return avlen_gemm(dY, ldy, 0, L.w, L.in_f, 1, dX, ldx, nullptr, add, ldadd, M, L.in_f, L.out_f, 0, c.prec, 1, 0.f,
                    c.gws, c.gws_bytes, c.st);
}
// dW[out_f, in_f] += dY^T X   (reduction over the M rows, split over the chip)
int linear_dw(const Ctx& c, const avlen_linear& G, const float* dY, int ldy, const float* X, int ldx, int M) {
  if (big_path(c, M)) {
    const long Mp = pad8(M);
    const int np = c.prec == AVLEN_PREC_BF16X3 ? 2 : 1;
    XsBump b(c);
    bf16* dYT = b.take((size_t)np * G.out_f * Mp); bf16* XT = b.take((size_t)np * G.in_f * Mp);
    if (b.good) {
      const long yl = np > 1 ? (long)G.out_f * Mp : 0, xl = np > 1 ? (long)G.in_f * Mp : 0;
      TRY(tcast(c, dY, ldy, dYT, Mp, M, G.out_f, yl));
      TRY(tcast(c, X, ldx, XT, Mp, M, G.in_f, xl));
      return big_gemm(c, dYT, (int)Mp, yl, XT, (int)Mp, xl, G.w, G.in_f, nullptr, G.w, G.in_f, G.out_f, G.in_f, (int)Mp, 0);
    }
  }
  int sk = avlen_gemm_pick_splitk(G.out_f, G.in_f, M);
  while (sk > 1 && avlen_gemm_workspace_bytes(G.out_f, G.in_f, M, sk) > c.gws_bytes) sk /= 2;
  return avlen_gemm(dY, ldy, 1, X, ldx, 1, G.w, G.in_f, nullptr, nullptr, 0, G.out_f, G.in_f, M, 0, c.prec, sk, 1.f,
                    c.gws, c.gws_bytes, c.st);
}

int colsum_acc(const Ctx& c, const float* dY, int ld, float* out, int rows, int N);
// The whole backward of Y = X W^T + b for one upstream gradient: G.w += dY^T X, G.b += colsum(dY), dX = dY W (+ add) when dX is
// given.  Large M (the 16-bit glds route): dY is read ONCE (dy_prep_kernel); otherwise -- or when the operand scratch cannot hold all
// four operands at a time -- the three separate steps, in the order the call sites always used.
// X16 (optional): the forward already holds X as a row-major 16-bit operand (row stride ldx16 >= pad8(in_f), pad columns zero).
int linear_bwd(const Ctx& c, const avlen_linear& L, const avlen_linear& G, const float* dY, int ldy, const float* X, int ldx, float* dX,
               int lddx, int M, const float* add, int ldadd, const bf16* X16 = nullptr, int ldx16 = 0, const bf16* relu16 = nullptr,
               int ldr = 0, bool dy16_ready = false) {     // relu16: dY still has to pass the ReLU whose 16-bit output this is (see dy_prep_kernel)
  // dy16_ready: the PRODUCER of dY already left its row-major 16-bit operand [M][pad8(out_f)] at the head of the operand scratch
  // (dy16_slot) and added the column sums to G.b; dY (fp32) is not read
  if (big_path(c, M) && !(ldy & 3) && !((uintptr_t)dY & 15) && fused_dy_on()) {
    const int Np = pad8(L.out_f);
    const long Mp = pad8(M);
    const int np = c.prec == AVLEN_PREC_BF16X3 ? 2 : 1;
    if (np == 1 && tn_dw_on() && avlen_i_gemm_tn_workspace_bytes(M, L.out_f, L.in_f) <= c.gws_bytes) {
      // plain 16-bit operands (bf16 mode, and the compensated mode's backward at scale): NO transposed copies -- one pass over dY
      // (row-major 16-bit operand + bias gradient), one cast of X, and the row-contracted product of gemm_tn.hip on both
      const int Kp = pad8(L.in_f);
      XsBump b(c);
      const bool have_x = X16 != nullptr && ldx16 >= Kp;
      bf16* dY16 = b.take((size_t)M * Np); bf16* Xc = have_x ? nullptr : b.take((size_t)M * Kp);
      bf16* WT16 = dX ? b.take((size_t)L.in_f * Np) : nullptr;
      if (b.good) {
        const long rt = (Mp + 63) / 64;
        const int tiles = (int)(rt / 64 < 1 ? 1 : (rt / 64 > 16 ? 16 : rt / 64));
        const dim3 grid((unsigned)((rt + tiles - 1) / tiles), ceil_div(L.out_f, 64));
        if (!dy16_ready) {
          hipLaunchKernelGGL(dy_prep_kernel<1>, grid, dim3(256), 0, c.st, dY, ldy, dY16, Np, 0L, (bf16*)nullptr, Mp, 0L, G.b, (long)M, L.out_f, tiles,
                             relu16, ldr);
          TRY(avlen_launch_status());
        }
        if (!have_x) TRY(cast_pair(c, X, ldx, Xc, Kp, M, L.in_f, 0));
        TRY(avlen_i_gemm_tn_bf16(dY16, Np, have_x ? X16 : Xc, have_x ? ldx16 : Kp, M, L.out_f, L.in_f, G.w, L.in_f, 1.f, c.gws, c.gws_bytes,
                                 c.st));
        if (!dX) return AVLEN_OK;
        TRY(tcast(c, L.w, L.in_f, WT16, Np, L.out_f, L.in_f, 0));
        return big_gemm(c, dY16, Np, 0, WT16, Np, 0, dX, lddx, nullptr, add, ldadd, M, L.in_f, Np, 0);
      }
    }
    if (X16 || relu16 || dy16_ready) return AVLEN_ERR_WS;
    XsBump b(c);
    bf16* dY16 = dX ? b.take((size_t)np * M * Np) : nullptr;
    bf16* dYT = b.take((size_t)np * L.out_f * Mp); bf16* XT = b.take((size_t)np * L.in_f * Mp);
    bf16* WT16 = dX ? b.take((size_t)np * L.in_f * Np) : nullptr;
    if (b.good) {
      const long yl = np > 1 ? (long)M * Np : 0, tl = np > 1 ? (long)L.out_f * Mp : 0, xl = np > 1 ? (long)L.in_f * Mp : 0,
                 wl = np > 1 ? (long)L.in_f * Np : 0;
      const long rt = (Mp + 63) / 64;                       // 64-row tiles; up to 16 per block once the grid fills the chip anyway
      const int tiles = (int)(rt / 64 < 1 ? 1 : (rt / 64 > 16 ? 16 : rt / 64));
      const dim3 grid((unsigned)((rt + tiles - 1) / tiles), ceil_div(L.out_f, 64));
      // without dX only the transposed operand is wanted: the row-major stores are skipped by an empty column range (Np = 0)
      if (np > 1) hipLaunchKernelGGL(dy_prep_kernel<2>, grid, dim3(256), 0, c.st, dY, ldy, dY16, dX ? Np : 0, yl, dYT, Mp, tl, G.b, (long)M, L.out_f, tiles);
      else hipLaunchKernelGGL(dy_prep_kernel<1>, grid, dim3(256), 0, c.st, dY, ldy, dY16, dX ? Np : 0, yl, dYT, Mp, tl, G.b, (long)M, L.out_f, tiles);
      TRY(avlen_launch_status());
      TRY(tcast(c, X, ldx, XT, Mp, M, L.in_f, xl));
      TRY(big_gemm(c, dYT, (int)Mp, tl, XT, (int)Mp, xl, G.w, L.in_f, nullptr, G.w, L.in_f, L.out_f, L.in_f, (int)Mp, 0));
      if (!dX) return AVLEN_OK;
      TRY(tcast(c, L.w, L.in_f, WT16, Np, L.out_f, L.in_f, wl));
      return big_gemm(c, dY16, Np, yl, WT16, Np, wl, dX, lddx, nullptr, add, ldadd, M, L.in_f, Np, 0);
    }
  }
  if (X16 || relu16 || dy16_ready) return AVLEN_ERR_WS;        // the caller kept X only as a 16-bit plane: that route (above) must have been taken
  TRY(linear_dw(c, G, dY, ldy, X, ldx, M));
  TRY(colsum_acc(c, dY, ldy, G.b, M, L.out_f));
  return dX ? linear_dx(c, L, dY, ldy, dX, lddx, M, add, ldadd) : AVLEN_OK;
}

// Convolution weight gradient on the large-M bf16 route WITHOUT materialising im2col(X) in fp32: the gather is fused into the
// transposed cast (cols^T [K][Mp] bf16 is written straight from X), so the fp32 im2col buffer (1.2 GB for the GRU baseline's
// first visual conv) is neither written nor read back.  G.w [cout][K] (packed layout) += dY^T * im2col(X).
// Returns AVLEN_NOT_BIG when the route does not apply (fp32 mode / few rows): the caller takes im2col + linear_dw.
__global__ void tcast_im2col_kernel(const float* __restrict__ X, bf16* __restrict__ dst, long ldt, long M, int K, long Mp, int H, int W,
                                    int C, int OH, int OW, int KH, int KW, int s, int pad) {
  __shared__ float t[32][33];
  const long r0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int k = c0 + tx;
  int kh = 0, kw = 0, cc = 0;
  const bool kok = k < K;
  if (kok) { cc = k % C; const int tap = k / C; kw = tap % KW; kh = tap / KW; }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned m = (unsigned)r0 + ty + j * 8;          // M < 2^31 (checked by the launcher): 32-bit decode
    float v = 0.f;
    if (kok && m < (unsigned)M) {
      const unsigned q = m / (unsigned)OW; const int ow = (int)(m - q * OW);
      const unsigned b = q / (unsigned)OH; const int oh = (int)(q - b * OH);
      const int ih = oh * s + kh - pad, iw = ow * s + kw - pad;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = X[(((long)b * H + ih) * W + iw) * C + cc];
    }
    t[ty + j * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int kk = c0 + ty + j * 8; const long m = r0 + tx;
    if (kk < K && m < Mp) dst[(long)kk * ldt + m] = (bf16)t[tx][ty + j * 8];
  }
}
}  // namespace
int avlen_i_conv_dw16(const avlen_ctx& c, const avlen_linear& G, const float* dY, int ldy, const float* X, long B, int H, int W,
                      int C, int OH, int OW, int KH, int KW, int s, int pad) {
  const long M = B * OH * OW;
  const int K = KH * KW * C;
  if (!big_path(c, M) || c.prec != AVLEN_PREC_BF16 || M > 0x7ffffff0L) return AVLEN_NOT_BIG;     // plain bf16 only (no compensated form)
  const long Mp = pad8(M);
  XsBump b(c);
  bf16* dYT = b.take((size_t)G.out_f * Mp); bf16* XT = b.take((size_t)K * Mp);
  if (!b.good) return AVLEN_NOT_BIG;
  TRY(tcast(c, dY, ldy, dYT, Mp, M, G.out_f));
  hipLaunchKernelGGL(tcast_im2col_kernel, dim3((unsigned)((Mp + 31) / 32), ceil_div(K, 32)), dim3(32, 8), 0, c.st, X, XT, Mp, M, K, Mp,
                     H, W, C, OH, OW, KH, KW, s, pad);
  TRY(avlen_launch_status());
  return avlen_gemm_bf16(dYT, (int)Mp, XT, (int)Mp, G.w, K, nullptr, 0, nullptr, G.w, K, G.out_f, K, (int)Mp, 0, c.gws, c.gws_bytes,
                         c.st);
}
namespace {

// ---------------------------------------------------------------- small kernels used by the modules
__global__ void colsum_acc_kernel(const float* __restrict__ dY, int ld, float* __restrict__ out, int rows, int N,
                                  int rows_per_block) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  if (col < N) for (int r = r0 + w; r < r1; r += 4) s += dY[(long)r * ld + col];
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < N) atomicAdd(&out[col], sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
}
int colsum_acc(const Ctx& c, const float* dY, int ld, float* out, int rows, int N) {
  if (!out) return AVLEN_OK;
  int rpb = rows >= 65536 ? 1024 : rows >= 2048 ? 128 : 32;
  hipLaunchKernelGGL(colsum_acc_kernel, dim3(ceil_div(N, 64), ceil_div(rows, rpb)), dim3(256), 0, c.st, dY, ld, out,
                     rows, N, rpb);
  return avlen_launch_status();
}

__global__ void relu_bwd_kernel(float* __restrict__ dx, const float* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && y[i] <= 0.f) dx[i] = 0.f;
}
__global__ void relu_bwd16_kernel(float* __restrict__ dx, const bf16* __restrict__ y16, long n) {      // the sign survives the cast
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (float)y16[i] <= 0.f) dx[i] = 0.f;
}
int relu_bwd16(const Ctx& c, float* dx, const bf16* y16, long n) {
  hipLaunchKernelGGL(relu_bwd16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st, dx, y16, n);
  return avlen_launch_status();
}
int relu_bwd(const Ctx& c, float* dx, const float* y, long n) {
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.st, dx, y, n);
  return avlen_launch_status();
}

}  // namespace
// shared with train_gru.hip (internal.h)
int avlen_i_linear(const avlen_ctx& c, const avlen_linear& L, const float* X, int ldx, float* Y, int ldy, int M, int act,
                   const float* res, int ldr) { return linear(c, L, X, ldx, Y, ldy, M, act, res, ldr); }
int avlen_i_linear_dx(const avlen_ctx& c, const avlen_linear& L, const float* dY, int ldy, float* dX, int ldx, int M,
                      const float* add, int ldadd) { return linear_dx(c, L, dY, ldy, dX, ldx, M, add, ldadd); }
int avlen_i_linear_dw(const avlen_ctx& c, const avlen_linear& G, const float* dY, int ldy, const float* X, int ldx, int M) {
  return linear_dw(c, G, dY, ldy, X, ldx, M);
}
int avlen_i_colsum_acc(const avlen_ctx& c, const float* dY, int ld, float* out, int rows, int N) { return colsum_acc(c, dY, ld, out, rows, N); }
namespace {

// SMT fusion input.  Row (b, s): src = s < M ? memory[s, b, :] : x[b, :]
//   XF[row] = [ src[0:pc] | pose_encoder(format(relative_pose(x_pose[b] -> src_pose))) (16) | src[pc+4:F] ]
//   FMT[row] = formatted 5-vector (kept for the pose-encoder gradient)
//   maskx[b, s] = s < M ? (pretraining ? 0 : masks[b, s]) : 1
// cto (current token only): S = 1 and only the s = M row is produced.
__global__ void smt_build_kernel(const float* __restrict__ x, const float* __restrict__ memory,
                                 const int32_t* __restrict__ mem_index, int NC, const float* __restrict__ masks, const float* __restrict__ pw,
                                 const float* __restrict__ pb, float* __restrict__ XF, int ldxf, float* __restrict__ FMT,
                                 float* __restrict__ maskx, int B, int M, int F, int pc, int cto,
                                 bf16* __restrict__ XF16 = nullptr, int ld16 = 0, long lo16 = 0) {
  // XF16 (training forward at scale): the row as the 16-bit operand of the first product (hi plane; lo plane lo16 elements behind when
  // lo16 != 0; columns F + 12 .. ld16 - 1 zero); XF may then be null
  const int S = cto ? 1 : M + 1;
  const int row = blockIdx.x;                 // b * S + s
  const int b = row / S, s = cto ? M : row % S;
  const int col = mem_index ? mem_index[b] : b;
  const float* src = (s < M) ? memory + ((long)s * NC + col) * F : x + (long)b * F;
  const float* xp = x + (long)b * F + pc;
  __shared__ float fmt[5];
  const int t = threadIdx.x;
  if (t == 0) {
    float ax = xp[0], ay = xp[1], ah = xp[2];
    float bx = src[pc], by = src[pc + 1], bh = src[pc + 2], bt = src[pc + 3];
    float heading_a = -ah, heading_b = -bh;
    float dx = bx - ax, dy = by - ay;
    float r = sqrtf(dx * dx + dy * dy);
    float phi = atan2f(dy, dx) - heading_a;
    float x_ab = r * cosf(phi), y_ab = r * sinf(phi);
    float dh = heading_b - heading_a;
    dh = -atan2f(sinf(dh), cosf(dh));
    fmt[0] = x_ab; fmt[1] = y_ab; fmt[2] = cosf(dh); fmt[3] = sinf(dh); fmt[4] = expf(-bt);
    if (maskx) maskx[(long)b * S + (cto ? 0 : s)] = (s < M) ? masks[(long)b * M + s] : 1.f;
  }
  __syncthreads();
  float* o = XF ? XF + (long)row * ldxf : nullptr;
  bf16* o16 = XF16 ? XF16 + (long)row * ld16 : nullptr;
  const int ncol = XF16 && ld16 > F + 12 ? ld16 : F + 12;
  for (int i = t; i < ncol; i += blockDim.x) {
    float v;
    if (i < pc) v = src[i];
    else if (i < pc + 16) {
      int j = i - pc;
      v = pb[j];
#pragma unroll
      for (int k = 0; k < 5; k++) v += pw[j * 5 + k] * fmt[k];
    } else if (i < F + 12) v = src[i - 12];
    else v = 0.f;
    if (o && i < F + 12) o[i] = v;
    if (o16) {
      const bf16 hv = (bf16)v;
      o16[i] = hv;
      if (lo16) o16[lo16 + i] = (bf16)(v - (float)hv);
    }
  }
  if (FMT && t < 5) FMT[(long)row * 8 + t] = fmt[t];
}

// smt_build_kernel for the training forward at scale (722 k rows): ONE WAVE per row, four rows per block, no LDS / barrier -- the
// relative pose is computed by every lane (same values, no broadcast needed), a lane writes column pairs (4-byte stores of the hi
// and lo planes).  Same values as smt_build_kernel; the fp32 row is not written.
__global__ __launch_bounds__(256) void smt_build16w_kernel(const float* __restrict__ x, const float* __restrict__ memory,
                                                           const int32_t* __restrict__ mem_index, int NC, const float* __restrict__ masks,
                                                           const float* __restrict__ pw, const float* __restrict__ pb, float* __restrict__ FMT,
                                                           float* __restrict__ maskx, long R, int B, int M, int F, int pc,
                                                           bf16* __restrict__ XF16, int ld16, long lo16) {
  const int S = M + 1, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int b = (int)(row / S), s = (int)(row % S);
  const int col = mem_index ? mem_index[b] : b;
  const float* src = (s < M) ? memory + ((long)s * NC + col) * F : x + (long)b * F;
  const float* xp = x + (long)b * F + pc;
  float fmt[5];
  {
    const float ax = xp[0], ay = xp[1], ah = xp[2];
    const float bx = src[pc], by = src[pc + 1], bh = src[pc + 2], bt = src[pc + 3];
    const float heading_a = -ah, heading_b = -bh;
    const float dx = bx - ax, dy = by - ay;
    const float r = sqrtf(dx * dx + dy * dy);
    const float phi = atan2f(dy, dx) - heading_a;
    float dh = heading_b - heading_a;
    dh = -atan2f(sinf(dh), cosf(dh));
    fmt[0] = r * cosf(phi); fmt[1] = r * sinf(phi); fmt[2] = cosf(dh); fmt[3] = sinf(dh); fmt[4] = expf(-bt);
  }
  if (lane == 0) maskx[(long)b * S + s] = (s < M) ? masks[(long)b * M + s] : 1.f;
  if (lane < 5) FMT[row * 8 + lane] = fmt[lane];
  bf16* o16 = XF16 + row * ld16;
  for (int i = 2 * lane; i < ld16; i += 128) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int c = i + e;
      if (c < pc) v[e] = src[c];
      else if (c < pc + 16) {
        const int j = c - pc;
        float a = pb[j];
#pragma unroll
        for (int k = 0; k < 5; k++) a += pw[j * 5 + k] * fmt[k];
        v[e] = a;
      } else if (c < F + 12) v[e] = src[c - 12];
      else v[e] = 0.f;
    }
    const bf16 h0 = (bf16)v[0], h1 = (bf16)v[1];
    *reinterpret_cast<unsigned*>(o16 + i) = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    if (lo16) {
      const bf16 l0 = (bf16)(v[0] - (float)h0), l1 = (bf16)(v[1] - (float)h1);
      *reinterpret_cast<unsigned*>(o16 + lo16 + i) = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
  }
}

// dW_pose[16][5] += dPE^T FMT ; db_pose[16] += colsum(dPE)    (dPE: [R,16], FMT: [R,8])
// A thread takes whole rows (64 + 32 contiguous bytes each), keeps the 96 sums in registers; one wave reduction + 96 atomics per
// block.  (Until round 5: 96 threads, one output each, striding through the rows -- 0.8 ms per call at 722 k rows.)
__global__ __launch_bounds__(256) void pose_grad_kernel(const float* __restrict__ dPE, const float* __restrict__ FMT, float* __restrict__ gw,
                                                        float* __restrict__ gb, long R, int rows_per_block) {
  __shared__ float red[4][96];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float acc[96];
#pragma unroll
  for (int i = 0; i < 96; i++) acc[i] = 0.f;
  for (long r = r0 + t; r < r1; r += 256) {
    float pe[16], f[8];
#pragma unroll
    for (int u = 0; u < 4; u++) *reinterpret_cast<float4*>(&pe[4 * u]) = *reinterpret_cast<const float4*>(dPE + r * 16 + 4 * u);
#pragma unroll
    for (int u = 0; u < 2; u++) *reinterpret_cast<float4*>(&f[4 * u]) = *reinterpret_cast<const float4*>(FMT + r * 8 + 4 * u);
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[j * 5 + k] += pe[j] * f[k];
      acc[80 + j] += pe[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 96; i++) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (t < 96) {
    const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (t < 80) atomicAdd(&gw[t], v); else atomicAdd(&gb[t - 80], v);
  }
}

// dialog sequence rows: seq[b, s, :] = [ (s<M ? memory_state[s,b,:] : x_att[b,:]) | d_emb[b,:] (optional) ]
__global__ void dialog_build_kernel(const float* __restrict__ x_att, const float* __restrict__ mem,
                                    const float* __restrict__ masks, const float* __restrict__ d_emb,
                                    float* __restrict__ seq, int ldseq, float* __restrict__ maskx, int B, int M, int d) {
  const int S = M + 1, row = blockIdx.x, b = row / S, s = row % S;
  const float* src = s < M ? mem + ((long)s * B + b) * d : x_att + (long)b * d;
  float* o = seq + (long)row * ldseq;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    o[i] = src[i];
    if (d_emb) o[d + i] = d_emb[(long)b * d + i];
  }
  if (threadIdx.x == 0) maskx[(long)b * S + s] = s < M ? masks[(long)b * M + s] : 1.f;
}
// seq[b, s, :] += pe[int(agent_step[b]), :]
__global__ void add_pe_kernel(float* __restrict__ seq, const float* __restrict__ pe, const float* __restrict__ step,
                              int S, int d, int pe_len) {
  const int row = blockIdx.x, b = row / S;
  int idx = (int)step[b];
  idx = idx < 0 ? 0 : (idx >= pe_len ? pe_len - 1 : idx);
  for (int i = threadIdx.x; i < d; i += blockDim.x) seq[(long)row * d + i] += pe[(long)idx * d + i];
}

// CLIP: x[b, t, :] = tok_emb[tokens[b,t]] + pos_emb[t]; eot[b] = argmax_t tokens[b, t]
__global__ void clip_embed_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                  const float* __restrict__ pos_emb, float* __restrict__ x, int ctx, int width, int vocab) {
  const int row = blockIdx.x, t = row % ctx;
  long id = tokens[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* e = reinterpret_cast<const float4*>(tok_emb + id * width);
  const float4* p = reinterpret_cast<const float4*>(pos_emb + (long)t * width);
  float4* o = reinterpret_cast<float4*>(x + (long)row * width);
  for (int i = threadIdx.x; i < width / 4; i += blockDim.x) {
    float4 a = e[i], c = p[i];
    o[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  }
}
// Ragged batch bookkeeping: causal attention + EOT pooling means tokens AFTER the EOT token can never influence the
// output, so only the prefix [0, eot] of every dialog is computed.  seg[b] = first compact row of sample b,
// seg[B] = number of live rows; rowmap[r] = b*ctx + t of compact row r.
__global__ void clip_segments_kernel(const int64_t* __restrict__ tokens, int* __restrict__ seg, int* __restrict__ rowmap, int B,
                                     int ctx) {
  __shared__ int len[1024];
  __shared__ int off[1025];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  // one wave per sample: first position of the maximum token id (torch.argmax picks the first maximum)
  for (int b = wave; b < B; b += nw) {
    long best = -1; int bi = 0x7fffffff;
    for (int k = lane; k < ctx; k += 64) { long v = tokens[(long)b * ctx + k]; if (v > best) { best = v; bi = k; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      long ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) len[b] = bi + 1;
  }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < B; b++) { off[b] = acc; acc += len[b]; } off[B] = acc; }
  __syncthreads();
  for (int b = t; b <= B; b += blockDim.x) seg[b] = off[b];
  for (int b = wave; b < B; b += nw) {
    const int o = off[b];
    for (int k = lane; k < len[b]; k += 64) rowmap[o + k] = b * ctx + k;
  }
}
__global__ void clip_embed_ragged_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                         const float* __restrict__ pos_emb, float* __restrict__ x, const int* __restrict__ seg,
                                         const int* __restrict__ rowmap, int B, int ctx, int width, int vocab,
                                         __bf16* __restrict__ x16, float* __restrict__ stats, int f16) {
  __shared__ float sh[16];
  const int row = blockIdx.x;
  if (row >= seg[B]) return;
  const int src = rowmap[row], t = src % ctx;
  long id = tokens[src];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* e = reinterpret_cast<const float4*>(tok_emb + id * width);
  const float4* p = reinterpret_cast<const float4*>(pos_emb + (long)t * width);
  float4* o = reinterpret_cast<float4*>(x + (long)row * width);
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < width / 4; i += blockDim.x) {
    float4 a = e[i], c = p[i];
    float4 v = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    o[i] = v;
    if (x16) {           // 16-bit copy + LayerNorm statistics of the row (LayerNorm folded into the first projection)
      __bf16* d = x16 + (long)row * width + i * 4;
      if (f16) {
        d[0] = __builtin_bit_cast(__bf16, (_Float16)v.x); d[1] = __builtin_bit_cast(__bf16, (_Float16)v.y);
        d[2] = __builtin_bit_cast(__bf16, (_Float16)v.z); d[3] = __builtin_bit_cast(__bf16, (_Float16)v.w);
      } else {
        d[0] = (__bf16)v.x; d[1] = (__bf16)v.y; d[2] = (__bf16)v.z; d[3] = (__bf16)v.w;
      }
      s1 += (v.x + v.y) + (v.z + v.w); s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  }
  if (stats) {
    s1 = block_sum(s1, sh); s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { stats[(long)row * 2] = s1; stats[(long)row * 2 + 1] = s2; }
  }
}
__global__ void clip_gather_last_kernel(const float* __restrict__ x, float* __restrict__ out, const int* __restrict__ seg,
                                        int width) {
  const int b = blockIdx.x;
  const float* src = x + (long)(seg[b + 1] - 1) * width;
  for (int i = threadIdx.x; i < width; i += blockDim.x) out[(long)b * width + i] = src[i];
}

// Last-layer pruning: only the EOT row of every dialog leaves the tower, so after the last attention the residual stream
// shrinks to one row per sample.  xe = last live row of x (fp32), aoe = the same row of the attention output (bf16);
// the row-statistics slot of the compact stream is cleared for the out_proj epilogue's atomics.
__global__ void clip_gather_last2_kernel(const float* __restrict__ x, const __bf16* __restrict__ ao, const int* __restrict__ seg,
                                         float* __restrict__ xe, __bf16* __restrict__ aoe, float* __restrict__ stats, int width) {
  const int b = blockIdx.x;
  const long r = seg[b + 1] - 1;
  for (int i = threadIdx.x; i < width; i += blockDim.x) {
    xe[(long)b * width + i] = x[r * width + i];
    aoe[(long)b * width + i] = ao[r * width + i];
  }
  if (stats && threadIdx.x < 2) stats[b * 2 + threadIdx.x] = 0.f;
}

__global__ void clip_gather_eot_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ x,
                                       float* __restrict__ out, int ctx, int width) {
  const int b = blockIdx.x;
  __shared__ int s_eot;
  if (threadIdx.x == 0) {
    long best = tokens[(long)b * ctx]; int bi = 0;
    for (int t = 1; t < ctx; t++) { long v = tokens[(long)b * ctx + t]; if (v > best) { best = v; bi = t; } }
    s_eot = bi;
  }
  __syncthreads();
  const float* src = x + ((long)b * ctx + s_eot) * width;
  for (int i = threadIdx.x; i < width; i += blockDim.x) out[(long)b * width + i] = src[i];
}

// GRU gates (rnn_state_encoder.py via nn.GRU): r,z,n ordering; h' = (1-z)*n + z*h
__global__ void gru_mask_kernel(const float* __restrict__ h, const float* __restrict__ mask, float* __restrict__ hm, int N, int H) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)N * H) hm[i] = h[i] * mask[i / H];
}
__global__ void gru_gate_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ hm,
                                float* __restrict__ hout, float* __restrict__ out, int N, int H) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * H) return;
  int n = (int)(i / H), j = (int)(i % H);
  const float* a = gi + (long)n * 3 * H; const float* b = gh + (long)n * 3 * H;
  float r = 1.f / (1.f + expf(-(a[j] + b[j])));
  float z = 1.f / (1.f + expf(-(a[H + j] + b[H + j])));
  float nn = tanhf(a[2 * H + j] + r * b[2 * H + j]);
  float hv = (1.f - z) * nn + z * hm[i];
  hout[i] = hv; out[i] = hv;
}

}  // namespace

// =====================================================================================================
// bf16 fast path (rollout "perf mode"): activations that feed a GEMM are kept in bf16, weights come from the
// bf16 shadows (w16), tiles are staged with global_load_lds (igemm2.hip), GroupNorm statistics are produced by
// the conv epilogue.  Same math as the fp32-staged path with prec = BF16 (operands rounded to bf16, fp32
// accumulate); inference only (nothing is kept for backward).
// =====================================================================================================
namespace {


bool lin16_ok(const avlen_linear& L) { return L.w16 != nullptr && (L.ld16 % 8) == 0; }

// Y = act(X16 * W16^T + b) + res  -> fp32 (Y32) and/or bf16 (Y16)
// c.x3: X16 / Y16 are compensated pairs, their low planes xlo / ylo ELEMENTS behind the high planes; the weights' low plane is L.w16lo
int linear16(const Ctx& c, const avlen_linear& L, const bf16* X16, int ldx, float* Y32, int ld32, bf16* Y16, int ld16, int M,
             int act, const float* res, int ldr, long xlo = 0, long ylo = 0) {
  if (c.x3) {
    if (!L.w16lo || !xlo || (Y16 && !ylo)) return AVLEN_ERR_ARG;
    avlen_g2_opts o; o.x3 = 1; o.a_lo = xlo * 2; o.b_lo = (const char*)L.w16lo - (const char*)L.w16; o.c16_lo = ylo;
    return avlen_gemm_bf16_dyn(X16, ldx, L.w16, L.ld16, Y32, ld32, Y16, ld16, L.b, res, ldr, M, c.live, L.out_f, L.ld16, act, c.gws,
                               c.gws_bytes, c.st, &o);
  }
  if (c.live)          // ragged batch: only the first *live rows exist
    return avlen_gemm_bf16_dyn(X16, ldx, L.w16, L.ld16, Y32, ld32, Y16, ld16, L.b, res, ldr, M, c.live, L.out_f, L.ld16, act, c.gws,
                               c.gws_bytes, c.st);
  return avlen_gemm_bf16(X16, ldx, L.w16, L.ld16, Y32, ld32, Y16, ld16, L.b, res, ldr, M, L.out_f, L.ld16, act, c.gws,
                         c.gws_bytes, c.st);
}
int ln16(const Ctx& c, const float* x, const avlen_affine& a, float* y, bf16* y16, int rows, int d, long ylo = 0) {
  if (c.x3 && y16 && !ylo) return AVLEN_ERR_ARG;
  return avlen_layernorm_fwd16_dyn(x, nullptr, a.g, a.b, y, y16, nullptr, nullptr, rows, c.live, d, 1e-5f, c.st, c.x3 ? ylo : 0);
}
int linear16_rows(const Ctx& c, const avlen_linear& L, int r0, int n, const bf16* X16, int ldx, float* Y32, int ld32,
                  bf16* Y16, int ld16, int M, int act, long xlo = 0, long ylo = 0) {
  avlen_linear S = L;
  S.w16 = (char*)L.w16 + (size_t)r0 * L.ld16 * 2; S.b = L.b ? L.b + r0 : nullptr; S.out_f = n;
  if (L.w16lo) S.w16lo = (char*)L.w16lo + (size_t)r0 * L.ld16 * 2;
  return linear16(c, S, X16, ldx, Y32, ld32, Y16, ld16, M, act, nullptr, 0, xlo, ylo);
}

// ---- program builder for the fused row-batch chain (chain.hip) ----
struct ChainB {
  avlen_chain p; bool ok; bool x3;                  // x3: a compensated-bf16 program (run with avlen_chain_run(..., 1))
  explicit ChainB(bool x3_ = false) : ok(true), x3(x3_) { p.n = 0; }
  avlen_chain_op& next(int kind) {
    static avlen_chain_op dummy;
    if (p.n >= AVLEN_CHAIN_MAX_OPS) { ok = false; return dummy; }
    avlen_chain_op& o = p.op[p.n++];
    o = avlen_chain_op{kind, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0.f, 0, nullptr, nullptr, nullptr};
    return o;
  }
  int run(int B, hipStream_t st) const { return avlen_chain_run(&p, B, st, x3 ? 1 : 0); }
  void load_x16(const bf16* x, int ld, int k, int buf, const bf16* xlo = nullptr) {
    auto& o = next(AVLEN_CH_LOAD_X16); o.p0 = x; o.p1 = xlo; o.ld = ld; o.k = k; o.buf = buf;
    if (x3 && !xlo) ok = false;
  }
  void load_cur(const float* x, int ld, int buf, int div = 1) { auto& o = next(AVLEN_CH_LOAD_CUR); o.p0 = x; o.ld = ld; o.buf = buf; o.div = div; }
  void linear(const avlen_linear& L, int r0, int act, int res, int buf, int out_buf) {
    auto& o = next(AVLEN_CH_LINEAR);
    o.p0 = (const char*)L.w16 + (size_t)r0 * L.ld16 * 2; o.p1 = L.b ? L.b + r0 : nullptr;
    o.p2 = L.w16lo ? (const char*)L.w16lo + (size_t)r0 * L.ld16 * 2 : nullptr;
    o.k = L.ld16; o.ld = L.ld16; o.act = act; o.res = res; o.buf = buf; o.out_buf = out_buf;
    if (!L.w16 || (L.ld16 % 8) || L.ld16 > (x3 ? 352 : 320) || L.out_f < r0 + 256 || (x3 && !L.w16lo)) ok = false;    // chain.hip: KMAX / KMAX64
  }
  void ln(const avlen_affine& a, int out_buf) { auto& o = next(AVLEN_CH_LAYERNORM); o.p0 = a.g; o.p1 = a.b; o.out_buf = out_buf; }
  void save(int slot = 0) { auto& o = next(AVLEN_CH_SAVE); o.res = slot; }
  void recall(int slot, int out_buf) { auto& o = next(AVLEN_CH_RECALL); o.res = slot; o.out_buf = out_buf; }
  void store(float* y, int ld, bf16* y16, int ld2, int div = 1) {
    auto& o = next(AVLEN_CH_STORE); o.p0 = y; o.ld = ld; o.p1 = y16; o.ld2 = ld2; o.div = div;
  }
  void attn(int qslot, int kimg, int vimg, int out_buf, const float* key_mask, int seq, float scale) {
    auto& o = next(AVLEN_CH_ATTN); o.res = qslot; o.buf = kimg; o.ld2 = vimg; o.out_buf = out_buf; o.p0 = key_mask; o.seq = seq;
    o.scale = scale;
  }
  void add_pe(const float* table, int len, const float* step, int div, int out_buf) {
    auto& o = next(AVLEN_CH_ADD_PE); o.p0 = table; o.k = len; o.p1 = step; o.div = div; o.out_buf = out_buf;
  }
  // nn.Transformer(1 encoder + 1 decoder layer, post-norm) on groups of S <= 4 rows (tokens of one sample), the single
  // decoder target replicated on the group's rows; expects the fused sequence in registers, image 0 and save slot 0.
  void small_transformer(const avlen_transformer& tr, const float* goal, const float* key_mask, int S, float* out) {
    const int d = tr.d;
    const float scale = 1.0f / sqrtf((float)(d / tr.nhead));
    const avlen_enc_layer& e = tr.enc; const avlen_dec_layer& q = tr.dec;
    linear(e.self_attn.in_proj, 0, 0, 0, 0, 1); save(1);                 // Q (registers)
    linear(e.self_attn.in_proj, d, 0, 0, 0, 1);                          // K -> image 1
    linear(e.self_attn.in_proj, 2 * d, 0, 0, 0, 0);                      // V -> image 0 (in place)
    attn(1, 1, 0, 1, key_mask, S, scale);
    linear(e.self_attn.out_proj, 0, 0, 1, 1, 0);                         // + Z
    ln(e.norm1, 0); save();                                              // X1
    linear(e.lin1, 0, AVLEN_ACT_RELU, 0, 0, 1);
    linear(e.lin2, 0, 0, 1, 1, 0);
    ln(e.norm2, 0); ln(tr.enc_norm, 0);                                  // memory tokens in image 0
    load_cur(goal, d, 1, S); save();                                     // decoder target (one per sample), in place on image 1
    linear(q.self_attn.in_proj, 2 * d, 0, 0, 1, 1);                      // one target token: self attention == V projection
    linear(q.self_attn.out_proj, 0, 0, 1, 1, 1);
    ln(q.norm1, 1); save();                                              // Y1
    linear(q.cross_attn.in_proj, 0, 0, 0, 1, 1); save(1);                // cross-attention query
    linear(q.cross_attn.in_proj, d, 0, 0, 0, 1);                         // K(memory) -> image 1
    linear(q.cross_attn.in_proj, 2 * d, 0, 0, 0, 0);                     // V(memory) -> image 0 (in place)
    attn(1, 1, 0, 1, key_mask, S, scale);
    linear(q.cross_attn.out_proj, 0, 0, 1, 1, 0);                        // + Y1
    ln(q.norm2, 0); save();
    linear(q.lin1, 0, AVLEN_ACT_RELU, 0, 0, 1);
    linear(q.lin2, 0, 0, 1, 1, 0);
    ln(q.norm3, 0); ln(tr.dec_norm, 0);
    store(out, d, nullptr, 0, S);
    if (d != 256 || tr.nhead != 8 || e.lin1.out_f != 256 || q.lin1.out_f != 256) ok = false;
  }
};

bool ragged_enabled() {
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_SMT_RAGGED", 1);
  return v != 0;
}

bool chain_enabled() {
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_CHAIN", 1);
  return v != 0;
}

// Live tokens of every sample: the valid memory slots (mask != 0, in slot order) followed by the current token.  Masked slots
// are dead work for the whole encoder: no other token attends to them (src_key_padding_mask) and the decoder's cross
// attention skips them (memory_key_padding_mask), smt_state_encoder.py:126-129,186.  seg[b] = first compact row of sample
// b, seg[B] = number of live rows; rowmap[r] = b*(M+1) + s.  One wave per sample.
__global__ void smt_segments_kernel(const float* __restrict__ masks, int* __restrict__ seg, int* __restrict__ rowmap, int B,
                                    int M) {
  __shared__ int cnt[1024];
  __shared__ int off[1025];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  for (int b = wave; b < B; b += nw) {
    int n = 0;
    for (int s0 = 0; s0 < M; s0 += 64) {
      const int s = s0 + lane;
      const bool v = s < M && masks[(long)b * M + s] != 0.f;
      n += __popcll(__ballot(v));
    }
    if (lane == 0) cnt[b] = n + 1;
  }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < B; b++) { off[b] = acc; acc += cnt[b]; } off[B] = acc; }
  __syncthreads();
  for (int b = t; b <= B; b += blockDim.x) seg[b] = off[b];
  for (int b = wave; b < B; b += nw) {
    int o = off[b];
    for (int s0 = 0; s0 < M; s0 += 64) {
      const int s = s0 + lane;
      const bool v = s < M && masks[(long)b * M + s] != 0.f;
      const unsigned long long bal = __ballot(v);
      if (v) rowmap[o + __popcll(bal & ((1ull << lane) - 1ull))] = b * (M + 1) + s;
      o += __popcll(bal);
    }
    if (lane == 0) rowmap[o] = b * (M + 1) + M;
  }
}

__global__ void smt_build16_kernel(const float* __restrict__ x, const float* __restrict__ memory,
                                   const int32_t* __restrict__ mem_index, int NC, const float* __restrict__ masks,
                                   const float* __restrict__ pw, const float* __restrict__ pb, bf16* __restrict__ XF, int ldxf,
                                   float* __restrict__ maskx, int B, int M, int F, int pc, int cto,
                                   const int* __restrict__ seg, const int* __restrict__ rowmap, bf16* __restrict__ XFlo) {
  const int S = cto ? 1 : M + 1;
  int row = blockIdx.x, b = row / S, s = cto ? M : row % S;
  if (rowmap) {                       // ragged: compact row -> (sample, slot)
    if (row >= seg[B]) return;
    const int rs = rowmap[row];
    b = rs / S; s = rs - b * S;
  }
  const int col = mem_index ? mem_index[b] : b;
  const float* src = (s < M) ? memory + ((long)s * NC + col) * F : x + (long)b * F;
  const float* xp = x + (long)b * F + pc;
  __shared__ float fmt[5];
  const int t = threadIdx.x;
  if (t == 0) {
    float ax = xp[0], ay = xp[1], ah = xp[2];
    float bx = src[pc], by = src[pc + 1], bh = src[pc + 2], bt = src[pc + 3];
    float heading_a = -ah, heading_b = -bh;
    float dx = bx - ax, dy = by - ay;
    float r = sqrtf(dx * dx + dy * dy);
    float phi = atan2f(dy, dx) - heading_a;
    float dh = heading_b - heading_a;
    dh = -atan2f(sinf(dh), cosf(dh));
    fmt[0] = r * cosf(phi); fmt[1] = r * sinf(phi); fmt[2] = cosf(dh); fmt[3] = sinf(dh); fmt[4] = expf(-bt);
    if (!rowmap) maskx[(long)b * S + (cto ? 0 : s)] = (s < M) ? masks[(long)b * M + s] : 1.f;
  }
  __syncthreads();
  bf16* o = XF + (long)row * ldxf;
  for (int i = t; i < ldxf; i += blockDim.x) {
    float v = 0.f;
    if (i < pc) v = src[i];
    else if (i < pc + 16) {
      int j = i - pc;
      v = pb[j];
#pragma unroll
      for (int k = 0; k < 5; k++) v += pw[j * 5 + k] * fmt[k];
    } else if (i < F + 12) v = src[i - 12];
    const bf16 hv = (bf16)v;
    o[i] = hv;
    if (XFlo) XFlo[(long)row * ldxf + i] = (bf16)(v - (float)hv);
  }
}

// ---- ResNet-18 tower ----
size_t resnet18_ws_bf16(int B) {
  size_t px = (size_t)B * 4096;
  return px * 8 * 2 + 3 * (px * 16 * 4 + 256) + 4 * (px * 16 * 2 + 256) + 21 * ((size_t)B * 2 * 128 * 4 + 256) +
         avlen_gemm_bf16_workspace_bytes(B, 64) + 8192;
}

bool resnet18_has16(const avlen_resnet18* n) {
  if (!n->conv1.w16 || !n->fc.w16) return false;
  for (int i = 0; i < 8; i++) {
    if (!n->block[i].conv1.w16 || !n->block[i].conv2.w16) return false;
    if (n->block[i].has_down && !n->block[i].down.w16) return false;
  }
  return true;
}

// G towers of identical shape in lock-step: every conv / GroupNorm / fc is ONE grouped launch (blockIdx.y = tower).
int resnet18_group_fwd_bf16(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                            const float* divisors, float* const* outs, int ld_out, int G, int B, int S, void* ws,
                            size_t ws_bytes, hipStream_t st, const int* row_index = nullptr) {
  if (G < 1 || G > 8 || ws_bytes < (size_t)G * resnet18_ws_bf16(B)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes);
  size_t px = (size_t)B * 4096;
  size_t stat_stride = align_up((size_t)B * 2 * 128, 64);
  bf16* x0[8]; bf16* raw[3][8]; bf16* act[4][8];
  for (int g = 0; g < G; g++) {
    x0[g] = w.take<bf16>(px * 8);
    for (int i = 0; i < 3; i++) raw[i][g] = w.take<bf16>(px * 16);     // raw conv outputs kept in bf16
    for (int i = 0; i < 4; i++) act[i][g] = w.take<bf16>(px * 16);
  }
  float* stats_all = w.take<float>((size_t)G * 21 * stat_stride);
  size_t gwsb = (size_t)G * avlen_gemm_bf16_workspace_bytes(B, 64);
  void* gws = w.take<char>(gwsb);
  TRY(avlen_zero_bytes(stats_all, (size_t)G * 21 * stat_stride * sizeof(float), st));
  int si = 0;
  auto next_stats = [&](float** out) { for (int g = 0; g < G; g++) out[g] = stats_all + ((size_t)g * 21 + si) * stat_stride; si++; };
  // Preprocessing + stem + layers 1-2 as one launch per tower group (tower_head.hip: the 64x64x16 / 32x32x32 activations never leave the CU)
  static int head = -1;                        // AVLEN_TOWER_HEAD=0 (lab build): the launch-per-layer path below
  if (head < 0) head = (int)avlen_knob("AVLEN_TOWER_HEAD", 1);
  bool use_head = head != 0;
  for (int g = 0; g < G; g++) {
    if (!resnet18_has16(nets[g]) || channels[g] > 8) return AVLEN_ERR_ARG;
    use_head = use_head && avlen_tower_head_supported(nets[g], S, channels[g]);
  }
  for (int g = 0; g < G && !use_head; g++) {
    int same = -1;                      // towers of different policies read the same image: preprocess it once
    for (int h = 0; h < g && same < 0; h++)
      if (imgs[h] == imgs[g] && channels[h] == channels[g] && divisors[h] == divisors[g]) same = h;
    if (same >= 0) { x0[g] = x0[same]; continue; }
    TRY(avlen_preprocess_image_bf16(imgs[g], img_u8 ? img_u8[g] : 0, x0[g], B, S, channels[g], divisors[g], st, row_index));
  }
  const void* X[8]; const void* Wt[8]; void* Y[8]; float* ST[8]; float* ST2[8]; float* ST3[8];
  const float* GA[8]; const float* BE[8]; const void* RES[8]; void* OUT[8]; const void* XR[8];
  auto conv = [&](auto getk, bf16** xin, bf16** rawo, float** stt, int H) -> int {
    const avlen_conv& k0 = getk(nets[0]);
    for (int g = 0; g < G; g++) { X[g] = xin[g]; Wt[g] = getk(nets[g]).w16; Y[g] = rawo[g]; }
    if (avlen_dconv_supported(H, k0.cin16, k0.cout, k0.kh, k0.kw, k0.stride, k0.pad))      // small-channel stages: direct conv
      return avlen_dconv_bf16_grouped(X, Wt, Y, stt, G, B, H, k0.cin16, k0.cout, k0.kh, st);
    return avlen_conv2d_nhwc_bf16_grouped(X, Wt, nullptr, Y, stt, G, B, H, H, k0.cin16, k0.cout, k0.kh, k0.kw, k0.stride, k0.pad, gws,
                                          gwsb, st);
  };
  // rn != nullptr: the residual is a RAW conv output normalised on the fly with (rst, rn's affine, rrelu)
  const float* RGA[8]; const float* RBE[8];
  auto gn = [&](auto getn, bf16** rawi, float** stt, bf16** res, bf16** yo, int HW, int C, int relu,
                const avlen_affine* (*rn)(const avlen_resnet18*, int) = nullptr, int rblk = 0, float** rst = nullptr, int rrelu = 0) -> int {
    for (int g = 0; g < G; g++) {
      XR[g] = rawi[g]; GA[g] = getn(nets[g]).g; BE[g] = getn(nets[g]).b; RES[g] = res ? res[g] : nullptr; OUT[g] = yo[g];
      if (rn) { RGA[g] = rn(nets[g], rblk)->g; RBE[g] = rn(nets[g], rblk)->b; }
    }
    return avlen_groupnorm_apply_bf16_grouped(XR, 1, (const float* const*)stt, GA, BE, res ? RES : nullptr, OUT, G, B, HW, C, 16,
                                              relu, 1e-5f, st, rn ? (const float* const*)rst : nullptr, rn ? RGA : nullptr,
                                              rn ? RBE : nullptr, rrelu);
  };
  static int fuse_gn = -1;                     // AVLEN_DCONV_FUSE_GN=0: run bn1 as its own pass (A/B knob)
  if (fuse_gn < 0) fuse_gn = (int)avlen_knob("AVLEN_DCONV_FUSE_GN", 1);
  auto fuse_gn_on = [&]() { return fuse_gn != 0; };
  static int resfuse = -1;                     // AVLEN_GN_RESFUSE=0: the stem's bn1 and the downsample norm as their own passes (A/B knob)
  if (resfuse < 0) resfuse = (int)avlen_knob("AVLEN_GN_RESFUSE", 1);
  float* STS[8];                               // the stem's statistics outlive block 0 when its norm is applied by the consumers
  next_stats(STS);
  const avlen_resblock& kb0 = nets[0]->block[0];
  // The stem's bn1 + ReLU has two consumers: block 0's conv1 (a direct conv: applied in its halo staging) and block 0's
  // residual add (applied inside that GroupNorm pass) -- the normalised stem output never exists in HBM.
  const bool stem_fused = resfuse && fuse_gn_on() && !kb0.has_down && kb0.conv1.cout == 16 && nets[0]->conv1.cout == 16 &&
                          avlen_dconv_supported(64, kb0.conv1.cin16, kb0.conv1.cout, kb0.conv1.kh, kb0.conv1.kw, kb0.conv1.stride, kb0.conv1.pad);
  if (use_head) {
    void* HY[8];
    for (int g = 0; g < G; g++) HY[g] = act[0][g];
    TRY(avlen_tower_head_bf16(nets, imgs, img_u8, channels, divisors, row_index, HY, G, B, S, st));
  } else {
    TRY(conv([](const avlen_resnet18* n) -> const avlen_conv& { return n->conv1; }, x0, stem_fused ? raw[2] : raw[0], STS, 64));
    if (!stem_fused)
      TRY(gn([](const avlen_resnet18* n) -> const avlen_affine& { return n->bn1; }, raw[0], STS, nullptr, act[0], 4096, 16, 1));
  }
  bf16** cur = (stem_fused && !use_head) ? raw[2] : act[0]; bf16** a1 = act[1]; bf16** idt = act[2]; bf16** nxt = act[3];
  int H = use_head ? 32 : 64;
  static int tail = -1;                        // AVLEN_TOWER_TAIL=0: layers 3-4 as separate conv / GroupNorm launches (A/B knob)
  if (tail < 0) tail = (int)avlen_knob("AVLEN_TOWER_TAIL", 1);
  for (int i = use_head ? 4 : 0; i < 8; i++) {
    if (i == 4 && tail && H == 32) {
      // layers 3 + 4 (four basic blocks): one launch, one workgroup per image, activations resident in LDS
      for (int g = 0; g < G; g++) { X[g] = cur[g]; OUT[g] = nxt[g]; }
      int rc = avlen_tower_tail_bf16(nets, X, OUT, G, B, st);
      if (rc == AVLEN_OK) { cur = nxt; break; }
      if (rc != AVLEN_ERR_ARG) return rc;      // unsupported channel plan: fall through to the layer-by-layer path
    }
    const avlen_resblock& k = nets[0]->block[i];
    int s = k.conv1.stride, OH = (H + 2 - 3) / s + 1, Co = k.conv1.cout;
    next_stats(ST); next_stats(ST2);
    const bool raw_in = i == 0 && stem_fused;     // cur is the stem's RAW output
    if (raw_in) {
      for (int g = 0; g < G; g++) {
        X[g] = cur[g]; Wt[g] = nets[g]->block[0].conv1.w16; Y[g] = raw[0][g]; GA[g] = nets[g]->bn1.g; BE[g] = nets[g]->bn1.b;
      }
      TRY(avlen_dconv_bf16_grouped(X, Wt, Y, ST, G, B, H, k.conv1.cin16, k.conv1.cout, k.conv1.kh, st, (const float* const*)STS, GA, BE));
    } else {
      TRY(conv([i](const avlen_resnet18* n) -> const avlen_conv& { return n->block[i].conv1; }, cur, raw[0], ST, H));
    }
    const avlen_conv& k2 = k.conv2;
    if (fuse_gn && (Co == 16 || Co == 32) && avlen_dconv_supported(OH, k2.cin16, k2.cout, k2.kh, k2.kw, k2.stride, k2.pad)) {
      // bn1 + ReLU feed conv2 only: applied inside the direct conv's halo staging (no separate pass over the activation)
      for (int g = 0; g < G; g++) {
        X[g] = raw[0][g]; Wt[g] = nets[g]->block[i].conv2.w16; Y[g] = raw[1][g];
        GA[g] = nets[g]->block[i].bn1.g; BE[g] = nets[g]->block[i].bn1.b;
      }
      TRY(avlen_dconv_bf16_grouped(X, Wt, Y, ST2, G, B, OH, k2.cin16, k2.cout, k2.kh, st, (const float* const*)ST, GA, BE));
    } else {
      TRY(gn([i](const avlen_resnet18* n) -> const avlen_affine& { return n->block[i].bn1; }, raw[0], ST, nullptr, a1, OH * OH, Co, 1));
      TRY(conv([i](const avlen_resnet18* n) -> const avlen_conv& { return n->block[i].conv2; }, a1, raw[1], ST2, OH));
    }
    bf16** identity = cur;
    bool down_fused = false;
    if (k.has_down) {
      next_stats(ST3);
      TRY(conv([i](const avlen_resnet18* n) -> const avlen_conv& { return n->block[i].down; }, cur, raw[2], ST3, H));
      down_fused = resfuse != 0;
      if (!down_fused)
        TRY(gn([i](const avlen_resnet18* n) -> const avlen_affine& { return n->block[i].bnd; }, raw[2], ST3, nullptr, idt, OH * OH, Co, 0));
      identity = down_fused ? raw[2] : idt;
    }
    auto bn2 = [i](const avlen_resnet18* n) -> const avlen_affine& { return n->block[i].bn2; };
    if (raw_in)
      TRY(gn(bn2, raw[1], ST2, identity, nxt, OH * OH, Co, 1,
             [](const avlen_resnet18* n, int) -> const avlen_affine* { return &n->bn1; }, 0, STS, 1));
    else if (down_fused)
      TRY(gn(bn2, raw[1], ST2, identity, nxt, OH * OH, Co, 1,
             [](const avlen_resnet18* n, int blk) -> const avlen_affine* { return &n->block[blk].bnd; }, i, ST3, 0));
    else
      TRY(gn(bn2, raw[1], ST2, identity, nxt, OH * OH, Co, 1));
    if (raw_in) { cur = nxt; nxt = act[0]; }       // raw[2] goes back to being the downsample scratch
    else { bf16** old = cur; cur = nxt; nxt = old; }
    H = OH;
  }
  const void* FA[8]; const void* FB[8]; const float* FBI[8];
  for (int g = 0; g < G; g++) { FA[g] = cur[g]; FB[g] = nets[g]->fc.w16; FBI[g] = nets[g]->fc.b; }
  const avlen_linear& fc = nets[0]->fc;
  return avlen_gemm_bf16_grouped(FA, fc.ld16, FB, fc.ld16, outs, ld_out, FBI, G, B, fc.out_f, fc.ld16, 0, gws, gwsb, st);
}

int resnet18_fwd_bf16(const avlen_resnet18* net, const void* img, int img_u8, int B, int S, int C, float divisor, float* out,
                      int ld_out, void* ws, size_t ws_bytes, hipStream_t st) {
  return resnet18_group_fwd_bf16(&net, &img, &img_u8, &C, &divisor, &out, ld_out, 1, B, S, ws, ws_bytes, st);
}

// ---- 3-conv CNNs ----
int g_audio3 = 1;                // avlen_set_audio3(0): cast + one implicit-GEMM launch per conv (tests compare the two)
bool audio3_enabled() { return g_audio3 != 0; }
void cnn3_dims2(const avlen_cnn3* n, int H, int W, int oh[3], int ow[3]) {
  for (int i = 0; i < 3; i++) {
    oh[i] = (H - n->conv[i].kh) / n->conv[i].stride + 1;
    ow[i] = (W - n->conv[i].kw) / n->conv[i].stride + 1;
    H = oh[i]; W = ow[i];
  }
}
// conv0 in "super-pixel" form (see avlen_conv::w16c): stride pixels x cin channels = one 8-channel pixel
bool cnn3_superpixel(const avlen_cnn3* n, int W) {
  const avlen_conv& k = n->conv[0];
  return k.w16c && k.cin * k.stride == 8 && k.kw % k.stride == 0 && k.pad == 0 && W >= k.kw;
}
bool cnn3_has16(const avlen_cnn3* n) { return n->conv[0].w16 && n->conv[1].w16 && n->conv[2].w16 && n->fc.w16; }
size_t cnn3_ws_bf16(const avlen_cnn3* n, int B, int H, int W) {
  int oh[3], ow[3]; cnn3_dims2(n, H, W, oh, ow);
  size_t tot = (size_t)B * H * W * 8 * 2 + 256;
  size_t mx = 0;
  for (int i = 0; i < 3; i++) {
    tot += (size_t)B * oh[i] * ow[i] * n->conv[i].cout * 2 + 256;
    mx = zmax(mx, avlen_gemm_bf16_workspace_bytes(B * oh[i] * ow[i], n->conv[i].cout));
  }
  return tot + zmax(mx, avlen_gemm_bf16_workspace_bytes(B, n->fc.out_f)) + 4096;
}
// f16: the 16-bit shadows of this CNN (n->half_fmt == 1) and every 16-bit activation are IEEE half
int cnn3_fwd_bf16(const avlen_cnn3* n, const float* x, int B, int H, int W, float* out, int ld_out, void* ws,
                  size_t ws_bytes, hipStream_t st, const int* row_index = nullptr, bool f16 = false) {
  if (ws_bytes < cnn3_ws_bf16(n, B, H, W)) return AVLEN_ERR_WS;
  if (f16 != (n->half_fmt == 1)) return AVLEN_ERR_ARG;
  avlen_g2_opts go; go.f16 = f16;
  const int fmt = f16 ? 1 : 0;
  int oh[3], ow[3]; cnn3_dims2(n, H, W, oh, ow);
  WsBump w(ws, ws_bytes);
  bf16* x16 = w.take<bf16>((size_t)B * H * W * 8);
  bf16* a[3];
  size_t mx = avlen_gemm_bf16_workspace_bytes(B, n->fc.out_f);
  for (int i = 0; i < 3; i++) {
    a[i] = w.take<bf16>((size_t)B * oh[i] * ow[i] * n->conv[i].cout);
    mx = zmax(mx, avlen_gemm_bf16_workspace_bytes(B * oh[i] * ow[i], n->conv[i].cout));
  }
  void* gws = w.take<char>(mx);
  if (audio3_enabled() && avlen_i_audio3_ok(n, H, W) && n->fc.ld16 == oh[2] * ow[2] * n->conv[2].cout) {
    void* co = a[2];
    TRY(avlen_i_audio3_fwd(&n, x, row_index, 1, B, H, W, &co, st));
    return avlen_gemm_bf16_dyn(a[2], n->fc.ld16, n->fc.w16, n->fc.ld16, out, ld_out, nullptr, 0, n->fc.b, nullptr, 0, B, nullptr,
                               n->fc.out_f, n->fc.ld16, AVLEN_ACT_RELU, gws, mx, st, &go);
  }
  const bool sp = cnn3_superpixel(n, W);
  const int wsp = sp ? ((ow[0] - 1) * n->conv[0].stride + n->conv[0].kw) / n->conv[0].stride : 0;   // super-pixels per row
  if (sp) TRY(avlen_cast_bf16_indexed(x, W * n->conv[0].cin, x16, wsp * 8, (long)B * H, wsp * 8, row_index, H, st, fmt));     // drop the unused columns
  else TRY(avlen_cast_bf16_indexed(x, n->conv[0].cin, x16, 8, (long)B * H * W, n->conv[0].cin, row_index, H * W, st, fmt));         // channel-pad to 8
  const bf16* cur = x16; int h = H, wd = W;
  for (int i = 0; i < 3; i++) {
    const avlen_conv& k = n->conv[i];
    const void* X1 = cur; void* Y1 = a[i]; const float* B1 = k.b;
    if (i == 0 && sp) {
      const void* W1 = k.w16c;
      TRY(avlen_conv2d_nhwc_bf16_grouped(&X1, &W1, nullptr, &Y1, nullptr, 1, B, h, wsp, 8, k.cout, k.kh, k.kw / k.stride, k.stride,
                                         0, gws, mx, st, &B1, AVLEN_ACT_RELU, 1, &go));
    } else {
      const void* W1 = k.w16;
      TRY(avlen_conv2d_nhwc_bf16_grouped(&X1, &W1, nullptr, &Y1, nullptr, 1, B, h, wd, k.cin16, k.cout, k.kh, k.kw, k.stride, 0, gws,
                                         mx, st, &B1, i < 2 ? AVLEN_ACT_RELU : AVLEN_ACT_NONE, 0, &go));
    }
    cur = a[i]; h = oh[i]; wd = ow[i];
  }
  return avlen_gemm_bf16_dyn(cur, n->fc.ld16, n->fc.w16, n->fc.ld16, out, ld_out, nullptr, 0, n->fc.b, nullptr, 0, B, nullptr,
                             n->fc.out_f, n->fc.ld16, AVLEN_ACT_RELU, gws, mx, st, &go);
}

}  // namespace
// The TRAINING forward of a 3-conv CNN on the same 16-bit kernels (train_gru.hip, bf16 mode): every conv stores its fp32 output
// (keep[i], what the backward reads) next to the bf16 copy the next layer consumes; x16 / a16: caller's scratch (B*H*W*8 and the
// three activation sizes, bf16).  The fp32-staged implicit GEMM it replaces gathers fp32 patches in the kernel: 244 us for the
// audio conv 1 of a 1200-row minibatch.  AVLEN_NOT_BIG: no 16-bit shadows / unsupported geometry.
int avlen_i_cnn3_fwd16_keep(const avlen_cnn3* n, const float* x, int B, int H, int W, float* const* keep, float* out, int ld_out,
                            void* x16v, void* const* a16v, void* gws, size_t gws_bytes, hipStream_t st) {
  if (!cnn3_has16(n) || n->conv[0].cin > 8 || !x16v || !a16v) return AVLEN_NOT_BIG;
  const int fmt = n->half_fmt == 1 ? 1 : 0;            // the net's 16-bit shadows: bf16, or fp16 (the accurate mode of the GRU baseline)
  avlen_g2_opts go; go.f16 = fmt;
  int oh[3], ow[3]; cnn3_dims2(n, H, W, oh, ow);
  if (oh[2] <= 0 || ow[2] <= 0) return AVLEN_ERR_ARG;
  bf16* x16 = (bf16*)x16v;
  const bool sp = cnn3_superpixel(n, W);
  const int wsp = sp ? ((ow[0] - 1) * n->conv[0].stride + n->conv[0].kw) / n->conv[0].stride : 0;
  if (sp) TRY(avlen_cast_bf16_indexed(x, W * n->conv[0].cin, x16, wsp * 8, (long)B * H, wsp * 8, nullptr, H, st, fmt));
  else TRY(avlen_cast_bf16_indexed(x, n->conv[0].cin, x16, 8, (long)B * H * W, n->conv[0].cin, nullptr, H * W, st, fmt));
  const bf16* cur = x16; int h = H, wd = W;
  for (int i = 0; i < 3; i++) {
    const avlen_conv& k = n->conv[i];
    const void* X1 = cur; void* Y1 = a16v[i]; float* Y32 = keep[i]; const float* B1 = k.b;
    if (i == 0 && sp) {
      const void* W1 = k.w16c;
      TRY(avlen_conv2d_nhwc_bf16_grouped(&X1, &W1, &Y32, &Y1, nullptr, 1, B, h, wsp, 8, k.cout, k.kh, k.kw / k.stride, k.stride,
                                         0, gws, gws_bytes, st, &B1, AVLEN_ACT_RELU, 1, &go));
    } else {
      const void* W1 = k.w16;
      TRY(avlen_conv2d_nhwc_bf16_grouped(&X1, &W1, &Y32, &Y1, nullptr, 1, B, h, wd, k.cin16, k.cout, k.kh, k.kw, k.stride, 0, gws,
                                         gws_bytes, st, &B1, i < 2 ? AVLEN_ACT_RELU : AVLEN_ACT_NONE, 0, &go));
    }
    cur = (const bf16*)a16v[i]; h = oh[i]; wd = ow[i];
  }
  return avlen_gemm_bf16_dyn(cur, n->fc.ld16, n->fc.w16, n->fc.ld16, out, ld_out, nullptr, 0, n->fc.b, nullptr, 0, B, nullptr,
                             n->fc.out_f, n->fc.ld16, AVLEN_ACT_RELU, gws, gws_bytes, st, &go);
}
namespace {

// `G` CNNs of identical architecture (the audio encoders of pi_q / pi_g / pi_l) on the SAME input: one cast, one grouped
// launch per layer.
int cnn3_group_fwd_bf16(const avlen_cnn3* const* nets, const float* x, int G, int B, int H, int W, float* const* outs,
                        int ld_out, void* ws, size_t ws_bytes, hipStream_t st, bool f16 = false) {
  const avlen_cnn3* n = nets[0];
  if (ws_bytes < (size_t)G * cnn3_ws_bf16(n, B, H, W)) return AVLEN_ERR_WS;
  for (int g = 0; g < G; g++) if (f16 != (nets[g]->half_fmt == 1)) return AVLEN_ERR_ARG;
  avlen_g2_opts go; go.f16 = f16;
  const int fmt = f16 ? 1 : 0;
  int oh[3], ow[3]; cnn3_dims2(n, H, W, oh, ow);
  WsBump w(ws, ws_bytes);
  bf16* x16 = w.take<bf16>((size_t)B * H * W * 8);
  bf16* a[3][8];
  size_t mx = avlen_gemm_bf16_workspace_bytes(B, n->fc.out_f);
  for (int i = 0; i < 3; i++) {
    for (int g = 0; g < G; g++) a[i][g] = w.take<bf16>((size_t)B * oh[i] * ow[i] * n->conv[i].cout);
    mx = zmax(mx, avlen_gemm_bf16_workspace_bytes(B * oh[i] * ow[i], n->conv[i].cout));
  }
  mx *= G;
  void* gws = w.take<char>(mx);
  if (!w.ok()) return AVLEN_ERR_WS;
  bool fused = audio3_enabled() && n->fc.ld16 == oh[2] * ow[2] * n->conv[2].cout;
  for (int g = 0; g < G; g++) fused = fused && avlen_i_audio3_ok(nets[g], H, W);
  if (fused) {       // the three convs in one launch per (encoder, spectrogram): activations never leave LDS (audio3.hip)
    void* co[8]; const void* FA[8]; const void* FB[8]; const float* FBI[8];
    for (int g = 0; g < G; g++) { co[g] = a[2][g]; FA[g] = a[2][g]; FB[g] = nets[g]->fc.w16; FBI[g] = nets[g]->fc.b; }
    TRY(avlen_i_audio3_fwd(nets, x, nullptr, G, B, H, W, co, st));
    return avlen_gemm_bf16_grouped(FA, n->fc.ld16, FB, n->fc.ld16, outs, ld_out, FBI, G, B, n->fc.out_f, n->fc.ld16, AVLEN_ACT_RELU,
                                   gws, mx, st, &go);
  }
  bool sp = true;
  for (int g = 0; g < G; g++) sp = sp && cnn3_superpixel(nets[g], W);
  const int wsp = sp ? ((ow[0] - 1) * n->conv[0].stride + n->conv[0].kw) / n->conv[0].stride : 0;   // super-pixels per row
  if (sp) TRY(avlen_cast_h16(x, W * n->conv[0].cin, x16, wsp * 8, (long)B * H, wsp * 8, fmt, st));     // drop the unused columns
  else TRY(avlen_cast_h16(x, n->conv[0].cin, x16, 8, (long)B * H * W, n->conv[0].cin, fmt, st));         // channel-pad to 8
  const void* X[8]; const void* Wt[8]; void* Y[8]; const float* BI[8];
  int h = H, wd = W;
  for (int i = 0; i < 3; i++) {
    const avlen_conv& k = n->conv[i];
    const bool spi = sp && i == 0;
    for (int g = 0; g < G; g++) {
      X[g] = i == 0 ? (const void*)x16 : (const void*)a[i - 1][g];
      Wt[g] = spi ? nets[g]->conv[i].w16c : nets[g]->conv[i].w16; Y[g] = a[i][g]; BI[g] = nets[g]->conv[i].b;
    }
    TRY(avlen_conv2d_nhwc_bf16_grouped(X, Wt, nullptr, Y, nullptr, G, B, h, spi ? wsp : wd, spi ? 8 : k.cin16, k.cout, k.kh,
                                       spi ? k.kw / k.stride : k.kw, k.stride, 0, gws, mx, st, BI,
                                       i < 2 ? AVLEN_ACT_RELU : AVLEN_ACT_NONE, spi ? 1 : 0, &go));
    h = oh[i]; wd = ow[i];
  }
  const void* FA[8]; const void* FB[8]; const float* FBI[8];
  for (int g = 0; g < G; g++) { FA[g] = a[2][g]; FB[g] = nets[g]->fc.w16; FBI[g] = nets[g]->fc.b; }
  return avlen_gemm_bf16_grouped(FA, n->fc.ld16, FB, n->fc.ld16, outs, ld_out, FBI, G, B, n->fc.out_f, n->fc.ld16, AVLEN_ACT_RELU,
                                 gws, mx, st, &go);
}

}  // namespace

extern "C" size_t avlen_cnn3_group_workspace_bytes(const avlen_cnn3* n, int groups, int B, int H, int W) {
  return (size_t)groups * cnn3_ws_bf16(n, B, H, W) + 4096;
}

extern "C" int avlen_cnn3_group_fwd(const avlen_cnn3* const* nets, const float* x, int groups, int B, int H, int W,
                                    float* const* outs, int ld_out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!nets || groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  const bool f16 = nets[0]->half_fmt == 1;              // the format of the nets' 16-bit shadows selects bf16 / fp16 arithmetic
  for (int g = 0; g < groups; g++) {
    if (!cnn3_has16(nets[g])) return AVLEN_ERR_ARG;
    for (int i = 0; i < 3; i++) {
      const avlen_conv &a = nets[g]->conv[i], &b = nets[0]->conv[i];
      if (a.cin != b.cin || a.cout != b.cout || a.kh != b.kh || a.kw != b.kw || a.stride != b.stride) return AVLEN_ERR_ARG;
    }
  }
  return cnn3_group_fwd_bf16(nets, x, groups, B, H, W, outs, ld_out, ws, ws_bytes, st, f16);
}

// =====================================================================================================
// ResNet-18 tower
// =====================================================================================================
extern "C" size_t avlen_resnet18_workspace_bytes(int B) {
  size_t act = (size_t)B * 64 * 64 * 16 * sizeof(float);
  size_t v1 = 5 * (act + 256) + avlen_groupnorm_workspace_bytes(B, 128) + GEMM_SCRATCH + 4096;
  return zmax(v1, resnet18_ws_bf16(B));
}

extern "C" size_t avlen_resnet18_group_workspace_bytes(int groups, int B) { return (size_t)groups * resnet18_ws_bf16(B) + 4096; }

extern "C" int avlen_resnet18_group_fwd(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                                        const float* divisors, float* const* outs, int ld_out, int groups, int B, int S,
                                        void* ws, size_t ws_bytes, hipStream_t st) {
  if (!nets || groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  return resnet18_group_fwd_bf16(nets, imgs, img_u8, channels, divisors, outs, ld_out, groups, B, S, ws, ws_bytes, st);
}

extern "C" int avlen_resnet18_group_fwd_indexed(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                                                const float* divisors, float* const* outs, int ld_out, int groups, int B, int S,
                                                const int32_t* row_index, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!nets || groups < 1 || groups > 8 || B <= 0) return AVLEN_ERR_ARG;
  return resnet18_group_fwd_bf16(nets, imgs, img_u8, channels, divisors, outs, ld_out, groups, B, S, ws, ws_bytes, st, row_index);
}

// ---- compensated bf16 (AVLEN_PREC_BF16X3): `groups` towers in lock-step on tower_x3.hip, then fc (8192 -> 64) on the
// compensated staged GEMM.  row_index (optional): image b of the batch is image row_index[b] of imgs[g].
extern "C" size_t avlen_resnet18_group_x3_workspace_bytes(int groups, int B) {
  return avlen_tower_x3_workspace_bytes(groups, B) + (size_t)groups * ((size_t)B * 8192 * sizeof(float) + 256) + GEMM_SCRATCH + 4096;
}
// phase: 1 = the towers only (layer-4 outputs stay in the workspace), 2 = the fc only (on the workspace the same call with phase 1
// filled), 3 = both.  The step sequencer replays the two as separate graph pieces: the fc beside the AudioCNNs, not in front of them.
extern "C" int avlen_resnet18_group_fwd_x3_phase(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8,
                                                 const int* channels, const float* divisors, float* const* outs, int ld_out, int groups,
                                                 int B, int S, const int32_t* row_index, int phase, void* ws, size_t ws_bytes,
                                                 hipStream_t st) {
  if (!nets || groups < 1 || groups > 8 || B <= 0 || phase < 1 || phase > 3) return AVLEN_ERR_ARG;
  if (ws_bytes < avlen_resnet18_group_x3_workspace_bytes(groups, B)) return AVLEN_ERR_WS;
  for (int g = 0; g < groups; g++)
    if (!avlen_tower_x3_supported(nets[g], S, channels[g])) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes);
  void* Y[8];
  for (int g = 0; g < groups; g++) Y[g] = w.take<bf16>((size_t)2 * B * 8192);           // hi plane, lo plane
  void* gws = w.take<char>(GEMM_SCRATCH);
  const size_t tb = avlen_tower_x3_workspace_bytes(groups, B);
  void* tws = w.take<char>(tb);
  if (phase & 1) TRY(avlen_tower_x3_fwd(nets, imgs, img_u8, channels, divisors, row_index, Y, groups, B, S, tws, tb, st));
  if (!(phase & 2)) return AVLEN_OK;
  // fc (8192 -> 64) of all towers: one grouped compensated GEMM (the weights' low planes lie at one common distance)
  const void* FA[8]; const void* FB[8]; const float* FBI[8];
  const avlen_linear& fc = nets[0]->fc;
  avlen_g2_opts o; o.x3 = 1; o.a_lo = (long)B * 8192 * 2; o.b_lo = (const char*)fc.w16lo - (const char*)fc.w16;
  for (int g = 0; g < groups; g++) {
    const avlen_linear& f = nets[g]->fc;
    if (!f.w16 || !f.w16lo || f.ld16 != 8192 || (const char*)f.w16lo - (const char*)f.w16 != o.b_lo) return AVLEN_ERR_ARG;
    FA[g] = Y[g]; FB[g] = f.w16; FBI[g] = f.b;
  }
  return avlen_gemm_bf16_grouped(FA, 8192, FB, 8192, outs, ld_out, FBI, groups, B, fc.out_f, 8192, 0, gws, GEMM_SCRATCH, st, &o);
}

extern "C" int avlen_resnet18_group_fwd_x3(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8,
                                           const int* channels, const float* divisors, float* const* outs, int ld_out, int groups,
                                           int B, int S, const int32_t* row_index, void* ws, size_t ws_bytes, hipStream_t st) {
  return avlen_resnet18_group_fwd_x3_phase(nets, imgs, img_u8, channels, divisors, outs, ld_out, groups, B, S, row_index, 3, ws, ws_bytes, st);
}

extern "C" int avlen_resnet18_fwd(const avlen_resnet18* net, const void* img, int img_u8, int B, int S, int C, float divisor,
                                  float* out, int ld_out, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!net || B <= 0 || ws_bytes < avlen_resnet18_workspace_bytes(B)) return AVLEN_ERR_WS;
  if (prec == AVLEN_PREC_BF16 && resnet18_has16(net) && C <= 8)
    return resnet18_fwd_bf16(net, img, img_u8, B, S, C, divisor, out, ld_out, ws, ws_bytes, st);
  WsBump w(ws, ws_bytes);
  size_t act = (size_t)B * 64 * 64 * 16;
  float* x0 = w.take<float>(act);
  float* buf[4];
  for (int i = 0; i < 4; i++) buf[i] = w.take<float>(act);
  float* part = w.take<float>(avlen_groupnorm_workspace_bytes(B, 128) / sizeof(float));
  void* gws = w.take<char>(GEMM_SCRATCH);
  Ctx c{st, prec, gws, GEMM_SCRATCH};

  TRY(avlen_preprocess_image(img, img_u8, x0, B, S, C, divisor, st));
  const avlen_conv& c1 = net->conv1;
  TRY(avlen_conv2d_nhwc(x0, c1.w, nullptr, nullptr, buf[0], B, 64, 64, c1.cin, c1.cout, c1.kh, c1.kw, c1.stride, c1.pad,
                        0, prec, st));
  TRY(avlen_groupnorm_nhwc_ws(buf[0], net->bn1.g, net->bn1.b, nullptr, buf[0], B, 4096, 16, 16, 1, 1e-5f, part, st));
  float* cur = buf[0]; float* t1 = buf[1]; float* t2 = buf[2]; float* t3 = buf[3];
  int H = 64, Cc = 16;
  for (int i = 0; i < 8; i++) {
    const avlen_resblock& k = net->block[i];
    int s = k.conv1.stride, OH = (H + 2 - 3) / s + 1, Co = k.conv1.cout;
    TRY(avlen_conv2d_nhwc(cur, k.conv1.w, nullptr, nullptr, t1, B, H, H, Cc, Co, 3, 3, s, 1, 0, prec, st));
    TRY(avlen_groupnorm_nhwc_ws(t1, k.bn1.g, k.bn1.b, nullptr, t1, B, OH * OH, Co, 16, 1, 1e-5f, part, st));
    TRY(avlen_conv2d_nhwc(t1, k.conv2.w, nullptr, nullptr, t2, B, OH, OH, Co, Co, 3, 3, 1, 1, 0, prec, st));
    const float* idt = cur;
    if (k.has_down) {
      TRY(avlen_conv2d_nhwc(cur, k.down.w, nullptr, nullptr, t3, B, H, H, Cc, Co, 1, 1, s, 0, 0, prec, st));
      TRY(avlen_groupnorm_nhwc_ws(t3, k.bnd.g, k.bnd.b, nullptr, t3, B, OH * OH, Co, 16, 0, 1e-5f, part, st));
      idt = t3;
    }
    TRY(avlen_groupnorm_nhwc_ws(t2, k.bn2.g, k.bn2.b, idt, t2, B, OH * OH, Co, 16, 1, 1e-5f, part, st));
    float* old = cur; cur = t2; t2 = old;
    H = OH; Cc = Co;
  }
  // flatten (NHWC order; fc.w was packed to match) + fc
  return linear(c, net->fc, cur, H * H * Cc, out, ld_out, B, 0, nullptr, 0);
}

// =====================================================================================================
// 3-conv CNNs
// =====================================================================================================
static void cnn3_dims(const avlen_cnn3* n, int H, int W, int oh[3], int ow[3]) {
  for (int i = 0; i < 3; i++) {
    oh[i] = (H - n->conv[i].kh) / n->conv[i].stride + 1;
    ow[i] = (W - n->conv[i].kw) / n->conv[i].stride + 1;
    H = oh[i]; W = ow[i];
  }
}
extern "C" size_t avlen_cnn3_workspace_bytes(const avlen_cnn3* n, int B, int H, int W) {
  int oh[3], ow[3]; cnn3_dims(n, H, W, oh, ow);
  size_t tot = 0;
  for (int i = 0; i < 3; i++) tot += (size_t)B * oh[i] * ow[i] * n->conv[i].cout * sizeof(float) + 256;
  return zmax(tot + GEMM_SCRATCH + 1024, cnn3_ws_bf16(n, B, H, W));
}
extern "C" int avlen_cnn3_fwd_indexed(const avlen_cnn3* n, const float* x, const int32_t* row_index, int B, int H, int W, float* out,
                                      int ld_out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!n || B <= 0 || ws_bytes < avlen_cnn3_workspace_bytes(n, B, H, W)) return AVLEN_ERR_WS;
  int oh[3], ow[3]; cnn3_dims(n, H, W, oh, ow);
  if (oh[2] <= 0 || ow[2] <= 0 || n->fc.in_f != oh[2] * ow[2] * n->conv[2].cout) return AVLEN_ERR_ARG;
  if (!(cnn3_has16(n) && n->conv[0].cin <= 8)) return AVLEN_ERR_ARG;          // 16-bit fast paths only (bf16, or fp16 shadows: half_fmt 1)
  return cnn3_fwd_bf16(n, x, B, H, W, out, ld_out, ws, ws_bytes, st, row_index, n->half_fmt == 1);
}

extern "C" int avlen_cnn3_fwd(const avlen_cnn3* n, const float* x, int B, int H, int W, float* out, int ld_out,
                              int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!n || B <= 0 || ws_bytes < avlen_cnn3_workspace_bytes(n, B, H, W)) return AVLEN_ERR_WS;
  int oh[3], ow[3]; cnn3_dims(n, H, W, oh, ow);
  if (oh[2] <= 0 || ow[2] <= 0 || n->fc.in_f != oh[2] * ow[2] * n->conv[2].cout) return AVLEN_ERR_ARG;
  if (prec == AVLEN_PREC_BF16 && n->half_fmt == 0 && cnn3_has16(n) && n->conv[0].cin <= 8)
    return cnn3_fwd_bf16(n, x, B, H, W, out, ld_out, ws, ws_bytes, st);
  if (prec == AVLEN_PREC_FP16) {
    if (n->half_fmt != 1 || !cnn3_has16(n) || n->conv[0].cin > 8) return AVLEN_ERR_ARG;     // no fp16 fallback
    return cnn3_fwd_bf16(n, x, B, H, W, out, ld_out, ws, ws_bytes, st, nullptr, true);
  }
  WsBump w(ws, ws_bytes);
  float* a[3];
  for (int i = 0; i < 3; i++) a[i] = w.take<float>((size_t)B * oh[i] * ow[i] * n->conv[i].cout);
  void* gws = w.take<char>(GEMM_SCRATCH);
  Ctx c{st, prec, gws, GEMM_SCRATCH};
  const float* cur = x; int h = H, wd = W;
  for (int i = 0; i < 3; i++) {
    const avlen_conv& k = n->conv[i];
    TRY(avlen_conv2d_nhwc(cur, k.w, k.b, nullptr, a[i], B, h, wd, k.cin, k.cout, k.kh, k.kw, k.stride, 0,
                          i < 2 ? AVLEN_ACT_RELU : AVLEN_ACT_NONE, prec, st));
    cur = a[i]; h = oh[i]; wd = ow[i];
  }
  return linear(c, n->fc, cur, n->fc.in_f, out, ld_out, B, AVLEN_ACT_RELU, nullptr, 0);
}

// =====================================================================================================
// Transformer (1 enc + 1 dec layer, post-norm) over batch-major rows
// =====================================================================================================
namespace {

struct TrWs {          // forward activations kept for backward (R = B*S rows, d model dim)
  float *QKV, *AO, *LSE, *T1, *X1, *F1, *T2, *X2, *MEM;
  float *m1, *r1, *m2, *r2, *me, *re;
  float *V0, *U1, *Y1, *Qc, *KVc, *AOc, *LSEc, *U2, *Y2, *G1, *U3, *Y3;
  float *md1, *rd1, *md2, *rd2, *md3, *rd3, *mf, *rf;
  float *cA, *cM, *cDM, *cDA, *cP;     // cross attention in memory space (cross1.hip): [B][H][d] x 4, probabilities [B][H][S]
};

void tr_layout(WsBump& w, TrWs& t, long B, long S, int d, int H, bool cto) {
  long R = B * S;
  t.QKV = w.take<float>(R * (cto ? d : 3 * d));
  t.AO = cto ? t.QKV : w.take<float>(R * d);
  t.LSE = w.take<float>(B * H * S);
  t.T1 = w.take<float>(R * d); t.X1 = w.take<float>(R * d); t.F1 = w.take<float>(R * d);
  t.T2 = w.take<float>(R * d); t.X2 = w.take<float>(R * d); t.MEM = w.take<float>(R * d);
  t.m1 = w.take<float>(R); t.r1 = w.take<float>(R); t.m2 = w.take<float>(R); t.r2 = w.take<float>(R);
  t.me = w.take<float>(R); t.re = w.take<float>(R);
  t.V0 = w.take<float>(B * d); t.U1 = w.take<float>(B * d); t.Y1 = w.take<float>(B * d); t.Qc = w.take<float>(B * d);
  t.KVc = w.take<float>(R * (cto ? d : 2 * d));
  t.AOc = cto ? t.KVc : w.take<float>(B * d);
  t.LSEc = w.take<float>(B * H);
  t.U2 = w.take<float>(B * d); t.Y2 = w.take<float>(B * d); t.G1 = w.take<float>(B * d);
  t.U3 = w.take<float>(B * d); t.Y3 = w.take<float>(B * d);
  t.md1 = w.take<float>(B); t.rd1 = w.take<float>(B); t.md2 = w.take<float>(B); t.rd2 = w.take<float>(B);
  t.md3 = w.take<float>(B); t.rd3 = w.take<float>(B); t.mf = w.take<float>(B); t.rf = w.take<float>(B);
  t.cA = t.cM = t.cDM = t.cDA = t.cP = nullptr;
  if (!cto) {
    t.cA = w.take<float>(B * H * d); t.cM = w.take<float>(B * H * d); t.cDM = w.take<float>(B * H * d); t.cDA = w.take<float>(B * H * d);
    t.cP = w.take<float>(B * H * S);
  }
}

// Encoder layer + final encoder norm (fp32-staged kernels; keeps every activation for backward).
// Z: [R, d] encoder input (batch-major), maskx: [B, S] (1 = valid)
int enc_fwd(const Ctx& c, const avlen_transformer& tr, TrWs& t, const float* Z, const float* maskx, int B, int S,
            bool cto) {
  const int d = tr.d, H = tr.nhead, D = d / H;
  const long R = (long)B * S;
  const float scale = 1.0f / sqrtf((float)D);
  const avlen_enc_layer& e = tr.enc;
  if (cto) {      // single valid key: softmax == 1, attention output == V(token)
    TRY(linear_rows(c, e.self_attn.in_proj, 2 * d, d, Z, d, t.QKV, d, (int)R, 0, nullptr, 0));
  } else {
    TRY(linear(c, e.self_attn.in_proj, Z, d, t.QKV, 3 * d, (int)R, 0, nullptr, 0));
    TRY(avlen_attention_fwd(t.QKV, 3 * d, t.QKV + d, 3 * d, t.QKV + 2 * d, 3 * d, t.AO, d, maskx, t.LSE, B, H, S, S, D,
                            0, scale, c.st));
  }
  TRY(linear(c, e.self_attn.out_proj, t.AO, d, t.T1, d, (int)R, 0, Z, d));
  TRY(avlen_layernorm_fwd(t.T1, nullptr, e.norm1.g, e.norm1.b, t.X1, t.m1, t.r1, (int)R, d, 1e-5f, c.st));
  TRY(linear(c, e.lin1, t.X1, d, t.F1, e.lin1.out_f, (int)R, AVLEN_ACT_RELU, nullptr, 0));
  TRY(linear(c, e.lin2, t.F1, e.lin1.out_f, t.T2, d, (int)R, 0, t.X1, d));
  TRY(avlen_layernorm_fwd(t.T2, nullptr, e.norm2.g, e.norm2.b, t.X2, t.m2, t.r2, (int)R, d, 1e-5f, c.st));
  TRY(avlen_layernorm_fwd(t.X2, nullptr, tr.enc_norm.g, tr.enc_norm.b, t.MEM, t.me, t.re, (int)R, d, 1e-5f, c.st));
  // K/V (cto: V only) projections of the encoder memory for the decoder's cross attention
  const avlen_dec_layer& q = tr.dec;
  if (cto) return linear_rows(c, q.cross_attn.in_proj, 2 * d, d, t.MEM, d, t.KVc, d, (int)R, 0, nullptr, 0);
  return linear_rows(c, q.cross_attn.in_proj, d, 2 * d, t.MEM, d, t.KVc, 2 * d, (int)R, 0, nullptr, 0);
}

// Decoder layer over ONE target token per sample, given t.KVc (K|V projections of the encoder memory) -- or, mem16 given (training at
// scale, cross1.hip), the encoder memory itself as 16-bit rows [B * S][d] (lo plane mem_lo elements behind; 0: none): the cross
// attention then runs in memory space and t.KVc is never read.
int dec_fwd(const Ctx& c, const avlen_transformer& tr, TrWs& t, const float* maskx, const float* tgt, float* out, int B,
            int S, bool cto, const void* mem16 = nullptr, long mem_lo = 0) {
  const int d = tr.d, H = tr.nhead, D = d / H;
  const float scale = 1.0f / sqrtf((float)D);
  const avlen_dec_layer& q = tr.dec;
  TRY(linear_rows(c, q.self_attn.in_proj, 2 * d, d, tgt, d, t.V0, d, B, 0, nullptr, 0));       // self-attn over 1 token
  TRY(linear(c, q.self_attn.out_proj, t.V0, d, t.U1, d, B, 0, tgt, d));
  TRY(avlen_layernorm_fwd(t.U1, nullptr, q.norm1.g, q.norm1.b, t.Y1, t.md1, t.rd1, B, d, 1e-5f, c.st));
  if (!cto) {
    TRY(linear_rows(c, q.cross_attn.in_proj, 0, d, t.Y1, d, t.Qc, d, B, 0, nullptr, 0));
    if (mem16) {
      const avlen_linear& win = q.cross_attn.in_proj;
      TRY(avlen_i_cross1_expand(t.Qc, d, win.w + (size_t)d * d, d, t.cA, B, c.st));                         // A = q_h W_k[h]
      TRY(avlen_i_cross1_fwd(t.cA, mem16, mem_lo, maskx, t.cP, t.cM, B, S, scale, c.st));
      TRY(avlen_i_cross1_reduce(t.cM, win.w + (size_t)2 * d * d, d, win.b ? win.b + 2 * d : nullptr, t.AOc, d, B, c.st));
    } else
      TRY(avlen_attention_fwd(t.Qc, d, t.KVc, 2 * d, t.KVc + d, 2 * d, t.AOc, d, maskx, t.LSEc, B, H, 1, S, D, 0, scale,
                              c.st));
  }
  TRY(linear(c, q.cross_attn.out_proj, t.AOc, d, t.U2, d, B, 0, t.Y1, d));
  TRY(avlen_layernorm_fwd(t.U2, nullptr, q.norm2.g, q.norm2.b, t.Y2, t.md2, t.rd2, B, d, 1e-5f, c.st));
  TRY(linear(c, q.lin1, t.Y2, d, t.G1, q.lin1.out_f, B, AVLEN_ACT_RELU, nullptr, 0));
  TRY(linear(c, q.lin2, t.G1, q.lin1.out_f, t.U3, d, B, 0, t.Y2, d));
  TRY(avlen_layernorm_fwd(t.U3, nullptr, q.norm3.g, q.norm3.b, t.Y3, t.md3, t.rd3, B, d, 1e-5f, c.st));
  TRY(avlen_layernorm_fwd(t.Y3, nullptr, tr.dec_norm.g, tr.dec_norm.b, out, t.mf, t.rf, B, d, 1e-5f, c.st));
  return AVLEN_OK;
}

int transformer_fwd(const Ctx& c, const avlen_transformer& tr, TrWs& t, const float* Z, const float* maskx,
                    const float* tgt, float* out, int B, int S, bool cto) {
  TRY(enc_fwd(c, tr, t, Z, maskx, B, S, cto));
  return dec_fwd(c, tr, t, maskx, tgt, out, B, S, cto);
}

struct TrBwdWs { float *dA, *dB, *dC, *dD, *dE, *delta; };     // dA: [R,3d]; dB..dE: [R,d]
// 16-bit activation planes of the training forward at scale ("big16": the 2nd-stage update, 722 k token rows): every producer emits the
// operand of the next product (hi plane, and the lo plane right behind it in compensated mode), the hi planes stay for the backward's
// weight gradients.  No per-product cast of an fp32 activation is left in the forward, none of X in the backward.
struct Big16 { bf16 *XF = nullptr, *H1 = nullptr, *Z = nullptr, *QKV = nullptr, *AO = nullptr, *X1 = nullptr, *F1 = nullptr, *MEM = nullptr;
               int ldxf = 0; long lo1 = 0; };      // lo1: elements from the hi to the lo plane of the [R][d] buffers (0: plain bf16)
int g_big16 = 1;                 // avlen_set_big16(0): fp32-staged training forward at every size (tests compare the two)
bool big16_enabled() { return g_big16 != 0; }
bool cross1_enabled() { return (g_big16 & 2) == 0; }       // avlen_set_big16(3): big16 with the K | V projection route (tests compare the two)

void trb_layout(WsBump& w, TrBwdWs& b, long B, long S, int d, int H) {
  long R = B * S;
  b.dA = w.take<float>(R * 3 * d); b.dB = w.take<float>(R * d); b.dC = w.take<float>(R * d);
  b.dD = w.take<float>(R * d); b.dE = w.take<float>(R * d); b.delta = w.take<float>(B * H * S);
}

// Gradients w.r.t. the transformer parameters (accumulated into g) and w.r.t. Z (written to dZ [R,d]).
int transformer_bwd(const Ctx& c, const avlen_transformer& tr, const avlen_transformer& g, TrWs& t, TrBwdWs& s,
                    const float* Z, const float* maskx, const float* tgt, const float* d_out, float* dZ, int B, int S,
                    bool cto, const Big16* h16 = nullptr) {
  const int d = tr.d, H = tr.nhead, D = d / H;
  const long R = (long)B * S;
  const float scale = 1.0f / sqrtf((float)D);
  const avlen_dec_layer& q = tr.dec; const avlen_dec_layer& gq = g.dec;
  float* dY3 = s.dB; float* dU3 = s.dC;
  // ---- decoder: final LN, norm3, FFN
  TRY(avlen_layernorm_bwd(d_out, t.Y3, tr.dec_norm.g, t.mf, t.rf, dY3, g.dec_norm.g, g.dec_norm.b, B, d, c.st));
  TRY(avlen_layernorm_bwd(dY3, t.U3, q.norm3.g, t.md3, t.rd3, dU3, gq.norm3.g, gq.norm3.b, B, d, c.st));
  float* dG1 = s.dD; float* dY2 = s.dE;
  TRY(linear_bwd(c, q.lin2, gq.lin2, dU3, d, t.G1, q.lin1.out_f, dG1, q.lin1.out_f, B, nullptr, 0));
  TRY(relu_bwd(c, dG1, t.G1, (long)B * q.lin1.out_f));
  TRY(linear_bwd(c, q.lin1, gq.lin1, dG1, q.lin1.out_f, t.Y2, d, dY2, d, B, dU3, d));               // dY2 = dU3 + dG1 W1
  // ---- decoder: norm2, cross attention
  float* dU2 = s.dB;
  TRY(avlen_layernorm_bwd(dY2, t.U2, q.norm2.g, t.md2, t.rd2, dU2, gq.norm2.g, gq.norm2.b, B, d, c.st));
  float* dAOc = s.dC;
  TRY(linear_bwd(c, q.cross_attn.out_proj, gq.cross_attn.out_proj, dU2, d, t.AOc, d, dAOc, d, B, nullptr, 0));
  float* dMEM = s.dD;                    // [R, d]
  float* dY1 = s.dE;                     // [B, d]
  avlen_linear gin = gq.cross_attn.in_proj; const avlen_linear& win = q.cross_attn.in_proj;
  if (cto) {
    // AOc == Vc(MEM): dVc = dAOc ; q,k projections get exactly zero gradient (softmax over one key)
    avlen_linear gv = gin; gv.w = gin.w + (size_t)2 * d * d; gv.b = gin.b + 2 * d; gv.out_f = d;
    avlen_linear wv = win; wv.w = win.w + (size_t)2 * d * d; wv.out_f = d;
    TRY(linear_bwd(c, wv, gv, dAOc, d, t.MEM, d, dMEM, d, (int)R, nullptr, 0));
    TRY(avlen_copy_rows(dU2, d, dY1, d, B, d, c.st));                          // dY1 = dU2 (residual)
  } else if (h16 && avlen_i_cross1_ok(d, H, S) && cross1_enabled()) {
    // cross attention in memory space (cross1.hip): no K | V rows, no gradient rows of them, no backward of their projection
    const float* Wk = win.w + (size_t)d * d; const float* Wv = win.w + (size_t)2 * d * d;
    float* dQc = s.dA;                   // [B, d]
    if (gin.b) TRY(colsum_acc(c, dAOc, d, gin.b + 2 * d, B, d));                                    // db_v (db_k = 0)
    TRY(avlen_i_cross1_dw(dAOc, d, t.cM, gin.w + (size_t)2 * d * d, d, B, c.st));                      // dW_v[h] += dout_h^T m_h
    TRY(avlen_i_cross1_expand(dAOc, d, Wv, d, t.cDM, B, c.st));                                        // dm = dout_h W_v[h]
    // (the mixed backward reads the memory rows as their hi plane: bf16 operands, as every other product of this backward)
    TRY(avlen_i_cross1_bwd(t.cP, t.cDM, t.cA, h16->MEM, c.prec == AVLEN_PREC_BF16 ? 0L : h16->lo1, t.cDA, dMEM, B, S, scale, c.st));
    TRY(avlen_i_cross1_reduce(t.cDA, Wk, d, nullptr, dQc, d, B, c.st));                                // dq_h = dA W_k[h]^T
    TRY(avlen_i_cross1_dw(t.Qc, d, t.cDA, gin.w + (size_t)d * d, d, B, c.st));                         // dW_k[h] += q_h^T dA
    avlen_linear gqp = gin; gqp.out_f = d;
    avlen_linear wqp = win; wqp.out_f = d;
    TRY(linear_bwd(c, wqp, gqp, dQc, d, t.Y1, d, dY1, d, B, dU2, d));                         // dY1 = dU2 + dQc Wq
  } else {
    float* dQc = s.dA;                   // [B, d]
    float* dKVc = s.dA + (size_t)B * d;   // [R, 2d]
    TRY(avlen_attention_bwd(t.Qc, d, t.KVc, 2 * d, t.KVc + d, 2 * d, t.AOc, d, dAOc, d, maskx, t.LSEc, s.delta, dQc, d,
                            dKVc, 2 * d, dKVc + d, 2 * d, B, H, 1, S, D, 0, scale, c.st));
    avlen_linear gqp = gin; gqp.out_f = d;
    avlen_linear wqp = win; wqp.out_f = d;
    TRY(linear_bwd(c, wqp, gqp, dQc, d, t.Y1, d, dY1, d, B, dU2, d));                         // dY1 = dU2 + dQc Wq
    avlen_linear gkv = gin; gkv.w = gin.w + (size_t)d * d; gkv.b = gin.b + d; gkv.out_f = 2 * d;
    avlen_linear wkv = win; wkv.w = win.w + (size_t)d * d; wkv.out_f = 2 * d;
    TRY(linear_bwd(c, wkv, gkv, dKVc, 2 * d, t.MEM, d, dMEM, d, (int)R, nullptr, 0, h16 ? h16->MEM : nullptr, d));
  }
  // ---- decoder: norm1, self attention over the single target token
  float* dU1 = s.dB;
  TRY(avlen_layernorm_bwd(dY1, t.U1, q.norm1.g, t.md1, t.rd1, dU1, gq.norm1.g, gq.norm1.b, B, d, c.st));
  float* dV0 = s.dC;
  TRY(linear_bwd(c, q.self_attn.out_proj, gq.self_attn.out_proj, dU1, d, t.V0, d, dV0, d, B, nullptr, 0));
  {
    avlen_linear gv = gq.self_attn.in_proj; gv.w += (size_t)2 * d * d; gv.b += 2 * d; gv.out_f = d;
    TRY(linear_bwd(c, gv, gv, dV0, d, tgt, d, nullptr, 0, B, nullptr, 0));
  }
  // ---- encoder: final norm, norm2, FFN
  const avlen_enc_layer& e = tr.enc; const avlen_enc_layer& ge = g.enc;
  float* dX2 = s.dB; float* dT2 = s.dC;
  TRY(avlen_layernorm_bwd(dMEM, t.X2, tr.enc_norm.g, t.me, t.re, dX2, g.enc_norm.g, g.enc_norm.b, (int)R, d, c.st));
  // (h16: the LayerNorm backward leaves its dx also as the next product's bf16 operand + bias gradient -- no cast pass over dT2 / dT1)
  const bool lnf = h16 != nullptr && c.prec == AVLEN_PREC_BF16 && d % 8 == 0;
  TRY(avlen_layernorm_bwd16(dX2, t.T2, e.norm2.g, t.m2, t.r2, dT2, ge.norm2.g, ge.norm2.b, (int)R, d, c.st, lnf ? dy16_slot(c) : nullptr,
                            lnf ? ge.lin2.b : nullptr));
  float* dF1 = s.dD; float* dX1 = s.dE;
  TRY(linear_bwd(c, e.lin2, ge.lin2, dT2, d, t.F1, e.lin1.out_f, dF1, e.lin1.out_f, (int)R, nullptr, 0, h16 ? h16->F1 : nullptr,
                 e.lin1.out_f, nullptr, 0, lnf));
  if (!h16) TRY(relu_bwd(c, dF1, t.F1, R * e.lin1.out_f));          // (h16: relu' is applied where dF1 is cast for its two products)
  TRY(linear_bwd(c, e.lin1, ge.lin1, dF1, e.lin1.out_f, t.X1, d, dX1, d, (int)R, dT2, d, h16 ? h16->X1 : nullptr, d,
                 h16 ? h16->F1 : nullptr, e.lin1.out_f));                                            // dX1 = dT2 + dF1 W1
  // ---- encoder: norm1, self attention
  float* dT1 = s.dB;
  TRY(avlen_layernorm_bwd16(dX1, t.T1, e.norm1.g, t.m1, t.r1, dT1, ge.norm1.g, ge.norm1.b, (int)R, d, c.st, lnf ? dy16_slot(c) : nullptr,
                            lnf ? ge.self_attn.out_proj.b : nullptr));
  float* dAO = s.dC;
  TRY(linear_bwd(c, e.self_attn.out_proj, ge.self_attn.out_proj, dT1, d, t.AO, d, dAO, d, (int)R, nullptr, 0, h16 ? h16->AO : nullptr, d,
                 nullptr, 0, lnf));
  if (cto) {
    avlen_linear gv = ge.self_attn.in_proj; gv.w += (size_t)2 * d * d; gv.b += 2 * d; gv.out_f = d;
    avlen_linear wv = e.self_attn.in_proj; wv.w += (size_t)2 * d * d; wv.out_f = d;
    TRY(linear_bwd(c, wv, gv, dAO, d, Z, d, dZ, d, (int)R, dT1, d));                         // dZ = dT1 + dV Wv
  } else {
    float* dQKV = s.dA;
    // bf16 mode: the matrix-core backward (P / dS in registers); shapes outside its envelope take the fp32 kernels
    int rc = AVLEN_ERR_ARG;
    if (h16) {
      if (c.prec != AVLEN_PREC_BF16) return AVLEN_ERR_ARG;
      // the 3d-wide gradient goes straight into the in-projection's operand slot as bf16 rows (+ its bias gradient): no fp32 copy
      TRY(avlen_attention_bwd_p16(nullptr, 0, nullptr, 0, nullptr, 0, t.AO, d, dAO, d, maskx, t.LSE, s.delta, nullptr, 0, nullptr, 0,
                                  nullptr, 0, B, H, S, S, D, 0, scale, c.st, h16->QKV, 3 * d, dy16_slot(c), 3 * d, ge.self_attn.in_proj.b));
      return linear_bwd(c, e.self_attn.in_proj, ge.self_attn.in_proj, nullptr, 3 * d, Z, d, dZ, d, (int)R, dT1, d, h16->Z, d, nullptr, 0,
                        true);                                                                      // dZ = dT1 + dQKV Win
    } else if (c.prec == AVLEN_PREC_BF16 && attn_bwd16_on())
      rc = avlen_attention_bwd_bf16(t.QKV, 3 * d, t.QKV + d, 3 * d, t.QKV + 2 * d, 3 * d, t.AO, d, dAO, d, maskx, t.LSE, s.delta,
                                    dQKV, 3 * d, dQKV + d, 3 * d, dQKV + 2 * d, 3 * d, B, H, S, S, D, 0, scale, c.st);
    if (rc == AVLEN_ERR_ARG)
      rc = avlen_attention_bwd(t.QKV, 3 * d, t.QKV + d, 3 * d, t.QKV + 2 * d, 3 * d, t.AO, d, dAO, d, maskx, t.LSE, s.delta,
                               dQKV, 3 * d, dQKV + d, 3 * d, dQKV + 2 * d, 3 * d, B, H, S, S, D, 0, scale, c.st);
    TRY(rc);
    TRY(linear_bwd(c, e.self_attn.in_proj, ge.self_attn.in_proj, dQKV, 3 * d, Z, d, dZ, d, (int)R, dT1, d, h16 ? h16->Z : nullptr, d));   // dZ = dT1 + dQKV Win
  }
  return AVLEN_OK;
}

struct SmtWs { float *XF, *FMT, *maskx, *H1, *Z; TrWs tr; TrBwdWs tb; float *dZ, *dH1, *dPE, *dXc; void* gws; int ldxf;
               void* xs; size_t xs_bytes; Big16 h; };

void smt_layout(WsBump& w, SmtWs& s, const avlen_smt* p, long B, long M, int F, bool cto) {
  long S = cto ? 1 : M + 1, R = B * S;
  int d = p->tr.d;
  s.ldxf = (int)align_up(F + 12, 4);
  s.XF = w.take<float>(R * s.ldxf); s.FMT = w.take<float>(R * 8); s.maskx = w.take<float>(B * S);
  s.H1 = w.take<float>(R * d); s.Z = w.take<float>(R * d);
  tr_layout(w, s.tr, B, S, d, p->tr.nhead, cto);
  trb_layout(w, s.tb, B, S, d, p->tr.nhead);
  s.dZ = w.take<float>(R * d); s.dH1 = w.take<float>(R * d); s.dPE = w.take<float>(R * 16);
  s.dXc = w.take<float>(B * s.ldxf);             // gradient w.r.t. the current token's fusion input (update_dialog)
  s.gws = w.take<char>(GEMM_SCRATCH);
  s.xs = nullptr; s.xs_bytes = 0;
  if (R >= big_m()) {            // operand scratch of the large-M bf16 products: two operands of up to 3d + ldxf columns
    s.xs_bytes = 2 * ((size_t)(R + 8) * (size_t)(4 * d + s.ldxf + 16) * 2 + ((size_t)4 << 20));     // x 2: hi + lo planes (bf16x3)
    s.xs = w.take<char>(s.xs_bytes);
    if (!cto) {                    // big16 planes (hi + lo each)
      const int ff = p->tr.enc.lin1.out_f;
      s.h.ldxf = pad8(F + 12);
      s.h.XF = w.take<bf16>(2 * R * s.h.ldxf); s.h.H1 = w.take<bf16>(2 * R * d); s.h.Z = w.take<bf16>(2 * R * d);
      s.h.QKV = w.take<bf16>(2 * R * 3 * d); s.h.AO = w.take<bf16>(2 * R * d); s.h.X1 = w.take<bf16>(2 * R * d);
      s.h.F1 = w.take<bf16>(2 * R * ff); s.h.MEM = w.take<bf16>(2 * R * d);
    }
  }
}
// both directions decide alike: plain bf16, or the compensated mode where its backward runs on plain operands (R >= g_mixed_rows)
bool big16_on(const avlen_smt* p, const SmtWs& s, int prec, long R, int S, bool cto) {
  const int d = p->tr.d, H = p->tr.nhead;
  if (cto || !s.xs || !s.h.XF || !big16_enabled() || R < big_m() || !tn_dw_on()) return false;
  if (!(prec == AVLEN_PREC_BF16 || (prec == AVLEN_PREC_BF16X3 && g_mixed_rows > 0 && R >= g_mixed_rows))) return false;
  return H > 0 && d / H == 32 && d % 8 == 0 && S <= 320 && (d == 256 || d == 512 || d == 128 || d == 64) && p->fus0.out_f == d &&
         p->tr.enc.lin1.out_f % 8 == 0;
}
// Y = act(X16 W[r0 : r0 + n]^T + b) + res from 16-bit activation planes (xlo / ylo: ELEMENTS from the hi to the lo plane, compensated
// mode) -> fp32 (Y32) and / or planes (Y16).  The weights are cast here, per call: they change with every optimiser step.
int linear16t(const Ctx& c, const avlen_linear& L, int r0, int n, const bf16* X16, int ldx, long xlo, float* Y32, int ld32, bf16* Y16,
              int ld16, long ylo, long M, int act, const float* res, int ldr) {
  const int Kp = pad8(L.in_f);
  const int np = c.prec == AVLEN_PREC_BF16X3 ? 2 : 1;
  if (ldx < Kp || (np > 1 && (!xlo || (Y16 && !ylo)))) return AVLEN_ERR_ARG;
  XsBump b(c);
  bf16* W16 = b.take((size_t)np * n * Kp);
  if (!b.good) return AVLEN_ERR_WS;
  const long wl = np > 1 ? (long)n * Kp : 0;
  TRY(cast_pair(c, L.w + (size_t)r0 * L.in_f, L.in_f, W16, Kp, n, L.in_f, wl));
  avlen_g2_opts o;
  if (np > 1) { o.x3 = 1; o.a_lo = xlo * 2; o.b_lo = wl * 2; o.c16_lo = ylo; }
  return avlen_gemm_bf16_dyn(X16, ldx, W16, Kp, Y32, ld32, Y16, ld16, L.b ? L.b + r0 : nullptr, res, ldr, (int)M, nullptr, n, Kp, act,
                             c.gws, c.gws_bytes, c.st, &o);
}
// The training forward at scale on 16-bit activation planes: same products, same formats as the fp32-staged forward (an operand plane
// is the cast of the fp32 value either way), the self attention on the matrix cores (attn_smt16_kernel; it also leaves the fp32
// output and the log-sum-exp the backward reads).  Fills what transformer_bwd / avlen_smt_bwd read: QKV, AO, LSE, T1, T2, X2, KVc, the
// LayerNorm statistics and the planes of s.h.
int smt_fwd_big16(const Ctx& c, const avlen_smt* p, SmtWs& s, const float* goal, float* out, int B, int S) {
  const avlen_transformer& tr = p->tr; const avlen_enc_layer& e = tr.enc;
  const int d = tr.d, H = tr.nhead, ff = e.lin1.out_f;
  const long R = (long)B * S;
  const float scale = 1.0f / sqrtf((float)(d / H));
  const bool x3 = c.prec == AVLEN_PREC_BF16X3;
  Big16& h = s.h; TrWs& t = s.tr;
  const long lx = x3 ? R * h.ldxf : 0, l1 = x3 ? R * d : 0, l3 = x3 ? R * 3 * d : 0, lf = x3 ? R * ff : 0;      // hi -> lo plane
  TRY(linear16t(c, p->fus0, 0, d, h.XF, h.ldxf, lx, nullptr, 0, h.H1, d, l1, R, AVLEN_ACT_RELU, nullptr, 0));
  TRY(linear16t(c, p->fus2, 0, d, h.H1, d, l1, s.Z, d, h.Z, d, l1, R, 0, nullptr, 0));
  TRY(linear16t(c, e.self_attn.in_proj, 0, 3 * d, h.Z, d, l1, nullptr, 0, h.QKV, 3 * d, l3, R, 0, nullptr, 0));   // (no fp32 q | k | v: the backward
  // stages the hi plane -- the bf16 values it would round the fp32 ones to)
  TRY(avlen_attention_smt16(h.QKV, 3 * d, h.AO, d, B, H, S, scale, s.maskx, nullptr, c.st, l3, l1, t.AO, d, t.LSE));
  TRY(linear16t(c, e.self_attn.out_proj, 0, d, h.AO, d, l1, t.T1, d, nullptr, 0, 0, R, 0, s.Z, d));
  TRY(avlen_layernorm_fwd16_dyn(t.T1, nullptr, e.norm1.g, e.norm1.b, t.X1, h.X1, t.m1, t.r1, (int)R, nullptr, d, 1e-5f, c.st, l1));
  TRY(linear16t(c, e.lin1, 0, ff, h.X1, d, l1, nullptr, 0, h.F1, ff, lf, R, AVLEN_ACT_RELU, nullptr, 0));
  TRY(linear16t(c, e.lin2, 0, d, h.F1, ff, lf, t.T2, d, nullptr, 0, 0, R, 0, t.X1, d));
  TRY(avlen_layernorm_fwd(t.T2, nullptr, e.norm2.g, e.norm2.b, t.X2, t.m2, t.r2, (int)R, d, 1e-5f, c.st));
  TRY(avlen_layernorm_fwd16_dyn(t.X2, nullptr, tr.enc_norm.g, tr.enc_norm.b, nullptr, h.MEM, t.me, t.re, (int)R, nullptr, d, 1e-5f, c.st, l1));
  if (avlen_i_cross1_ok(d, H, S) && cross1_enabled())     // the decoder's single-query cross attention reads the memory rows themselves
    return dec_fwd(c, tr, t, s.maskx, goal, out, B, S, false, h.MEM, l1);
  TRY(linear16t(c, tr.dec.cross_attn.in_proj, d, 2 * d, h.MEM, d, l1, t.KVc, 2 * d, nullptr, 0, 0, R, 0, nullptr, 0));
  return dec_fwd(c, tr, t, s.maskx, goal, out, B, S, false);
}

}  // namespace

namespace {


bool tr_has16(const avlen_transformer& t) {
  return lin16_ok(t.enc.self_attn.in_proj) && lin16_ok(t.enc.self_attn.out_proj) && lin16_ok(t.enc.lin1) &&
         lin16_ok(t.enc.lin2) && lin16_ok(t.dec.cross_attn.in_proj) && lin16_ok(t.dec.cross_attn.out_proj) &&
         lin16_ok(t.dec.self_attn.in_proj) && lin16_ok(t.dec.self_attn.out_proj) && lin16_ok(t.dec.lin1) &&
         lin16_ok(t.dec.lin2);
}

// bf16-path transformer scratch (inference only)
struct Tr16Ws {
  bf16 *Z16, *AO16, *X116, *F116, *MEM16;            // [R, d]
  float *Z, *QKV, *T1, *X1, *T2, *X2, *KVc;          // [R, d] ([R,3d] QKV, [R,2d] KVc)
  bf16 *tgt16, *V016, *Y116, *AOc16, *Y216, *G116;   // [B, d]
  float *U1, *Y1, *Qc, *U2, *Y2, *U3, *Y3;           // [B, d]
};

void tr16_layout(WsBump& w, Tr16Ws& t, long B, long S, int d, bool cto) {
  long R = B * S;
  // every bf16 buffer has room for the low plane of a compensated pair right behind it (lo = hi + R*d / B*d elements)
  t.Z16 = w.take<bf16>(2 * R * d); t.AO16 = w.take<bf16>(2 * R * d); t.X116 = w.take<bf16>(2 * R * d); t.F116 = w.take<bf16>(2 * R * d);
  t.MEM16 = w.take<bf16>(2 * R * d);
  t.Z = w.take<float>(R * d); t.QKV = cto ? nullptr : w.take<float>(R * 3 * d); t.T1 = w.take<float>(R * d);
  t.X1 = w.take<float>(R * d); t.T2 = w.take<float>(R * d); t.X2 = w.take<float>(R * d);
  t.KVc = cto ? nullptr : w.take<float>(R * 2 * d);
  t.tgt16 = w.take<bf16>(2 * B * d); t.V016 = w.take<bf16>(2 * B * d); t.Y116 = w.take<bf16>(2 * B * d); t.AOc16 = w.take<bf16>(2 * B * d);
  t.Y216 = w.take<bf16>(2 * B * d); t.G116 = w.take<bf16>(2 * B * d);
  t.U1 = w.take<float>(B * d); t.Y1 = w.take<float>(B * d); t.Qc = w.take<float>(B * d); t.U2 = w.take<float>(B * d);
  t.Y2 = w.take<float>(B * d); t.U3 = w.take<float>(B * d); t.Y3 = w.take<float>(B * d);
}

// Encoder layer + final norm + K/V projection for the decoder; consumes t.Z (fp32) / t.Z16.
int enc_fwd16(const Ctx& c, const avlen_transformer& tr, Tr16Ws& t, const float* maskx, int B, int S, bool cto) {
  const int d = tr.d, H = tr.nhead, D = d / H;
  const long R = (long)B * S;
  const long lo = R * d;          // compensated mode: the low plane of every [R][d] bf16 buffer lies R*d elements behind it
  const float scale = 1.0f / sqrtf((float)D);
  const avlen_enc_layer& e = tr.enc;
  if (c.x3 && (cto || !(D == 32 && S <= 320))) return AVLEN_ERR_ARG;
  if (cto) {                // one valid key: attention output == V(token)
    TRY(linear16_rows(c, e.self_attn.in_proj, 2 * d, d, t.Z16, d, nullptr, 0, t.AO16, d, (int)R, 0));
  } else if (D == 32 && S <= 320) {    // packed bf16 q|k|v straight into the MFMA attention (ragged: live tokens only)
    // (compensated: the fp32-sized QKV buffer holds the hi plane [R][3d] followed by the lo plane)
    TRY(linear16(c, e.self_attn.in_proj, t.Z16, d, nullptr, 0, (bf16*)t.QKV, 3 * d, (int)R, 0, nullptr, 0, lo, 3 * lo));
    TRY(avlen_attention_smt16(t.QKV, 3 * d, t.AO16, d, B, H, S, scale, c.seg ? nullptr : maskx, c.seg, c.st, c.x3 ? 3 * lo : 0,
                              c.x3 ? lo : 0));
  } else {
    if (c.live) return AVLEN_ERR_ARG;
    TRY(linear16(c, e.self_attn.in_proj, t.Z16, d, t.QKV, 3 * d, nullptr, 0, (int)R, 0, nullptr, 0));
    TRY(avlen_attention_fwd16(t.QKV, 3 * d, t.QKV + d, 3 * d, t.QKV + 2 * d, 3 * d, nullptr, 0, t.AO16, d, maskx, nullptr, B,
                              H, S, S, D, 0, scale, c.st));
  }
  TRY(linear16(c, e.self_attn.out_proj, t.AO16, d, t.T1, d, nullptr, 0, (int)R, 0, t.Z, d, lo, 0));
  TRY(ln16(c, t.T1, e.norm1, t.X1, t.X116, (int)R, d, lo));
  TRY(linear16(c, e.lin1, t.X116, d, nullptr, 0, t.F116, d, (int)R, AVLEN_ACT_RELU, nullptr, 0, lo, lo));
  TRY(linear16(c, e.lin2, t.F116, d, t.T2, d, nullptr, 0, (int)R, 0, t.X1, d, lo, 0));
  TRY(ln16(c, t.T2, e.norm2, t.X2, nullptr, (int)R, d));
  TRY(ln16(c, t.X2, tr.enc_norm, nullptr, t.MEM16, (int)R, d, lo));
  if (cto)                  // cross attention over one valid key == V projection of that token
    return linear16_rows(c, tr.dec.cross_attn.in_proj, 2 * d, d, t.MEM16, d, nullptr, 0, t.AOc16, d, (int)R, 0);
  return linear16_rows(c, tr.dec.cross_attn.in_proj, d, 2 * d, t.MEM16, d, t.KVc, 2 * d, nullptr, 0, (int)R, 0, lo, 0);
}

// Decoder layer for one target token per sample on the bf16 path.
int dec_fwd16(const Ctx& c, const avlen_transformer& tr, Tr16Ws& t, const float* maskx, const float* tgt, float* out, int B,
              int S, bool cto) {
  const int d = tr.d, H = tr.nhead, D = d / H;
  const float scale = 1.0f / sqrtf((float)D);
  const avlen_dec_layer& q = tr.dec;
  if (c.x3 && (cto || d != 256 || !(D == 32 && S <= 320))) return AVLEN_ERR_ARG;
  if (!cto && d == 256 && (chain_enabled() || c.x3)) {          // two fused chains around the cross attention
    const long blo = (long)B * d;                     // low plane of the [B][d] bf16 buffers
    ChainB a(c.x3 != 0);
    a.load_cur(tgt, d, 0); a.save();
    a.linear(q.self_attn.in_proj, 2 * d, 0, 0, 0, 1);                // one target token: self attention == V projection
    a.linear(q.self_attn.out_proj, 0, 0, 1, 1, 0);
    a.ln(q.norm1, 0); a.store(t.Y1, d, nullptr, 0);
    a.linear(q.cross_attn.in_proj, 0, 0, 0, 0, 1); a.store(t.Qc, d, nullptr, 0);
    ChainB b(c.x3 != 0);
    b.load_cur(t.Y1, d, 0); b.save();
    b.load_x16(t.AOc16, d, d, 1, c.x3 ? t.AOc16 + blo : nullptr);
    b.linear(q.cross_attn.out_proj, 0, 0, 1, 1, 0);
    b.ln(q.norm2, 0); b.save();
    b.linear(q.lin1, 0, AVLEN_ACT_RELU, 0, 0, 1);
    b.linear(q.lin2, 0, 0, 1, 1, 0);
    b.ln(q.norm3, 0); b.ln(tr.dec_norm, 0); b.store(out, d, nullptr, 0);
    if (a.ok && b.ok && q.lin1.out_f == 256) {
      TRY(a.run(B, c.st));
      if (D == 32 && S <= 320)
        TRY(avlen_attention_q1(t.Qc, d, t.KVc, 2 * d, t.KVc + d, 2 * d, t.AOc16, d, B, H, S, scale, c.seg ? nullptr : maskx, c.seg,
                               c.st, c.x3 ? blo : 0));
      else
        TRY(avlen_attention_fwd16(t.Qc, d, t.KVc, 2 * d, t.KVc + d, 2 * d, nullptr, 0, t.AOc16, d, maskx, nullptr, B, H, 1, S, D,
                                  0, scale, c.st));
      return b.run(B, c.st);
    }
    if (c.x3) return AVLEN_ERR_ARG;
  }
  TRY(avlen_cast_bf16(tgt, d, t.tgt16, d, B, d, c.st));
  TRY(linear16_rows(c, q.self_attn.in_proj, 2 * d, d, t.tgt16, d, nullptr, 0, t.V016, d, B, 0));
  TRY(linear16(c, q.self_attn.out_proj, t.V016, d, t.U1, d, nullptr, 0, B, 0, tgt, d));
  TRY(avlen_layernorm_fwd16(t.U1, nullptr, q.norm1.g, q.norm1.b, t.Y1, t.Y116, nullptr, nullptr, B, d, 1e-5f, c.st));
  if (!cto) {
    TRY(linear16_rows(c, q.cross_attn.in_proj, 0, d, t.Y116, d, t.Qc, d, nullptr, 0, B, 0));
    TRY(avlen_attention_fwd16(t.Qc, d, t.KVc, 2 * d, t.KVc + d, 2 * d, nullptr, 0, t.AOc16, d, maskx, nullptr, B, H, 1, S, D, 0,
                              scale, c.st));
  }
  TRY(linear16(c, q.cross_attn.out_proj, t.AOc16, d, t.U2, d, nullptr, 0, B, 0, t.Y1, d));
  TRY(avlen_layernorm_fwd16(t.U2, nullptr, q.norm2.g, q.norm2.b, t.Y2, t.Y216, nullptr, nullptr, B, d, 1e-5f, c.st));
  TRY(linear16(c, q.lin1, t.Y216, d, nullptr, 0, t.G116, d, B, AVLEN_ACT_RELU, nullptr, 0));
  TRY(linear16(c, q.lin2, t.G116, d, t.U3, d, nullptr, 0, B, 0, t.Y2, d));
  TRY(avlen_layernorm_fwd16(t.U3, nullptr, q.norm3.g, q.norm3.b, t.Y3, nullptr, nullptr, nullptr, B, d, 1e-5f, c.st));
  return avlen_layernorm_fwd16(t.Y3, nullptr, tr.dec_norm.g, tr.dec_norm.b, out, nullptr, nullptr, nullptr, B, d, 1e-5f, c.st);
}

bool smt_has16(const avlen_smt* p) { return lin16_ok(p->fus0) && lin16_ok(p->fus2) && tr_has16(p->tr); }
bool tr_has16lo(const avlen_transformer& t) {
  return t.enc.self_attn.in_proj.w16lo && t.enc.self_attn.out_proj.w16lo && t.enc.lin1.w16lo && t.enc.lin2.w16lo &&
         t.dec.cross_attn.in_proj.w16lo && t.dec.cross_attn.out_proj.w16lo && t.dec.self_attn.in_proj.w16lo &&
         t.dec.self_attn.out_proj.w16lo && t.dec.lin1.w16lo && t.dec.lin2.w16lo;
}
bool smt_has16lo(const avlen_smt* p) { return smt_has16(p) && p->fus0.w16lo && p->fus2.w16lo && tr_has16lo(p->tr); }

struct Smt16Ws { bf16 *XF, *H1; float* maskx; Tr16Ws tr; void* gws; size_t gwsb; int ldxf; int *seg, *rowmap; };

void smt16_layout(WsBump& w, Smt16Ws& s, const avlen_smt* p, long B, long M, int F, bool cto) {
  long S = cto ? 1 : M + 1, R = B * S; int d = p->tr.d;
  (void)F;
  s.ldxf = p->fus0.ld16;
  s.XF = w.take<bf16>(2 * R * s.ldxf); s.H1 = w.take<bf16>(2 * R * d); s.maskx = w.take<float>(B * S);
  s.seg = w.take<int>(B + 1); s.rowmap = w.take<int>(R);
  tr16_layout(w, s.tr, B, S, d, cto);
  s.gwsb = zmax((size_t)(32u << 20), avlen_gemm_bf16_workspace_bytes(128, 768));
  s.gws = w.take<char>(s.gwsb);
}

size_t smt16_ws_bytes(const avlen_smt* p, int B, int M, int F, bool cto) {
  WsBump w(nullptr, 0); Smt16Ws s; smt16_layout(w, s, p, B, M, F, cto); return w.off + 4096;
}

// SMTStateEncoder forward, inference only, bf16 operands (all rollout calls of pi_g / pi_l / pi_q)
// x3: compensated bf16 -- every 16-bit operand is a (hi, lo) pair, the weights' low planes are avlen_linear::w16lo.  Returns
// AVLEN_NOT_BIG when the shape has no compensated fast path yet (caller: fp32-staged path with compensated products).
int smt_fwd_infer_bf16(const avlen_smt* p, const float* x, const float* memory, const int32_t* mem_index, int NC,
                       const float* masks, const float* goal, float* out, int B, int M, int F, int pose_col, bool cto,
                       void* ws, size_t ws_bytes, hipStream_t st, bool x3 = false) {
  if (ws_bytes < smt16_ws_bytes(p, B, M, F, cto)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes); Smt16Ws s; smt16_layout(w, s, p, B, M, F, cto);
  Ctx c{st, AVLEN_PREC_BF16, s.gws, s.gwsb};
  c.x3 = x3 ? 1 : 0;
  const avlen_transformer& tr = p->tr;
  const int S = cto ? 1 : M + 1, d = tr.d;
  const long R = (long)B * S;
  if (!mem_index) NC = B;
  // Ragged mode (memory history in use): only the live tokens -- valid memory slots + the current one -- go through the
  // fusion MLP, the encoder layer and the decoder's K/V projection; with a 150-slot window of a 300-slot ring that is half
  // of the rows.  Row counts live on the device (seg[B]); grids are sized for the maximum.
  const bool ragged = !cto && masks && M > 4 && d / tr.nhead == 32 && S <= 320 && ragged_enabled();
  const bool chain_cto = cto && d == 256 && chain_enabled();
  const bool chain_small = !cto && (S == 2 || S == 4) && d == 256 && s.ldxf <= 320 && chain_enabled();
  const bool general_x3 = !cto && d == 256 && d / tr.nhead == 32 && S <= 320 && p->fus0.out_f == 256;
  if (x3 && !chain_cto && !chain_small && !general_x3) return AVLEN_NOT_BIG;
  bf16* XFlo = x3 ? s.XF + R * s.ldxf : nullptr;
  if (ragged) {
    hipLaunchKernelGGL(smt_segments_kernel, dim3(1), dim3(1024), 0, st, masks, s.seg, s.rowmap, B, M);
    c.live = s.seg + B; c.seg = s.seg;
  }
  hipLaunchKernelGGL(smt_build16_kernel, dim3((unsigned)R), dim3(128), 0, st, x, memory, mem_index, NC, masks, p->pose.w,
                     p->pose.b, s.XF, s.ldxf, s.maskx, B, M, F, pose_col, cto ? 1 : 0, ragged ? s.seg : (const int*)nullptr,
                     ragged ? s.rowmap : (const int*)nullptr, XFlo);
  TRY(avlen_launch_status());
  if (chain_cto) {
    // `current_token_only`: every attention sees one valid key, so the whole encoder + decoder is a chain of row-wise
    // steps (attention output == V projection); one launch.
    const avlen_enc_layer& e = tr.enc; const avlen_dec_layer& q = tr.dec;
    ChainB ch(x3);
    ch.load_x16(s.XF, s.ldxf, s.ldxf, 0, XFlo);
    ch.linear(p->fus0, 0, AVLEN_ACT_RELU, 0, 0, 1);
    ch.linear(p->fus2, 0, 0, 0, 1, 0); ch.save();                     // Z
    ch.linear(e.self_attn.in_proj, 2 * d, 0, 0, 0, 1);               // V(Z)
    ch.linear(e.self_attn.out_proj, 0, 0, 1, 1, 0);
    ch.ln(e.norm1, 0); ch.save();                                    // X1
    ch.linear(e.lin1, 0, AVLEN_ACT_RELU, 0, 0, 1);
    ch.linear(e.lin2, 0, 0, 1, 1, 0);
    ch.ln(e.norm2, 0); ch.ln(tr.enc_norm, 0);                        // memory token
    ch.linear(q.cross_attn.in_proj, 2 * d, 0, 0, 0, 1); ch.save(1);  // cross attention output, parked in save slot 1
    ch.load_cur(goal, d, 0); ch.save();
    ch.linear(q.self_attn.in_proj, 2 * d, 0, 0, 0, 1);
    ch.linear(q.self_attn.out_proj, 0, 0, 1, 1, 0);
    ch.ln(q.norm1, 0); ch.save();                                    // Y1
    ch.recall(1, 1);
    ch.linear(q.cross_attn.out_proj, 0, 0, 1, 1, 0);
    ch.ln(q.norm2, 0); ch.save();                                    // Y2
    ch.linear(q.lin1, 0, AVLEN_ACT_RELU, 0, 0, 1);
    ch.linear(q.lin2, 0, 0, 1, 1, 0);
    ch.ln(q.norm3, 0); ch.ln(tr.dec_norm, 0); ch.store(out, d, nullptr, 0);
    if (ch.ok && e.lin1.out_f == 256 && q.lin1.out_f == 256 && p->fus0.out_f == 256) return ch.run(B, st);
    if (x3) return AVLEN_NOT_BIG;
  }
  if (chain_small) {
    // a short memory (pi_l: M = 3): the whole state encoder -- fusion MLP, encoder layer, decoder layer -- is one chain
    // launch; attention runs inside each sample's group of S rows
    ChainB ch(x3);
    ch.load_x16(s.XF, s.ldxf, s.ldxf, 0, XFlo);
    ch.linear(p->fus0, 0, AVLEN_ACT_RELU, 0, 0, 1);
    ch.linear(p->fus2, 0, 0, 0, 1, 0); ch.save();                     // Z
    ch.small_transformer(tr, goal, s.maskx, S, out);
    if (ch.ok && p->fus0.out_f == 256) return ch.run((int)R, st);
  }
  if (x3 && !general_x3) return AVLEN_NOT_BIG;
  TRY(linear16(c, p->fus0, s.XF, s.ldxf, nullptr, 0, s.H1, d, (int)R, AVLEN_ACT_RELU, nullptr, 0, R * s.ldxf, R * d));
  TRY(linear16(c, p->fus2, s.H1, d, s.tr.Z, d, s.tr.Z16, d, (int)R, 0, nullptr, 0, R * d, R * d));
  TRY(enc_fwd16(c, tr, s.tr, s.maskx, B, S, cto));
  return dec_fwd16(c, tr, s.tr, s.maskx, goal, out, B, S, cto);
}

// ---- dialog state encoder on the bf16 path ----
__global__ void dialog_build16_kernel(const float* __restrict__ x_att, const float* __restrict__ mem,
                                      const float* __restrict__ masks, const float* __restrict__ d_emb,
                                      bf16* __restrict__ seq16, float* __restrict__ seq32, int ldseq, float* __restrict__ maskx,
                                      int B, int M, int d, bf16* __restrict__ seq16lo) {
  const int S = M + 1, row = blockIdx.x, b = row / S, s = row % S;
  const float* src = s < M ? mem + ((long)s * B + b) * d : x_att + (long)b * d;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    if (seq16) {
      const float v = src[i]; const bf16 hv = (bf16)v;
      seq16[(long)row * ldseq + i] = hv;
      if (seq16lo) seq16lo[(long)row * ldseq + i] = (bf16)(v - (float)hv);
      if (d_emb) {
        const float e = d_emb[(long)b * d + i]; const bf16 he = (bf16)e;
        seq16[(long)row * ldseq + d + i] = he;
        if (seq16lo) seq16lo[(long)row * ldseq + d + i] = (bf16)(e - (float)he);
      }
    }
    if (seq32) seq32[(long)row * d + i] = src[i];
  }
  if (threadIdx.x == 0) maskx[(long)b * S + s] = s < M ? masks[(long)b * M + s] : 1.f;
}
// z[row] += pe[step[b]]  ->  fp32 (in place) and bf16 copy
__global__ void add_pe16_kernel(float* __restrict__ z, bf16* __restrict__ z16, const float* __restrict__ pe,
                                const float* __restrict__ step, int S, int d, int pe_len) {
  const int row = blockIdx.x, b = row / S;
  int idx = (int)step[b];
  idx = idx < 0 ? 0 : (idx >= pe_len ? pe_len - 1 : idx);
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    float v = z[(long)row * d + i] + pe[(long)idx * d + i];
    z[(long)row * d + i] = v; z16[(long)row * d + i] = (bf16)v;
  }
}

struct Dlg16Ws { bf16 *SEQ16, *H1; float* maskx; Tr16Ws tr; void* gws; size_t gwsb; };
void dlg16_layout(WsBump& w, Dlg16Ws& s, const avlen_dialog* p, long B, long M) {
  long S = M + 1, R = B * S; int d = p->tr.d;
  s.SEQ16 = w.take<bf16>(2 * R * 2 * d); s.H1 = w.take<bf16>(2 * R * d); s.maskx = w.take<float>(B * S);
  tr16_layout(w, s.tr, B, S, d, false);
  s.gwsb = 32u << 20; s.gws = w.take<char>(s.gwsb);
}
size_t dlg16_ws_bytes(const avlen_dialog* p, int B, int M) { WsBump w(nullptr, 0); Dlg16Ws s; dlg16_layout(w, s, p, B, M); return w.off + 4096; }
bool dlg_has16(const avlen_dialog* p) { return lin16_ok(p->fus0) && lin16_ok(p->fus2) && tr_has16(p->tr); }
bool dlg_has16lo(const avlen_dialog* p) { return dlg_has16(p) && p->fus0.w16lo && p->fus2.w16lo && tr_has16lo(p->tr); }

int dialog_fwd_bf16(const avlen_dialog* p, const float* x_att, const float* memory_state, const float* masks,
                    const float* d_emb, const float* agent_step, const float* goal, float* out, int B, int M, void* ws,
                    size_t ws_bytes, hipStream_t st, bool x3 = false) {
  if (ws_bytes < dlg16_ws_bytes(p, B, M)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes); Dlg16Ws s; dlg16_layout(w, s, p, B, M);
  Ctx c{st, AVLEN_PREC_BF16, s.gws, s.gwsb};
  c.x3 = x3 ? 1 : 0;
  const int S = M + 1, d = p->tr.d; const long R = (long)B * S;
  const bool small = (S == 2 || S == 4) && d == 256 && chain_enabled();
  if (x3 && !small) return AVLEN_NOT_BIG;
  if (d_emb) {
    const long seqlo = R * 2 * d, h1lo = R * d;
    hipLaunchKernelGGL(dialog_build16_kernel, dim3((unsigned)R), dim3(128), 0, st, x_att, memory_state, masks, d_emb, s.SEQ16,
                       (float*)nullptr, 2 * d, s.maskx, B, M, d, x3 ? s.SEQ16 + seqlo : (bf16*)nullptr);
    TRY(linear16(c, p->fus0, s.SEQ16, 2 * d, nullptr, 0, s.H1, d, (int)R, AVLEN_ACT_RELU, nullptr, 0, seqlo, h1lo));
    if (small) {                    // fusion output -> + positional row -> encoder + decoder: one chain launch
      ChainB ch(x3);
      ch.load_x16(s.H1, d, d, 0, x3 ? s.H1 + h1lo : nullptr);
      ch.linear(p->fus2, 0, 0, 0, 0, 1);
      ch.add_pe(p->pe, p->pe_len, agent_step, S, 0); ch.save();
      ch.small_transformer(p->tr, goal, s.maskx, S, out);
      if (ch.ok) return ch.run((int)R, st);
      if (x3) return AVLEN_ERR_ARG;
    }
    TRY(linear16(c, p->fus2, s.H1, d, s.tr.Z, d, nullptr, 0, (int)R, 0, nullptr, 0));
  } else {
    hipLaunchKernelGGL(dialog_build16_kernel, dim3((unsigned)R), dim3(128), 0, st, x_att, memory_state, masks,
                       (const float*)nullptr, (bf16*)nullptr, s.tr.Z, d, s.maskx, B, M, d, (bf16*)nullptr);
    if (small) {
      ChainB ch(x3);
      ch.load_cur(s.tr.Z, d, 0);
      ch.add_pe(p->pe, p->pe_len, agent_step, S, 0); ch.save();
      ch.small_transformer(p->tr, goal, s.maskx, S, out);
      if (ch.ok) { TRY(avlen_launch_status()); return ch.run((int)R, st); }
      if (x3) return AVLEN_ERR_ARG;
    }
  }
  hipLaunchKernelGGL(add_pe16_kernel, dim3((unsigned)R), dim3(128), 0, st, s.tr.Z, s.tr.Z16, p->pe, agent_step, S, d, p->pe_len);
  TRY(avlen_launch_status());
  TRY(enc_fwd16(c, p->tr, s.tr, s.maskx, B, S, false));
  return dec_fwd16(c, p->tr, s.tr, s.maskx, goal, out, B, S, false);
}

}  // namespace

extern "C" size_t avlen_smt_workspace_bytes(const avlen_smt* p, int B, int M, int F, int cto) {
  WsBump w(nullptr, 0); SmtWs s;
  smt_layout(w, s, p, B, M, F, cto != 0);
  size_t v1 = w.off + 4096;
  return smt_has16(p) ? zmax(v1, smt16_ws_bytes(p, B, M, F, cto != 0)) : v1;
}

extern "C" int avlen_smt_fwd(const avlen_smt* p, const float* x, const float* memory, const int32_t* mem_index, int NC,
                             const float* masks, const float* goal, float* out, int B, int M, int F, int pose_col,
                             int cto, int save_for_backward, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || B <= 0 || M < 0 || p->fus0.in_f != F + 12 || p->pose.in_f != 5 || p->pose.out_f != 16) return AVLEN_ERR_ARG;
  if (!cto && M > 0 && (!memory || !masks)) return AVLEN_ERR_ARG;
  if (ws_bytes < avlen_smt_workspace_bytes(p, B, M, F, cto)) return AVLEN_ERR_WS;
  if (prec == AVLEN_PREC_BF16 && !save_for_backward && (cto || M > 0) && smt_has16(p))
    return smt_fwd_infer_bf16(p, x, memory, mem_index, NC, masks, goal, out, B, M, F, pose_col, cto != 0, ws, ws_bytes, st);
  if (prec == AVLEN_PREC_BF16X3 && !save_for_backward && (cto || M > 0) && smt_has16lo(p)) {
    const int rc = smt_fwd_infer_bf16(p, x, memory, mem_index, NC, masks, goal, out, B, M, F, pose_col, cto != 0, ws, ws_bytes, st, true);
    if (rc != AVLEN_NOT_BIG) return rc;
  }
  WsBump w(ws, ws_bytes); SmtWs s;
  smt_layout(w, s, p, B, M, F, cto != 0);
  Ctx c{st, prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  const int S = cto ? 1 : M + 1, d = p->tr.d;
  const long R = (long)B * S;
  if (!mem_index) NC = B;
  s.h.lo1 = prec == AVLEN_PREC_BF16X3 ? R * d : 0;
  if (save_for_backward && big16_on(p, s, prec, R, S, cto != 0)) {       // the fusion input straight into its operand planes
    hipLaunchKernelGGL(smt_build16w_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, memory, mem_index, NC, masks, p->pose.w,
                       p->pose.b, s.FMT, s.maskx, R, B, M, F, pose_col, s.h.XF, s.h.ldxf, prec == AVLEN_PREC_BF16X3 ? R * s.h.ldxf : 0L);
    TRY(avlen_launch_status());
    return smt_fwd_big16(c, p, s, goal, out, B, S);
  }
  hipLaunchKernelGGL(smt_build_kernel, dim3((unsigned)R), dim3(128), 0, st, x, memory, mem_index, NC, masks, p->pose.w, p->pose.b, s.XF,
                     s.ldxf, s.FMT, s.maskx, B, M, F, pose_col, cto);
  TRY(avlen_launch_status());
  TRY(linear(c, p->fus0, s.XF, s.ldxf, s.H1, d, (int)R, AVLEN_ACT_RELU, nullptr, 0));
  TRY(linear(c, p->fus2, s.H1, d, s.Z, d, (int)R, 0, nullptr, 0));
  return transformer_fwd(c, p->tr, s.tr, s.Z, s.maskx, goal, out, B, S, cto != 0);
}

// d_x[b][0:pc] = dXF_cur[0:pc], d_x[b][pc:pc+4] = 0 (the pose is data), d_x[b][pc+4:F] = dXF_cur[pc+16:F+12]
__global__ void smt_dx_scatter_kernel(const float* __restrict__ dXc, int ldxf, float* __restrict__ d_x, int ld_dx, int B, int F, int pc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * F) return;
  const int b = (int)(i / F), col = (int)(i % F);
  float v = 0.f;
  if (col < pc) v = dXc[(long)b * ldxf + col];
  else if (col >= pc + 4) v = dXc[(long)b * ldxf + col + 12];
  d_x[(long)b * ld_dx + col] = v;
}

extern "C" int avlen_smt_bwd(const avlen_smt* p, const avlen_smt* g, const float* goal, const float* d_out, int B, int M,
                             int F, int pose_col, int cto, int prec, float* d_x, int ld_dx, void* ws, size_t ws_bytes,
                             hipStream_t st) {
  if (!p || !g || ws_bytes < avlen_smt_workspace_bytes(p, B, M, F, cto)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes); SmtWs s;
  smt_layout(w, s, p, B, M, F, cto != 0);
  const int S = cto ? 1 : M + 1, d = p->tr.d;
  const long R = (long)B * S;
  // Compensated mode at scale (2nd stage: 722 k token rows per minibatch): the FORWARD stays compensated -- it decides the logits,
  // the ratio, the losses -- while the backward's products and its attention run on plain bf16 operands with fp32 accumulation
  // (standard mixed-precision training: gradient terms carry ~1e-2 relative instead of ~1e-4), 1 pass instead of 3 and the
  // matrix-core attention backward instead of the fp32 VALU one: 257 -> ~170 ms per update.  Below g_mixed_rows nothing changes.
  const bool mixed = prec == AVLEN_PREC_BF16X3 && g_mixed_rows > 0 && R >= g_mixed_rows;
  Ctx c{st, mixed ? AVLEN_PREC_BF16 : prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  s.h.lo1 = prec == AVLEN_PREC_BF16X3 ? R * d : 0;
  const Big16* h16 = big16_on(p, s, prec, R, S, cto != 0) ? &s.h : nullptr;      // the forward left 16-bit operand planes
  TRY(transformer_bwd(c, p->tr, g->tr, s.tr, s.tb, s.Z, s.maskx, goal, d_out, s.dZ, B, S, cto != 0, h16));
  // fusion MLP
  TRY(linear_bwd(c, p->fus2, g->fus2, s.dZ, d, s.H1, d, s.dH1, d, (int)R, nullptr, 0, h16 ? h16->H1 : nullptr, d));
  if (h16) TRY(relu_bwd16(c, s.dH1, h16->H1, R * d));
  else TRY(relu_bwd(c, s.dH1, s.H1, R * d));
  {
    avlen_linear g0 = g->fus0;            // dW0[d][F+12] += dH1^T XF  (XF rows are ldxf apart)
    TRY(linear_bwd(c, g0, g0, s.dH1, d, s.XF, s.ldxf, nullptr, 0, (int)R, nullptr, 0, h16 ? h16->XF : nullptr, h16 ? h16->ldxf : 0));
  }
  // pose encoder: dPE[R,16] = dH1 * W0[:, pc:pc+16]
  TRY(avlen_gemm(s.dH1, d, 0, p->fus0.w + pose_col, p->fus0.in_f, 1, s.dPE, 16, nullptr, nullptr, 0, (int)R, 16, d, 0,
                 c.prec, 1, 0.f, s.gws, GEMM_SCRATCH, st));
  int rpb = R >= 65536 ? 2048 : 256;
  hipLaunchKernelGGL(pose_grad_kernel, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0, st, s.dPE, s.FMT, g->pose.w,
                     g->pose.b, R, rpb);
  if (d_x) {
    // gradient w.r.t. the CURRENT observation's features (pi_l under update_dialog: the encoders are trained through it; the
    // memory rows are stored data).  Current token = row b*S + (S-1): dXF_cur [B, F+12] = dH1_cur * W0
    const float* cur = s.dH1 + (long)(S - 1) * d;
    TRY(avlen_gemm(cur, S * d, 0, p->fus0.w, p->fus0.in_f, 1, s.dXc, s.ldxf, nullptr, nullptr, 0, B, p->fus0.in_f, d, 0, prec, 1,
                   0.f, s.gws, GEMM_SCRATCH, st));
    const long n = (long)B * F;
    hipLaunchKernelGGL(smt_dx_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.dXc, s.ldxf, d_x, ld_dx, B, F,
                       pose_col);
  }
  return avlen_launch_status();
}

// =====================================================================================================
// Dialog state encoder
// =====================================================================================================
namespace {
struct DlgWs { float *SEQ, *maskx, *H1, *Z; TrWs tr; void* gws; TrBwdWs tb; float *dZ, *dH1, *dSEQ; };
void dlg_layout(WsBump& w, DlgWs& s, const avlen_dialog* p, long B, long M, bool train = false) {
  long S = M + 1, R = B * S; int d = p->tr.d;
  s.SEQ = w.take<float>(R * 2 * d); s.maskx = w.take<float>(B * S);
  s.H1 = w.take<float>(R * d); s.Z = w.take<float>(R * d);
  tr_layout(w, s.tr, B, S, d, p->tr.nhead, false);
  s.gws = w.take<char>(GEMM_SCRATCH);
  if (train) {                                    // the inference layout is a prefix of the training layout
    trb_layout(w, s.tb, B, S, d, p->tr.nhead);
    s.dZ = w.take<float>(R * d); s.dH1 = w.take<float>(R * d); s.dSEQ = w.take<float>(R * 2 * d);
  }
}
// d_x_att[b] = dSEQ[(b, M)][0:d];   d_demb[b] = sum_s dSEQ[(b, s)][d:2d]   (the dialog embedding is repeated on every token)
__global__ void dialog_split_grad_kernel(const float* __restrict__ dseq, int ldseq, float* __restrict__ d_x_att,
                                         float* __restrict__ d_demb, int S, int d) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    d_x_att[(long)b * d + i] = dseq[((long)b * S + (S - 1)) * ldseq + i];
    if (d_demb) {
      float a = 0.f;
      for (int s2 = 0; s2 < S; s2++) a += dseq[((long)b * S + s2) * ldseq + d + i];
      d_demb[(long)b * d + i] = a;
    }
  }
}
}  // namespace

extern "C" size_t avlen_dialog_workspace_bytes(const avlen_dialog* p, int B, int M) {
  WsBump w(nullptr, 0); DlgWs s; dlg_layout(w, s, p, B, M);
  return dlg_has16(p) ? zmax(w.off + 4096, dlg16_ws_bytes(p, B, M)) : w.off + 4096;
}

extern "C" int avlen_dialog_fwd(const avlen_dialog* p, const float* x_att, const float* memory_state,
                                const float* masks, const float* d_emb, const float* agent_step, const float* goal,
                                float* out, int B, int M, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || B <= 0 || ws_bytes < avlen_dialog_workspace_bytes(p, B, M)) return AVLEN_ERR_WS;
  if (prec == AVLEN_PREC_BF16 && dlg_has16(p))
    return dialog_fwd_bf16(p, x_att, memory_state, masks, d_emb, agent_step, goal, out, B, M, ws, ws_bytes, st);
  if (prec == AVLEN_PREC_BF16X3 && dlg_has16lo(p)) {
    const int rc = dialog_fwd_bf16(p, x_att, memory_state, masks, d_emb, agent_step, goal, out, B, M, ws, ws_bytes, st, true);
    if (rc != AVLEN_NOT_BIG) return rc;
  }
  WsBump w(ws, ws_bytes); DlgWs s; dlg_layout(w, s, p, B, M);
  Ctx c{st, prec, s.gws, GEMM_SCRATCH};
  const int S = M + 1, d = p->tr.d; const long R = (long)B * S;
  const int ldseq = d_emb ? 2 * d : d;
  float* seq = d_emb ? s.SEQ : s.Z;
  hipLaunchKernelGGL(dialog_build_kernel, dim3((unsigned)R), dim3(128), 0, st, x_att, memory_state, masks, d_emb, seq,
                     ldseq, s.maskx, B, M, d);
  TRY(avlen_launch_status());
  if (d_emb) {
    TRY(linear(c, p->fus0, s.SEQ, 2 * d, s.H1, d, (int)R, AVLEN_ACT_RELU, nullptr, 0));
    TRY(linear(c, p->fus2, s.H1, d, s.Z, d, (int)R, 0, nullptr, 0));
  }
  hipLaunchKernelGGL(add_pe_kernel, dim3((unsigned)R), dim3(128), 0, st, s.Z, p->pe, agent_step, S, d, p->pe_len);
  TRY(avlen_launch_status());
  return transformer_fwd(c, p->tr, s.tr, s.Z, s.maskx, goal, out, B, S, false);
}

// ---- training form (PPO.update_dialog, ppo.py:99-154): fp32-staged kernels, every activation kept in the workspace ----
extern "C" size_t avlen_dialog_train_workspace_bytes(const avlen_dialog* p, int B, int M) {
  WsBump w(nullptr, 0); DlgWs s; dlg_layout(w, s, p, B, M, true);
  return w.off + 4096;
}

extern "C" int avlen_dialog_train_fwd(const avlen_dialog* p, const float* x_att, const float* memory_state, const float* masks,
                                      const float* d_emb, const float* agent_step, const float* goal, float* out, int B, int M,
                                      int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || B <= 0 || ws_bytes < avlen_dialog_train_workspace_bytes(p, B, M)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes); DlgWs s; dlg_layout(w, s, p, B, M, true);
  Ctx c{st, prec, s.gws, GEMM_SCRATCH};
  const int S = M + 1, d = p->tr.d; const long R = (long)B * S;
  const int ldseq = d_emb ? 2 * d : d;
  float* seq = d_emb ? s.SEQ : s.Z;
  hipLaunchKernelGGL(dialog_build_kernel, dim3((unsigned)R), dim3(128), 0, st, x_att, memory_state, masks, d_emb, seq, ldseq, s.maskx,
                     B, M, d);
  TRY(avlen_launch_status());
  if (d_emb) {
    TRY(linear(c, p->fus0, s.SEQ, 2 * d, s.H1, d, (int)R, AVLEN_ACT_RELU, nullptr, 0));
    TRY(linear(c, p->fus2, s.H1, d, s.Z, d, (int)R, 0, nullptr, 0));
  }
  hipLaunchKernelGGL(add_pe_kernel, dim3((unsigned)R), dim3(128), 0, st, s.Z, p->pe, agent_step, S, d, p->pe_len);
  TRY(avlen_launch_status());
  return transformer_fwd(c, p->tr, s.tr, s.Z, s.maskx, goal, out, B, S, false);
}

// g: gradient views (accumulated).  d_x_att (B,d) out; d_demb (B,d) out when the forward had a dialog embedding (else NULL).
extern "C" int avlen_dialog_bwd(const avlen_dialog* p, const avlen_dialog* g, const float* goal, const float* d_out, int has_dialog,
                                float* d_x_att, float* d_demb, int B, int M, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || !g || !d_out || !d_x_att || B <= 0 || ws_bytes < avlen_dialog_train_workspace_bytes(p, B, M)) return AVLEN_ERR_WS;
  if (has_dialog && !d_demb) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes); DlgWs s; dlg_layout(w, s, p, B, M, true);
  Ctx c{st, prec, s.gws, GEMM_SCRATCH};
  const int S = M + 1, d = p->tr.d; const long R = (long)B * S;
  // the positional encoding is an additive constant: dZ is the gradient of the fusion output as well
  TRY(transformer_bwd(c, p->tr, g->tr, s.tr, s.tb, s.Z, s.maskx, goal, d_out, s.dZ, B, S, false));
  if (has_dialog) {
    TRY(linear_bwd(c, p->fus2, g->fus2, s.dZ, d, s.H1, d, s.dH1, d, (int)R, nullptr, 0));
    TRY(relu_bwd(c, s.dH1, s.H1, R * d));
    TRY(linear_bwd(c, p->fus0, g->fus0, s.dH1, d, s.SEQ, 2 * d, s.dSEQ, 2 * d, (int)R, nullptr, 0));
    hipLaunchKernelGGL(dialog_split_grad_kernel, dim3((unsigned)B), dim3(128), 0, st, s.dSEQ, 2 * d, d_x_att, d_demb, S, d);
  } else {
    hipLaunchKernelGGL(dialog_split_grad_kernel, dim3((unsigned)B), dim3(128), 0, st, s.dZ, d, d_x_att, (float*)nullptr, S, d);
  }
  return avlen_launch_status();
}

// Generic Linear backward for single layers that sit between modules (pi_l's dialog_layer, policy.py:849):
// G.w += dY^T X, G.b += colsum(dY), dX = dY W (optional)
extern "C" int avlen_linear_bwd(const avlen_linear* L, const avlen_linear* G, const float* X, int ldx, const float* dY, int ldy,
                                float* dX, int lddx, int M, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!L || !G || !X || !dY || M <= 0 || ws_bytes < GEMM_SCRATCH) return AVLEN_ERR_WS;
  Ctx c{st, prec, ws, GEMM_SCRATCH};
  TRY(linear_bwd(c, *L, *G, dY, ldy, X, ldx, dX, lddx, M, nullptr, 0));
  return AVLEN_OK;
}
extern "C" size_t avlen_linear_bwd_workspace_bytes(void) { return GEMM_SCRATCH; }

// action_encoder = Linear(one_hot(prev_action)) (policy.py:662-667): gw[j][a] += sum_{b: a_b = a} d[b][j], gb[j] += sum_b d[b][j]
namespace {
// One block per output feature j: the rows are split over the block's 256 threads, each keeps per-action partial sums in
// registers (n_act <= 8), a fixed-order LDS reduction combines them (deterministic), one write per (j, action).
__global__ __launch_bounds__(256) void action_encoder_bwd_kernel(const float* __restrict__ d, int ld, const int64_t* __restrict__ prev_actions,
                                                                 float* __restrict__ gw, float* __restrict__ gb, int B, int n_out, int n_act) {
  __shared__ float red[256][9];
  const int j = blockIdx.x, t = threadIdx.x;
  float acc[9];
#pragma unroll
  for (int a = 0; a < 9; a++) acc[a] = 0.f;
  for (int b = t; b < B; b += 256) {
    const float v = d[(long)b * ld + j];
    const long a = prev_actions[b];
    acc[8] += v;
#pragma unroll
    for (int k = 0; k < 8; k++) if (a == k) acc[k] += v;
  }
#pragma unroll
  for (int a = 0; a < 9; a++) red[t][a] = acc[a];
  __syncthreads();
  if (t < 9) {
    float s = 0.f;
    for (int i = 0; i < 256; i++) s += red[i][t];
    if (t == 8) gb[j] += s;
    else if (t < n_act) gw[(long)j * n_act + t] += s;
  }
}
}  // namespace
extern "C" int avlen_action_encoder_bwd(const float* d_feats, int ld, const int64_t* prev_actions, const avlen_linear* G, int B,
                                        hipStream_t st) {
  if (!d_feats || !prev_actions || !G || B <= 0 || G->out_f > 256 || G->in_f > 8) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(action_encoder_bwd_kernel, dim3(G->out_f), dim3(256), 0, st, d_feats, ld, prev_actions, G->w, G->b, B, G->out_f, G->in_f);
  return avlen_launch_status();
}

// =====================================================================================================
// CLIP text tower
// =====================================================================================================
extern "C" size_t avlen_clip_text_workspace_bytes(const avlen_clip_text* p, int B) {
  size_t R = (size_t)B * p->ctx, wd = p->width;
  return (R * (wd * 3 + 3 * wd + 4 * wd) + (size_t)B * wd * 2) * sizeof(float) + GEMM_SCRATCH + 8192 +
         (p->wstream ? avlen_clip_tower_stream_ws_bytes(B) + 256 : 0);
}

// out = E2 @ text_proj   (text_proj stored [width][out_dim]); a handful of row tiles -> split K over the chip
static int clip_project(const avlen_clip_text* p, const float* E2, float* out, int B, int prec, void* gws, hipStream_t st) {
  // a handful of rows (<= 16): one wave per output column, fp32 FMA, one launch.  Not for the 64-row rollout batch: the kernel
  // walks 16 rows at a time, every wave re-reads all of x, and a kernel trace showed 92 us against 37 us for the split-K tile GEMM
  if (p->text_proj_t && B <= 16 && avlen_i_skinny_linear_ok(B, p->width))
    return avlen_i_skinny_linear(E2, p->width, p->text_proj_t, nullptr, out, p->out_dim, B, p->out_dim, p->width, st);
  int sk = avlen_gemm_pick_splitk(B, p->out_dim, p->width);
  while (sk > 1 && avlen_gemm_workspace_bytes(B, p->out_dim, p->width, sk) > GEMM_SCRATCH) sk /= 2;
  return avlen_gemm(E2, p->width, 0, p->text_proj, p->out_dim, 1, out, p->out_dim, nullptr, nullptr, 0, B, p->out_dim, p->width,
                    0, prec, sk, 0.f, gws, GEMM_SCRATCH, st);
}

// Largest batch one pass takes: the one-launch tower's flag block holds 6 words per dialog (clip_tower.hip) and the ragged
// launch-per-GEMM path scans the EOT positions in one 1024-thread block; larger batches (PPO.update_dialog evaluates T * N rows,
// ppo.py:99-154) run as consecutive passes over the same workspace -- same kernels, same per-row arithmetic.
constexpr int CLIP_PASS_ROWS = 512;
extern "C" int avlen_clip_text_fwd(const avlen_clip_text* p, const int64_t* tokens, float* out, int B, int prec,
                                   void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || B <= 0 || p->width % p->heads || ws_bytes < avlen_clip_text_workspace_bytes(p, B)) return AVLEN_ERR_WS;
  if (B > CLIP_PASS_ROWS) {
    const int od = p->text_proj ? p->out_dim : p->width;
    for (int b0 = 0; b0 < B; b0 += CLIP_PASS_ROWS) {
      const int nb = B - b0 < CLIP_PASS_ROWS ? B - b0 : CLIP_PASS_ROWS;
      TRY(avlen_clip_text_fwd(p, tokens + (long)b0 * p->ctx, out + (long)b0 * od, nb, prec, ws, ws_bytes, st));
    }
    return AVLEN_OK;
  }
  const int wd = p->width, H = p->heads, D = wd / H, ctx = p->ctx;
  const long R = (long)B * ctx;
  WsBump w(ws, ws_bytes);
  float* X = w.take<float>(R * wd); float* Hn = w.take<float>(R * wd); float* AO = w.take<float>(R * wd);
  float* QKV = w.take<float>(R * 3 * wd); float* Fh = w.take<float>(R * 4 * wd);
  float* E = w.take<float>((size_t)B * wd); float* E2 = w.take<float>((size_t)B * wd);
  void* gws = w.take<char>(GEMM_SCRATCH);
  Ctx c{st, prec, gws, GEMM_SCRATCH};
  const float scale = 1.0f / sqrtf((float)D);
  // the 16-bit shadows of this module's weights are in ONE format (p->half_fmt: 0 = bf16, 1 = fp16); the fast path runs when
  // the requested arithmetic matches it
  const bool f16 = prec == AVLEN_PREC_FP16;
  bool fast = ((prec == AVLEN_PREC_BF16 && p->half_fmt == 0) || (f16 && p->half_fmt == 1)) && B <= 1024;
  if (f16 && !fast) return AVLEN_ERR_ARG;             // no fp16 fallback: the caller asked for shadows it did not build
  avlen_g2_opts go; go.f16 = f16;
  for (int l = 0; l < p->layers && fast; l++) {
    const avlen_clip_block& b = p->block[l];
    fast = lin16_ok(b.attn.in_proj) && lin16_ok(b.attn.out_proj) && lin16_ok(b.fc) && lin16_ok(b.proj);
  }
  if (fast && p->wstream && avlen_clip_stream_bytes(p)) {
    // the 12 blocks as ONE sequence-stationary launch (clip_tower.hip) -> the EOT rows; ln_final + projection as before
    const size_t sb = avlen_clip_tower_stream_ws_bytes(B);
    void* sws = w.take<char>(sb);
    if (!w.ok()) return AVLEN_ERR_WS;
    TRY(avlen_clip_tower_stream_fwd(p, tokens, E, B, f16 ? 1 : 0, sws, sb, st));
    if (!p->text_proj)            // caller folded the projection into the Linear that follows (policy.py: dialog_layer): out = ln_final(E)
      return avlen_layernorm_fwd(E, nullptr, p->ln_final.g, p->ln_final.b, out, nullptr, nullptr, B, wd, 1e-5f, st);
    TRY(avlen_layernorm_fwd(E, nullptr, p->ln_final.g, p->ln_final.b, E2, nullptr, nullptr, B, wd, 1e-5f, st));
    return clip_project(p, E2, out, B, f16 ? AVLEN_PREC_BF16X3 : prec, gws, st);
  }
  if (fast) {      // bf16 operands from HBM; ragged batch: only the tokens up to each EOT are computed
    bf16* Hn16 = (bf16*)Hn; bf16* AO16 = (bf16*)AO; bf16* F16 = (bf16*)Fh;
    int* seg = (int*)E2;                              // [B+1] (E2 is used again only after the layers)
    int* rowmap = (int*)QKV;                          // [R]   (consumed by the embedding, before QKV is written)
    const int* live = seg + B;
    hipLaunchKernelGGL(clip_segments_kernel, dim3(1), dim3(1024), 0, st, tokens, seg, rowmap, B, ctx);
    bool fold = D == 64 && ctx <= 96;                 // LayerNorms folded into the projections that follow them
    for (int l = 0; l < p->layers && fold; l++)
      fold = p->block[l].attn_fold.w16f && p->block[l].fc_fold.w16f && p->block[l].attn.in_proj.ld16 == wd;
    // fold mode: Hn16 holds the raw residual stream in bf16, the rest of Hn the per-layer row statistics: 2 * layers arrays of
    // SL slabs [R][2] -- a producing GEMM STORES one partial per 128-column tile (SL = width / 128), the consumer adds them in
    // slab order: no atomics, bit-reproducible, nothing to zero
    float* stats = (float*)((char*)Hn + (size_t)R * wd * 2);
    const int SL = (wd + 127) / 128;
    const size_t st_stride = (size_t)R * 2 * SL;
    fold = fold && (size_t)wd * 2 >= (size_t)16 * p->layers * SL;
    { static int en = -1; if (en < 0) en = (int)avlen_knob("AVLEN_CLIP_FOLD", 1); fold = fold && en; }
    if (f16 && !fold) return AVLEN_ERR_ARG;           // the fp16 tower exists in the folded-LayerNorm form only
    avlen_g2_opts g_in1 = go, g_inS = go, g_out = go;   // consumer of the embedding's statistics (1 slab) / of a GEMM's (SL); producer
    g_in1.ln_slabs = 1; g_inS.ln_slabs = SL; g_out.rs_slabs = SL;
    hipLaunchKernelGGL(clip_embed_ragged_kernel, dim3((unsigned)R), dim3(128), 0, st, tokens, p->tok_emb, p->pos_emb, X, seg,
                       rowmap, B, ctx, wd, p->vocab, fold ? Hn16 : (bf16*)nullptr, fold ? stats : (float*)nullptr, f16 ? 1 : 0);
    TRY(avlen_launch_status());
    auto lin = [&](const avlen_linear& L, const bf16* X16, int ldx, float* Y32, int ld32, bf16* Y16, int ld16, int act,
                   const float* res) {
      return avlen_gemm_bf16_dyn(X16, ldx, L.w16, L.ld16, Y32, ld32, Y16, ld16, L.b, res, ld32, (int)R, live, L.out_f, L.ld16,
                                 act, c.gws, c.gws_bytes, c.st, &go);
    };
    bool pruned = false;
    static int prune = -1;                            // AVLEN_CLIP_PRUNE=0: last layer over every live token (A/B knob)
    if (prune < 0) prune = (int)avlen_knob("AVLEN_CLIP_PRUNE", 1);
    if (fold) {
      // per layer: 4 GEMMs + attention, no LayerNorm launch: the out_proj / c_proj epilogues emit the new residual
      // stream in fp32 + bf16 and its row statistics; in_proj / c_fc apply the normalisation in their epilogue
      for (int l = 0; l < p->layers; l++) {
        const avlen_clip_block& b = p->block[l];
        float* st1 = stats + (size_t)(2 * l) * st_stride; float* st2 = st1 + st_stride;
        float* st_next = l + 1 < p->layers ? st2 + st_stride : nullptr;
        TRY(avlen_gemm_bf16_ln(Hn16, wd, b.attn_fold.w16f, wd, nullptr, 0, QKV, 3 * wd, b.attn_fold.c, nullptr, 0, (int)R, live,
                               3 * wd, wd, 0, st1, b.attn_fold.s, nullptr, c.gws, c.gws_bytes, st, l == 0 ? &g_in1 : &g_inS));
        TRY(avlen_attention_qkv16(QKV, 3 * wd, AO16, wd, B, H, ctx, 1, scale, seg, st, f16 ? 1 : 0));
        if (prune && l + 1 == p->layers) {
          // one row per sample from here on: out_proj, c_fc and c_proj shrink from the live token count to B rows
          bf16* Fe16 = F16; bf16* AOe16 = Fe16 + (size_t)B * b.fc.out_f; bf16* He16 = AOe16 + (size_t)B * wd;
          float* ste = (float*)(He16 + (size_t)B * wd);                     // SL slabs [B][2]
          hipLaunchKernelGGL(clip_gather_last2_kernel, dim3(B), dim3(128), 0, st, X, AO16, seg, E, AOe16, (float*)nullptr, wd);
          TRY(avlen_launch_status());
          TRY(avlen_gemm_bf16_ln(AOe16, wd, b.attn.out_proj.w16, b.attn.out_proj.ld16, E, wd, He16, wd, b.attn.out_proj.b, E, wd,
                                 B, nullptr, wd, b.attn.out_proj.ld16, 0, nullptr, nullptr, ste, c.gws, c.gws_bytes, st, &g_out));
          TRY(avlen_gemm_bf16_ln(He16, wd, b.fc_fold.w16f, wd, nullptr, 0, Fe16, b.fc.out_f, b.fc_fold.c, nullptr, 0, B, nullptr,
                                 b.fc.out_f, wd, AVLEN_ACT_QUICKGELU, ste, b.fc_fold.s, nullptr, c.gws, c.gws_bytes, st, &g_inS));
          TRY(avlen_gemm_bf16_ln(Fe16, b.fc.out_f, b.proj.w16, b.proj.ld16, E, wd, nullptr, 0, b.proj.b, E, wd, B, nullptr, wd,
                                 b.proj.ld16, 0, nullptr, nullptr, nullptr, c.gws, c.gws_bytes, st, &go));
          pruned = true;
          break;
        }
        TRY(avlen_gemm_bf16_ln(AO16, wd, b.attn.out_proj.w16, b.attn.out_proj.ld16, X, wd, Hn16, wd, b.attn.out_proj.b, X, wd,
                               (int)R, live, wd, b.attn.out_proj.ld16, 0, nullptr, nullptr, st2, c.gws, c.gws_bytes, st, &g_out));
        TRY(avlen_gemm_bf16_ln(Hn16, wd, b.fc_fold.w16f, wd, nullptr, 0, F16, b.fc.out_f, b.fc_fold.c, nullptr, 0, (int)R, live,
                               b.fc.out_f, wd, AVLEN_ACT_QUICKGELU, st2, b.fc_fold.s, nullptr, c.gws, c.gws_bytes, st, &g_inS));
        TRY(avlen_gemm_bf16_ln(F16, b.fc.out_f, b.proj.w16, b.proj.ld16, X, wd, Hn16, wd, b.proj.b, X, wd, (int)R, live, wd,
                               b.proj.ld16, 0, nullptr, nullptr, st_next, c.gws, c.gws_bytes, st, st_next ? &g_out : &go));
      }
    } else
    for (int l = 0; l < p->layers; l++) {
      const avlen_clip_block& b = p->block[l];
      TRY(avlen_layernorm_fwd16_dyn(X, nullptr, b.ln1.g, b.ln1.b, nullptr, Hn16, nullptr, nullptr, (int)R, live, wd, 1e-5f, st));
      if (D == 64 && ctx <= 96) {                      // packed bf16 q|k|v straight into the MFMA attention
        TRY(lin(b.attn.in_proj, Hn16, wd, nullptr, 0, (bf16*)QKV, 3 * wd, 0, nullptr));
        TRY(avlen_attention_qkv16(QKV, 3 * wd, AO16, wd, B, H, ctx, 1, scale, seg, st, f16 ? 1 : 0));
      } else {
        TRY(lin(b.attn.in_proj, Hn16, wd, QKV, 3 * wd, nullptr, 0, 0, nullptr));
        TRY(avlen_attention_fwd16_seg(QKV, 3 * wd, QKV + wd, 3 * wd, QKV + 2 * wd, 3 * wd, nullptr, 0, AO16, wd, nullptr,
                                      nullptr, B, H, ctx, ctx, D, 1, scale, seg, st));
      }
      TRY(lin(b.attn.out_proj, AO16, wd, X, wd, nullptr, 0, 0, X));
      TRY(avlen_layernorm_fwd16_dyn(X, nullptr, b.ln2.g, b.ln2.b, nullptr, Hn16, nullptr, nullptr, (int)R, live, wd, 1e-5f, st));
      TRY(lin(b.fc, Hn16, wd, nullptr, 0, F16, b.fc.out_f, AVLEN_ACT_QUICKGELU, nullptr));
      TRY(lin(b.proj, F16, b.fc.out_f, X, wd, nullptr, 0, 0, X));
    }
    if (!pruned) hipLaunchKernelGGL(clip_gather_last_kernel, dim3(B), dim3(128), 0, st, X, E, seg, wd);
    TRY(avlen_launch_status());
    TRY(avlen_layernorm_fwd(E, nullptr, p->ln_final.g, p->ln_final.b, E2, nullptr, nullptr, B, wd, 1e-5f, st));
    return clip_project(p, E2, out, B, f16 ? AVLEN_PREC_BF16X3 : prec, gws, st);
  }
  hipLaunchKernelGGL(clip_embed_kernel, dim3((unsigned)R), dim3(128), 0, st, tokens, p->tok_emb, p->pos_emb, X, ctx, wd,
                     p->vocab);
  TRY(avlen_launch_status());
  for (int l = 0; l < p->layers; l++) {
    const avlen_clip_block& b = p->block[l];
    TRY(avlen_layernorm_fwd(X, nullptr, b.ln1.g, b.ln1.b, Hn, nullptr, nullptr, (int)R, wd, 1e-5f, st));
    TRY(linear(c, b.attn.in_proj, Hn, wd, QKV, 3 * wd, (int)R, 0, nullptr, 0));
    TRY(avlen_attention_fwd(QKV, 3 * wd, QKV + wd, 3 * wd, QKV + 2 * wd, 3 * wd, AO, wd, nullptr, nullptr, B, H, ctx, ctx,
                            D, 1, scale, st));
    TRY(linear(c, b.attn.out_proj, AO, wd, X, wd, (int)R, 0, X, wd));                 // x += attn
    TRY(avlen_layernorm_fwd(X, nullptr, b.ln2.g, b.ln2.b, Hn, nullptr, nullptr, (int)R, wd, 1e-5f, st));
    TRY(linear(c, b.fc, Hn, wd, Fh, b.fc.out_f, (int)R, AVLEN_ACT_QUICKGELU, nullptr, 0));
    TRY(linear(c, b.proj, Fh, b.fc.out_f, X, wd, (int)R, 0, X, wd));                  // x += mlp
  }
  hipLaunchKernelGGL(clip_gather_eot_kernel, dim3(B), dim3(128), 0, st, tokens, X, E, ctx, wd);
  TRY(avlen_launch_status());
  TRY(avlen_layernorm_fwd(E, nullptr, p->ln_final.g, p->ln_final.b, E2, nullptr, nullptr, B, wd, 1e-5f, st));
  return clip_project(p, E2, out, B, prec, gws, st);
}

// ---- memoised tower: the text tower is frozen and a dialog is constant for NUM_DIALOG_STEPS steps, all-zero for an env without a
// query (ppo_trainer.py:347, 582-586) -- only rows whose tokens differ from the previous call's go through the 12 blocks.  The memo
// is keyed on the CONTENT of a row (its 77 tokens), not on the env it belongs to: a row that compares equal holds, by construction,
// the embedding of exactly these tokens, so shrinking / reordering the batch (pop_at) cannot make it stale; only a weight change
// does (the caller zero-fills the state block: valid = 0).
// State block: [valid | count | zcount | pad to 256 B] [tokens of the last call (B + 1) x ctx, row B all zero] [E (B + 1) x width:
// residual row at the EOT token, BEFORE ln_final] [work list (B + 1)] [zero-row list (B + 1)].  Row B is the all-zero dialog: every
// all-zero row of the batch copies its E instead of running the tower.
namespace {
struct TextCache { int* hdr; int64_t* prev; float* E; int* idx; int* zidx; };
inline size_t text_cache_bytes(int B, int ctx, int wd) {
  return 256 + align_up((size_t)(B + 1) * ctx * 8, 256) + align_up((size_t)(B + 1) * wd * 4, 256) + 2 * align_up((size_t)(B + 1) * 4, 256);
}
inline TextCache text_cache_map(void* state, int B, int ctx, int wd) {
  TextCache c;
  char* s = (char*)state;
  c.hdr = (int*)s; s += 256;
  c.prev = (int64_t*)s; s += align_up((size_t)(B + 1) * ctx * 8, 256);
  c.E = (float*)s; s += align_up((size_t)(B + 1) * wd * 4, 256);
  c.idx = (int*)s; s += align_up((size_t)(B + 1) * 4, 256);
  c.zidx = (int*)s;
  return c;
}
// (which rows changed is decided by the tower launch's own work-list kernel: clip_tower.hip, clip_worklist_kernel)
__global__ __launch_bounds__(128) void text_cache_zero_rows_kernel(float* __restrict__ E, const int* __restrict__ hdr,
                                                                   const int* __restrict__ zidx, int B, int wd) {
  if ((int)blockIdx.x >= hdr[2]) return;
  const int r = zidx[blockIdx.x];
  for (int c = threadIdx.x * 4; c < wd; c += 512)
    *reinterpret_cast<float4*>(E + (long)r * wd + c) = *reinterpret_cast<const float4*>(E + (long)B * wd + c);
}
// The tail of the rollout's text graph as ONE launch: the shared embedding of the all-zero rows copied into the memo (what
// text_cache_zero_rows_kernel does), ln_final of every row, the cast to the tower's 16-bit format and dialog_layer o text_projection
// as one 16-row MFMA product (policy.py:847-849: dialog_layer(encode_text(x)); the projection is folded into the Linear's weight,
// engine.Packed.proj_fold).  Was four launches of 5-9 us each behind the text tower, on the step's critical path.
// Block (x, y) = 16 rows x 64 output features, 4 waves; wave w owns features 64 y + 16 w .. + 15 (one MFMA tile), K = 512 in 16
// steps; the A operand is the weight fragment ([16 features][32 k], 16-byte loads from the row-major 16-bit weight), the B operand
// the rows' 16-bit image in LDS (every block of a row group normalises its 16 rows itself: 32 KB of reads against a launch of its
// own) -- the result tile is lane (c = lane & 15: row, q = lane >> 4: features 4 q .. 4 q + 3), as in chain.hip.  Blocks with
// x >= ceil(B / 16) do no arithmetic: they warm the L2s with the weight ranges in `pf` (the dialog state encoder's fused chain
// runs right behind this launch and the text tower has just swept the caches: elementwise.hip, prefetch_l2_kernel).
struct TailPrefetch { const char* p[4]; long n16[4]; int n; };
typedef __attribute__((ext_vector_type(8))) _Float16 mf16x8;
typedef __attribute__((ext_vector_type(4))) float mf32x4;
template <bool F16>
__global__ __launch_bounds__(256) void text_tail_kernel(float* __restrict__ E, const int* __restrict__ hdr, const int* __restrict__ zidx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const unsigned short* __restrict__ w16, int ldw, const float* __restrict__ bias,
                                                        float* __restrict__ out, int ldo, int B, int N, int row_blocks, TailPrefetch pf,
                                                        unsigned* __restrict__ sink) {
  constexpr int WD = 512, XLD = WD + 8;
  __shared__ __attribute__((aligned(16))) unsigned short xs[16 * XLD];
  __shared__ int zrow[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r0 = blockIdx.x * 16;
  if ((int)blockIdx.x >= row_blocks) {                     // prefetch role: block b serves XCD b % 8, slice b / 8
    const int pb = (blockIdx.x - row_blocks) * gridDim.y + blockIdx.y, np = (gridDim.x - row_blocks) * gridDim.y;
    const int slices = np >> 3, slice = pb >> 3;
    unsigned acc = 0;
    if (slices > 0 && slice < slices)
      for (int r = 0; r < pf.n; r++) {
        const uint4* src = reinterpret_cast<const uint4*>(pf.p[r]);
        const long per = (pf.n16[r] + slices - 1) / slices, lo = per * slice, hi = lo + per < pf.n16[r] ? lo + per : pf.n16[r];
        long i = lo + tid;
        for (; i + 3 * 256 < hi; i += 4 * 256) { const uint4 v0 = src[i], v1 = src[i + 256], v2 = src[i + 512], v3 = src[i + 768]; acc ^= v0.x ^ v1.x ^ v2.x ^ v3.x; }
        for (; i < hi; i += 256) acc ^= src[i].x;
      }
    if (acc == 0x9e3779b9u && sink) *sink = acc;
    return;
  }
  if (tid < 16) zrow[tid] = 0;
  __syncthreads();
  const int nz = hdr ? hdr[2] : 0;
  for (int i = tid; i < nz; i += 256) { const int r = zidx[i] - r0; if (r >= 0 && r < 16) zrow[r] = 1; }
  __syncthreads();
  // ---- ln_final of the block's rows (a wave per row, four rows per wave): ln_fwd_kernel<8>'s arithmetic -> 16-bit image
  for (int rr = wave * 4; rr < wave * 4 + 4; rr++) {
    const int row = r0 + rr;
    unsigned short* xr = xs + rr * XLD;
    if (row >= B) {
      for (int i = 0; i < 8; i++) xr[lane + i * 64] = 0;
      continue;
    }
    const float* src = E + (long)(zrow[rr] ? B : row) * WD;             // an all-zero dialog: the shared row (row B of the memo)
    float v[8], s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = src[lane + i * 64]; s += v[i]; }
    if (zrow[rr] && blockIdx.y == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) E[(long)row * WD + lane + i * 64] = v[i];   // the memo keeps every row's tower output
    }
    const float mean = wave_sum(s) * (1.f / WD);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) { const float t = v[i] - mean; qq += t * t; }
    const float rstd = rsqrtf(wave_sum(qq) * (1.f / WD) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int c = lane + i * 64;
      const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
      if (F16) { const _Float16 hv = (_Float16)o; xr[c] = __builtin_bit_cast(unsigned short, hv); }
      else { const __bf16 hv = (__bf16)o; xr[c] = __builtin_bit_cast(unsigned short, hv); }
    }
  }
  __syncthreads();
  // ---- out[row][n] = bias[n] + sum_k x16[row][k] * w16[n][k]
  const int c = lane & 15, q = lane >> 4;
  {
    const int n0 = blockIdx.y * 64 + wave * 16;
    if (n0 >= N) return;
    mf32x4 acc = (mf32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned short* wr = w16 + (long)(n0 + c) * ldw + q * 8;
    const unsigned short* xr = xs + c * XLD + q * 8;
#pragma unroll
    for (int k = 0; k < WD; k += 32) {
      const uint4 wv = *reinterpret_cast<const uint4*>(wr + k);
      const uint4 xv = *reinterpret_cast<const uint4*>(xr + k);
      if (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mf16x8, wv), __builtin_bit_cast(mf16x8, xv), acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xv), acc, 0, 0, 0);
    }
    const int row = r0 + c;
    if (row < B) {
      const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(out + (long)row * ldo + n0 + q * 4) = make_float4(acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w);
    }
  }
}
}  // namespace
// dialog_layer(CLIP.encode_text(tokens)) of the rollout (policy.py:844-851) on the memoised one-launch tower, the tail fused:
// p = the tower WITHOUT its text projection (text_proj == NULL), fold = dialog_layer with the projection folded into its weight
// ([out_f][width], 16-bit shadow in the tower's format).  out (B, fold->out_f) fp32.  16-bit modes and B + 1 <= 512 rows only
// (AVLEN_ERR_ARG otherwise: the caller takes avlen_clip_text_cached_fwd + its own product).
extern "C" int avlen_clip_text_dialog_fwd(const avlen_clip_text* p, const avlen_linear* fold, const int64_t* tokens, void* state,
                                          size_t state_bytes, float* out, int B, int prec, void* ws, size_t ws_bytes,
                                          const void* const* warm_ptrs, const int64_t* warm_bytes, int n_warm, hipStream_t st) {
  if (!p || !fold || !tokens || !state || !out || B <= 0) return AVLEN_ERR_ARG;
  const int wd = p->width;
  const bool f16 = prec == AVLEN_PREC_FP16;
  const bool fast = ((prec == AVLEN_PREC_BF16 && p->half_fmt == 0) || (f16 && p->half_fmt == 1)) && B + 1 <= CLIP_PASS_ROWS &&
                    p->wstream && avlen_clip_stream_bytes(p);
  if (!fast || p->text_proj || wd != 512 || fold->in_f != wd || !fold->w16 || fold->ld16 % 8 || fold->out_f % 16 || fold->out_f > 256)
    return AVLEN_ERR_ARG;
  if (state_bytes < text_cache_bytes(B, p->ctx, wd) || ws_bytes < avlen_clip_text_workspace_bytes(p, B + 1)) return AVLEN_ERR_WS;
  const TextCache c = text_cache_map(state, B, p->ctx, wd);
  WsBump w(ws, ws_bytes);
  w.take<float>((size_t)B * wd);
  w.take<char>(GEMM_SCRATCH);
  const size_t sb = avlen_clip_tower_stream_ws_bytes(B + 1);
  void* sws = w.take<char>(sb);
  if (!w.ok()) return AVLEN_ERR_WS;
  const avlen_clip_memo memo{tokens, c.prev, c.hdr, c.zidx};
  TRY(avlen_clip_tower_stream_fwd(p, c.prev, c.E, B + 1, f16 ? 1 : 0, sws, sb, st, &memo));
  TailPrefetch pf = {};
  if (n_warm < 0 || n_warm > 4 || (n_warm > 0 && (!warm_ptrs || !warm_bytes))) return AVLEN_ERR_ARG;
  for (int i = 0; i < n_warm; i++) {
    if (!warm_ptrs[i] || warm_bytes[i] < 0 || ((uintptr_t)warm_ptrs[i] & 15)) return AVLEN_ERR_ARG;
    pf.p[pf.n] = (const char*)warm_ptrs[i]; pf.n16[pf.n] = warm_bytes[i] >> 4; pf.n++;
  }
  const int rb = (B + 15) / 16, ny = (fold->out_f + 63) / 64;
  // prefetch blocks: 8 XCDs x 32 slices as elementwise.hip's avlen_prefetch_l2, rounded up to whole grid columns
  const int pcols = pf.n ? (256 + ny - 1) / ny : 0;
  const dim3 grid((unsigned)(rb + pcols), (unsigned)ny);
  if (f16)
    hipLaunchKernelGGL(text_tail_kernel<true>, grid, dim3(256), 0, st, c.E, c.hdr, c.zidx, p->ln_final.g, p->ln_final.b,
                       (const unsigned short*)fold->w16, fold->ld16, fold->b, out, fold->out_f, B, fold->out_f, rb, pf, (unsigned*)nullptr);
  else
    hipLaunchKernelGGL(text_tail_kernel<false>, grid, dim3(256), 0, st, c.E, c.hdr, c.zidx, p->ln_final.g, p->ln_final.b,
                       (const unsigned short*)fold->w16, fold->ld16, fold->b, out, fold->out_f, B, fold->out_f, rb, pf, (unsigned*)nullptr);
  return avlen_launch_status();
}
extern "C" size_t avlen_clip_text_cache_bytes(const avlen_clip_text* p, int B) {
  return p && B > 0 ? text_cache_bytes(B, p->ctx, p->width) : 0;
}
extern "C" int avlen_clip_text_cached_fwd(const avlen_clip_text* p, const int64_t* tokens, void* state, size_t state_bytes, float* out,
                                          int B, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || !tokens || !state || !out || B <= 0) return AVLEN_ERR_ARG;
  const int wd = p->width;
  const bool f16 = prec == AVLEN_PREC_FP16;
  bool fast = ((prec == AVLEN_PREC_BF16 && p->half_fmt == 0) || (f16 && p->half_fmt == 1)) && B + 1 <= CLIP_PASS_ROWS &&
              p->wstream && avlen_clip_stream_bytes(p);
  // the memo exists for the one-launch tower (16-bit modes, <= 511 rows); anything else computes every row
  if (!fast) return avlen_clip_text_fwd(p, tokens, out, B, prec, ws, ws_bytes, st);
  if (state_bytes < text_cache_bytes(B, p->ctx, wd) || ws_bytes < avlen_clip_text_workspace_bytes(p, B + 1)) return AVLEN_ERR_WS;
  const TextCache c = text_cache_map(state, B, p->ctx, wd);
  WsBump w(ws, ws_bytes);
  float* E2 = w.take<float>((size_t)B * wd);
  void* gws = w.take<char>(GEMM_SCRATCH);
  const size_t sb = avlen_clip_tower_stream_ws_bytes(B + 1);
  void* sws = w.take<char>(sb);
  if (!w.ok()) return AVLEN_ERR_WS;
  const avlen_clip_memo memo{tokens, c.prev, c.hdr, c.zidx};
  TRY(avlen_clip_tower_stream_fwd(p, c.prev, c.E, B + 1, f16 ? 1 : 0, sws, sb, st, &memo));
  hipLaunchKernelGGL(text_cache_zero_rows_kernel, dim3(B), dim3(128), 0, st, c.E, c.hdr, c.zidx, B, wd);
  TRY(avlen_launch_status());
  if (!p->text_proj)
    return avlen_layernorm_fwd(c.E, nullptr, p->ln_final.g, p->ln_final.b, out, nullptr, nullptr, B, wd, 1e-5f, st);
  TRY(avlen_layernorm_fwd(c.E, nullptr, p->ln_final.g, p->ln_final.b, E2, nullptr, nullptr, B, wd, 1e-5f, st));
  return clip_project(p, E2, out, B, f16 ? AVLEN_PREC_BF16X3 : prec, gws, st);
}

// =====================================================================================================
// GRU
// =====================================================================================================
extern "C" size_t avlen_gru_workspace_bytes(const avlen_gru* p, int T, int N) {
  size_t H = p->hidden;
  return ((size_t)T * N * 3 * H + (size_t)N * 3 * H + 2 * (size_t)N * H) * sizeof(float) + GEMM_SCRATCH + 4096;
}

extern "C" int avlen_gru_fwd(const avlen_gru* p, const float* x, const float* h0, const float* masks, float* out,
                             float* h_out, int T, int N, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!p || T <= 0 || N <= 0 || ws_bytes < avlen_gru_workspace_bytes(p, T, N)) return AVLEN_ERR_WS;
  const int H = p->hidden;
  WsBump w(ws, ws_bytes);
  float* GI = w.take<float>((size_t)T * N * 3 * H); float* GH = w.take<float>((size_t)N * 3 * H);
  float* hm = w.take<float>((size_t)N * H); float* hc = w.take<float>((size_t)N * H);
  void* gws = w.take<char>(GEMM_SCRATCH);
  Ctx c{st, prec, gws, GEMM_SCRATCH};
  avlen_linear ih{p->w_ih, p->b_ih, 3 * H, p->in_f}, hh{p->w_hh, p->b_hh, 3 * H, H};
  if (avlen_i_skinny_linear_ok(T * N, p->in_f) && (p->in_f & 7))      // a rollout step: 16 rows, K = 1045 (not 8-aligned)
    TRY(avlen_i_skinny_linear(x, p->in_f, p->w_ih, p->b_ih, GI, 3 * H, T * N, 3 * H, p->in_f, st));
  else
    TRY(linear(c, ih, x, p->in_f, GI, 3 * H, T * N, 0, nullptr, 0));
  const float* hprev = h0;
  dim3 g((unsigned)(((long)N * H + 255) / 256));
  if (avlen_i_gru_step_ok(N, H)) {          // few rows: one fused launch per step, one wave per hidden unit (train_gru.hip)
    for (int t = 0; t < T; t++) {
      TRY(avlen_i_gru_step_fwd(p, GI + (size_t)t * N * 3 * H, hprev, masks + (size_t)t * N, out + (size_t)t * N * H, N, st));
      hprev = out + (size_t)t * N * H;
    }
    return avlen_copy_rows(hprev, H, h_out, H, N, H, st);
  }
  for (int t = 0; t < T; t++) {
    hipLaunchKernelGGL(gru_mask_kernel, g, dim3(256), 0, st, hprev, masks + (size_t)t * N, hm, N, H);
    TRY(linear(c, hh, hm, H, GH, 3 * H, N, 0, nullptr, 0));
    hipLaunchKernelGGL(gru_gate_kernel, g, dim3(256), 0, st, GI + (size_t)t * N * 3 * H, GH, hm, hc,
                       out + (size_t)t * N * H, N, H);
    hprev = hc;
  }
  TRY(avlen_launch_status());
  return avlen_copy_rows(hc, H, h_out, H, N, H, st);
}

// Row count from which the training path's Linear products take the cast-to-bf16 + glds GEMM route (default 4096: the benched
// minibatch of 4800 rows runs 2.5 ms per update faster on it than on the fp32-staged kernel; lab env AVLEN_BIGM); <= 0 restores
// the default.  A tuning / test knob, not part of the reference's interface.
extern "C" void avlen_set_big_m(long rows) { g_big_m = rows > 0 ? rows : 4096; }
// bf16x3: token rows (B x (M + 1)) from which the SMT backward uses plain bf16 operands (forward stays compensated); 0 = never,
// < 0 restores the default (65536).
extern "C" void avlen_set_big16(int on) { g_big16 = on; }
extern "C" void avlen_set_audio3(int on) { g_audio3 = on ? 1 : 0; }
extern "C" void avlen_set_x3_mixed_backward_rows(long rows) { g_mixed_rows = rows < 0 ? 65536 : rows; }
