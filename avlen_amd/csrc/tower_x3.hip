// CustomResNet tower (smt_resnet.py:56-149; preprocessing smt_cnn.py:83-93) in COMPENSATED bf16 ("bf16x3", AVLEN_PREC_BF16X3):
// every MFMA operand is a pair hi = bf16(x), lo = bf16(x - hi) and every product is three bf16 MFMAs
// (W_hi X_lo + W_lo X_hi + W_hi X_hi) into one fp32 accumulator -- fp32-grade results (~2^-17 per product) on the matrix cores.
// Raw conv outputs, GroupNorm statistics, the normalisation and the residual stream are fp32.
//
// Pairs double the activation footprint: the 64x64x16 stage (136 KiB as a bf16 LDS image in tower_head.hip) no longer fits a CU's
// LDS, so the tower runs as
//   stem_x3   : preprocessing + 7x7 stem, one workgroup per 8-row band          -> raw0 (fp32, HBM) + statistics
//   c16_x3    : a 3x3 16 -> 16 conv over 8-row bands; the PRODUCER's GroupNorm + ReLU (+ residual) is applied while the halo is
//               staged, so no normalised 64x64x16 activation is ever written      -> raw_k (fp32, HBM) + statistics   (x 4)
//   rest_x3   : layers 2-4 (15 convs, 15 GroupNorms) in ONE launch, one workgroup per image: the stride-2 entry of layer 2 reads
//               the layer-1 output in two half-image passes; from there on the activation is a pair of zero-framed bf16 images
//               in LDS (34x34x32, 18x18x64, 10x10x128), raw outputs and the residual stay in registers as fp32.
// Convs run TAP-OUTER: the weight fragments of one tap (hi + lo, a few KiB per wave) are fetched from L2 one tap ahead and the
// activation fragments are re-read from LDS per tap -- with three MFMAs per fragment pair the matrix pipe, not LDS, is the bound
// (the bf16 kernels are input-row stationary with register-resident weights: 2 x 144 registers would not fit here).
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include "tower_util.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f2;

__device__ __forceinline__ f32x4 mma(const bf16x8& w, const bf16x8& x, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
}
// acc += W X with W = wh + wl, X = xh + xl (the wl * xl term, 2^-18 relative, is dropped); small terms first
__device__ __forceinline__ f32x4 mma3(const bf16x8& wh, const bf16x8& wl, const bf16x8& xh, const bf16x8& xl, f32x4 c) {
  c = mma(wh, xl, c);
  c = mma(wl, xh, c);
  return mma(wh, xh, c);
}
// four fp32 values -> packed hi and lo bf16 quadruples (8 bytes each)
__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo) {
  const unsigned h0 = pack2((f32x2){v[0], v[1]}), h1 = pack2((f32x2){v[2], v[3]});
  const f32x2 a = {__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
  const f32x2 b = {__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
  hi = make_uint2(h0, h1);
  lo = make_uint2(pack2((f32x2){v[0] - a[0], v[1] - a[1]}), pack2((f32x2){v[2] - b[0], v[3] - b[1]}));
}
__device__ __forceinline__ bf16x8 zero_frag() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; e++) z[e] = (bf16)0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------------------------------
// GroupNorm(16) scale / shift of a channel from the producer's (sum, sum of squares): the arithmetic of tower_head.hip
// (moments combined in double, 1 / sqrtf(var + eps) in fp32)
__device__ __forceinline__ void gn_coef(double sum, double sq, double inv_n, float gamma, float beta, float& sc, float& sh) {
  const double mean = sum * inv_n;
  double var = sq * inv_n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
  sc = gamma * rstd;
  sh = beta - (float)mean * sc;
}

constexpr int HCOLS = 66;                                   // 64 pixels + the 3x3 zero frame
__device__ __forceinline__ int h16(int y, int p, int chunk) { return (y * HCOLS + p) * 32 + ((chunk ^ ((p >> 3) & 1)) << 4); }
template <int K, int C, typename T>
__device__ __forceinline__ void l1_fill(const T* __restrict__ img, float scale, char* lds, int tid);

// ========================================================================================================================
// stem + layer 1 fused: one workgroup (512 threads) per image
// ========================================================================================================================
// The 64 x 64 x 16 activation as a compensated pair is 256 KiB: twice a CU's LDS.  The workgroup therefore walks every conv in
// two half-image passes over ONE LDS frame (34 rows x 66 pixels x 32 B per plane, hi + lo): the top half of a conv's input is
// written into the frame by the previous conv's normalise-and-split pass (registers -> LDS), and the BOTTOM half never leaves the
// CU either: the normalised, split values of rows 32 .. 63 stay in the 64 registers their raw accumulators occupied (a packed
// hi | lo pair takes the space of the fp32 value) until the next conv's bottom pass writes them into the frame (image row 31, its
// halo, waits in a one-row side buffer).  (Until round 4 the bottom half made a global round trip per conv -- 128 KB out, 128 KB
// back by LDS-DMA, ~10 k cycles of exposed latency each.)  Raw outputs of both halves stay in registers (128 fp32 per lane) until
// the GroupNorm statistics of the whole image are known.  Only BLOCK outputs (stem, block 0, block 1) are written to the scratch
// images: the next block's residual re-reads them, the layer 2-4 body reads the last one.
constexpr int RTH = 512;
__device__ __attribute__((aligned(16))) unsigned int g_zero_page_x3[4096];
constexpr int L1_PLANE = 71 * 1024;                         // 34 x 66 x 32 B = 71,808 B, rounded up to whole 1 KiB DMA pieces
constexpr int L1_PART_OFF = 2 * L1_PLANE;                   // [8 waves][16 ch][2] fp32
constexpr int L1_COEF_OFF = L1_PART_OFF + 8 * 16 * 2 * 4;   // scale[16] shift[16]
constexpr int L1_GB_OFF = L1_COEF_OFF + 32 * 4;             // gamma | beta of the 5 GroupNorms
constexpr int L1_SIDE_OFF = L1_GB_OFF + 5 * 32 * 4;         // image row 31 of the current activation (one frame row, hi | lo): the halo
constexpr int L1_SIDE_PLANE = HCOLS * 32;                   // row of the bottom-half pass
constexpr int L1_LDS = L1_SIDE_OFF + 2 * L1_SIDE_PLANE;
constexpr int S70 = 70;
static_assert((S70 * S70 + 2) * 8 <= L1_PLANE && L1_LDS <= 160 * 1024, "layer-1 LDS budget");
constexpr long APLANE = 64L * 64 * 16;                      // elements of one plane of the scratch activation image

struct L1Tower {
  const void* img; int u8; int C; float div;
  const bf16* wh[5]; const bf16* wl[5];                     // stem [16][49][8], then the four [16][9][16] convs: hi / lo
  const float* g[5]; const float* b[5];
  bf16* a0; bf16* a1;                                       // scratch activation images (B x 2 planes x 64 x 64 x 16): ping / pong
};
struct L1Args { L1Tower t[8]; const int* row_index; int S; };
#ifdef AVLEN_X3_PROF            // tools/x3_lab.hip: phase timestamps of one thread of every workgroup
#define X3_STAMP(k) do { if (prof && tid == AVLEN_X3_PROF) prof[(long)item * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define X3_WALL(k) do { if (args.prof && tid == AVLEN_X3_PROF) args.prof[(long)item * 32 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define X3_STAMP(k) do { } while (0)
#define X3_WALL(k) do { } while (0)
#endif


// one half of a scratch activation image -> the LDS frame (both planes): frame row fr <-> image row 32 h - 1 + fr, frame pixel
// p <-> image pixel p - 1; rows / pixels outside the image read the zero page.  The chunk swizzle of h16() is applied on the
// SOURCE address (the LDS side of an LDS-DMA is lane-linear).
template <int LPLANE>
__device__ __forceinline__ void l1_load_half(const bf16* __restrict__ act, int h, char* lds, int wave, int lane) {
#pragma unroll 1
  for (int pc = wave; pc < 2 * 71; pc += 8) {
    const int pl = pc >= 71, piece = pc - pl * 71;
    const int L = piece * 64 + lane;                        // 16-byte slot inside the plane
    const int row = L / 132, c16 = L - row * 132, px = c16 >> 1, chunk = (c16 & 1) ^ ((px >> 3) & 1);
    const int iy = 32 * h - 1 + row;
    const bool ok = row < 34 && px >= 1 && px <= 64 && iy >= 0 && iy < 64;
    const char* src = ok ? (const char*)(act + pl * APLANE + ((long)iy * 64 + (px - 1)) * 16 + chunk * 8)
                         : (const char*)g_zero_page_x3 + lane * 16;
    __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(lds + pl * LPLANE + piece * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ void l1_finish16(float (&s1)[4], float (&s2)[4], char* lds, int gi, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + L1_GB_OFF) + gi * 32;
  float* part = reinterpret_cast<float*>(lds + L1_PART_OFF);
  float* coef = reinterpret_cast<float*>(lds + L1_COEF_OFF);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float a = row16_sum(s1[r]), c = row16_sum(s2[r]);
    if (r16 == 0) *reinterpret_cast<float2*>(&part[(wave * 16 + q * 4 + r) * 2]) = make_float2(a, c);
  }
  lds_barrier();                                            // every wave has also finished reading the frame
  if (tid < 16) {
    double sum = 0.0, sq = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) { const float2 v = *reinterpret_cast<const float2*>(&part[(w * 16 + tid) * 2]); sum += v.x; sq += v.y; }
    gn_coef(sum, sq, 1.0 / 4096.0, gamma[tid], gamma[16 + tid], coef[tid], coef[16 + tid]);
  }
  lds_barrier();
}

// normalise (+ residual) + ReLU of the 32 raw tiles; split into hi / lo; -> the scratch image `dst` (both planes, all rows),
// the residual image (END: this is a block output) and -- rows -1 .. 32 -- straight into the LDS frame as the next conv's top half
// RES: `dst` still holds the block's input (this block's residual) as a compensated pair; each lane reads the elements it is
// about to overwrite.  The loads run three rows of tiles ahead of their use through a ring of four; the first three rows are
// requested before the GroupNorm statistics are combined (l1_res_prefetch), so their latency hides behind that phase's barriers.
struct L1Ring { uint2 h[4][4], l[4][4]; };
__device__ __forceinline__ void l1_res_fetch(L1Ring& ring, const bf16* __restrict__ dst, unsigned le, int g) {
#pragma unroll
  for (int mt = 0; mt < 4; mt++) {
    const int te = ((32 * (g >> 2) + (g & 3)) * 64 + mt * 16) * 16;
    ring.h[g & 3][mt] = *reinterpret_cast<const uint2*>((dst + te) + le);
    ring.l[g & 3][mt] = *reinterpret_cast<const uint2*>((dst + APLANE + te) + le);
  }
}
__device__ __forceinline__ unsigned l1_lane_elem(int wave, int r16, int q) {
  // one 32-bit element offset per lane; a tile's offset is a compile-time constant folded into the (uniform) base pointer, so
  // no per-tile 64-bit addresses are kept alive across the kernel
  unsigned le = (unsigned)((4 * wave * 64 + r16) * 16 + q * 4);
  asm volatile("" : "+v"(le));
  return le;
}
__device__ __forceinline__ void l1_res_prefetch(L1Ring& ring, const bf16* __restrict__ dst, int wave, int r16, int q) {
  const unsigned le = l1_lane_elem(wave, r16, q);
#pragma unroll
  for (int g = 0; g < 3; g++) l1_res_fetch(ring, dst, le, g);
}

template <bool RES, bool ALLROWS, bool TOLDS>
__device__ __forceinline__ void l1_apply(f32x4 (&acc)[32], L1Ring& ring, char* lds, bf16* __restrict__ dst, int wave, int r16, int q) {
  const float* coef = reinterpret_cast<const float*>(lds + L1_COEF_OFF);
  float sc[4], sh[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { sc[r] = coef[q * 4 + r]; sh[r] = coef[16 + q * 4 + r]; }
  const unsigned le = l1_lane_elem(wave, r16, q);
  unsigned ll0 = (unsigned)(h16(4 * wave + 1, r16 + 1, q >> 1) + (q & 1) * 8);
  asm volatile("" : "+v"(ll0));
#pragma unroll
  for (int g = 0; g < 8; g++) {                             // g = (h, rr): image row 32 h + 4 wave + rr
    const int h = g >> 2, rr = g & 3;
    if (RES && g + 3 < 8) l1_res_fetch(ring, dst, le, g + 3);
    // an intermediate activation (a block's conv1 output) never leaves the CU: top half -> LDS frame, bottom half -> registers
    const bool to_global = ALLROWS;
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      const int te = ((32 * h + rr) * 64 + mt * 16) * 16;
      const f32x4& a = acc[(h * 4 + rr) * 4 + mt];
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = a[r] * sc[r] + sh[r];
      if (RES) {
        const uint2 uh = ring.h[g & 3][mt], ul = ring.l[g & 3][mt];
        v[0] += __uint_as_float(uh.x << 16) + __uint_as_float(ul.x << 16);
        v[1] += __uint_as_float(uh.x & 0xffff0000u) + __uint_as_float(ul.x & 0xffff0000u);
        v[2] += __uint_as_float(uh.y << 16) + __uint_as_float(ul.y << 16);
        v[3] += __uint_as_float(uh.y & 0xffff0000u) + __uint_as_float(ul.y & 0xffff0000u);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      uint2 hh, ll;
      split4(v, hh, ll);
      if (to_global) {
        *reinterpret_cast<uint2*>((dst + te) + le) = hh;
        *reinterpret_cast<uint2*>((dst + APLANE + te) + le) = ll;
      }
      // rows -1 .. 32 of the image are the next conv's top half: frame row = image row + 1 (h = 1 reaches it with wave 0, rr 0)
      if (TOLDS && (h == 0 || (rr == 0 && wave == 0))) {
        const int tl = ((32 * h + rr) * HCOLS + mt * 16) * 32;
        *reinterpret_cast<uint2*>(lds + ll0 + tl) = hh;
        *reinterpret_cast<uint2*>(lds + ll0 + tl + L1_PLANE) = ll;
      }
      if (TOLDS && h == 0 && rr == 3 && wave == 7) {        // image row 31: the halo row of the next conv's bottom pass
        const int sl = L1_SIDE_OFF + h16(0, r16 + 1, q >> 1) + (q & 1) * 8 + mt * 16 * 32;
        *reinterpret_cast<uint2*>(lds + sl) = hh;
        *reinterpret_cast<uint2*>(lds + sl + L1_SIDE_PLANE) = ll;
      }
      // the bottom half of the next conv's input waits in the registers of its own raw values (l1_store_bottom)
      if (TOLDS && h == 1) acc[(h * 4 + rr) * 4 + mt] = (f32x4){__uint_as_float(hh.x), __uint_as_float(hh.y), __uint_as_float(ll.x), __uint_as_float(ll.y)};
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Bottom pass of a conv: the packed pairs of image rows 32 .. 63 (acc[16 .. 31], see l1_apply) -> frame rows 1 .. 32; image row 31
// (side buffer) -> frame row 0; frame row 33 (image row 64) = zero.  The left / right border pixels of rows 1 .. 33 are still zero
// from zero_top_border (nothing but that and the direct writes of pixels 1 .. 64 ever touches the frame).
__device__ __forceinline__ void l1_store_bottom(const f32x4 (&acc)[32], char* lds, int tid, int wave, int r16, int q) {
  unsigned ll0 = (unsigned)(h16(4 * wave + 1, r16 + 1, q >> 1) + (q & 1) * 8);
  asm volatile("" : "+v"(ll0));
#pragma unroll
  for (int rr = 0; rr < 4; rr++)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      const f32x4& a = acc[(4 + rr) * 4 + mt];
      const int tl = (rr * HCOLS + mt * 16) * 32;
      *reinterpret_cast<uint2*>(lds + ll0 + tl) = make_uint2(__float_as_uint(a[0]), __float_as_uint(a[1]));
      *reinterpret_cast<uint2*>(lds + ll0 + tl + L1_PLANE) = make_uint2(__float_as_uint(a[2]), __float_as_uint(a[3]));
    }
  for (int i = tid; i < 4 * 132; i += RTH) {                // 16-byte chunks: 132 per frame row and plane
    const int zero = i >= 2 * 132, j = i - zero * 2 * 132, pl = j >= 132, c = j - pl * 132;
    if (!zero) {                                            // frame row 0 <- side buffer (whose border pixels 0 / 65 are never written)
      uint4 v = *reinterpret_cast<const uint4*>(lds + L1_SIDE_OFF + pl * L1_SIDE_PLANE + c * 16);
      if (c < 2 || c >= 130) v = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(lds + pl * L1_PLANE + c * 16) = v;
    } else                                                    // frame row 33 (image row 64) <- zero
      *reinterpret_cast<uint4*>(lds + pl * L1_PLANE + 33 * HCOLS * 32 + c * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// preprocessing of the whole image (x / div, K x K mean) into the stem's LDS image: 70 x 70 pixels x 4 channels, hi / lo planes
template <int K, int C, typename T>
__device__ __forceinline__ void l1_fill(const T* __restrict__ img, float scale, char* lds, int tid) {
  constexpr int S = 64 * K, E = K * C;
  static_assert(E % 2 == 0 && C <= 4, "vector loads need an even span");
  typedef __attribute__((ext_vector_type(2))) T T2;
  // One or two batches of frame pixels per thread (ten pixels in all): first ALL loads of a batch (coordinates clamped, no branch -- a load under
  // `if (inside the image)` compiles to branch + loads + s_waitcnt per pixel, and even branch-free the scheduler keeps each unrolled
  // iteration's loads next to its own arithmetic: ten dependent memory round trips per thread, 21.9k cycles for a 48 KB image),
  // then the window sums, the split and the LDS stores.
  constexpr int TOT = S70 * S70 + 2, NB = sizeof(T) * E <= 8 ? 10 : 5;     // one batch when a pixel's window is <= 8 bytes per row
#pragma unroll 1
  for (int i0 = tid; i0 < TOT; i0 += 512 * NB) {
    T2 v[NB][K][E / 2];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int i = i0 + 512 * b < TOT ? i0 + 512 * b : TOT - 1;
      const int hr = i / S70, col = i - hr * S70;
      const int oy = hr - 3, ox = col - 3;
      const int cy = oy < 0 ? 0 : oy > 63 ? 63 : oy, cx = ox < 0 ? 0 : ox > 63 ? 63 : ox;
      const T* p = img + ((long)cy * K * S + cx * K) * C;
#pragma unroll
      for (int dy = 0; dy < K; dy++)
#pragma unroll
        for (int j = 0; j < E / 2; j++) v[b][dy][j] = *reinterpret_cast<const T2*>(p + (long)dy * S * C + 2 * j);
    }
    __builtin_amdgcn_sched_barrier(0);                      // the loads above are issued before any of the arithmetic below
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int i = i0 + 512 * b;
      const int hr = i / S70, col = i - hr * S70;
      const int oy = hr - 3, ox = col - 3;
      const bool ok = hr < S70 && oy >= 0 && oy < 64 && ox >= 0 && ox < 64;
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C; c++) {
        float sm = 0.f;
#pragma unroll
        for (int dy = 0; dy < K; dy++)
#pragma unroll
          for (int dx = 0; dx < K; dx++) { const int e = dx * C + c; sm += (float)v[b][dy][e >> 1][e & 1]; }
        o[c] = ok ? sm * scale : 0.f;
      }
      uint2 h, l;
      split4(o, h, l);
      if (i < TOT) {
        *reinterpret_cast<uint2*>(lds + i * 8) = h;
        *reinterpret_cast<uint2*>(lds + L1_PLANE + i * 8) = l;
      }
    }
  }
}

__device__ __forceinline__ void l1_body(const L1Args& args, int g, int img, char* lds, long long* prof, int item) {
  // the thread index is made opaque per work item: everything derived from it is recomputed inside the item instead of being
  // hoisted out of the queue loop and kept alive (spilled) across both bodies
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const L1Tower& t = args.t[g];
  bf16* __restrict__ A0 = t.a0 + (long)img * 2 * APLANE;
  bf16* __restrict__ A1 = t.a1 + (long)img * 2 * APLANE;
  if (tid < 160) reinterpret_cast<float*>(lds + L1_GB_OFF)[tid] = (tid & 31) < 16 ? t.g[tid >> 5][tid & 15] : t.b[tid >> 5][tid & 15];
  X3_STAMP(0);
  f32x4 acc[32];
  L1Ring ring;
  float s1[4], s2[4];
  auto zero_stats = [&]() {
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[r] = 0.f; s2[r] = 0.f; }
  };
  auto add_stats = [&]() {
#pragma unroll
    for (int i = 0; i < 32; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) { s1[r] += acc[i][r]; s2[r] = __builtin_fmaf(acc[i][r], acc[i][r], s2[r]); }
  };
  // ---- stem: the preprocessed 4-channel image (hi / lo, 3-pixel zero frame) in LDS, whole image at once
  {
    bf16x8 wh[7], wl[7];
#pragma unroll
    for (int ky = 0; ky < 7; ky++) {
      const int kx = 2 * q;
      const long o0 = (long)r16 * 392 + (ky * 7 + kx) * 8;
      const uint2 a0 = *reinterpret_cast<const uint2*>(t.wh[0] + o0), b0 = *reinterpret_cast<const uint2*>(t.wl[0] + o0);
      uint2 a1 = make_uint2(0u, 0u), b1 = a1;
      if (kx + 1 < 7) { a1 = *reinterpret_cast<const uint2*>(t.wh[0] + o0 + 8); b1 = *reinterpret_cast<const uint2*>(t.wl[0] + o0 + 8); }
      wh[ky] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
      wl[ky] = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
    }
    const int S = args.S, k = S / 64, C = t.C;
    const long bs = args.row_index ? args.row_index[img] : img;
    // mean over the k x k window of x / div as (sum x) * (1 / (div k^2)): one rounding instead of k^2 + 1 (a few 1e-8 relative, far
    // inside the 2^-17 of the compensated products) and one multiply per channel instead of k^2 divisions
    const float scale = 1.f / (t.div * (float)(k * k));
    if (k == 2 && C == 3 && t.u8) l1_fill<2, 3, unsigned char>((const unsigned char*)t.img + bs * S * S * 3, scale, lds, tid);
    else if (k == 2 && C == 3) l1_fill<2, 3, float>((const float*)t.img + bs * S * S * 3, scale, lds, tid);
    else if (k == 2 && C == 1 && !t.u8) l1_fill<2, 1, float>((const float*)t.img + bs * S * S, scale, lds, tid);
    else
      for (int i = tid; i < S70 * S70 + 2; i += RTH) {
        const int hr = i / S70, col = i - hr * S70;
        const int oy = hr - 3, ox = col - 3;
        uint2 h = make_uint2(0u, 0u), l = h;
        if (hr < S70 && oy >= 0 && oy < 64 && ox >= 0 && ox < 64) {
          float o[4] = {0.f, 0.f, 0.f, 0.f};
          const long base = ((bs * S + (long)oy * k) * S + (long)ox * k) * C;
#pragma unroll
          for (int c = 0; c < 4; c++) {
            if (c >= C) break;
            float sm = 0.f;
            for (int dy = 0; dy < k; dy++)
              for (int dx = 0; dx < k; dx++) {
                const long idx = base + ((long)dy * S + dx) * C + c;
                sm += t.u8 ? (float)((const unsigned char*)t.img)[idx] : ((const float*)t.img)[idx];
              }
            o[c] = sm * scale;
          }
          split4(o, h, l);
        }
        *reinterpret_cast<uint2*>(lds + i * 8) = h;
        *reinterpret_cast<uint2*>(lds + L1_PLANE + i * 8) = l;
      }
    lds_barrier();
    X3_STAMP(1);
    // a wave's four output rows of one column tile share their input rows: each of the 10 input rows is read once and feeds the
    // (up to 4) accumulators whose kernel row it is
#pragma unroll
    for (int hm = 0; hm < 8; hm++) {
      const int h = hm >> 2, mt = hm & 3;
      const int base = ((32 * h + 4 * wave) * S70 + mt * 16 + r16 + 2 * q) * 8;
      f32x4 a[4];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) a[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ir = 0; ir < 10; ir++) {
        const int off = base + ir * S70 * 8;
        const uint2 x0 = *reinterpret_cast<const uint2*>(lds + off), x1 = *reinterpret_cast<const uint2*>(lds + off + 8);
        const uint2 l0 = *reinterpret_cast<const uint2*>(lds + L1_PLANE + off), l1 = *reinterpret_cast<const uint2*>(lds + L1_PLANE + off + 8);
        const bf16x8 X = __builtin_bit_cast(bf16x8, make_uint4(x0.x, x0.y, x1.x, x1.y)), L = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
          if (ir - rr >= 0 && ir - rr < 7) a[rr] = mma3(wh[ir - rr], wl[ir - rr], X, L, a[rr]);
      }
#pragma unroll
      for (int rr = 0; rr < 4; rr++) acc[(h * 4 + rr) * 4 + mt] = a[rr];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  X3_STAMP(2);
  zero_stats(); add_stats();
  l1_finish16(s1, s2, lds, 0, tid, wave, r16, q);
  X3_STAMP(3);
  // zero the frame's border: rows 0 and the left / right pixel columns are rewritten by every DMA, the direct writes leave them alone
  l1_apply<false, true, true>(acc, ring, lds, A0, wave, r16, q);
  auto zero_top_border = [&]() {                             // frame row 0 (image row -1) and pixels 0 / 65 of rows 1 .. 33, both planes
    for (int i = tid; i < 132 + 33 * 4; i += RTH) {
      int ad;
      if (i < 132) ad = h16(0, i >> 1, i & 1);
      else { const int j = i - 132; ad = h16(1 + (j >> 2), (j & 2) ? 65 : 0, j & 1); }
      *reinterpret_cast<uint4*>(lds + ad) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(lds + L1_PLANE + ad) = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  zero_top_border();
  X3_STAMP(4);
  // ---- the four 3x3 convs of layer 1
  bf16* src = A0; bf16* dst = A1;
#pragma unroll 1
  for (int blk = 0; blk < 2; blk++)
#pragma unroll
  for (int cj = 0; cj < 2; cj++) {                          // cj is static: the residual ring lives only inside a block's conv2
    const int ci = 2 * blk + cj;
    // INPUT-ROW-STATIONARY k-steps (K = 9 taps x 16 channels = 4.5 steps of 32): a step pairs two taps through the lane quarter
    // (q >> 1 selects the tap, q & 1 the 8-channel chunk).  Pairing taps of ONE kernel row makes a fragment of input row R serve
    // three output rows:
    //   A(R) = row R, pixels x | x + 1   -> output rows R, R - 1, R - 2 with the weights of (ky, 0 | 1), ky = 0, 1, 2
    //   B(R) = row R | row R + 1, pixel x + 2 -> output row R with the weights of (0, 2) | (1, 2); its first half is all step
    //          C = (2, 2) | zero needs: output row R - 2 reuses B(R)'s registers with the weights [(2, 2) | 0]
    // 12 fragment pairs (hi + lo) per 16-pixel column of a wave's four output rows instead of 20 -- the output-stationary loop was
    // LDS-bandwidth-bound (2 KB of fragments per three MFMAs) -- and still 5 steps per output tile.
    bf16x8 wh[5], wl[5];                                    // A0 A1 A2 B C
#pragma unroll
    for (int s = 0; s < 5; s++) {
      const int hi_tap = q >> 1;
      const int tap = s < 3 ? 3 * s + hi_tap : s == 3 ? (hi_tap ? 5 : 2) : 8;
      if (s < 4 || hi_tap == 0) {
        wh[s] = *reinterpret_cast<const bf16x8*>(t.wh[1 + ci] + (long)r16 * 144 + tap * 16 + (q & 1) * 8);
        wl[s] = *reinterpret_cast<const bf16x8*>(t.wl[1 + ci] + (long)r16 * 144 + tap * 16 + (q & 1) * 8);
      } else { wh[s] = zero_frag(); wl[s] = zero_frag(); }
    }
    const int rdA = h16(4 * wave, r16 + (q >> 1), q & 1);   // + (R * HCOLS + 16 mt) * 32: input row R of the wave, column tile mt
    const int rdB = h16(4 * wave + (q >> 1), r16 + 2, q & 1);
    const int rdC = h16(4 * wave, r16 + 2, q & 1);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (h == 1) {
        lds_barrier();                                      // everyone has left the top half
        l1_store_bottom(acc, lds, tid, wave, r16, q);
      }
      lds_barrier();
      if (h == 1) X3_STAMP(6 + 5 * ci);
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        f32x4 a[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) a[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int R = 0; R < 6; R++) {
          const int off = (R * HCOLS + mt * 16) * 32;
          const bf16x8 xah = *reinterpret_cast<const bf16x8*>(lds + rdA + off), xal = *reinterpret_cast<const bf16x8*>(lds + L1_PLANE + rdA + off);
          const int rb = (R < 4 ? rdB : rdC) + off;         // rows 4, 5 are read for step C only (its second half has zero weights)
          const bf16x8 xbh = *reinterpret_cast<const bf16x8*>(lds + rb), xbl = *reinterpret_cast<const bf16x8*>(lds + L1_PLANE + rb);
#pragma unroll
          for (int ky = 0; ky < 3; ky++)
            if (R - ky >= 0 && R - ky < 4) a[R - ky] = mma3(wh[ky], wl[ky], xah, xal, a[R - ky]);
          if (R < 4) a[R] = mma3(wh[3], wl[3], xbh, xbl, a[R]);
          if (R >= 2) a[R - 2] = mma3(wh[4], wl[4], xbh, xbl, a[R - 2]);
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) acc[(h * 4 + rr) * 4 + mt] = a[rr];
        __builtin_amdgcn_sched_barrier(0);
      }
      X3_STAMP(5 + 5 * ci + 2 * h);
    }
    if (cj == 1) l1_res_prefetch(ring, dst, wave, r16, q);
    zero_stats(); add_stats();
    l1_finish16(s1, s2, lds, 1 + ci, tid, wave, r16, q);
    X3_STAMP(8 + 5 * ci);
    if (cj == 0) l1_apply<false, false, true>(acc, ring, lds, dst, wave, r16, q);
    else if (blk == 0) l1_apply<true, true, true>(acc, ring, lds, dst, wave, r16, q);
    else l1_apply<true, true, false>(acc, ring, lds, dst, wave, r16, q);
    if (ci < 3) zero_top_border();
    X3_STAMP(9 + 5 * ci);
    bf16* tmp = src; src = dst; dst = tmp;
  }
}

// ========================================================================================================================
// layers 2-4: one workgroup (512 threads) per image
// ========================================================================================================================
#ifndef AVLEN_X3_STAGGER
#define AVLEN_X3_STAGGER 8       // s_sleep units (64 cycles) by which waves 4-7 enter a conv's tap loop late: their LDS read bursts then
#endif                           // fall under the MFMA segments of their SIMD partners (waves 0-3) instead of colliding with them
constexpr int R32 = 34, R64 = 18, R128 = 10;
constexpr int PLANE = R32 * R32 * 64;                       // 73984 B: one plane of the largest frame; the later frames reuse the space
constexpr int HALF16 = 34 * HCOLS * 32;                     // 71808 B: one plane of a half image of the 16-channel stage
static_assert(HALF16 <= PLANE && R64 * R64 * 128 <= PLANE && R128 * R128 * 256 <= PLANE, "frames share one LDS region");
constexpr int XPART_OFF = 2 * PLANE;                        // statistics partials: [8 waves][16 slots][2] fp32
constexpr int XCOEF_OFF = XPART_OFF + 8 * 16 * 2 * 4;       // scale[128], shift[128]
constexpr int XGB_OFF = XCOEF_OFF + 1024;                   // gamma | beta of the 15 GroupNorms: 5 x 64, 5 x 128, 5 x 256 fp32
constexpr int REST_LDS = XGB_OFF + (5 * 64 + 5 * 128 + 5 * 256) * 4;
static_assert(REST_LDS <= 160 * 1024, "tower x3 LDS budget");

__device__ __forceinline__ int a32(int y, int p, int chunk) { return (y * R32 + p) * 64 + ((chunk ^ ((p >> 1) & 3)) << 4); }
__device__ __forceinline__ int a64(int y, int p, int chunk) { return (y * R64 + p) * 128 + ((chunk ^ (p & 7)) << 4); }
// 128 channels: a pixel is one 256-byte bank row; a ds_read_b128 lane group holds 8 lanes of each of two lane quarters (chunks
// c, c + 1) on image rows y, y + 1 -- the row parity moves the slot into the other half, so the 16 lanes hit 16 different slots
__device__ __forceinline__ int a128(int y, int p, int chunk) { return (y * R128 + p) * 256 + ((chunk ^ (p & 7) ^ ((y & 1) << 3)) << 4); }

// GroupNorm index per stage: 0 downsample, 1 block 0 conv1 (stride 2), 2 block 0 conv2, 3 block 1 conv1, 4 block 1 conv2
struct RestTower {
  const bf16* a;                                                              // layer-1 output (B x 2 planes x 64 x 64 x 16): hi / lo
  const bf16* wh[15]; const bf16* wl[15];                                     // [stage * 5 + conv]
  const float* g[15]; const float* b[15];
  bf16* y;                                                                    // layer-4 output NHWC (B, 8, 8, 128) as a bf16 pair
};
struct RestArgs { RestTower t[8]; long y_lo; };              // y_lo: elements from the hi plane to the lo plane

__device__ __forceinline__ void stat_pair(const f32x4& v, float& g0, float& g1, float& h0, float& h1) {     // 32 channels: 2 per group
  g0 += v[0] + v[1]; g1 += v[2] + v[3];
  h0 = __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], h0)); h1 = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], h1));
}
__device__ __forceinline__ void stat_quad(const f32x4& v, float& s1, float& s2) {                            // 64 / 128 channels
  s1 += (v[0] + v[1]) + (v[2] + v[3]);
  s2 = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], s2))));
}

// 32 channels: s1 / s2 = the lane's four partial sums, groups (r >> 1) * 8 + q * 2 + (r & 1) (r >> 1 = cout tile)
__device__ __forceinline__ void finish32(float (&s1)[4], float (&s2)[4], char* lds, int gb, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + XGB_OFF) + gb;
  const float* beta = gamma + 32;
  float* part = reinterpret_cast<float*>(lds + XPART_OFF);
  float* coef = reinterpret_cast<float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float a = row16_sum(s1[r]), c = row16_sum(s2[r]);
    if (r16 == 0) *reinterpret_cast<float2*>(&part[(wave * 16 + (r >> 1) * 8 + q * 2 + (r & 1)) * 2]) = make_float2(a, c);
  }
  lds_barrier();
  if (tid < 32) {
    const int g = tid >> 1;
    double sum = 0.0, sq = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) { const float2 v = *reinterpret_cast<const float2*>(&part[(w * 16 + g) * 2]); sum += v.x; sq += v.y; }
    gn_coef(sum, sq, 1.0 / 2048.0, gamma[tid], beta[tid], coef[tid], coef[128 + tid]);
  }
  lds_barrier();
}
// 64 channels: the lane's sums are those of group (cout tile) * 4 + q over the wave's 8 rows (other half: wave + 4);
// 128 channels: of half a group, (cout tile = wave) * 2 + (q >> 1)
template <int NCH>
__device__ __forceinline__ void finish_q(float s1, float s2, char* lds, int gb, int tid, int wave, int r16, int q) {
  const float* gamma = reinterpret_cast<const float*>(lds + XGB_OFF) + gb;
  const float* beta = gamma + NCH;
  float2* part = reinterpret_cast<float2*>(lds + XPART_OFF);
  float* coef = reinterpret_cast<float*>(lds + XCOEF_OFF);
  const float a = row16_sum(s1), c = row16_sum(s2);
  if (r16 == 0) part[wave * 4 + q] = make_float2(a, c);
  lds_barrier();
  if (tid < NCH) {
    float2 u, v;
    if (NCH == 64) { const int g = tid >> 2, ct = g >> 2, qq = g & 3; u = part[ct * 4 + qq]; v = part[(ct + 4) * 4 + qq]; }
    else { const int w = tid >> 4, qq = (tid >> 2) & 2; u = part[w * 4 + qq]; v = part[w * 4 + qq + 1]; }
    gn_coef((double)u.x + (double)v.x, (double)u.y + (double)v.y, NCH == 64 ? 1.0 / 1024.0 : 1.0 / 512.0, gamma[tid], beta[tid],
            coef[tid], coef[128 + tid]);
  }
  lds_barrier();
}

// y = [relu](raw * scale + shift [+ res]) of the lane's four channels c0 .. c0 + 3 -> hi / lo frames at `ad` (8 bytes each);
// MODE 0: relu, no residual; 1: + res, relu, result becomes the new residual; 2: no relu, result -> res only (downsample branch)
template <int MODE>
__device__ __forceinline__ void apply4(const f32x4& raw, f32x4& res, const float* coef, int c0, char* lds, int ad) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    v[r] = raw[r] * coef[c0 + r] + coef[128 + c0 + r];
    if (MODE == 1) v[r] += res[r];
    if (MODE != 2) v[r] = fmaxf(v[r], 0.f);
  }
  if (MODE != 0) res = (f32x4){v[0], v[1], v[2], v[3]};
  if (MODE != 2) {
    uint2 h, l;
    split4(v, h, l);
    *reinterpret_cast<uint2*>(lds + ad) = h;
    *reinterpret_cast<uint2*>(lds + PLANE + ad) = l;
  }
}

// fragment-order weights (avlen_conv::w16f): [cout tile][k-step][lane][8]
__device__ __forceinline__ bf16x8 wfrag(const bf16* __restrict__ w, int ct, int ksteps, int i, int lane) {
  return *reinterpret_cast<const bf16x8*>(w + ((long)(ct * ksteps + i) * 64 + lane) * 8);
}

// The stride-1 convs below share one schedule.  Per tap: the weight fragments of the NEXT tap are requested first (two register
// sets, the tap loop unrolled by two: no copies, no wait at the loop end), then the tap's tiles are processed in groups with the
// LDS reads of group g + 1 in flight under the MFMAs of group g (sched_barrier pins "reads issued, then MFMAs": left alone the
// scheduler issued two reads, waited, one MFMA).  Waves 4-7 enter the tap loop half a tap late (AVLEN_X3_STAGGER): their read
// bursts fall under the MFMA segments of their SIMD partners instead of colliding with them.
#define X3_PIN() __builtin_amdgcn_sched_barrier(0)

// ---- 32 channels @ 32 x 32: wave owns rows oy(rr) = 16 (rr >> 1) + 2 wave + (rr & 1), both column tiles, both cout tiles
__device__ __forceinline__ int row32(int wave, int rr) { return 16 * (rr >> 1) + 2 * wave + (rr & 1); }
template <int MODE>
__device__ __forceinline__ void conv32(const bf16* __restrict__ wh, const bf16* __restrict__ wl, int gb, char* lds, f32x4 (&res)[16],
                                       int tid, int wave, int lane, int r16, int q) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto loadw = [&](bf16x8 (&D)[4], int tap) {             // [hi ct0, hi ct1, lo ct0, lo ct1]
    D[0] = wfrag(wh, 0, 9, tap, lane); D[1] = wfrag(wh, 1, 9, tap, lane);
    D[2] = wfrag(wl, 0, 9, tap, lane); D[3] = wfrag(wl, 1, 9, tap, lane);
  };
  // group g = tile (rr = g >> 1, pt = g & 1): 2 reads, 6 MFMAs (64 + 64 accumulator / residual registers leave room for no more)
  auto body = [&](const bf16x8 (&W)[4], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    int ad[8];
#pragma unroll
    for (int g = 0; g < 8; g++) ad[g] = a32(row32(wave, g >> 1) + ky, (g & 1) * 16 + r16 + kx, q);
    bf16x8 F[2][2];                                       // [set][hi, lo]
    auto rd = [&](int g, bf16x8 (&D)[2]) {
      D[0] = *reinterpret_cast<const bf16x8*>(lds + ad[g]); D[1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad[g]);
    };
    auto mm = [&](int g, const bf16x8 (&D)[2]) {
      acc[2 * g] = mma3(W[0], W[2], D[0], D[1], acc[2 * g]);
      acc[2 * g + 1] = mma3(W[1], W[3], D[0], D[1], acc[2 * g + 1]);
    };
    rd(0, F[0]);
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g + 1 < 8) rd(g + 1, F[(g + 1) & 1]);
      X3_PIN();
      mm(g, F[g & 1]);
    }
  };
  bf16x8 W0[4], W1[4];
  loadw(W0, 0);
  if (AVLEN_X3_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(AVLEN_X3_STAGGER);
#pragma unroll 1
  for (int t2 = 0; t2 < 8; t2 += 2) {
    loadw(W1, t2 + 1); X3_PIN();
    body(W0, t2);
    loadw(W0, t2 + 2); X3_PIN();
    body(W1, t2 + 1);
  }
  body(W0, 8);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; i++) { const int ct = i & 1; stat_pair(acc[i], s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]); }
  finish32(s1, s2, lds, gb, tid, wave, r16, q);
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int rr = 0; rr < 4; rr++)
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        const int ti = (rr * 2 + pt) * 2 + ct;
        apply4<MODE>(acc[ti], res[ti], coef, ct * 16 + q * 4, lds, a32(row32(wave, rr) + 1, pt * 16 + r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
      }
  lds_barrier();
}

// ---- 64 channels @ 16 x 16: wave owns cout tile wave & 3 and rows 8 (wave >> 2) .. + 7
template <int MODE>
__device__ __forceinline__ void conv64(const bf16* __restrict__ wh, const bf16* __restrict__ wl, int gb, char* lds, f32x4 (&res)[8],
                                       int tid, int wave, int lane, int r16, int q) {
  const int ct = wave & 3, half = wave >> 2;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto loadw = [&](bf16x8 (&D)[4], int tap) {             // [hi hf0, hi hf1, lo hf0, lo hf1]
    D[0] = wfrag(wh, ct, 18, tap * 2, lane); D[1] = wfrag(wh, ct, 18, tap * 2 + 1, lane);
    D[2] = wfrag(wl, ct, 18, tap * 2, lane); D[3] = wfrag(wl, ct, 18, tap * 2 + 1, lane);
  };
  // group g = rows 2 g, 2 g + 1 (8 reads, 12 MFMAs)
  auto body = [&](const bf16x8 (&W)[4], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int ad0 = a64(half * 8 + ky, r16 + kx, q), ad1 = a64(half * 8 + ky, r16 + kx, 4 + q);      // + rr rows: the swizzle does not depend on the row
    bf16x8 F[2][8];                                       // [set][row j: hf0 hi, hf0 lo, hf1 hi, hf1 lo]
    auto rd = [&](int g, bf16x8 (&D)[8]) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int o = (2 * g + j) * (R64 * 128);
        D[4 * j] = *reinterpret_cast<const bf16x8*>(lds + ad0 + o); D[4 * j + 1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad0 + o);
        D[4 * j + 2] = *reinterpret_cast<const bf16x8*>(lds + ad1 + o); D[4 * j + 3] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad1 + o);
      }
    };
    auto mm = [&](int g, const bf16x8 (&D)[8]) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        acc[2 * g + j] = mma3(W[0], W[2], D[4 * j], D[4 * j + 1], acc[2 * g + j]);
        acc[2 * g + j] = mma3(W[1], W[3], D[4 * j + 2], D[4 * j + 3], acc[2 * g + j]);
      }
    };
    rd(0, F[0]);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g + 1 < 4) rd(g + 1, F[(g + 1) & 1]);
      X3_PIN();
      mm(g, F[g & 1]);
    }
  };
  bf16x8 W0[4], W1[4];
  loadw(W0, 0);
  if (AVLEN_X3_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(AVLEN_X3_STAGGER);
#pragma unroll 1
  for (int t2 = 0; t2 < 8; t2 += 2) {
    loadw(W1, t2 + 1); X3_PIN();
    body(W0, t2);
    loadw(W0, t2 + 2); X3_PIN();
    body(W1, t2 + 1);
  }
  body(W0, 8);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int rr = 0; rr < 8; rr++) stat_quad(acc[rr], s1, s2);
  finish_q<64>(s1, s2, lds, gb, tid, wave, r16, q);
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int rr = 0; rr < 8; rr++)
    apply4<MODE>(acc[rr], res[rr], coef, ct * 16 + q * 4, lds, a64(half * 8 + rr + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
  lds_barrier();
}

// ---- 128 channels @ 8 x 8: wave owns cout tile `wave` and the four column tiles (rows 2 pt, 2 pt + 1; lane: row r16 >> 3, pixel r16 & 7)
template <int MODE>
__device__ __forceinline__ void conv128(const bf16* __restrict__ wh, const bf16* __restrict__ wl, int gb, char* lds, f32x4 (&res)[4],
                                        int tid, int wave, int lane, int r16, int q) {
  const int ct = wave, ly = r16 >> 3, lx = r16 & 7;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto loadw = [&](bf16x8 (&D)[8], int tap) {             // [hi j0..3, lo j0..3]
#pragma unroll
    for (int j = 0; j < 4; j++) { D[j] = wfrag(wh, ct, 36, tap * 4 + j, lane); D[4 + j] = wfrag(wl, ct, 36, tap * 4 + j, lane); }
  };
  // group g = column tile pt = g (8 reads, 12 MFMAs); + 2 rows per tile: the row parity (the swizzle's top bit) does not change
  auto body = [&](const bf16x8 (&W)[8], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    int ad[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ad[j] = a128(ly + ky, lx + kx, 4 * j + q);
    bf16x8 F[2][8];                                       // [set][j hi x4, j lo x4]
    auto rd = [&](int g, bf16x8 (&D)[8]) {
      const int o = 2 * g * (R128 * 256);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        D[j] = *reinterpret_cast<const bf16x8*>(lds + ad[j] + o); D[4 + j] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad[j] + o);
      }
    };
    auto mm = [&](int g, const bf16x8 (&D)[8]) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc[g] = mma3(W[j], W[4 + j], D[j], D[4 + j], acc[g]);
    };
    rd(0, F[0]);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g + 1 < 4) rd(g + 1, F[(g + 1) & 1]);
      X3_PIN();
      mm(g, F[g & 1]);
    }
  };
  bf16x8 W0[8], W1[8];
  loadw(W0, 0);
  if (AVLEN_X3_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(AVLEN_X3_STAGGER);
#pragma unroll 1
  for (int t2 = 0; t2 < 8; t2 += 2) {
    loadw(W1, t2 + 1); X3_PIN();
    body(W0, t2);
    loadw(W0, t2 + 2); X3_PIN();
    body(W1, t2 + 1);
  }
  body(W0, 8);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int pt = 0; pt < 4; pt++) stat_quad(acc[pt], s1, s2);
  finish_q<128>(s1, s2, lds, gb, tid, wave, r16, q);
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
#pragma unroll
  for (int pt = 0; pt < 4; pt++)
    apply4<MODE>(acc[pt], res[pt], coef, ct * 16 + q * 4, lds, a128(2 * pt + ly + 1, lx + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
  lds_barrier();
}

template <int CH>                                          // zero the one-pixel frame of the CH-channel image (both planes)
__device__ __forceinline__ void zero_frame(char* lds, int tid) {
  constexpr int R = CH == 32 ? R32 : CH == 64 ? R64 : R128, PB = CH * 2, CPP = PB / 16;     // chunks per pixel
  constexpr int NROW = 2 * R * CPP, NCOL = (R - 2) * 2 * CPP;
  for (int i = tid; i < NROW + NCOL; i += RTH) {
    int off;
    if (i < NROW) off = ((i / (R * CPP)) * (R - 1) * R) * PB + (i % (R * CPP)) * 16;
    else { const int j = i - NROW, row = 1 + j / (2 * CPP), rem = j % (2 * CPP); off = (row * R + (rem / CPP) * (R - 1)) * PB + (rem % CPP) * 16; }
    *reinterpret_cast<uint4*>(lds + off) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(lds + PLANE + off) = make_uint4(0u, 0u, 0u, 0u);
  }
}

__device__ __forceinline__ void rest_body(const RestArgs& args, int g, int img, char* lds, long long* prof, int item) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                             // as in l1_body
  const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const RestTower& t = args.t[g];
  const float* coef = reinterpret_cast<const float*>(lds + XCOEF_OFF);
  // ---- GroupNorm affine parameters -> LDS
  for (int i = tid; i < 5 * 64 + 5 * 128 + 5 * 256; i += RTH) {
    int n, j, nch;
    if (i < 320) { n = i >> 6; j = i & 63; nch = 32; }
    else if (i < 960) { n = 5 + ((i - 320) >> 7); j = (i - 320) & 127; nch = 64; }
    else { n = 10 + ((i - 960) >> 8); j = (i - 960) & 255; nch = 128; }
    reinterpret_cast<float*>(lds + XGB_OFF)[i] = j < nch ? t.g[n][j] : t.b[n][j - nch];
  }
  lds_barrier();
  X3_STAMP(0);

  // =========================================================== layer 2 entry ===========================================================
  // in = relu(GN(x) + r) (64 x 64 x 16) read in two half-image passes; 3x3 stride-2 conv 16 -> 32 and the 1x1 stride-2 downsample.
  // Wave owns output rows 16 h + 2 wave, + 1 of half h (the row mapping of conv32), both column tiles, both cout tiles.
  f32x4 raw2[16], res2[16];
  {
    const bf16* __restrict__ wAh = t.wh[1]; const bf16* __restrict__ wAl = t.wl[1];          // [32][9][16]
    const bf16* __restrict__ wDh = t.wh[0]; const bf16* __restrict__ wDl = t.wl[0];          // [32][16]
    // weight fragments are fetched per k-step (2 taps x 16 channels) inside each half: [hi ct0, hi ct1, lo ct0, lo ct1]
    auto loadA = [&](bf16x8 (&W)[4], int s) {
      const int k = 32 * s + 8 * q;
#pragma unroll
      for (int ct = 0; ct < 2; ct++) {
        W[ct] = k < 144 ? *reinterpret_cast<const bf16x8*>(wAh + (long)(ct * 16 + r16) * 144 + k) : zero_frag();
        W[2 + ct] = k < 144 ? *reinterpret_cast<const bf16x8*>(wAl + (long)(ct * 16 + r16) * 144 + k) : zero_frag();
      }
    };
    bf16x8 WD[4];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      WD[ct] = (q >> 1) == 0 ? *reinterpret_cast<const bf16x8*>(wDh + (long)(ct * 16 + r16) * 16 + 8 * (q & 1)) : zero_frag();
      WD[2 + ct] = (q >> 1) == 0 ? *reinterpret_cast<const bf16x8*>(wDl + (long)(ct * 16 + r16) * 16 + 8 * (q & 1)) : zero_frag();
    }
    const bf16* __restrict__ act = t.a + (long)img * 2 * APLANE;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      // frame row fr <-> image row 32 h - 1 + fr, fr = 0 .. 33; frame column = image column + 1; borders come from the zero page
      l1_load_half<PLANE>(act, h, lds, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      X3_STAMP(1 + 2 * h);
      const int rdD = h16(4 * wave + 1, 2 * r16 + 1, q & 1);
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          const int ti = ((2 * h + j) * 2 + pt) * 2, o = (2 * j * HCOLS + 32 * pt) * 32;
          raw2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f}; raw2[ti + 1] = raw2[ti];
          const bf16x8 dh = *reinterpret_cast<const bf16x8*>(lds + rdD + o), dl = *reinterpret_cast<const bf16x8*>(lds + PLANE + rdD + o);
          res2[ti] = mma3(WD[0], WD[2], dh, dl, (f32x4){0.f, 0.f, 0.f, 0.f});
          res2[ti + 1] = mma3(WD[1], WD[3], dh, dl, (f32x4){0.f, 0.f, 0.f, 0.f});
        }
      bf16x8 WA[2][4];
      loadA(WA[0], 0);
#pragma unroll
      for (int s = 0; s < 5; s++) {
        if (s + 1 < 5) loadA(WA[(s + 1) & 1], s + 1);
        int tap = 2 * s + (q >> 1);
        if (tap > 8) tap = 8;                             // zero weights there
        const int ky = tap / 3, kx = tap - ky * 3;
        const int rdA = h16(4 * wave + ky, 2 * r16 + kx, q & 1);            // output row 16 h + 2 wave + j reads frame rows 2 (2 wave + j) + ky
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int pt = 0; pt < 2; pt++) {
            const int ti = ((2 * h + j) * 2 + pt) * 2, o = (2 * j * HCOLS + 32 * pt) * 32;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(lds + rdA + o), xl = *reinterpret_cast<const bf16x8*>(lds + PLANE + rdA + o);
            raw2[ti] = mma3(WA[s & 1][0], WA[s & 1][2], xh, xl, raw2[ti]);
            raw2[ti + 1] = mma3(WA[s & 1][1], WA[s & 1][3], xh, xl, raw2[ti + 1]);
          }
      }
      lds_barrier();                                      // every wave is done with this half before it is overwritten
      X3_STAMP(2 + 2 * h);
    }
  }
  {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; i++) { const int ct = i & 1; stat_pair(res2[i], s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]); }
    finish32(s1, s2, lds, 0, tid, wave, r16, q);                          // downsample norm (no ReLU) -> residual
#pragma unroll
    for (int i = 0; i < 16; i++) { f32x4 dummy = res2[i]; apply4<2>(dummy, res2[i], coef, (i & 1) * 16 + q * 4, lds, 0); }
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; i++) { const int ct = i & 1; stat_pair(raw2[i], s1[ct * 2], s1[ct * 2 + 1], s2[ct * 2], s2[ct * 2 + 1]); }
    finish32(s1, s2, lds, 64, tid, wave, r16, q);                         // block 0 bn1 + ReLU -> the 32-channel frame
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int pt = 0; pt < 2; pt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          f32x4 dummy;
          apply4<0>(raw2[(rr * 2 + pt) * 2 + ct], dummy, coef, ct * 16 + q * 4, lds,
                    a32(row32(wave, rr) + 1, pt * 16 + r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
        }
    zero_frame<32>(lds, tid);
  }
  lds_barrier();
  X3_STAMP(5);
  conv32<1>(t.wh[2], t.wl[2], 128, lds, res2, tid, wave, lane, r16, q);                // block 0 conv2 + skip
  X3_STAMP(6);
  conv32<0>(t.wh[3], t.wl[3], 192, lds, res2, tid, wave, lane, r16, q);                // block 1 conv1
  X3_STAMP(7);
  conv32<1>(t.wh[4], t.wl[4], 256, lds, res2, tid, wave, lane, r16, q);                // block 1 conv2 + identity
  X3_STAMP(8);

  // =========================================================== layer 3 ===========================================================
  f32x4 res3[8];
  {
    const int ct = wave & 3, half = wave >> 2;
    f32x4 raw3[8];
    const bf16x8 wDh = wfrag(t.wh[5], ct, 1, 0, lane), wDl = wfrag(t.wl[5], ct, 1, 0, lane);
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      const int ad = a32(2 * (half * 8 + rr) + 1, 2 * r16 + 1, q);
      res3[rr] = mma3(wDh, wDl, *reinterpret_cast<const bf16x8*>(lds + ad), *reinterpret_cast<const bf16x8*>(lds + PLANE + ad),
                      (f32x4){0.f, 0.f, 0.f, 0.f});
      raw3[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto body3 = [&](const bf16x8& Wh, const bf16x8& Wl, int tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int ad = a32(2 * (half * 8) + ky, 2 * r16 + kx, q);               // + 2 rr rows: the swizzle does not depend on the row
      bf16x8 F[8][2];
#pragma unroll
      for (int rr = 0; rr < 8; rr++) {
        F[rr][0] = *reinterpret_cast<const bf16x8*>(lds + ad + 2 * rr * (R32 * 64));
        F[rr][1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad + 2 * rr * (R32 * 64));
      }
      X3_PIN();
#pragma unroll
      for (int rr = 0; rr < 8; rr++) raw3[rr] = mma3(Wh, Wl, F[rr][0], F[rr][1], raw3[rr]);
    };
    bf16x8 W0h = wfrag(t.wh[6], ct, 9, 0, lane), W0l = wfrag(t.wl[6], ct, 9, 0, lane), W1h, W1l;
#pragma unroll 1
    for (int t2 = 0; t2 < 8; t2 += 2) {
      W1h = wfrag(t.wh[6], ct, 9, t2 + 1, lane); W1l = wfrag(t.wl[6], ct, 9, t2 + 1, lane); X3_PIN();
      body3(W0h, W0l, t2);
      W0h = wfrag(t.wh[6], ct, 9, t2 + 2, lane); W0l = wfrag(t.wl[6], ct, 9, t2 + 2, lane); X3_PIN();
      body3(W1h, W1l, t2 + 1);
    }
    body3(W0h, W0l, 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) stat_quad(res3[rr], s1, s2);
    finish_q<64>(s1, s2, lds, 320, tid, wave, r16, q);
#pragma unroll
    for (int rr = 0; rr < 8; rr++) { f32x4 dummy = res3[rr]; apply4<2>(dummy, res3[rr], coef, ct * 16 + q * 4, lds, 0); }
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) stat_quad(raw3[rr], s1, s2);
    finish_q<64>(s1, s2, lds, 320 + 128, tid, wave, r16, q);             // the barrier inside: every wave has left the 32-channel frame
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      f32x4 dummy;
      apply4<0>(raw3[rr], dummy, coef, ct * 16 + q * 4, lds, a64(half * 8 + rr + 1, r16 + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
    }
    zero_frame<64>(lds, tid);
  }
  lds_barrier();
  X3_STAMP(9);
  conv64<1>(t.wh[7], t.wl[7], 320 + 256, lds, res3, tid, wave, lane, r16, q);
  X3_STAMP(10);
  conv64<0>(t.wh[8], t.wl[8], 320 + 384, lds, res3, tid, wave, lane, r16, q);
  X3_STAMP(11);
  conv64<1>(t.wh[9], t.wl[9], 320 + 512, lds, res3, tid, wave, lane, r16, q);
  X3_STAMP(12);

  // =========================================================== layer 4 ===========================================================
  f32x4 res4[4];
  {
    const int ct = wave, ly = r16 >> 3, lx = r16 & 7;
    f32x4 raw4[4];
    bf16x8 wD[4];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) { wD[hf] = wfrag(t.wh[10], ct, 2, hf, lane); wD[2 + hf] = wfrag(t.wl[10], ct, 2, hf, lane); }
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int ad = a64(2 * (2 * pt + ly) + 1, 2 * lx + 1, 4 * hf + q);
        v = mma3(wD[hf], wD[2 + hf], *reinterpret_cast<const bf16x8*>(lds + ad), *reinterpret_cast<const bf16x8*>(lds + PLANE + ad), v);
      }
      res4[pt] = v;
      raw4[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto loadw = [&](bf16x8 (&D)[4], int tap) {
#pragma unroll
      for (int hf = 0; hf < 2; hf++) { D[hf] = wfrag(t.wh[11], ct, 18, tap * 2 + hf, lane); D[2 + hf] = wfrag(t.wl[11], ct, 18, tap * 2 + hf, lane); }
    };
    auto body4 = [&](const bf16x8 (&W)[4], int tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int ad0 = a64(2 * ly + ky, 2 * lx + kx, q), ad1 = a64(2 * ly + ky, 2 * lx + kx, 4 + q);
      bf16x8 F[4][4];
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        const int o = 4 * pt * (R64 * 128);
        F[pt][0] = *reinterpret_cast<const bf16x8*>(lds + ad0 + o); F[pt][1] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad0 + o);
        F[pt][2] = *reinterpret_cast<const bf16x8*>(lds + ad1 + o); F[pt][3] = *reinterpret_cast<const bf16x8*>(lds + PLANE + ad1 + o);
      }
      X3_PIN();
#pragma unroll
      for (int pt = 0; pt < 4; pt++) {
        raw4[pt] = mma3(W[0], W[2], F[pt][0], F[pt][1], raw4[pt]);
        raw4[pt] = mma3(W[1], W[3], F[pt][2], F[pt][3], raw4[pt]);
      }
    };
    bf16x8 W0[4], W1[4];
    loadw(W0, 0);
#pragma unroll 1
    for (int t2 = 0; t2 < 8; t2 += 2) {
      loadw(W1, t2 + 1); X3_PIN();
      body4(W0, t2);
      loadw(W0, t2 + 2); X3_PIN();
      body4(W1, t2 + 1);
    }
    body4(W0, 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) stat_quad(res4[pt], s1, s2);
    finish_q<128>(s1, s2, lds, 960, tid, wave, r16, q);
#pragma unroll
    for (int pt = 0; pt < 4; pt++) { f32x4 dummy = res4[pt]; apply4<2>(dummy, res4[pt], coef, ct * 16 + q * 4, lds, 0); }
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) stat_quad(raw4[pt], s1, s2);
    finish_q<128>(s1, s2, lds, 960 + 256, tid, wave, r16, q);
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      f32x4 dummy;
      apply4<0>(raw4[pt], dummy, coef, ct * 16 + q * 4, lds, a128(2 * pt + ly + 1, lx + 1, ct * 2 + (q >> 1)) + (q & 1) * 8);
    }
    zero_frame<128>(lds, tid);
  }
  lds_barrier();
  X3_STAMP(13);
  conv128<1>(t.wh[12], t.wl[12], 960 + 512, lds, res4, tid, wave, lane, r16, q);
  X3_STAMP(14);
  conv128<0>(t.wh[13], t.wl[13], 960 + 768, lds, res4, tid, wave, lane, r16, q);
  X3_STAMP(15);
  conv128<1>(t.wh[14], t.wl[14], 960 + 1024, lds, res4, tid, wave, lane, r16, q);
  X3_STAMP(16);
  // ---- layer-4 output (post ReLU) = the residual registers of the last block: NHWC fp32 (8, 8, 128)
  {
    const int ly = r16 >> 3, lx = r16 & 7;
    bf16* __restrict__ y = t.y + (long)img * 64 * 128;
#pragma unroll
    for (int pt = 0; pt < 4; pt++) {
      const float v[4] = {res4[pt][0], res4[pt][1], res4[pt][2], res4[pt][3]};
      uint2 hh, ll;
      split4(v, hh, ll);
      const long o = ((long)(2 * pt + ly) * 8 + lx) * 128 + wave * 16 + q * 4;
      *reinterpret_cast<uint2*>(y + o) = hh;
      *reinterpret_cast<uint2*>(y + args.y_lo + o) = ll;
    }
  }
}

// ========================================================================================================================
// One PERSISTENT launch for the whole tower group: a work queue of 2 n items (n = images x towers)
// ========================================================================================================================
// Both bodies need (almost) a whole CU's LDS, so a launch of n one-image workgroups runs in ceil(n / 256) rounds: the step's
// 6 towers x 64 images = 384 workgroups cost two rounds per kernel, the second half empty.  Here ~256 resident workgroups pull
// items from ONE queue: items [0, n) are the stem + layer-1 bodies, items [n, 2 n) the layer 2-4 bodies of the same images in the
// same order -- 768 half-sized items fill 256 CUs in three rounds instead of four.  A layer 2-4 item waits for its image's flag:
// that image's layer-1 item has a smaller ticket, so it is held by a workgroup that is already running and never waits itself --
// the queue cannot deadlock whatever part of the grid is resident.  Hand-off (cdna_hip_programming.md, Guideline 16): every storing
// wave drains its stores, the workgroup's barrier, ONE lane's agent-scope release, a relaxed agent-scope flag store; the consumer
// polls the flag relaxed, ONE lane's agent-scope acquire, barrier, then plain (LDS-DMA) loads.  The queue head and the flags are
// zeroed by a memset node in front of every launch.
struct TowerArgs { L1Args l1; RestArgs rest; unsigned* q; int B; int n; long long* prof; };   // q[0]: head, q[1]: give-up word, q[4 + i]: flags
constexpr int QSLOT_OFF = REST_LDS > L1_LDS ? REST_LDS : L1_LDS;
constexpr int TOWER_LDS = QSLOT_OFF + 16;
static_assert(TOWER_LDS <= 160 * 1024, "tower x3 LDS budget");
typedef __attribute__((address_space(1))) unsigned gu32;

// In-situ duration of the launch, for bench.py's roofline record (the kernel's time INSIDE the running rollout step, beside the
// other streams' work, without a profiler): the workgroup that draws ticket 0 stamps the constant-rate wall clock, the last workgroup
// to leave adds (its clock - that stamp) to a running sum.  Two atomics per workgroup and launch.
__device__ unsigned long long g_x3_t0, g_x3_ticks, g_x3_launches;

__global__ __launch_bounds__(RTH) void tower_x3_kernel(TowerArgs args) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  volatile int* slot = reinterpret_cast<volatile int*>(lds + QSLOT_OFF);
  gu32* qw = (gu32*)args.q;
  const int n = args.n;
  for (;;) {
    if (tid == 0) *slot = (int)__hip_atomic_fetch_add(qw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(*slot);
    __syncthreads();                                        // the slot is rewritten only after every wave has read it
    if (item == 0 && tid == 0) __hip_atomic_store(&g_x3_t0, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (item >= 2 * n) {
      if (tid == 0 && __hip_atomic_fetch_add(qw + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        const unsigned long long t0 = __hip_atomic_load(&g_x3_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&g_x3_ticks, (unsigned long long)wall_clock64() - t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&g_x3_launches, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      break;
    }
    X3_WALL(29);
    if (item < n) {
      l1_body(args.l1, item / args.B, item % args.B, lds, args.prof, item);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(qw + 4 + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const int j = item - n;
      if (tid == 0) {
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(qw + 4 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(32);
          // ~2^22 polls (seconds) without the producer: give up rather than hang the GPU -- the image's output is poisoned
          // below and the give-up word tells every later item to do the same at once
          if (++spins > (1u << 22) || __hip_atomic_load(qw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
        }
        if (!ok) { __hip_atomic_store(qw + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *slot = -1; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(*slot) == -1) {
        // the producer never arrived: this image's output is poisoned with NaN (both planes) instead of being left stale, and the
        // loop goes on -- every remaining layer 2-4 item sees the give-up word at its first poll and does the same, so after a
        // lost hand-off each output row is either valid or NaN (as the GRU sequence kernels and the CLIP tower do)
        bf16* y = args.rest.t[j / args.B].y + (long)(j % args.B) * 8192;
        for (int i = tid; i < 4096; i += RTH) {
          reinterpret_cast<unsigned*>(y)[i] = 0x7fc07fc0u;
          reinterpret_cast<unsigned*>(y + args.rest.y_lo)[i] = 0x7fc07fc0u;
        }
      } else {
        X3_WALL(30);
        rest_body(args.rest, j / args.B, j % args.B, lds, args.prof, item);
      }
    }
    X3_WALL(31);
    __syncthreads();                                        // LDS (and the slot) are free for the next item
  }
}

int g_reserved_cus = 0;
static int launch_tower_x3(TowerArgs& a, size_t q_bytes, hipStream_t stream) {
  static unsigned long long attr_done = 0;
  static int n_cu = 0;
  if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&tower_x3_kernel), TOWER_LDS, &attr_done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  if (!n_cu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  if (avlen_zero_bytes(a.q, q_bytes, stream) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  // one workgroup per CU (LDS): more would only queue behind them; fewer (avlen_set_tower_x3_reserved_cus) leave CUs to the
  // work of other streams for the whole launch -- a persistent workgroup never gives its CU back
  int cap = n_cu - g_reserved_cus;
  if (cap < 32) cap = n_cu < 32 ? n_cu : 32;
  const int grid = 2 * a.n < cap ? 2 * a.n : cap;
  hipLaunchKernelGGL(tower_x3_kernel, dim3(grid), dim3(RTH), TOWER_LDS, stream, a);
  return avlen_launch_status();
}
static inline size_t tower_x3_queue_bytes(int n) { return align_up(sizeof(unsigned) * (4 + (size_t)n), 256); }

}  // namespace

bool avlen_tower_x3_supported(const avlen_resnet18* n, int S, int C) {
  if (!n || S % 64 || S < 64 || C < 1 || C > 4) return false;
  const avlen_conv& k = n->conv1;
  if (!k.w16 || !k.w16lo || k.cin16 != 8 || k.cout != 16 || k.kh != 7 || k.kw != 7 || k.stride != 1 || k.pad != 3) return false;
  auto conv3 = [](const avlen_conv& c, int cin, int cout, int stride, bool frag) {
    return c.w16 && c.w16lo && (!frag || (c.w16f && c.w16flo)) && c.cin16 == cin && c.cout == cout && c.kh == 3 && c.kw == 3 &&
           c.stride == stride && c.pad == 1;
  };
  auto down = [](const avlen_conv& d, int cin, int cout, bool frag) {
    return d.w16 && d.w16lo && (!frag || (d.w16f && d.w16flo)) && d.cin16 == cin && d.cout == cout && d.kh == 1 && d.kw == 1 &&
           d.stride == 2 && d.pad == 0;
  };
  for (int i = 0; i < 2; i++)
    if (n->block[i].has_down || !conv3(n->block[i].conv1, 16, 16, 1, false) || !conv3(n->block[i].conv2, 16, 16, 1, false)) return false;
  int cin = 16;
  for (int l = 0; l < 3; l++) {
    const int co = 32 << l;
    const avlen_resblock& b0 = n->block[2 + 2 * l]; const avlen_resblock& b1 = n->block[3 + 2 * l];
    if (!b0.has_down || b1.has_down || !down(b0.down, cin, co, l > 0) || !conv3(b0.conv1, cin, co, 2, l > 0) ||
        !conv3(b0.conv2, co, co, 1, true) || !conv3(b1.conv1, co, co, 1, true) || !conv3(b1.conv2, co, co, 1, true)) return false;
    cin = co;
  }
  return true;
}

// scratch per tower and image: two activation images (hi + lo planes of 64 x 64 x 16 bf16)
size_t avlen_tower_x3_workspace_bytes(int groups, int B) {
  const size_t img = (size_t)APLANE * 2 * 2 * sizeof(bf16);
  return tower_x3_queue_bytes(B * groups) + (size_t)groups * (2 * 256 + (size_t)B * img) + 4096;
}

// Y[g] = layer-4 output NHWC (B, 8, 8, 128) of tower g as a compensated bf16 pair (hi plane, lo plane B * 8192 elements behind;
// the caller applies fc); imgs[g]: B (or, with row_index, more) images
int avlen_tower_x3_fwd(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                       const float* divisors, const int* row_index, void* const* Y, int groups, int B, int S, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
  if (groups < 1 || groups > 8 || B <= 0 || ws_bytes < avlen_tower_x3_workspace_bytes(groups, B)) return AVLEN_ERR_WS;
  WsBump w(ws, ws_bytes);
  const int n = B * groups;
  TowerArgs a = {};
  a.q = w.take<unsigned>(tower_x3_queue_bytes(n) / sizeof(unsigned));      // first: the memset block starts at the workspace's start
  a.B = B; a.n = n;
  a.l1.row_index = row_index; a.l1.S = S;
  a.rest.y_lo = (long)B * 8192;
  for (int g = 0; g < groups; g++) {
    const avlen_resnet18* nn = nets[g];
    if (!avlen_tower_x3_supported(nn, S, channels[g])) return AVLEN_ERR_ARG;
    // a0 = relu(GN0(stem)); block 0: a1 = relu(GN1(conv1(a0))), a0 = relu(GN2(conv2(a1)) + a0) [in place: the residual is what
    // a0 held]; block 1: a1 = relu(GN3(conv1(a0))), a0 = relu(GN4(conv2(a1)) + a0) = the layer-1 output
    L1Tower& t = a.l1.t[g];
    t.img = imgs[g]; t.u8 = img_u8 ? img_u8[g] : 0; t.C = channels[g]; t.div = divisors[g];
    const avlen_conv* cv[5] = {&nn->conv1, &nn->block[0].conv1, &nn->block[0].conv2, &nn->block[1].conv1, &nn->block[1].conv2};
    const avlen_affine* gn[5] = {&nn->bn1, &nn->block[0].bn1, &nn->block[0].bn2, &nn->block[1].bn1, &nn->block[1].bn2};
    for (int i = 0; i < 5; i++) { t.wh[i] = (const bf16*)cv[i]->w16; t.wl[i] = (const bf16*)cv[i]->w16lo; t.g[i] = gn[i]->g; t.b[i] = gn[i]->b; }
    t.a0 = w.take<bf16>((size_t)B * 2 * APLANE);
    t.a1 = w.take<bf16>((size_t)B * 2 * APLANE);
    RestTower& r = a.rest.t[g];
    r.a = t.a0; r.y = (bf16*)Y[g];
    for (int l = 0; l < 3; l++) {
      const avlen_resblock& b0 = nn->block[2 + 2 * l]; const avlen_resblock& b1 = nn->block[3 + 2 * l];
      const int o = 5 * l;
      const bool frag = l > 0;
      r.wh[o] = (const bf16*)(frag ? b0.down.w16f : b0.down.w16); r.wl[o] = (const bf16*)(frag ? b0.down.w16flo : b0.down.w16lo);
      r.wh[o + 1] = (const bf16*)(frag ? b0.conv1.w16f : b0.conv1.w16); r.wl[o + 1] = (const bf16*)(frag ? b0.conv1.w16flo : b0.conv1.w16lo);
      r.wh[o + 2] = (const bf16*)b0.conv2.w16f; r.wl[o + 2] = (const bf16*)b0.conv2.w16flo;
      r.wh[o + 3] = (const bf16*)b1.conv1.w16f; r.wl[o + 3] = (const bf16*)b1.conv1.w16flo;
      r.wh[o + 4] = (const bf16*)b1.conv2.w16f; r.wl[o + 4] = (const bf16*)b1.conv2.w16flo;
      r.g[o] = b0.bnd.g; r.b[o] = b0.bnd.b; r.g[o + 1] = b0.bn1.g; r.b[o + 1] = b0.bn1.b; r.g[o + 2] = b0.bn2.g; r.b[o + 2] = b0.bn2.b;
      r.g[o + 3] = b1.bn1.g; r.b[o + 3] = b1.bn1.b; r.g[o + 4] = b1.bn2.g; r.b[o + 4] = b1.bn2.b;
    }
  }
  if (!w.ok()) return AVLEN_ERR_WS;
  return launch_tower_x3(a, tower_x3_queue_bytes(n), stream);
}

// CUs the persistent tower launch leaves free (default 0).  The CLIP text tower of the same rollout step runs on another stream:
// with every CU held by a tower workgroup it cannot start before the towers end; 64 reserved CUs cost the towers a fourth round
// (0.50 -> 0.62 ms) and bring the step's text tower forward by more (step 1.97 -> 1.87 ms).
extern "C" void avlen_set_tower_x3_reserved_cus(int n) { g_reserved_cus = n > 0 ? n : 0; }

// Mean in-situ duration (microseconds) of the persistent tower launches since the last reset, and their count; reset != 0 clears
// the counters afterwards.  SYNCHRONISES the device (a measurement call: bench.py, never the rollout).
extern "C" int avlen_tower_x3_timing(double* mean_us, long long* launches, int reset) {
  unsigned long long ticks = 0, n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return AVLEN_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(&ticks, HIP_SYMBOL(g_x3_ticks), 8) != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_x3_launches), 8) != hipSuccess)
    return AVLEN_ERR_LAUNCH;
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  if (mean_us) *mean_us = n ? (double)ticks / (double)n / ((double)khz * 1e-3) : 0.0;
  if (launches) *launches = (long long)n;
  if (reset) {
    const unsigned long long z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_x3_ticks), &z, 8) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_x3_launches), &z, 8) != hipSuccess)
      return AVLEN_ERR_LAUNCH;
  }
  return AVLEN_OK;
}
