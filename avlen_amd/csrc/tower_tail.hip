// Layers 3 and 4 of the CustomResNet tower (smt_resnet.py:37-53, 100-131) as ONE launch: one workgroup per image keeps every
// activation of the four basic blocks in LDS (16x16x64 and 8x8x128 bf16 tensors: 32 / 16 KiB each), streams the ten convs'
// weights from L2 through a global_load_lds ring, takes the GroupNorm statistics from its own fp32 accumulators
// (deterministic, no atomics: the whole image is in the block) and applies normalisation / residual / ReLU in place.
// Replaces 10 implicit-GEMM launches + 10 GroupNorm-apply launches per tower group and all their HBM round trips: the
// image's layer-2 output (64 KiB) is read once, the layer-4 output (16 KiB) written once.
//
// Layout of an activation in LDS: [pixel][C] bf16, 16-byte chunks XOR-swizzled by the pixel index so that the 16 pixels of
// an MFMA row tile (consecutive, or every other one for the stride-2 convs) read conflict-free with ds_read_b128:
// physical chunk = chunk ^ ((pixel / (16 / CP)) & (CP - 1)), CP = C / 8 chunks per pixel.  Zero padding is a bounds test on
// the fragment read.  MFMA is issued transposed (W fragment first): a lane ends up with 4 consecutive channels of one
// pixel -> 8-byte LDS stores.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_tt[4096];     // K tails / surplus pieces of the weight tiles

#ifdef AVLEN_TT_LAB
__device__ long long g_tt_stamps[64];      // tools/tower_lab.hip: phase time stamps of block (0, 0)
__device__ int g_tt_n;
#define TT_STAMP() do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_tt_stamps[g_tt_n++] = clock64(); } while (0)
#else
#define TT_STAMP() do {} while (0)
#endif

namespace {

constexpr int NTH = 512, NW = 8;
constexpr int ACT = 32768;                    // one activation buffer (16x16x64 bf16)
constexpr int RING3 = 4 * ACT;                // layer-3 weight ring: 3 stages of [64][64] (8 KiB) behind the four buffers
constexpr int SCRATCH3 = RING3 + 3 * 8192;    // 1 KiB landing area of the layer-3 tiles' surplus pieces
constexpr int STATS = SCRATCH3 + 1024;        // per-wave channel partials [8 waves][64 channels][2] fp32 = 4 KiB
constexpr int COEF = STATS + 4096;            // scale / shift tables [4][128] fp32 = 2 KiB
constexpr int WTAB = COEF + 2048;             // the ten weight pointers in stream order (an indexed private array would
constexpr int LDS_BYTES = WTAB + 128;         // live in scratch: a VMEM load + vmcnt(0) per tile drains the whole ring)
static_assert(2 * ACT + 16384 + 4 * 16384 <= SCRATCH3 && 3 * ACT + 7 * 8192 <= SCRATCH3 && LDS_BYTES <= 160 * 1024,
              "tower tail LDS budget");

struct TailTower { const bf16* x; bf16* y; const bf16* w[10]; const float* g[10]; const float* b[10]; };
struct TailArgs { TailTower t[6]; };

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
template <int CP> __device__ __forceinline__ int swz(int pix) { return (pix / (16 / CP)) & (CP - 1); }

// The weight stream of one layer: five convs back to back, tiles of [COUT][64 k], 3 stages, two 1 KiB pieces per wave and
// tile (COUT = 64 fills only the first; the second then reads the zero page so that every tile counts the same in vmcnt).
template <int COUT, int NS, int K0, int K1, int K2>     // reduction lengths of the stream's convs: compile-time, or the
struct WStream {                                          // select below turns into an indexed load from a spilled struct
  const bf16* const* wt;                                   // weight pointers (LDS table)
  char* ring; char* scratch; const char* zero; int nconv, conv, kt, issued, consumed;
  __device__ __forceinline__ void init(char* r, char* scr, int n, const bf16* const* w, const char* z) {
    ring = r; scratch = scr; zero = z; nconv = n; conv = 0; kt = 0; issued = 0; consumed = 0; wt = w;
  }
  __device__ __forceinline__ void issue(int tid, int wave, int lane) {
    if (conv >= nconv) return;
    char* stage = ring + (issued % NS) * (COUT * 128);
    const int Kc = conv == 0 ? K0 : conv == 1 ? K1 : K2;
    const bf16* wc = wt[conv];                               // LDS read (lgkmcnt): does not touch the VMEM counter
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int piece = r * NW + wave;                       // 8 weight rows per piece
      const int row = piece * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
      const int kcol = kt * 64 + ch * 8;
      const bool ok = piece * 8 < COUT && kcol < Kc;
      const char* src = ok ? (const char*)(wc + (long)row * Kc + kcol) : (const char*)g_zero_page_tt + tid * 16;
      char* dst = piece * 8 < COUT ? stage + piece * 1024 : scratch;      // surplus piece (COUT = 64): a scratch KiB
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    issued++;
    if (++kt == (Kc + 63) / 64) { kt = 0; conv++; }
  }
  __device__ __forceinline__ void prime(int tid, int wave, int lane) {
    for (int i = 0; i < NS - 1; i++) issue(tid, wave, lane);
  }
  __device__ __forceinline__ const char* acquire() {                         // wait for the oldest tile in flight, make it visible to the block
    const int ahead = issued - consumed - 1;                 // tiles issued after it: 2 pieces each may stay outstanding
    if (ahead >= 6) wait_vmcnt<12>(); else if (ahead == 5) wait_vmcnt<10>(); else if (ahead == 4) wait_vmcnt<8>();
    else if (ahead == 3) wait_vmcnt<6>(); else if (ahead == 2) wait_vmcnt<4>(); else if (ahead == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    bar();
    const char* stage = ring + (consumed % NS) * (COUT * 128);
    consumed++;
    return stage;
  }
};

// One convolution out of LDS into LDS.  in: [HIN*HIN][CIN] swizzled; out: [HOUT*HOUT][COUT] swizzled (raw conv output, bf16);
// part: per-wave channel sums of the fp32 accumulators [NW][64 channels of the wave][2].
// Per-lane gather plan of a conv geometry: for every tap the LDS byte offset of this lane's input pixel (-1 when the tap falls
// into the padding) and the pixel's swizzle.  The K loop of conv_lds is fully unrolled, so the tap of every K-step is a
// compile-time index into these registers and a fragment read costs three VALU ops -- computed per K-step (div / mod /
// bounds / swizzle per lane) the address math alone made the conv VALU-bound at ~2k cycles per tile.  The three stride-1
// 3x3 convs of a stage share one plan.
template <int TAPS, int MI> struct Plan { int aoff[TAPS][MI], asw[TAPS][MI]; };
template <int CIN, int COUT, int HIN, int HOUT, int KS, int STRIDE>
__device__ __forceinline__ void make_plan(Plan<KS * KS, (HOUT * HOUT / 16) * (COUT / 16) / NW / 4>& pl, int tid) {
  constexpr int PAD = KS / 2, MI = (HOUT * HOUT / 16) * (COUT / 16) / NW / 4, CPI = CIN / 8;
  const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15;
  const int mt0 = MI == 2 ? wave * 2 : (wave & 3);
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const int p = (mt0 + i) * 16 + r16, oy = p / HOUT, ox = p % HOUT;
#pragma unroll
    for (int tp = 0; tp < KS * KS; tp++) {
      const int iy = oy * STRIDE + tp / KS - PAD, ix = ox * STRIDE + tp % KS - PAD;
      const bool ok = (unsigned)iy < (unsigned)HIN && (unsigned)ix < (unsigned)HIN;
      const int pix = iy * HIN + ix;
      pl.aoff[tp][i] = ok ? pix * (CIN * 2) : -1;
      pl.asw[tp][i] = swz<CPI>(pix);
    }
  }
}

template <int CIN, int COUT, int HIN, int HOUT, int KS, int STRIDE, class WS>
__device__ __forceinline__ void conv_lds(const char* in, char* out, WS& ws, float* part, int tid,
                                         const Plan<KS * KS, (HOUT * HOUT / 16) * (COUT / 16) / NW / 4>& pl) {
  constexpr int M = HOUT * HOUT, MT = M / 16, K = KS * KS * CIN, NKT = (K + 63) / 64, TAPS = KS * KS;
  constexpr int CPO = COUT / 8;
  constexpr int MI = MT * (COUT / 16) / NW / 4;             // m-tiles per wave; 4 n-tiles per wave
  static_assert(MI == 1 || MI == 2, "wave tiling");
  const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  // layer 3 (M = 256, N = 64): wave -> m-tiles 2w, 2w+1, n-tiles 0..3;  layer 4 (M = 64, N = 128): m-tile w & 3, n-tiles 4(w>>2)..
  const int mt0 = MI == 2 ? wave * 2 : (wave & 3), nt0 = MI == 2 ? 0 : (wave >> 2) * 4;
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (gather plan `pl`: see make_plan)
  const char* zero16 = ws.zero;                // a zeroed 16-byte chunk in LDS (padding taps, K tails)
#pragma unroll
  for (int kt = 0; kt < NKT; kt++) {
    const char* stage = ws.acquire();
    ws.issue(tid, wave, lane);
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      const int kbase = kt * 64 + kh * 32;                   // compile-time: tap and first channel of this K-step
      const int tp = kbase / CIN < TAPS ? kbase / CIN : TAPS - 1;
      const bool live = kbase < K;                           // K tail (layer-3 conv1: 4.5 tiles; 1x1 convs: half a tile)
      const int cbase = (kbase % CIN) / 8;                   // chunk of channel 0 of the step; this lane adds q4
      bf16x8 af[MI];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int o = pl.aoff[tp][i];
        const char* ap = (live && o >= 0) ? in + o + (((cbase + q4) ^ pl.asw[tp][i]) << 4) : zero16;
        af[i] = *reinterpret_cast<const bf16x8*>(ap);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int wrow = (nt0 + j) * 16 + r16;
        bf16x8 wf = *reinterpret_cast<const bf16x8*>(stage + wrow * 128 + (((kh * 4 + q4) ^ ((wrow >> 1) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < MI; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[i], acc[i][j], 0, 0, 0);
      }
    }
  }
  // lane (r16, q4) holds pixel (mt0+i)*16 + r16, channels (nt0+j)*16 + 4*q4 + r.  Channel sums over the wave's pixels
  // (reduce-scatter over the 16 pixel lanes), then the raw output as bf16.
  const bool hi = r16 & 8, hi2 = r16 & 4;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float v1[4], v2[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int i = 0; i < MI; i++) { const float v = acc[i][j][r]; a += v; b += v * v; }
      v1[r] = a; v2[r] = b;
    }
    float a0 = hi ? v1[2] : v1[0], a1 = hi ? v1[3] : v1[1], b0 = hi ? v1[0] : v1[2], b1 = hi ? v1[1] : v1[3];
    float c0 = hi ? v2[2] : v2[0], c1 = hi ? v2[3] : v2[1], d0 = hi ? v2[0] : v2[2], d1 = hi ? v2[1] : v2[3];
    a0 += __shfl_xor(b0, 8, 64); a1 += __shfl_xor(b1, 8, 64); c0 += __shfl_xor(d0, 8, 64); c1 += __shfl_xor(d1, 8, 64);
    float s1 = hi2 ? a1 : a0, t1 = hi2 ? a0 : a1, s2 = hi2 ? c1 : c0, t2 = hi2 ? c0 : c1;
    s1 += __shfl_xor(t1, 4, 64); s2 += __shfl_xor(t2, 4, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    const int ch = (nt0 + j) * 16 + q4 * 4 + (hi ? 2 : 0) + (hi2 ? 1 : 0);
    if ((r16 & 3) == 0) { part[(wave * 64 + ch - nt0 * 16) * 2] = s1; part[(wave * 64 + ch - nt0 * 16) * 2 + 1] = s2; }
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int pix = (mt0 + i) * 16 + r16, chn = (nt0 + j) * 16 + q4 * 4;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] = (bf16)acc[i][j][r];
      *reinterpret_cast<bf16x4*>(out + pix * (COUT * 2) + (((chn >> 3) ^ swz<CPO>(pix)) << 4) + (chn & 7) * 2) = o;
    }
  }
}

// GroupNorm(16) scale / shift of one raw tensor from the per-wave partial sums (deterministic order).
template <int COUT, int M, int MI>
__device__ __forceinline__ void gn_coeffs(const float* part, const float* gamma, const float* beta, float* sc, float* sh, int tid) {
  constexpr int CPG = COUT / 16;
  if (tid < 16) {
    float sum = 0.f, sq = 0.f;
    for (int c = tid * CPG; c < (tid + 1) * CPG; c++)
      for (int w = 0; w < NW; w++) {
        // layer 3: every wave covers all channels; layer 4: waves 0-3 cover channels 0-63, waves 4-7 channels 64-127
        if (MI == 1 && (w >> 2) != (c >> 6)) continue;
        sum += part[(w * 64 + (c & 63)) * 2]; sq += part[(w * 64 + (c & 63)) * 2 + 1];
      }
    const float inv_n = 1.f / (float)(M * CPG), mean = sum * inv_n;
    const float var = fmaxf(sq * inv_n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + 1e-5f);
    for (int c = tid * CPG; c < (tid + 1) * CPG; c++) {
      const float s = gamma[c] * rstd;
      sc[c] = s; sh[c] = beta[c] - mean * s;
    }
  }
}

// y = [relu]( x * sc + sh  [+ r * rsc + rsh | + r] ) in place on x, all tensors [M][C] with the same swizzle.
template <int C, int M>
__device__ __forceinline__ void gn_apply_lds(char* x, const float* sc, const float* sh, const char* res, const float* rsc, const float* rsh,
                             int relu, int tid) {
  constexpr int CP = C / 8;
  for (int i = tid; i < M * CP; i += NTH) {
    const int pix = i / CP, pc = i % CP, c0 = (pc ^ swz<CP>(pix)) * 8;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 16);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (float)v[e] * sc[c0 + e] + sh[c0 + e];
    if (res) {
      bf16x8 r = *reinterpret_cast<const bf16x8*>(res + i * 16);
#pragma unroll
      for (int e = 0; e < 8; e++) f[e] += rsc ? (float)r[e] * rsc[c0 + e] + rsh[c0 + e] : (float)r[e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (bf16)(relu ? fmaxf(f[e], 0.f) : f[e]);
    *reinterpret_cast<bf16x8*>(x + i * 16) = v;
  }
}

// One ResNet stage (two basic blocks, the first with stride 2 and a 1x1 downsample): input in `xin` ([HIN*HIN][CIN]),
// result in `bufB`.  Buffers A, B, C are [HOUT*HOUT][COUT]; xin may overlap C/D (it is dead after the first two convs).
template <int CIN, int COUT, int HIN, int HOUT, class WS1, class WS2>
__device__ __forceinline__ void stage(const char* xin, char* A, char* B, char* C, WS1& ws1, WS2& ws2, const TailTower& t, int base, float* part,
                      float* sc, float* sh, float* sc2, float* sh2, int tid) {
  constexpr int M = HOUT * HOUT, MI = (M / 16) * (COUT / 16) / NW / 4;
  const int wave = tid >> 6, lane = tid & 63;
  ws1.prime(tid, wave, lane);
  TT_STAMP();
  // block 0
  {
    Plan<9, MI> p1; make_plan<CIN, COUT, HIN, HOUT, 3, 2>(p1, tid);
    conv_lds<CIN, COUT, HIN, HOUT, 3, 2>(xin, A, ws1, part, tid, p1);             // conv1 (stride 2)
  }
  bar();
  TT_STAMP();
  gn_coeffs<COUT, M, MI>(part, t.g[base + 0], t.b[base + 0], sc, sh, tid);
  bar();
  {
    Plan<1, MI> pd; make_plan<CIN, COUT, HIN, HOUT, 1, 2>(pd, tid);
    conv_lds<CIN, COUT, HIN, HOUT, 1, 2>(xin, B, ws1, part, tid, pd);             // downsample (1x1, stride 2): xin dead after this
  }
  bar();
  TT_STAMP();
  ws2.prime(tid, wave, lane);                                                     // (its ring may overlap xin)
  gn_coeffs<COUT, M, MI>(part, t.g[base + 2], t.b[base + 2], sc2, sh2, tid);
  gn_apply_lds<COUT, M>(A, sc, sh, nullptr, nullptr, nullptr, 1, tid);            // a1 = relu(gn1(raw1))
  bar();
  TT_STAMP();
  Plan<9, MI> p3; make_plan<COUT, COUT, HOUT, HOUT, 3, 1>(p3, tid);                // shared by the three stride-1 convs
  conv_lds<COUT, COUT, HOUT, HOUT, 3, 1>(A, C, ws2, part, tid, p3);               // conv2
  bar();
  TT_STAMP();
  gn_coeffs<COUT, M, MI>(part, t.g[base + 1], t.b[base + 1], sc, sh, tid);
  bar();
  gn_apply_lds<COUT, M>(C, sc, sh, B, sc2, sh2, 1, tid);                          // out0 = relu(gn2(raw2) + gn_d(rawd))  -> C
  bar();
  TT_STAMP();
  // block 1
  conv_lds<COUT, COUT, HOUT, HOUT, 3, 1>(C, A, ws2, part, tid, p3);
  bar();
  gn_coeffs<COUT, M, MI>(part, t.g[base + 3], t.b[base + 3], sc, sh, tid);
  bar();
  gn_apply_lds<COUT, M>(A, sc, sh, nullptr, nullptr, nullptr, 1, tid);
  bar();
  conv_lds<COUT, COUT, HOUT, HOUT, 3, 1>(A, B, ws2, part, tid, p3);
  bar();
  gn_coeffs<COUT, M, MI>(part, t.g[base + 4], t.b[base + 4], sc, sh, tid);
  bar();
  gn_apply_lds<COUT, M>(B, sc, sh, C, nullptr, nullptr, 1, tid);                  // out1 = relu(gn2(raw) + out0)  -> B
  bar();
  TT_STAMP();
}

__global__ __launch_bounds__(NTH) void tower_tail_kernel(TailArgs args, int Bn) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TailTower& t = args.t[blockIdx.y];
  const int img = blockIdx.x;
  char* A = lds; char* B = lds + ACT; char* C = lds + 2 * ACT; char* D = lds + 3 * ACT;
  float* part = reinterpret_cast<float*>(lds + STATS);
  float* sc = reinterpret_cast<float*>(lds + COEF); float* sh = sc + 128; float* sc2 = sh + 128; float* sh2 = sc2 + 128;
  const bf16** wtab = reinterpret_cast<const bf16**>(lds + WTAB);
  if (tid >= 64 && tid < 68) reinterpret_cast<unsigned*>(lds + WTAB + 96)[tid - 64] = 0u;      // the zero chunk
  if (tid < 10) {                              // stream order: conv1, down, conv2, conv1', conv2' per layer
    const int r = tid % 5, o = (tid / 5) * 5 + (r == 1 ? 2 : r == 2 ? 1 : r);      // 0,2,1,3,4 | 5,7,6,8,9
    const bf16* wp = t.w[0];
#pragma unroll
    for (int i = 1; i < 10; i++) wp = o == i ? t.w[i] : wp;
    wtab[tid] = wp;
  }
  // ---- layer-2 output [32*32][32] (64 KiB) -> C..D, swizzled (CP = 4), by LDS-DMA: 64 pieces of 16 pixels
  {
    const char* x = (const char*)(t.x + (long)img * 32 * 32 * 32);
    for (int pc = wave; pc < 64; pc += NW) {
      const int pix = pc * 16 + (lane >> 2), chunk = (lane & 3) ^ swz<4>(pix);
      __builtin_amdgcn_global_load_lds((const void*)(x + pix * 64 + chunk * 16),
          (__attribute__((address_space(3))) void*)(C + pc * 1024), 16, 0, 0);
    }
    wait_vmcnt<0>();
  }
  bar();
  TT_STAMP();
  {
    // layer 3: the first two convs read the staged input (C..D) and stream through the 3-stage ring behind the buffers; once
    // the input is dead, buffer D joins the ring: 7 stages of 8 KiB -- 6 tiles (48 KiB) in flight, which is what it takes to
    // keep one CU's L2 -> LDS path busy (2 small tiles in flight ran this kernel 2.5x slower: latency-bound)
    WStream<64, 3, 288, 32, 32> w1; w1.init(lds + RING3, lds + SCRATCH3, 2, wtab, lds + WTAB + 96);
    WStream<64, 7, 576, 576, 576> w2; w2.init(D, lds + SCRATCH3, 3, wtab + 2, lds + WTAB + 96);
    stage<32, 64, 32, 16>(C, A, B, C, w1, w2, t, 0, part, sc, sh, sc2, sh2, tid);
  }
  {
    // layer 4 (16 KiB tensors: A, A + 16 KiB, lower half of C; input in B): ring of 4 x 16 KiB from the upper half of C on
    WStream<128, 4, 576, 64, 64> w1; w1.init(C + 16384, lds + SCRATCH3, 2, wtab + 5, lds + WTAB + 96);
    WStream<128, 4, 1152, 1152, 1152> w2; w2.init(C + 16384, lds + SCRATCH3, 3, wtab + 7, lds + WTAB + 96);
    stage<64, 128, 16, 8>(B, A, A + 16384, C, w1, w2, t, 5, part, sc, sh, sc2, sh2, tid);
  }
  // ---- layer-4 output (in A + 16 KiB): [64 pixels][128] -> global NHWC, un-swizzled
  {
    const char* o = A + 16384;
    bf16* y = t.y + (long)img * 64 * 128;
    for (int i = tid; i < 64 * 16; i += NTH) {
      const int pix = i >> 4, pc = i & 15, c0 = (pc ^ swz<16>(pix)) * 8;
      *reinterpret_cast<uint4*>(y + pix * 128 + c0) = *reinterpret_cast<const uint4*>(o + i * 16);
    }
  }
  (void)Bn; (void)D;
}

}  // namespace

// Layers 3 + 4 of `groups` (<= 6) towers: X[g] = layer-2 output NHWC bf16 (B, 32, 32, 32); Y[g] = layer-4 output NHWC bf16
// (B, 8, 8, 128).  Conv weights bf16 [Cout][kh][kw][Cin] (K contiguous) in block order {conv1, conv2, down} x2 per layer
// as in avlen_resblock; every tower must have the 32->64->128 channel plan.
int avlen_tower_tail_bf16(const avlen_resnet18* const* nets, const void* const* X, void* const* Y, int groups, int B,
                          hipStream_t stream) {
  if (groups < 1 || groups > 6 || B <= 0) return AVLEN_ERR_ARG;
  TailArgs a = {};
  for (int g = 0; g < groups; g++) {
    const avlen_resnet18* n = nets[g];
    TailTower& t = a.t[g];
    t.x = (const bf16*)X[g]; t.y = (bf16*)Y[g];
    for (int l = 0; l < 2; l++) {                          // l = 0: layer 3 (blocks 4, 5); l = 1: layer 4 (blocks 6, 7)
      const avlen_resblock& b0 = n->block[4 + 2 * l]; const avlen_resblock& b1 = n->block[5 + 2 * l];
      const int cin = l == 0 ? 32 : 64, co = l == 0 ? 64 : 128;
      if (!b0.has_down || b1.has_down || b0.conv1.cin16 != cin || b0.conv1.cout != co || b0.conv1.stride != 2 ||
          b0.conv2.cin16 != co || b0.down.cin16 != cin || b0.down.kh != 1 || b1.conv1.cin16 != co || b1.conv2.cout != co ||
          !b0.conv1.w16 || !b0.conv2.w16 || !b0.down.w16 || !b1.conv1.w16 || !b1.conv2.w16)
        return AVLEN_ERR_ARG;
      const int o = 5 * l;
      t.w[o + 0] = (const bf16*)b0.conv1.w16; t.w[o + 1] = (const bf16*)b0.conv2.w16; t.w[o + 2] = (const bf16*)b0.down.w16;
      t.w[o + 3] = (const bf16*)b1.conv1.w16; t.w[o + 4] = (const bf16*)b1.conv2.w16;
      t.g[o + 0] = b0.bn1.g; t.b[o + 0] = b0.bn1.b; t.g[o + 1] = b0.bn2.g; t.b[o + 1] = b0.bn2.b;
      t.g[o + 2] = b0.bnd.g; t.b[o + 2] = b0.bnd.b;
      t.g[o + 3] = b1.bn1.g; t.b[o + 3] = b1.bn1.b; t.g[o + 4] = b1.bn2.g; t.b[o + 4] = b1.bn2.b;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(tower_tail_kernel, dim3(B, groups), dim3(NTH), LDS_BYTES, stream, a, B);
  return avlen_launch_status();
}
