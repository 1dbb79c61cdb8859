// bf16-in-HBM MFMA GEMM / implicit-GEMM convolution for gfx950 (the rollout "perf mode" kernel).
//
//   C[m, n] = epilogue( sum_k A(m, k) * W(n, k) ),   A, W bf16 (K contiguous), accumulate fp32
//
// What differs from igemm.hip (fp32 operands, register staging):
//  * operands are already bf16 in HBM (producers emit bf16 copies), so tiles go HBM -> LDS with
//    `global_load_lds_dwordx4` (16 B per lane, no VGPR round trip, no conversion, no ds_write);
//  * BK = 64 (128-byte LDS rows), double-buffered LDS, ONE barrier per K-step: the loads of tile k+1 are in
//    flight while tile k is multiplied;
//  * the LDS image is lane-linear (a glds requirement), so the bank-conflict swizzle is applied to the per-lane
//    SOURCE address and undone on the fragment read: 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7);
//  * wave tile 64x64 (BN=128: waves 2x2) -> 32 MFMA per 16 ds_read_b128 per K-step;
//  * implicit-GEMM conv: the per-lane source address IS the gather (NHWC, Cin % 8 == 0, one 16-byte chunk = 8
//    channels of one tap); padding taps and K/M tails read a 16-byte zero page;
//  * small grids split K across blockIdx.z into fp32 slabs (deterministic reduce, no atomics).
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include <stdlib.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// zeros for padding taps / K tails / surplus rows: 16 KB so that the lanes of a block do not all hit one address
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4096];

namespace {

constexpr int BK = 64;

constexpr int MAXG = 8;
struct G2Grp { const bf16* A; const bf16* B; float* C32; bf16* C16; const float* bias; const float* residual; float* stats; };

struct G2 {
  G2Grp g[MAXG]; int groups;       // blockIdx.y selects the group (same shapes, different tensors: the 6 towers)
  const bf16* A; const bf16* B; float* C32; bf16* C16; const float* bias; const float* residual;
  int M, N, K;
  int lda, ldb, ldc32, ldc16, ldr;
  int act;
  int conv, H, W, Cin, cin_log2, OH, OW, KH, KW, kw_magic, stride, pad, stride_w;   // stride_w: 0 = same as stride
  int splitk, ksteps_per_split;
  int vec4;                   // N, ldc32, ldc16, ldr all multiples of 4: 16-byte / 8-byte epilogue accesses
  // LayerNorm folded into the GEMM (A = raw bf16 rows x, W = bf16(W * gamma)): C = rstd_m * (acc - mean_m * ln_s[n]) + bias[n]
  // with (sum, sum of squares) of row m in ln_stats[m][2] and mean/rstd over K columns; NULL -> plain epilogue
  const float* ln_stats; const float* ln_s;
  float* rowstats;            // optional output: rowstats[m][2] += (sum, sum of squares) of the final C row (N-tile partials)
  // Deterministic form of the two fields above: rs_slabs != 0 -> every N tile STORES its partial at
  // rowstats[(n_tile * M_alloc + m) * 2] (no atomics, nothing to zero); ln_slabs = s > 1 -> ln_stats holds s such slabs, added
  // in slab order by the consumer
  int rs_slabs, ln_slabs;
  float* slab;
  const int* M_dev;           // optional: the live row count (<= M) is read from device memory (ragged batches)
  float* stats; int ohw;      // optional GroupNorm statistics: stats[sample][0|1][N] += sum / sum of squares of C
  // 16-bit operand format: 0 = bf16, 1 = fp16 (IEEE half; v_mfma_f32_16x16x32_f16, C16 written as fp16)
  int f16;
  // Compensated bf16 ("bf16x3"): every operand is a PAIR of bf16 planes, hi = bf16(x) and lo = bf16(x - hi), the lo plane
  // `a_lo` / `b_lo` BYTES behind the hi plane (same layout).  The K loop runs three passes over K -- (A_lo, W_hi),
  // (A_hi, W_lo), (A_hi, W_hi) -- into the same fp32 accumulators: the product is exact to ~2^-17 per term instead of 2^-9.
  // c16_lo (elements): the bf16 output is written as a pair too, lo plane at C16 + c16_lo.  x3 == 0: plain bf16.
  int x3; long a_lo, b_lo, c16_lo;
#ifdef AVLEN_G2_LAB
  int ablate;                 // tools/gemm_lab.hip only: 1 = no fragment reads / MFMA, 2 = no staging loads, 4 = no C stores
#endif
};
#ifdef AVLEN_G2_LAB
int g_lab_ablate = 0;
int g_lab_cfg[5] = {0, 0, 0, 0, 0};     // bm, bn, threads, stages, split-K (0 = default)
#define LAB(bit) (p.ablate & (bit))
#else
#define LAB(bit) 0
#endif

__device__ __forceinline__ float act2(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v / (1.f + __expf(-1.702f * v));
  return v;
}

// NS = LDS stages: up to NS-1 K-tiles of global_load_lds in flight per block.  Large grids run NS=2 (small LDS
// footprint -> 2+ blocks per CU hide the latency); grids that cannot fill the chip run NS=4 (deep prefetch).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float4 ld4(const float* base, long off, int col, int N, bool vec) {
  if (vec) return col < N ? *reinterpret_cast<const float4*>(base + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v;
  v.x = col < N ? base[off] : 0.f; v.y = col + 1 < N ? base[off + 1] : 0.f;
  v.z = col + 2 < N ? base[off + 2] : 0.f; v.w = col + 3 < N ? base[off + 3] : 0.f;
  return v;
}

// NTH = 256: 4 waves, one per SIMD (small-N conv tiles; big grids overlap block against block).
// NTH = 512: 8 waves, two per SIMD, in two groups that run the K-step in opposite order -- waves 0-3 multiply tile t and
// then issue their share of the staging loads, waves 4-7 issue first and multiply after -- so that on every SIMD one wave's
// MFMAs run under the other wave's global_load_lds issue (a 1 KiB piece holds its wave for ~60-100 cycles; measured on the
// 4-wave kernel the two costs simply add: tools/gemm_lab.hip).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <bool F16> __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// fp32 -> the 16-bit storage format (bit pattern in a bf16-typed slot)
template <bool F16> __device__ __forceinline__ bf16 to16(float v) {
  if constexpr (F16) return __builtin_bit_cast(bf16, (_Float16)v);
  else return (bf16)v;
}

template <int BM, int BN, int WM, int WN, int NS, int NTH, bool CONV, bool F16 = false>
__global__ __launch_bounds__(NTH) void g2_kernel(G2 pp) {
  G2 p = pp;
  {
    const G2Grp gg = pp.g[blockIdx.y];
    p.A = gg.A; p.B = gg.B; p.C32 = gg.C32; p.C16 = gg.C16; p.bias = gg.bias; p.residual = gg.residual; p.stats = gg.stats;
    if (p.slab) p.slab += (size_t)blockIdx.y * p.splitk * p.M * p.N;
  }
  const int M_alloc = p.M;                     // slab layout uses the allocated row count
  if (p.M_dev) p.M = min(p.M, *p.M_dev);
  static_assert(WM * WN * 64 == NTH, "wave grid must cover the block");
  constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
  constexpr int MI = WTM / 16, NI = WTN / 16;
  static_assert((BM * 8) % NTH == 0, "A tile must be a whole number of block-wide rounds");
  constexpr int A_ROUNDS = BM * 8 / NTH;             // 16-byte chunks per thread for the A tile
  constexpr int B_SLOTS = BN * 8;
  constexpr int B_ROUNDS = (B_SLOTS + NTH - 1) / NTH;  // every wave issues the same number of loads per tile
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = (B_SLOTS >= NTH ? BN * 128 : NTH * 16);    // small BN: surplus lanes land in a pad area
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int LOADS = A_ROUNDS + B_ROUNDS;         // glds per wave per K-tile
  extern __shared__ __attribute__((aligned(16))) char lds[];       // [NS][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_tiles = (p.N + BN - 1) / BN;
  // XCD-aware tile order: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own
  // L2); label = id % 8 names the blocks that share an XCD, and each label walks a CONTIGUOUS range of tiles (row panel
  // major, column tile fastest) so the blocks that re-read one A panel -- or neighbouring conv halos -- share an L2.
  // With a device-side row count only the live tiles are dealt out (dead tiles would otherwise idle whole XCDs).
  int m0, n0;
  {
    const int nwg = ((p.M + BM - 1) / BM) * n_tiles;             // live tiles
    const int bid = blockIdx.x;
    if (bid >= nwg) return;
    const int label = bid & 7, idx = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
    const int t = (label < rr ? label * (qq + 1) : rr * (qq + 1) + (label - rr) * qq) + idx;
    m0 = (t / n_tiles) * BM; n0 = (t % n_tiles) * BN;
  }
  const int nk_pass = (p.K + BK - 1) / BK;                    // K-steps of one pass over K
  const int nk_total = p.x3 ? 3 * nk_pass : nk_pass;
  const int kt_beg = blockIdx.z * p.ksteps_per_split;
  const int kt_end = min(nk_total, kt_beg + p.ksteps_per_split);

  // ---- per-thread source descriptors (fixed across K-steps) ----
  const char* a_src[A_ROUNDS]; int a_iy0[A_ROUNDS], a_ix0[A_ROUNDS], a_sw[A_ROUNDS]; bool a_ok[A_ROUNDS];
#pragma unroll
  for (int r = 0; r < A_ROUNDS; r++) {
    int slot = r * NTH + tid, row = slot >> 3;
    a_sw[r] = ((slot & 7) ^ ((row >> 1) & 7)) * 8;           // logical k offset (elements) of this lane's chunk
    int m = m0 + row;
    a_ok[r] = m < p.M;
    if (CONV) {
      int ohw = p.OH * p.OW;
      int mm = a_ok[r] ? m : 0;
      int b = mm / ohw, rr = mm - b * ohw;
      int oy = rr / p.OW, ox = rr - oy * p.OW;
      a_src[r] = (const char*)(p.A + (long)b * p.H * p.W * p.Cin);
      a_iy0[r] = oy * p.stride - p.pad; a_ix0[r] = ox * (p.stride_w ? p.stride_w : p.stride) - p.pad;
    } else {
      a_src[r] = (const char*)(p.A + (long)(m < p.M ? m : m % p.M) * p.lda);     // surplus rows re-read distinct valid rows
      a_iy0[r] = 0; a_ix0[r] = 0;
    }
  }
  const char* b_src[B_ROUNDS]; int b_sw[B_ROUNDS];
#pragma unroll
  for (int r = 0; r < B_ROUNDS; r++) {
    int slot = r * NTH + tid, row = (slot >> 3) % BN;          // surplus lanes (slot >= B_SLOTS) re-read a valid row
    b_sw[r] = ((slot & 7) ^ ((row >> 1) & 7)) * 8;
    b_src[r] = (const char*)(p.B + (long)((n0 + row) < p.N ? (n0 + row) : (n0 + row) % p.N) * p.ldb);
  }
  const char* zero = (const char*)g_zero_page + (tid & 255) * 16;  // per-thread slice of the zero page

  auto issue = [&](int kt, int stage) {
    char* abase = lds + stage * STAGE;
    char* bbase = abase + A_BYTES;
    // bf16x3: pass 0 = (A_lo, W_hi), pass 1 = (A_hi, W_lo), pass 2 = (A_hi, W_hi)
    const int pass = (kt >= nk_pass) + (kt >= 2 * nk_pass);
    const long a_off = (p.x3 && pass == 0) ? p.a_lo : 0, b_off = (p.x3 && pass == 1) ? p.b_lo : 0;
    const int k0 = (kt - pass * nk_pass) * BK;
    // the swizzle term ((row>>1)&7) is the same for all of a thread's rows (they are >= 32 rows apart), so the k offset --
    // and, for convolutions, the tap decode -- is computed once per K-step and shared by the A_ROUNDS gathers
    const int ka = k0 + a_sw[0];
    int ky = 0, kx = 0, ci = 0;
    if (CONV) {
      int tap = ka >> p.cin_log2;
      ci = ka & (p.Cin - 1);
      ky = (tap * p.kw_magic) >> 16; kx = tap - ky * p.KW;
    }
    const bool k_ok = ka < p.K;
#pragma unroll
    for (int r = 0; r < A_ROUNDS; r++) {
      const char* src = zero;
      if (CONV) {
        int iy = a_iy0[r] + ky, ix = a_ix0[r] + kx;
        if (a_ok[r] && k_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          src = a_src[r] + (((long)iy * p.W + ix) * p.Cin + ci) * 2 + a_off;
      } else if (k_ok) {
        src = a_src[r] + (long)ka * 2 + a_off;
      }
      __builtin_amdgcn_global_load_lds((const void*)src,
          (__attribute__((address_space(3))) void*)(abase + (r * NTH + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < B_ROUNDS; r++) {
      int k = k0 + b_sw[r];
      const char* src = (k < p.K) ? b_src[r] + (long)k * 2 + b_off : zero;
      __builtin_amdgcn_global_load_lds((const void*)src,
          (__attribute__((address_space(3))) void*)(bbase + (r * NTH + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NI; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, q = lane >> 4;
  const bool loads_first = NTH == 512 && wave >= 4;

  // The deep-prefetch 64x128 instance runs the problems that cannot fill the chip (one block per CU: registers are free): its
  // epilogue operands -- bias and the fp32 residual tile -- are fetched NOW, under the K loop, instead of as a dependent round trip
  // behind it (the loads are older than every staging load, so the loop's counted vmcnt waits cover them).
  constexpr bool EARLY = NS == 4 && BM == 64 && BN == 128 && NTH == 512 && !CONV;
  float4 bv_e[EARLY ? NI : 1], rv_e[EARLY ? MI : 1][EARLY ? NI : 1];
  const bool early = EARLY && p.splitk == 1 && p.vec4 != 0 && !p.ln_stats;
  if (EARLY && early) {
#pragma unroll
    for (int j = 0; j < NI; j++) {
      const int col = n0 + wn * WTN + j * 16 + q * 4;
      bv_e[j] = p.bias ? ld4(p.bias, col, col, p.N, true) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int row = m0 + wm * WTM + i * 16 + r16;
        rv_e[i][j] = (p.residual && row < p.M) ? ld4(p.residual, (long)row * p.ldr + col, col, p.N, true) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }

  // ---- software pipeline: tiles kt .. kt+NS-2 in flight while tile kt is multiplied ----
  const int nkt = kt_end - kt_beg;
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; s0++)
    if (s0 < nkt) issue(kt_beg + s0, s0);
  for (int it = 0; it < nkt; it++) {
    // my loads of tile `it` have landed once at most the loads of the later in-flight tiles are outstanding
    if (NS > 2 && it + NS - 2 < nkt) wait_vmcnt<LOADS * (NS - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();      // everyone's part of tile `it` is in LDS; everyone has left stage (it-1)%NS
    const bool more = it + NS - 1 < nkt && !LAB(2);
    if (more && loads_first) issue(kt_beg + it + NS - 1, (it + NS - 1) % NS);
    if (!LAB(1)) {
      const char* abase = lds + (it % NS) * STAGE;
      const char* bbase = abase + A_BYTES;
#pragma unroll
      for (int kh = 0; kh < 2; kh++) {
        bf16x8 af[MI], bfr[NI];
        const int cc = kh * 4 + q;
#pragma unroll
        for (int i = 0; i < MI; i++) {
          int row = wm * WTM + i * 16 + r16;
          af[i] = *reinterpret_cast<const bf16x8*>(abase + row * 128 + ((cc ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; j++) {
          int row = wn * WTN + j * 16 + r16;
          bfr[j] = *reinterpret_cast<const bf16x8*>(bbase + row * 128 + ((cc ^ ((row >> 1) & 7)) << 4));
        }
        // W fragment as the first operand: the result comes out transposed, lane (r16, q) holds row m = r16 and the four
        // consecutive columns n = 4q .. 4q+3 of each 16x16 tile -> 16-byte fp32 / 8-byte bf16 stores in the epilogue
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
          for (int j = 0; j < NI; j++)
            acc[i][j] = mfma16<F16>(bfr[j], af[i], acc[i][j]);
      }
    }
    if (more && !loads_first) issue(kt_beg + it + NS - 1, (it + NS - 1) % NS);
  }

  // ---- fused GroupNorm statistics: per (sample, channel) sum and sum of squares of the raw conv output.
  // A wave tile (WTM rows) never straddles two samples (ohw % WTM == 0, checked on the host); rows >= M are zero.
  // Reduce-scatter over the 16 row lanes: 5 cross-lane moves per 4 columns, then one atomic per column per wave.
  if (p.stats) {
    const int sample = (m0 + wm * WTM) / p.ohw;
    if (m0 + wm * WTM < p.M) {
      const bool hi = r16 & 8, hi2 = r16 & 4;
#pragma unroll
      for (int j = 0; j < NI; j++) {
        float v1[4], v2[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int i = 0; i < MI; i++) { float v = acc[i][j][r]; a += v; b += v * v; }
          v1[r] = a; v2[r] = b;
        }
        float a0 = hi ? v1[2] : v1[0], a1 = hi ? v1[3] : v1[1], b0 = hi ? v1[0] : v1[2], b1 = hi ? v1[1] : v1[3];
        float c0 = hi ? v2[2] : v2[0], c1 = hi ? v2[3] : v2[1], d0 = hi ? v2[0] : v2[2], d1 = hi ? v2[1] : v2[3];
        a0 += __shfl_xor(b0, 8, 64); a1 += __shfl_xor(b1, 8, 64); c0 += __shfl_xor(d0, 8, 64); c1 += __shfl_xor(d1, 8, 64);
        float s1 = hi2 ? a1 : a0, t1 = hi2 ? a0 : a1, s2 = hi2 ? c1 : c0, t2 = hi2 ? c0 : c1;
        s1 += __shfl_xor(t1, 4, 64); s2 += __shfl_xor(t2, 4, 64);
        s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
        s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
        int col = n0 + wn * WTN + j * 16 + q * 4 + (hi ? 2 : 0) + (hi2 ? 1 : 0);
        if ((r16 & 3) == 0 && col < p.N) {
          atomicAdd(&p.stats[((long)sample * 2) * p.N + col], s1);
          atomicAdd(&p.stats[((long)sample * 2 + 1) * p.N + col], s2);
        }
      }
    }
  }

  // ---- epilogue: all bias / residual loads are issued before the first store (no serialized round trips) ----
  const bool vec = p.vec4 != 0;
  if (p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NI; j++) {
        int col = n0 + wn * WTN + j * 16 + q * 4;
        int row = m0 + wm * WTM + i * 16 + r16;
        if (row >= p.M) continue;
        float* dst = p.slab + ((long)blockIdx.z * M_alloc + row) * p.N + col;
        if (vec) { if (col < p.N) *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]); }
        else {
#pragma unroll
          for (int r = 0; r < 4; r++) if (col + r < p.N) dst[r] = acc[i][j][r];
        }
      }
    return;
  }
  float4 bv[NI];
#pragma unroll
  for (int j = 0; j < NI; j++) {
    int col = n0 + wn * WTN + j * 16 + q * 4;
    if (EARLY && early) bv[j] = bv_e[EARLY ? j : 0];
    else bv[j] = p.bias ? ld4(p.bias, col, col, p.N, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (p.ln_stats) {            // folded LayerNorm: acc <- rstd * (acc - mean * s)
    float4 sv[NI];
#pragma unroll
    for (int j = 0; j < NI; j++) {
      int col = n0 + wn * WTN + j * 16 + q * 4;
      sv[j] = ld4(p.ln_s, col, col, p.N, vec);
    }
    const float inv_k = 1.f / (float)p.K;
#pragma unroll
    for (int i = 0; i < MI; i++) {
      int row = m0 + wm * WTM + i * 16 + r16;
      float2 st = row < p.M ? *reinterpret_cast<const float2*>(p.ln_stats + (long)row * 2) : make_float2(0.f, 0.f);
      for (int sl = 1; sl < p.ln_slabs; sl++)
        if (row < p.M) { const float2 t2 = *reinterpret_cast<const float2*>(p.ln_stats + ((long)sl * M_alloc + row) * 2); st.x += t2.x; st.y += t2.y; }
      const float mean = st.x * inv_k;
      const float rstd = rsqrtf(fmaxf(st.y * inv_k - mean * mean, 0.f) + 1e-5f);
#pragma unroll
      for (int j = 0; j < NI; j++) {
        acc[i][j][0] = rstd * (acc[i][j][0] - mean * sv[j].x); acc[i][j][1] = rstd * (acc[i][j][1] - mean * sv[j].y);
        acc[i][j][2] = rstd * (acc[i][j][2] - mean * sv[j].z); acc[i][j][3] = rstd * (acc[i][j][3] - mean * sv[j].w);
      }
    }
  }
  if (p.residual) {
    float4 rv[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NI; j++) {
        int col = n0 + wn * WTN + j * 16 + q * 4;
        int row = m0 + wm * WTM + i * 16 + r16;
        if (EARLY && early) rv[i][j] = rv_e[EARLY ? i : 0][EARLY ? j : 0];
        else rv[i][j] = row < p.M ? ld4(p.residual, (long)row * p.ldr + col, col, p.N, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NI; j++) {
        acc[i][j][0] = act2(acc[i][j][0] + bv[j].x, p.act) + rv[i][j].x; acc[i][j][1] = act2(acc[i][j][1] + bv[j].y, p.act) + rv[i][j].y;
        acc[i][j][2] = act2(acc[i][j][2] + bv[j].z, p.act) + rv[i][j].z; acc[i][j][3] = act2(acc[i][j][3] + bv[j].w, p.act) + rv[i][j].w;
        if (p.act == AVLEN_ACT_RELU_POST) {        // ReLU after the residual add (torchvision BasicBlock)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NI; j++) {
        const int a1 = p.act == AVLEN_ACT_RELU_POST ? AVLEN_ACT_RELU : p.act;
        acc[i][j][0] = act2(acc[i][j][0] + bv[j].x, a1); acc[i][j][1] = act2(acc[i][j][1] + bv[j].y, a1);
        acc[i][j][2] = act2(acc[i][j][2] + bv[j].z, a1); acc[i][j][3] = act2(acc[i][j][3] + bv[j].w, a1);
      }
  }
  // C stores first: they are in flight while the row statistics are reduced (two barriers + one atomic per row)
  if (LAB(4)) { if (acc[0][0][0] == 123.456f) p.C32[0] = 1.f; return; }
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NI; j++) {
      int col = n0 + wn * WTN + j * 16 + q * 4;
      int row = m0 + wm * WTM + i * 16 + r16;
      if (row >= p.M || col >= p.N) continue;
      if (vec) {
        if (p.C32) *reinterpret_cast<float4*>(p.C32 + (long)row * p.ldc32 + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (p.C16) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = to16<F16>(acc[i][j][r]);
          *reinterpret_cast<bf16x4*>(p.C16 + (long)row * p.ldc16 + col) = o;
          if (!F16 && p.c16_lo) {
            bf16x4 l;
#pragma unroll
            for (int r = 0; r < 4; r++) l[r] = (bf16)(acc[i][j][r] - (float)o[r]);
            *reinterpret_cast<bf16x4*>(p.C16 + p.c16_lo + (long)row * p.ldc16 + col) = l;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (col + r < p.N) {
            if (p.C32) p.C32[(long)row * p.ldc32 + col + r] = acc[i][j][r];
            if (p.C16) {
              const bf16 hv = to16<F16>(acc[i][j][r]);
              p.C16[(long)row * p.ldc16 + col + r] = hv;
              if (!F16 && p.c16_lo) p.C16[p.c16_lo + (long)row * p.ldc16 + col + r] = (bf16)(acc[i][j][r] - (float)hv);
            }
          }
      }
    }
  if (p.rowstats) {            // per-row (sum, sum of squares) of the final values: the next layer's LayerNorm statistics
    // wave partials (4 columns x NI tiles per lane, 2 cross-lane moves) -> LDS -> one atomic per row and block: per-wave
    // atomics were 16-lane instructions, 4 * MI * WN of them per block
    float* rs = reinterpret_cast<float*>(lds);             // [BM][WN][2]; the staging ring is dead after the K loop
    __syncthreads();                                       // every wave has left its last fragment reads
#pragma unroll
    for (int i = 0; i < MI; i++) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NI; j++) {
        int col = n0 + wn * WTN + j * 16 + q * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) { float v = col + r < p.N ? acc[i][j][r] : 0.f; s1 += v; s2 += v * v; }
      }
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) {
        const int lr = wm * WTM + i * 16 + r16;
        rs[(lr * WN + wn) * 2] = s1; rs[(lr * WN + wn) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    for (int t = tid; t < BM * 2; t += NTH) {
      const int lr = t >> 1, which = t & 1, row = m0 + lr;
      if (row < p.M) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WN; w++) v += rs[(lr * WN + w) * 2 + which];
        if (p.rs_slabs) p.rowstats[((long)(n0 / BN) * M_alloc + row) * 2 + which] = v;
        else atomicAdd(&p.rowstats[(long)row * 2 + which], v);
      }
    }
  }
}

// Deterministic split-K reduction + epilogue; blockIdx.y = group (all groups of a grouped launch in one pass).
struct G2Red { G2Grp g[MAXG]; const float* slab; };
__device__ __forceinline__ void put16(bf16* C16, long idx, float v, int f16, long c16_lo) {
  if (f16) { C16[idx] = to16<true>(v); return; }
  const bf16 h = (bf16)v;
  C16[idx] = h;
  if (c16_lo) C16[c16_lo + idx] = (bf16)(v - (float)h);
}
__global__ void g2_reduce_kernel(G2Red rr, int M, int N, int ldc32, int ldc16, int ldr, int splits, int act,
                                 const int* __restrict__ M_dev, int f16, long c16_lo) {
  const G2Grp g = rr.g[blockIdx.y];
  const long tot = (long)M * N;
  const float* __restrict__ slab = rr.slab + (size_t)blockIdx.y * splits * tot;
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= tot) return;
  if ((N & 3) == 0 && ((size_t)slab & 15) == 0) {   // 4 consecutive columns of one row
    int row = (int)(i / N), col = (int)(i - (long)row * N);
    if (M_dev && row >= *M_dev) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight slabs requested at a time, added in slab order (the same sum as one by one: a dependent load per split made this
    // 4 K-element reduction a 20 us kernel on the rollout step's critical path)
    int z = 0;
    for (; z + 8 <= splits; z += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(slab + (long)(z + u) * tot + i);
#pragma unroll
      for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < splits; z++) {
      float4 v = *reinterpret_cast<const float4*>(slab + (long)z * tot + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float v = o[r];
      if (g.bias) v += g.bias[col + r];
      v = act2(v, act);
      if (g.residual) v += g.residual[(long)row * ldr + col + r];
      if (act == AVLEN_ACT_RELU_POST) v = fmaxf(v, 0.f);
      if (g.C32) g.C32[(long)row * ldc32 + col + r] = v;
      if (g.C16) put16(g.C16, (long)row * ldc16 + col + r, v, f16, c16_lo);
    }
    return;
  }
  for (int e = 0; e < 4 && i + e < tot; e++) {
    long ii = i + e;
    int row = (int)(ii / N), col = (int)(ii - (long)row * N);
    if (M_dev && row >= *M_dev) return;
    float s = 0.f;
    for (int z = 0; z < splits; z++) s += slab[(long)z * tot + ii];
    if (g.bias) s += g.bias[col];
    s = act2(s, act);
    if (g.residual) s += g.residual[(long)row * ldr + col];
    if (act == AVLEN_ACT_RELU_POST) s = fmaxf(s, 0.f);
    if (g.C32) g.C32[(long)row * ldc32 + col] = s;
    if (g.C16) put16(g.C16, (long)row * ldc16 + col, s, f16, c16_lo);
  }
}

// fp32 -> 16-bit storage: fmt 0 = bf16, 1 = fp16, 2 = the bf16 LOW plane of the compensated pair (bf16(x - bf16(x)))
__device__ __forceinline__ bf16 cvt16(float v, int fmt) {
  if (fmt == 1) return to16<true>(v);
  const bf16 h = (bf16)v;
  return fmt == 2 ? (bf16)(v - (float)h) : h;
}
// fp32 -> bf16 with row padding (pad columns zero-filled)
__global__ void cast_rows_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, long rows, int cols, int fmt) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ldd) return;
  long r = i / ldd; int c = (int)(i - r * ldd);
  dst[i] = c < cols ? cvt16(src[r * lds_ + c], fmt) : (bf16)0.f;
}

__global__ void cast_rows_indexed_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, long rows, int cols,
                                         const int* __restrict__ row_index, int rpi, int fmt) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ldd) return;
  long r = i / ldd; int c = (int)(i - r * ldd);
  const long item = r / rpi, rs = (long)row_index[item] * rpi + (r - item * rpi);
  dst[i] = c < cols ? cvt16(src[rs * lds_ + c], fmt) : (bf16)0.f;
}

// Same cast, 8 elements (32 B in, 16 B out) per thread: both leading dimensions multiples of 8 and 16-byte aligned bases
__global__ void cast_rows8_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, long rows, int cols, int fmt) {
  const int per_row = ldd >> 3;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const long r = i / per_row; const int c = (int)(i - r * per_row) << 3;
  bf16x8 o;
  if (c + 8 <= cols) {
    const float4 a = *reinterpret_cast<const float4*>(src + r * lds_ + c), b = *reinterpret_cast<const float4*>(src + r * lds_ + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (fmt == 0) {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (bf16)v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = cvt16(v[e], fmt);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = c + e < cols ? cvt16(src[r * lds_ + c + e], fmt) : (bf16)0.f;
  }
  *reinterpret_cast<bf16x8*>(dst + r * ldd + c) = o;
}

// OIHW fp32 -> [O][KH][KW][Cp] bf16 (channels zero-padded to Cp)
__global__ void pack_conv_bf16_kernel(const float* __restrict__ w, bf16* __restrict__ o, int O, int I, int KH, int KW, int Cp, int fmt) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)O * KH * KW * Cp;
  if (idx >= tot) return;
  int ci = (int)(idx % Cp); long r = idx / Cp;
  int kx = (int)(r % KW); r /= KW;
  int ky = (int)(r % KH); int oc = (int)(r / KH);
  o[idx] = ci < I ? cvt16(w[(((long)oc * I + ci) * KH + ky) * KW + kx], fmt) : (bf16)0.f;
}
// (O, C*HW) fp32 (NCHW flatten) -> (O, HW*C) bf16 (NHWC flatten)
__global__ void pack_fc_bf16_kernel(const float* __restrict__ w, bf16* __restrict__ o, int O, int C, int HW, int fmt) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)O * C * HW;
  if (idx >= tot) return;
  int c = (int)(idx % C); long r = idx / C; int pp = (int)(r % HW); int oc = (int)(r / HW);
  o[idx] = cvt16(w[((long)oc * C + c) * HW + pp], fmt);
}

// LayerNorm folding (derived data for avlen_gemm_bf16_ln): w16f[n][k] = bf16(W[n][k] * gamma[k]),
// s[n] = sum_k float(w16f[n][k]), c[n] = bias[n] + sum_k beta[k] * W[n][k].  One block per output row.
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      bf16* __restrict__ w16f, int ld16, float* __restrict__ s,
                                                      float* __restrict__ c, int K, int fmt) {
  __shared__ float sh[16];
  const int n = blockIdx.x;
  float sa = 0.f, ca = 0.f;
  for (int k = threadIdx.x; k < ld16; k += 256) {
    bf16 wf = (bf16)0.f;
    if (k < K) {
      const float w = W[(long)n * K + k];
      if (fmt == 1) { const _Float16 hv = (_Float16)(w * gamma[k]); wf = __builtin_bit_cast(bf16, hv); sa += (float)hv; }
      else { wf = (bf16)(w * gamma[k]); sa += (float)wf; }
      ca += beta[k] * w;
    }
    w16f[(long)n * ld16 + k] = wf;
  }
  sa = block_sum(sa, sh);
  ca = block_sum(ca, sh);
  if (threadIdx.x == 0) { s[n] = sa; c[n] = ca + (bias ? bias[n] : 0.f); }
}

template <int BM, int BN, int WM, int WN, int NS, int NTH, bool CONV, bool F16 = false>
int launch_cv(const G2& p, int m_tiles, int n_tiles, hipStream_t st) {
  size_t lds = (size_t)NS * (BM * 128 + (BN * 8 >= NTH ? BN * 128 : NTH * 16));
  static unsigned long long attr_done = 0;
  if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&g2_kernel<BM, BN, WM, WN, NS, NTH, CONV, F16>), (int)lds, &attr_done) != AVLEN_OK)
    return AVLEN_ERR_LAUNCH;
  hipLaunchKernelGGL((g2_kernel<BM, BN, WM, WN, NS, NTH, CONV, F16>), dim3(m_tiles * n_tiles, p.groups, p.splitk), dim3(NTH), lds, st, p);
  return avlen_launch_status();
}
// the implicit-GEMM gather (tap decode, bounds tests) is compiled out of the plain-GEMM instances; fp16 operands exist on every
// tile (CLIP text tower, dialog_layer, AudioCNN, BeliefPredictor's ResNets incl. their 2-column fc)
template <int BM, int BN, int WM, int WN, int NS, int NTH>
int launch_ns(const G2& p, int m_tiles, int n_tiles, hipStream_t st) {
  if (p.f16)
    return p.conv ? launch_cv<BM, BN, WM, WN, NS, NTH, true, true>(p, m_tiles, n_tiles, st)
                  : launch_cv<BM, BN, WM, WN, NS, NTH, false, true>(p, m_tiles, n_tiles, st);
  return p.conv ? launch_cv<BM, BN, WM, WN, NS, NTH, true>(p, m_tiles, n_tiles, st)
                : launch_cv<BM, BN, WM, WN, NS, NTH, false>(p, m_tiles, n_tiles, st);
}

// 4-wave tiles: NS = 4 (deep prefetch, one block per CU) when the grid cannot put 2+ blocks on every CU, else NS = 2.
template <int BM, int BN, int WM, int WN>
int launch4(const G2& p, int m_tiles, int n_tiles, hipStream_t st) {
  long blocks = (long)m_tiles * n_tiles * p.splitk * p.groups;
  static int forced = -1;                       // AVLEN_G2_NS=2|4 pins the stage count (A/B measurements)
  if (forced < 0) forced = (int)avlen_knob("AVLEN_G2_NS", 0);
  static long thresh = -1;
  if (thresh < 0) thresh = (long)avlen_knob("AVLEN_G2_NS_BLOCKS", 480);
  bool deep = forced ? forced == 4 : blocks < thresh;
  if (!deep) return launch_ns<BM, BN, WM, WN, 2, 256>(p, m_tiles, n_tiles, st);
  return launch_ns<BM, BN, WM, WN, 4, 256>(p, m_tiles, n_tiles, st);
}

bool aligned_to(const void* q, size_t a) { return ((uintptr_t)q & (a - 1)) == 0; }

void apply_opts(G2& p, const avlen_g2_opts* o) {
  if (!o) return;
  p.f16 = o->f16; p.x3 = o->x3; p.a_lo = o->a_lo; p.b_lo = o->b_lo; p.c16_lo = o->c16_lo;
  p.rs_slabs = o->rs_slabs; p.ln_slabs = o->ln_slabs;
}

int run_g2(G2 p, void* ws, size_t ws_bytes, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 7) || (p.lda & 7) || (p.ldb & 7)) return AVLEN_ERR_ARG;
  if (p.groups <= 0) {             // single problem: group 0 = the scalar fields
    p.groups = 1;
    p.g[0] = G2Grp{p.A, p.B, p.C32, p.C16, p.bias, p.residual, p.stats};
  }
  if (p.groups > MAXG) return AVLEN_ERR_ARG;
  p.vec4 = !(p.N & 3) && !(p.ldc32 & 3) && !(p.ldc16 & 3) && !(p.ldr & 3) && aligned_to(ws, 16) && aligned_to(p.ln_s, 16);
  for (int g = 0; g < p.groups; g++)
    p.vec4 = p.vec4 && aligned_to(p.g[g].C32, 16) && aligned_to(p.g[g].C16, 8) && aligned_to(p.g[g].bias, 16) &&
             aligned_to(p.g[g].residual, 16);
  int bn = p.N <= 16 ? 16 : p.N <= 32 ? 32 : p.N <= 64 ? 64 : 128;
  int n_tiles = ceil_div(p.N, bn);
  // GroupNorm statistics need every wave tile inside one sample: ohw % (wave rows) == 0
  const bool has_stats = p.stats || p.g[0].stats;
  int bm = 128, nth = 256, ns = 2;
  if (bn == 128) {
    // 128-column tiles run the 8-wave ping-pong kernel.  Tile height by how many tiles the problem yields (measured on
    // MI355X, tools/gemm_lab.hip): 256 rows only when even those fill the chip several times over, 64 rows when 128-row
    // tiles would leave CUs idle; deep prefetch (NS = 4) only for long-K problems that cannot fill the chip anyway.
    nth = 512;
    const int m_est = p.M_dev ? (p.M + 1) / 2 : p.M;      // ragged batches: about half the allocated rows are live
    const long t128 = (long)ceil_div(m_est, 128) * n_tiles * p.groups;
    const long t64 = (long)ceil_div(m_est, 64) * n_tiles * p.groups;
    if (t128 >= 4096) { bm = 256; ns = 3; }
    else if (t128 >= 400) { bm = 128; ns = 2; }
    else { bm = 64; ns = t64 < 256 ? 4 : 2; }          // < 256 tiles: one block per CU anyway -> deep prefetch, early epilogue operands
    if (has_stats && bm == 64 && (p.ohw % 32)) { bm = 128; ns = 2; }
    if (has_stats && (p.ohw % 64)) nth = 256;        // (not reached: the conv entry point requires ohw % 64 == 0)
  } else {
    // 64-row tiles when 128-row tiles would leave most of the 256 CUs idle (small rollout batches)
    if (bn >= 64 && (long)ceil_div(p.M, 128) * n_tiles * p.groups < 256 && p.M > 64) bm = 64;
    if (has_stats && bm == 64 && (p.ohw % 32)) bm = 128;
    // 128x64 tiles also run 8 waves (4x2, 32-row wave tiles): -23 % on the towers' 64-channel convs (AVLEN_G2_N64=256 restores
    // the 4-wave instance for A/B runs)
    static int n64 = -1;
    if (n64 < 0) n64 = (int)avlen_knob("AVLEN_G2_N64", 512);
    if (n64 == 512 && bn == 64 && bm == 128 && (!has_stats || p.ohw % 32 == 0)) { nth = 512; ns = 2; }
  }
#ifdef AVLEN_G2_LAB
  if (g_lab_cfg[0]) { bm = g_lab_cfg[0]; bn = g_lab_cfg[1]; nth = g_lab_cfg[2]; n_tiles = ceil_div(p.N, bn); }
#endif
  if (nth == 256 && bn == 128 && bm == 256) bm = 128;
  int m_tiles = ceil_div(p.M, bm);
  if (p.f16 && p.x3) return AVLEN_ERR_ARG;
  if (p.rs_slabs && (bn != 128 || p.rs_slabs != n_tiles)) return AVLEN_ERR_ARG;      // the caller sized rowstats for ceil(N / 128) slabs
  int nk = ceil_div(p.K, BK) * (p.x3 ? 3 : 1);
  long tiles = (long)m_tiles * n_tiles * p.groups;
  int split = 1;
  static long split_min_nk = -1;                  // fewest k-steps worth a second (reduce) launch: 8 -> 24 measured +2..4 % on cfg2 and --belief (AVLEN_G2_SPLIT_MIN_NK in lab builds)
  if (split_min_nk < 0) split_min_nk = avlen_knob("AVLEN_G2_SPLIT_MIN_NK", 24);
  if (tiles < 128 && nk >= split_min_nk && !has_stats && !p.ln_stats && !p.rowstats) {
    long a = nk / 4, b = (256 + tiles - 1) / tiles;
    split = (int)(a < b ? a : b);
    if (split < 1) split = 1;
    if (split > 32) split = 32;
    if (!ws || (size_t)split * p.groups * p.M * p.N * sizeof(float) > ws_bytes) split = 1;
  }
#ifdef AVLEN_G2_LAB
  if (g_lab_cfg[4] && ws && (size_t)g_lab_cfg[4] * p.groups * p.M * p.N * sizeof(float) <= ws_bytes) split = g_lab_cfg[4];
#endif
  p.ksteps_per_split = ceil_div(nk, split);
  p.splitk = ceil_div(nk, p.ksteps_per_split);
  p.slab = (float*)ws;
#ifdef AVLEN_G2_LAB
  p.ablate = g_lab_ablate;
#endif
  int rc = AVLEN_ERR_ARG;
  if (nth == 512) {
#ifdef AVLEN_G2_LAB
    if (g_lab_cfg[3]) ns = g_lab_cfg[3];
#endif
    if (bm == 64 && bn == 128) rc = ns == 2 ? launch_ns<64, 128, 2, 4, 2, 512>(p, m_tiles, n_tiles, st) : ns == 3 ? launch_ns<64, 128, 2, 4, 3, 512>(p, m_tiles, n_tiles, st) : launch_ns<64, 128, 2, 4, 4, 512>(p, m_tiles, n_tiles, st);
    else if (bm == 128 && bn == 128) rc = ns == 2 ? launch_ns<128, 128, 2, 4, 2, 512>(p, m_tiles, n_tiles, st) : ns == 3 ? launch_ns<128, 128, 2, 4, 3, 512>(p, m_tiles, n_tiles, st) : launch_ns<128, 128, 2, 4, 4, 512>(p, m_tiles, n_tiles, st);
    else if (bm == 256 && bn == 128) rc = ns == 2 ? launch_ns<256, 128, 4, 2, 2, 512>(p, m_tiles, n_tiles, st) : launch_ns<256, 128, 4, 2, 3, 512>(p, m_tiles, n_tiles, st);
#ifdef AVLEN_G2_LAB
    else if (bm == 64 && bn == 256) rc = ns == 2 ? launch_ns<64, 256, 2, 4, 2, 512>(p, m_tiles, n_tiles, st) : launch_ns<64, 256, 2, 4, 3, 512>(p, m_tiles, n_tiles, st);
    else if (bm == 128 && bn == 256) rc = ns == 2 ? launch_ns<128, 256, 2, 4, 2, 512>(p, m_tiles, n_tiles, st) : launch_ns<128, 256, 2, 4, 3, 512>(p, m_tiles, n_tiles, st);
#endif
    else if (bm == 128 && bn == 64) rc = ns == 2 ? launch_ns<128, 64, 4, 2, 2, 512>(p, m_tiles, n_tiles, st) : ns == 3 ? launch_ns<128, 64, 4, 2, 3, 512>(p, m_tiles, n_tiles, st) : launch_ns<128, 64, 4, 2, 4, 512>(p, m_tiles, n_tiles, st);
  } else if (bm == 64) {
    rc = (bn == 64) ? launch4<64, 64, 2, 2>(p, m_tiles, n_tiles, st) : launch4<64, 128, 2, 2>(p, m_tiles, n_tiles, st);
  } else {
    switch (bn) {
      case 16: rc = launch4<128, 16, 4, 1>(p, m_tiles, n_tiles, st); break;
      case 32: rc = launch4<128, 32, 4, 1>(p, m_tiles, n_tiles, st); break;
      case 64: rc = launch4<128, 64, 2, 2>(p, m_tiles, n_tiles, st); break;
      default: rc = launch4<128, 128, 2, 2>(p, m_tiles, n_tiles, st); break;
    }
  }
  if (rc != AVLEN_OK) return rc;
  if (p.splitk > 1) {
    long tot = (long)p.M * p.N;
    G2Red rr; rr.slab = p.slab;
    for (int g = 0; g < MAXG; g++) rr.g[g] = p.g[g < p.groups ? g : 0];
    hipLaunchKernelGGL(g2_reduce_kernel, dim3((unsigned)((tot / 4 + 256) / 256), p.groups), dim3(256), 0, st, rr, p.M, p.N,
                       p.ldc32, p.ldc16, p.ldr, p.splitk, p.act, p.M_dev, p.f16, p.x3 ? p.c16_lo : 0L);
    return avlen_launch_status();
  }
  return AVLEN_OK;
}

}  // namespace

extern "C" int avlen_ln_fold_weights_h16(const float* W, const float* bias, const float* gamma, const float* beta, void* w16f,
                                         int ld16, float* s, float* c, int N, int K, int fmt, hipStream_t stream) {
  if (!W || !gamma || !beta || !w16f || !s || !c || N <= 0 || K <= 0 || ld16 < K || (ld16 & 7) || fmt < 0 || fmt > 1) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(ln_fold_kernel, dim3(N), dim3(256), 0, stream, W, bias, gamma, beta, (bf16*)w16f, ld16, s, c, K, fmt);
  return avlen_launch_status();
}
extern "C" int avlen_ln_fold_weights(const float* W, const float* bias, const float* gamma, const float* beta, void* w16f,
                                     int ld16, float* s, float* c, int N, int K, hipStream_t stream) {
  return avlen_ln_fold_weights_h16(W, bias, gamma, beta, w16f, ld16, s, c, N, K, 0, stream);
}

extern "C" size_t avlen_gemm_bf16_workspace_bytes(int M, int N) { return (size_t)32 * M * N * sizeof(float) + 256; }

extern "C" int avlen_gemm_bf16(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16,
                               int ldc16, const float* bias, const float* residual, int ldr, int M, int N, int K,
                               int act, void* ws, size_t ws_bytes, hipStream_t stream) {
  G2 p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C32 = C32; p.C16 = (bf16*)C16; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc32 = ldc32; p.ldc16 = ldc16; p.ldr = ldr; p.act = act;
  return run_g2(p, ws, ws_bytes, stream);
}

extern "C" int avlen_conv2d_nhwc_bf16(const void* X, const void* Wp, const float* bias, const float* residual,
                                      float* Y32, void* Y16, float* gn_stats, int Bn, int H, int W, int Cin, int Cout,
                                      int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
  return avlen_conv2d_nhwc_h16(X, Wp, bias, residual, Y32, Y16, gn_stats, Bn, H, W, Cin, Cout, KH, KW, stride, pad, act, ws, ws_bytes,
                               stream, nullptr);
}
// the same convolution with options (o->f16: fp16 operands / 16-bit output in IEEE half)
int avlen_conv2d_nhwc_h16(const void* X, const void* Wp, const float* bias, const float* residual, float* Y32, void* Y16,
                          float* gn_stats, int Bn, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int act,
                          void* ws, size_t ws_bytes, hipStream_t stream, const avlen_g2_opts* o) {
  if (Cin < 8 || (Cin & (Cin - 1))) return AVLEN_ERR_ARG;       // power of two >= 8 (conv1 input is channel-padded)
  int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  if (OH <= 0 || OW <= 0) return AVLEN_ERR_ARG;
  G2 p = {};
  p.A = (const bf16*)X; p.B = (const bf16*)Wp; p.C32 = Y32; p.C16 = (bf16*)Y16; p.bias = bias; p.residual = residual;
  p.M = Bn * OH * OW; p.N = Cout; p.K = KH * KW * Cin; p.lda = 8; p.ldb = p.K; p.ldc32 = Cout; p.ldc16 = Cout; p.ldr = Cout;
  p.act = act; p.conv = 1; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad;
  if (gn_stats) {
    if ((OH * OW) % 64 || bias) return AVLEN_ERR_ARG;      // wave tiles (32 or 64 rows) must not straddle samples
    p.stats = gn_stats; p.ohw = OH * OW;
  }
  int l2 = 0; while ((1 << l2) < Cin) l2++;
  p.cin_log2 = l2; p.kw_magic = (65536 + KW - 1) / KW;
  apply_opts(p, o);
  return run_g2(p, ws, ws_bytes, stream);
}

// Grouped forms: `groups` independent problems of identical shape in ONE launch (blockIdx.y = group); used to run the
// six ResNet towers of the three policies (and the two towers of one policy) as single launches.
int avlen_conv2d_nhwc_bf16_grouped(const void* const* X, const void* const* Wp, float* const* Y32, void* const* Y16,
                                   float* const* gn_stats, int groups, int Bn, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                   void* ws, size_t ws_bytes, hipStream_t stream, const float* const* bias, int act, int stride_w,
                                   const avlen_g2_opts* o) {
  if (Cin < 8 || (Cin & (Cin - 1)) || groups < 1 || groups > MAXG) return AVLEN_ERR_ARG;
  int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / (stride_w ? stride_w : stride) + 1;
  if (OH <= 0 || OW <= 0 || (gn_stats && (OH * OW) % 64)) return AVLEN_ERR_ARG;
  G2 p = {};
  p.groups = groups;
  for (int g = 0; g < groups; g++)
    p.g[g] = G2Grp{(const bf16*)X[g], (const bf16*)Wp[g], Y32 ? Y32[g] : nullptr, Y16 ? (bf16*)Y16[g] : nullptr,
                   bias ? bias[g] : nullptr, nullptr, gn_stats ? gn_stats[g] : nullptr};
  p.act = act;
  if (gn_stats && bias) return AVLEN_ERR_ARG;
  p.M = Bn * OH * OW; p.N = Cout; p.K = KH * KW * Cin; p.lda = 8; p.ldb = p.K; p.ldc32 = Cout; p.ldc16 = Cout; p.ldr = Cout;
  p.conv = 1; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.stride_w = stride_w;
  p.ohw = OH * OW;
  int l2 = 0; while ((1 << l2) < Cin) l2++;
  p.cin_log2 = l2; p.kw_magic = (65536 + KW - 1) / KW;
  apply_opts(p, o);
  return run_g2(p, ws, ws_bytes, stream);
}

int avlen_gemm_bf16_grouped(const void* const* A, int lda, const void* const* B, int ldb, float* const* C32, int ldc32,
                            const float* const* bias, int groups, int M, int N, int K, int act, void* ws, size_t ws_bytes,
                            hipStream_t stream, const avlen_g2_opts* o) {
  if (groups < 1 || groups > MAXG) return AVLEN_ERR_ARG;
  G2 p = {};
  p.groups = groups;
  for (int g = 0; g < groups; g++)
    p.g[g] = G2Grp{(const bf16*)A[g], (const bf16*)B[g], C32[g], nullptr, bias ? bias[g] : nullptr, nullptr, nullptr};
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc32 = ldc32; p.act = act;
  apply_opts(p, o);
  return run_g2(p, ws, ws_bytes, stream);
}

// avlen_gemm_bf16_dyn + LayerNorm folding / row statistics (see G2::ln_stats, G2::rowstats).
int avlen_gemm_bf16_ln(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16, int ldc16,
                       const float* bias, const float* residual, int ldr, int M, const int* M_dev, int N, int K, int act,
                       const float* ln_stats, const float* ln_s, float* rowstats, void* ws, size_t ws_bytes,
                       hipStream_t stream, const avlen_g2_opts* o) {
  if (ln_stats && !ln_s) return AVLEN_ERR_ARG;
  G2 p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C32 = C32; p.C16 = (bf16*)C16; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc32 = ldc32; p.ldc16 = ldc16; p.ldr = ldr; p.act = act;
  p.M_dev = M_dev; p.ln_stats = ln_stats; p.ln_s = ln_s; p.rowstats = rowstats;
  apply_opts(p, o);
  return run_g2(p, ws, ws_bytes, stream);
}

// Same as avlen_gemm_bf16 with the live row count taken from device memory (*M_dev <= M): tiles beyond it exit at once.
int avlen_gemm_bf16_dyn(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16, int ldc16,
                        const float* bias, const float* residual, int ldr, int M, const int* M_dev, int N, int K, int act,
                        void* ws, size_t ws_bytes, hipStream_t stream, const avlen_g2_opts* o) {
  G2 p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C32 = C32; p.C16 = (bf16*)C16; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc32 = ldc32; p.ldc16 = ldc16; p.ldr = ldr; p.act = act;
  p.M_dev = M_dev;
  apply_opts(p, o);
  return run_g2(p, ws, ws_bytes, stream);
}

// avlen_gemm_bf16 with fp16 operands (fmt = 1): A, B and the 16-bit output hold IEEE half values
extern "C" int avlen_gemm_h16(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16,
                              int ldc16, const float* bias, const float* residual, int ldr, int M, int N, int K,
                              int act, int fmt, void* ws, size_t ws_bytes, hipStream_t stream) {
  avlen_g2_opts o; o.f16 = fmt == 1;
  return avlen_gemm_bf16_dyn(A, lda, B, ldb, C32, ldc32, C16, ldc16, bias, residual, ldr, M, nullptr, N, K, act, ws, ws_bytes, stream, &o);
}

__global__ void cast_rows_indexed8_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, long rows, int cols,
                                          const int* __restrict__ row_index, int rpi, int fmt);
extern "C" int avlen_cast_h16(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, int fmt, hipStream_t stream) {
  long tot = rows * ld_dst;
  if (tot <= 0 || fmt < 0 || fmt > 2) return AVLEN_ERR_ARG;
  if (!(ld_src & 3) && !(ld_dst & 7) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15)) {
    long t8 = tot >> 3;
    hipLaunchKernelGGL(cast_rows8_kernel, dim3((unsigned)((t8 + 255) / 256)), dim3(256), 0, stream, src, ld_src, (bf16*)dst, ld_dst, rows, cols, fmt);
    return avlen_launch_status();
  }
  if (!(ld_src & 1) && !(ld_dst & 7) && !((uintptr_t)src & 7) && !((uintptr_t)dst & 15)) {      // even row stride (a 101 x 2 spectrogram row): 8-byte loads
    long t8 = tot >> 3;
    hipLaunchKernelGGL(cast_rows_indexed8_kernel, dim3((unsigned)((t8 + 255) / 256)), dim3(256), 0, stream, src, ld_src, (bf16*)dst, ld_dst, rows,
                       cols, (const int*)nullptr, 1, fmt);
    return avlen_launch_status();
  }
  hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, src, ld_src, (bf16*)dst, ld_dst, rows, cols, fmt);
  return avlen_launch_status();
}
// fp32 rows -> the compensated pair in ONE pass over the source: hi plane at dst, lo plane `lo` elements behind it
__global__ void cast_pair8_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, long rows, int cols, long lo,
                                  int vec) {
  const int per_row = ldd >> 3;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const long r = i / per_row; const int c = (int)(i - r * per_row) << 3;
  float v[8];
  if (vec && c + 8 <= cols) {
    const float4 a = *reinterpret_cast<const float4*>(src + r * lds_ + c), b = *reinterpret_cast<const float4*>(src + r * lds_ + c + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = c + e < cols ? src[r * lds_ + c + e] : 0.f;
  }
  bf16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; e++) { h[e] = (bf16)v[e]; l[e] = (bf16)(v[e] - (float)h[e]); }
  *reinterpret_cast<bf16x8*>(dst + r * ldd + c) = h;
  *reinterpret_cast<bf16x8*>(dst + lo + r * ldd + c) = l;
}
int avlen_cast_pair(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, long lo, hipStream_t stream) {
  if (rows * ld_dst <= 0 || lo <= 0) return AVLEN_ERR_ARG;
  if (!(ld_dst & 7) && !((uintptr_t)dst & 15) && !(lo & 7)) {
    const long t8 = (rows * ld_dst) >> 3;
    const int vec = !(ld_src & 3) && !((uintptr_t)src & 15);
    hipLaunchKernelGGL(cast_pair8_kernel, dim3((unsigned)((t8 + 255) / 256)), dim3(256), 0, stream, src, ld_src, (bf16*)dst, ld_dst, rows, cols, lo, vec);
    return avlen_launch_status();
  }
  int rc = avlen_cast_h16(src, ld_src, dst, ld_dst, rows, cols, 0, stream);
  return rc ? rc : avlen_cast_h16(src, ld_src, (bf16*)dst + lo, ld_dst, rows, cols, 2, stream);
}
extern "C" int avlen_cast_bf16(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, hipStream_t stream) {
  return avlen_cast_h16(src, ld_src, dst, ld_dst, rows, cols, 0, stream);
}

// 8 output elements per thread, 8-byte loads (source rows only need an even leading dimension: 202 floats for the spectrogram)
__global__ void cast_rows_indexed8_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, long rows, int cols,
                                          const int* __restrict__ row_index, int rpi, int fmt) {
  const int per_row = ldd >> 3;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const long r = i / per_row; const int c = (int)(i - r * per_row) << 3;
  const long item = r / rpi, rs = row_index ? (long)row_index[item] * rpi + (r - item * rpi) : r;
  const float* s = src + rs * lds_ + c;
  bf16x8 o;
  if (c + 8 <= cols) {
#pragma unroll
    for (int e = 0; e < 4; e++) { const float2 v = *reinterpret_cast<const float2*>(s + 2 * e); o[2 * e] = cvt16(v.x, fmt); o[2 * e + 1] = cvt16(v.y, fmt); }
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = c + e < cols ? cvt16(s[e], fmt) : (bf16)0.f;
  }
  *reinterpret_cast<bf16x8*>(dst + r * ldd + c) = o;
}

int avlen_cast_bf16_indexed(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, const int* row_index,
                            int rows_per_item, hipStream_t stream, int fmt) {
  if (!row_index) return avlen_cast_h16(src, ld_src, dst, ld_dst, rows, cols, fmt, stream);
  long tot = rows * ld_dst;
  if (tot <= 0 || rows_per_item <= 0) return AVLEN_ERR_ARG;
  if (!(ld_src & 1) && !(ld_dst & 7) && !((uintptr_t)src & 7) && !((uintptr_t)dst & 15)) {
    long t8 = tot >> 3;
    hipLaunchKernelGGL(cast_rows_indexed8_kernel, dim3((unsigned)((t8 + 255) / 256)), dim3(256), 0, stream, src, ld_src, (bf16*)dst,
                       ld_dst, rows, cols, row_index, rows_per_item, fmt);
    return avlen_launch_status();
  }
  hipLaunchKernelGGL(cast_rows_indexed_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, src, ld_src, (bf16*)dst,
                     ld_dst, rows, cols, row_index, rows_per_item, fmt);
  return avlen_launch_status();
}

extern "C" int avlen_pack_conv_weight_h16(const float* w_oihw, void* w_packed, int O, int I, int KH, int KW, int Cpad, int fmt,
                                          hipStream_t stream) {
  if (fmt < 0 || fmt > 2) return AVLEN_ERR_ARG;
  long tot = (long)O * KH * KW * Cpad;
  hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, w_oihw, (bf16*)w_packed, O, I, KH, KW, Cpad, fmt);
  return avlen_launch_status();
}
extern "C" int avlen_pack_conv_weight_bf16(const float* w_oihw, void* w_packed, int O, int I, int KH, int KW, int Cpad,
                                           hipStream_t stream) {
  return avlen_pack_conv_weight_h16(w_oihw, w_packed, O, I, KH, KW, Cpad, 0, stream);
}

// w16 [cout][K] -> fragment order [cout/16][K/32][lane = 16 q + r][8]: lane's chunk = w16[16 t + r][32 i + 8 q ..]
__global__ void pack_conv_frag_kernel(const uint4* __restrict__ w16, uint4* __restrict__ w16f, int cout, int K) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;              // one 16-byte chunk
  const int ksteps = K / 32;
  if (idx >= (long)cout * ksteps * 4) return;
  const int lane = idx & 63, i = (idx >> 6) % ksteps, t = (int)((idx >> 6) / ksteps), r = lane & 15, q = lane >> 4;
  w16f[idx] = w16[((long)(t * 16 + r) * K + i * 32 + 8 * q) >> 3];
}
extern "C" int avlen_pack_conv_weight_frag(const void* w16, void* w16f, int cout, int K, hipStream_t stream) {
  if (!w16 || !w16f || cout % 16 || K % 32 || cout <= 0 || K <= 0) return AVLEN_ERR_ARG;
  const long tot = (long)cout * (K / 32) * 4;
  hipLaunchKernelGGL(pack_conv_frag_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, (const uint4*)w16, (uint4*)w16f, cout, K);
  return avlen_launch_status();
}

extern "C" int avlen_pack_fc_after_flatten_h16(const float* w, void* w_packed, int O, int C, int HW, int fmt, hipStream_t stream) {
  if (fmt < 0 || fmt > 2) return AVLEN_ERR_ARG;
  long tot = (long)O * C * HW;
  hipLaunchKernelGGL(pack_fc_bf16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, w, (bf16*)w_packed, O, C, HW, fmt);
  return avlen_launch_status();
}
extern "C" int avlen_pack_fc_after_flatten_bf16(const float* w, void* w_packed, int O, int C, int HW, hipStream_t stream) {
  return avlen_pack_fc_after_flatten_h16(w, w_packed, O, C, HW, 0, stream);
}
