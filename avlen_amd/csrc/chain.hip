// Fused "row-batch chain" kernel: a whole sequence of d=256 Linear / ReLU / residual / LayerNorm steps on a small batch
// (the single-target-token decoder, the collapsed `pretraining` encoder, the per-step tails of pi_q / pi_g / pi_l) in ONE
// launch instead of ~20 launches of 5-12 us each.
//
// One block = 16 batch rows, 16 waves; wave w owns output features 16w .. 16w+15.  The activation lives in registers
// between steps (fp32, layout of the transposed MFMA result: lane (c = lane&15, q = lane>>4) holds batch row c, features
// 16w + 4q + r, r in 0..3) and as a bf16 image in LDS (the B operand of the next step: X[row][k]).
//
// Everything a step needs except its weights is on chip before the first step runs:
//  * the program (step descriptors + the list of weight matrices) is copied from the kernel arguments into LDS -- a
//    dynamically indexed kernarg read is a ~300-cycle scalar round trip, and the chain would pay one or more per step;
//  * every bias / gamma / beta vector is copied into an LDS table by one cooperative pass (a global round trip per step
//    otherwise sits on the critical path: measured 2.6k of a LayerNorm step's 2.7k cycles).
// Weights: each step's [256][K] bf16 matrix is streamed through 3-stage LDS rings of [16][64] tiles, one ring per wave (a
// wave multiplies only its own 16 output features), by global_load_lds_dwordx4 (whole 128-byte lines, XOR swizzle on the
// source address), and the A fragments are read back with ds_read_b128; the stream itself needs no workgroup barrier.  (Fragment-shaped global loads straight to VGPRs,
// 16 rows x 64 B per instruction, ran at 13 B/clk per CU: 4.4 us per 128 KiB step.)  The tile stream is continuous ACROSS
// steps: while a step's epilogue / LayerNorm runs, the first tiles of the next Linear are already in flight (counted
// s_waitcnt vmcnt, raw s_barrier -- a __syncthreads() would drain them).  LayerNorm reduces inside the lane, across q by
// 2 shuffles and across the waves through 2 KB of LDS.  Residual operands and a parked activation live in two register
// save slots.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifdef AVLEN_CHAIN_LAB
__device__ long long g_chain_stamps[64];     // tools/chain_lab.hip: start time of every step (block 0)
#endif

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_ch[4096];     // K tails of the weight tiles

namespace {

constexpr int D = 256;            // feature width of every step's output
constexpr int KMAX = 352;         // widest input (pi_q fusion: features + pose encoding = 320; with the distractor's category input 341 -> 344).
constexpr int KMAX64 = 320;       // ... of a plain-bf16 program: its k blocks are 64 wide and the activation image is zero-filled to whole blocks
constexpr int XLD = KMAX + 8;     // LDS row stride (elements) of the bf16 activation image
constexpr int NTH = 1024, NW = 16;
constexpr int TILE_BYTES = D * 128;            // one ring stage: [256 features][64 k] bf16
constexpr int NS = 3;             // up to 2 tiles (64 KiB) in flight while one is multiplied
constexpr int NIMG = 2;           // activation images (a third operand is parked in a register save slot)
// Compensated bf16 (X3): the activation image is a PAIR of bf16 planes (hi, lo) and a block takes 8 batch rows instead of 16, so
// the pair fits the space of the plain image (the chain is bound by the weight stream, not by the half-used MFMA columns); the
// weight stream carries [16][32] hi + [16][32] lo per ring stage (the same 2 KiB) and a k-step is three MFMAs.
constexpr int XS_BYTES = NIMG * 16 * XLD * 2;
constexpr int RED_BYTES = NW * 16 * 2 * 4;
constexpr int MAX_LIN = 22, MAX_PAR = 32;      // Linear steps (largest program: 16) / 256-float parameter vectors per program
constexpr int ATT_BYTES = NW * 16 * 4 * 4;     // attention score partials [wave][row][key]

// what the kernel reads: compiled from avlen_chain on the host
struct DevOp { int kind, k, ld, ld2, act, res, buf, out_buf, par, div, seq; float scale; const void* p0; const void* p1; };
struct DevLin { const char* w; const char* wl; int ld, nkt, k, pad; };
struct DevProg {
  int n, n_lin, n_par, pad;
  DevOp op[AVLEN_CHAIN_MAX_OPS];
  DevLin lin[MAX_LIN];
  const float* par_src[MAX_PAR];              // par_src[i] -> LDS table row i (256 floats)
};
constexpr int PROG_BYTES = (sizeof(DevProg) + 15) / 16 * 16;
constexpr int PAR_BYTES = MAX_PAR * D * 4;
constexpr int LDS_BYTES = NS * TILE_BYTES + XS_BYTES + RED_BYTES + ATT_BYTES + PROG_BYTES + PAR_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "chain kernel LDS budget");
static_assert(sizeof(DevProg) <= 4000, "kernel argument limit");

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ float lo_of(float v, bf16 h) { return v - (float)h; }
// LDS-only barrier: does not drain the weight tiles in flight
__device__ __forceinline__ void bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// XS = log2 of the grid's stride between WORKING blocks: with XS = 3 only every 8th block of the grid works (the others return at
// once), so that -- blocks being dealt round-robin over the 8 XCDs -- all working blocks sit on ONE XCD and stream the program's
// weights out of that XCD's L2 after the first of them has missed, instead of eight L2s each fetching all of it from the fabric
// (placement is a speed assumption only: MI355X_MICROARCH.md, Workgroup dispatch).
template <bool X3>
__global__ __launch_bounds__(NTH) void chain_kernel(DevProg kprog, int B, int XS) {
  constexpr int RB = X3 ? 8 : 16;                                            // batch rows per block
  if (blockIdx.x & ((1u << XS) - 1)) return;
  const int blk = blockIdx.x >> XS;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ring = lds;                                                          // [NS][256][128 B]
  bf16* xs = reinterpret_cast<bf16*>(lds + NS * TILE_BYTES);                 // [NIMG][RB][XLD] (X3: hi planes, then the lo planes)
  bf16* xl = xs + NIMG * RB * XLD;                                           // X3 only
  float* red = reinterpret_cast<float*>(lds + NS * TILE_BYTES + XS_BYTES);   // [8][16 rows][4]
  float* att = reinterpret_cast<float*>(lds + NS * TILE_BYTES + XS_BYTES + RED_BYTES);                  // [NW][16][4]
  DevProg* sp = reinterpret_cast<DevProg*>(lds + NS * TILE_BYTES + XS_BYTES + RED_BYTES + ATT_BYTES);
  float* par = reinterpret_cast<float*>(lds + NS * TILE_BYTES + XS_BYTES + RED_BYTES + ATT_BYTES + PROG_BYTES);   // [MAX_PAR][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
  const int cr = c & (RB - 1);                         // X3: lanes 8..15 mirror rows 0..7 (their results are never stored)
  const int row = blk * RB + cr;                // this lane's batch row
  const bool rok = c < RB && row < B;
  const int n0 = wave * 16;

  // ---- program and parameter vectors -> LDS ----
  {
    const int* src = reinterpret_cast<const int*>(&kprog);
    int* dst = reinterpret_cast<int*>(sp);
    for (int i = tid; i < (int)(sizeof(DevProg) / 4); i += NTH) dst[i] = src[i];
    const int npar4 = kprog.n_par * (D / 4);           // float4 slots
    for (int i = tid; i < npar4; i += NTH) {
      const int v = i >> 6, o = (i & 63) * 4;          // 64 float4 per vector
      *reinterpret_cast<float4*>(par + v * D + o) = *reinterpret_cast<const float4*>(kprog.par_src[v] + o);
    }
  }
  __syncthreads();
  const int n_ops = sp->n, n_lin = sp->n_lin;

  float cur[4] = {0.f, 0.f, 0.f, 0.f}, sav0[4] = {0.f, 0.f, 0.f, 0.f}, sav1[4] = {0.f, 0.f, 0.f, 0.f};
  auto publish = [&](int buf) {                        // cur -> bf16 image xs[buf][row][feature] (X3: + the low plane)
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; r++) o[r] = (bf16)cur[r];
    if (c < RB) {
      *reinterpret_cast<bf16x4*>(&xs[(buf * RB + c) * XLD + n0 + q * 4]) = o;
      if (X3) {
        bf16x4 l;
#pragma unroll
        for (int r = 0; r < 4; r++) l[r] = (bf16)lo_of(cur[r], o[r]);
        *reinterpret_cast<bf16x4*>(&xl[(buf * RB + c) * XLD + n0 + q * 4]) = l;
      }
    }
  };

  // ---- the weight-tile stream: tile t of the program = (Linear step, 64-wide k block); ring stage = t % NS ----
  // Every thread issues 2 pieces per tile: slot = r*1024 + tid -> feature row slot>>3, 16-byte chunk slot&7.
  int ld_idx = 0, ld_kt = 0;           // next tile to issue: Linear index / k block   (ld_idx == n_lin: stream exhausted)
  DevLin ldl = sp->lin[0];
  int issued = 0, consumed = 0;        // tiles issued / tiles whose data has been waited for
  int st_issue = 0, st_cons = 0;       // their ring stages (issued % NS, consumed % NS)
  // Measured and rejected (end of round 4): a third tile in flight per wave through registers.  The X3 stream is bound by its 64 KiB
  // in flight (256 KiB per step in 6.4 us = 19 B/clk) and LDS is full, so every third tile went by fragment-shaped 16-byte loads
  // straight into registers -- held in C++ variables the compiler moved them (or parked them in scratch) while the load was in
  // flight; held in the top accumulator registers a56 .. a63 from asm it was correct (131 parity tests) and SLOWER: 83 -> 96 us per
  // launch.  Vector-memory results return in order, and the 64-byte-row fragment loads (13 B/clk on their own) hold back the DMA
  // tiles queued behind them.
  // Each wave streams ONLY the 16 weight rows it multiplies (2 KiB per 64-wide k block = two 1 KiB pieces) into a private
  // 3-stage ring, so the weight stream needs no workgroup barrier at all: the issuing wave's own counted vmcnt orders its
  // ds_reads behind its LDS-DMA, and the waves drift freely inside a step.
  char* wring = ring + wave * (NS * 2048);
  auto issue_one = [&]() {             // issue the next tile of the stream, if any
    if (ld_idx >= n_lin) return;
    char* stage = wring + st_issue * 2048;
    if (++st_issue == NS) st_issue = 0;
#if !defined(AVLEN_CHAIN_LAB) || AVLEN_CHAIN_LAB != 2
    if (X3) {                          // piece 0: [16 rows][32 k] of W_hi, piece 1: of W_lo (64-byte rows, chunk ^ ((row >> 1) & 3))
      const int frow = n0 + (lane >> 2), ch = (lane & 3) ^ ((frow >> 1) & 3);
      const int kcol = ld_kt * 32 + ch * 8;
      const long off = ((long)frow * ldl.ld + kcol) * 2;
      const bool ok = kcol < ldl.k;
      const char* zp = (const char*)g_zero_page_ch + tid * 16;
#if defined(AVLEN_CHAIN_LAB) && AVLEN_CHAIN_LAB == 4
      // lab: what a pre-packed stream would cost -- each piece ONE contiguous KiB (8 whole lines) instead of 16 half lines
      const long offc = ((long)n0 * ldl.ld) * 2 + (long)ld_kt * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const void*)(ldl.w + offc), (__attribute__((address_space(3))) void*)(stage), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(ldl.wl + offc), (__attribute__((address_space(3))) void*)(stage + 1024), 16, 0, 0);
#else
      __builtin_amdgcn_global_load_lds((const void*)(ok ? ldl.w + off : zp), (__attribute__((address_space(3))) void*)(stage), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(ok ? ldl.wl + off : zp), (__attribute__((address_space(3))) void*)(stage + 1024), 16, 0, 0);
#endif
    } else {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int frow = n0 + r * 8 + (lane >> 3), ch = (lane & 7) ^ ((frow >> 1) & 7);
      const int kcol = ld_kt * 64 + ch * 8;
      const char* src = kcol < ldl.k ? ldl.w + ((long)frow * ldl.ld + kcol) * 2 : (const char*)g_zero_page_ch + tid * 16;
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(stage + r * 1024), 16, 0, 0);
    }
    }
#endif
    issued++;
    if (++ld_kt == ldl.nkt) { ld_kt = 0; ld_idx++; ldl = sp->lin[ld_idx < n_lin ? ld_idx : 0]; }
  };
  issue_one(); issue_one();            // two tiles ahead from the start

  DevOp op = sp->op[0];
  for (int s = 0; s < n_ops; s++) {
    const DevOp nxt_op = sp->op[s + 1 < n_ops ? s + 1 : s];        // next step's descriptor: in flight during this step
#ifdef AVLEN_CHAIN_LAB
    if (tid == 0 && blk == 0) g_chain_stamps[s] = clock64();
#endif
    switch (op.kind) {
      case AVLEN_CH_LOAD_X16: {                        // bf16 global rows [B][ld] -> xs[buf][.][0:K)
        bar();
        const bf16* src = (const bf16*)op.p0;
        const bf16* srcl = (const bf16*)op.p1;         // X3: the low plane (same layout)
        const int cpr = X3 ? ((op.k + 31) >> 5) << 2 : ((op.k + 63) >> 6) << 3;   // 16-byte chunks per row, zero-filled up to a whole k block (32 / 64 wide)
        for (int i = tid; i < RB * cpr; i += NTH) {
          int rr = i / cpr, ch = i - rr * cpr;
          int gr = blk * RB + rr;
          uint4 v = make_uint4(0, 0, 0, 0), vl = v;
          if (gr < B && ch * 8 < op.k) {
            v = *reinterpret_cast<const uint4*>(src + (long)gr * op.ld + ch * 8);
            if (X3) vl = *reinterpret_cast<const uint4*>(srcl + (long)gr * op.ld + ch * 8);
          }
          *reinterpret_cast<uint4*>(&xs[(op.buf * RB + rr) * XLD + ch * 8]) = v;
          if (X3) *reinterpret_cast<uint4*>(&xl[(op.buf * RB + rr) * XLD + ch * 8]) = vl;
        }
        bar();
        break;
      }
      case AVLEN_CH_LOAD_CUR: {                        // fp32 global [B][ld] (256 features) -> cur (+ bf16 image)
        const float* src = (const float*)op.p0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok) v = *reinterpret_cast<const float4*>(src + (long)(row / op.div) * op.ld + n0 + q * 4);
        cur[0] = v.x; cur[1] = v.y; cur[2] = v.z; cur[3] = v.w;
        bar();
        publish(op.buf);
        bar();
        break;
      }
      case AVLEN_CH_LINEAR: {                          // cur = act(W x + b) [+ save slot]; x = xs[buf][.][0:K)
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nkt = X3 ? (op.k + 31) >> 5 : (op.k + 63) >> 6;
        const bf16* xrow = &xs[(op.buf * RB + cr) * XLD + q * 8];
        const bf16* xrowl = &xl[(op.buf * RB + cr) * XLD + q * 8];
        const int wr = n0 + c;                         // this lane's feature row inside the tile
        const int wsw = (wr >> 1) & 7;
        for (int kt = 0; kt < nkt; kt++) {
          // tile `consumed` has landed once only the pieces of the tiles issued after it are outstanding
          const int ahead = issued - consumed - 1;
          if (ahead >= 2) wait_vmcnt<4>(); else if (ahead == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();
          const char* stage = wring + st_cons * 2048;   // private to this wave: no barrier
          if (++st_cons == NS) st_cons = 0;
          consumed++;
          issue_one();
#if !defined(AVLEN_CHAIN_LAB) || AVLEN_CHAIN_LAB != 3
          if (X3) {
            const int sw3 = (wr >> 1) & 3;
            bf16x8 wh = *reinterpret_cast<const bf16x8*>(stage + c * 64 + ((q ^ sw3) << 4));
            bf16x8 wl = *reinterpret_cast<const bf16x8*>(stage + 1024 + c * 64 + ((q ^ sw3) << 4));
            bf16x8 xh = *reinterpret_cast<const bf16x8*>(xrow + kt * 32);
            bf16x8 xlo = *reinterpret_cast<const bf16x8*>(xrowl + kt * 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xlo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
          } else {                                     // all four fragment reads in flight before the first MFMA
            bf16x8 wf0 = *reinterpret_cast<const bf16x8*>(stage + c * 128 + ((q ^ wsw) << 4));
            bf16x8 wf1 = *reinterpret_cast<const bf16x8*>(stage + c * 128 + (((4 + q) ^ wsw) << 4));
            bf16x8 xf0 = *reinterpret_cast<const bf16x8*>(xrow + kt * 64);
            bf16x8 xf1 = *reinterpret_cast<const bf16x8*>(xrow + kt * 64 + 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0, xf0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, xf1, acc, 0, 0, 0);
          }
#else
          acc[0] += stage[0];
#endif
        }
        float4 bv = op.par >= 0 ? *reinterpret_cast<const float4*>(par + op.par * D + n0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float v = acc[r] + b4[r];
          if (op.act == AVLEN_ACT_RELU) v = fmaxf(v, 0.f);
          if (op.res == 1) v += sav0[r];
          if (op.res == 2) v += sav1[r];
          cur[r] = v;
        }
        if (op.out_buf == op.buf) bar();               // in place: every wave must have finished reading xs[buf]
        publish(op.out_buf);                           // (the other image was last read before the previous step's barrier)
        bar();
        break;
      }
      case AVLEN_CH_LAYERNORM: {                       // cur = LN(cur) * g + b over the 256 features of each row
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) { s1 += cur[r]; s2 += cur[r] * cur[r]; }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        // partials as [8 float4 slots][16 rows][4 waves]: a row's 16 + 16 partial sums are 8 conflict-free ds_read_b128
        if (q == 0) { red[(((wave >> 2) * 16 + c) << 2) + (wave & 3)] = s1; red[((((wave >> 2) + 4) * 16 + c) << 2) + (wave & 3)] = s2; }
        const float4 gv = *reinterpret_cast<const float4*>(par + op.par * D + n0 + q * 4);
        const float4 bv = *reinterpret_cast<const float4*>(par + (op.par + 1) * D + n0 + q * 4);
        bar();
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < NW / 4; w4++) {
          float4 a = *reinterpret_cast<const float4*>(&red[(w4 * 16 + c) << 2]);
          float4 b = *reinterpret_cast<const float4*>(&red[((w4 + 4) * 16 + c) << 2]);
          t1 += (a.x + a.y) + (a.z + a.w); t2 += (b.x + b.y) + (b.z + b.w);
        }
        const float mean = t1 * (1.f / D);
        const float var = fmaxf(t2 * (1.f / D) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + 1e-5f);
        float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int r = 0; r < 4; r++) cur[r] = (cur[r] - mean) * rstd * g4[r] + b4[r];
        publish(op.out_buf);                           // xs was last read before the barrier above
        bar();                                         // also: everyone has read `red` before the next LayerNorm writes it
        break;
      }
      case AVLEN_CH_SAVE: {                            // registers -> save slot `res` (0 | 1)
#pragma unroll
        for (int r = 0; r < 4; r++) { if (op.res == 0) sav0[r] = cur[r]; else sav1[r] = cur[r]; }
        break;
      }
      case AVLEN_CH_RECALL: {                          // save slot `res` -> registers and image `out_buf`
#pragma unroll
        for (int r = 0; r < 4; r++) cur[r] = op.res == 0 ? sav0[r] : sav1[r];
        bar();
        publish(op.out_buf);
        bar();
        break;
      }
      case AVLEN_CH_STORE: {                           // cur -> fp32 global and/or bf16 global (last row of every `div` rows)
        float* dst = (float*)op.p0; bf16* dst16 = (bf16*)op.p1;
        if (rok && (row % op.div) == op.div - 1) {
          const long orow = row / op.div;
          if (dst) *reinterpret_cast<float4*>(dst + orow * op.ld + n0 + q * 4) = make_float4(cur[0], cur[1], cur[2], cur[3]);
          if (dst16) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (bf16)cur[r];
            *reinterpret_cast<bf16x4*>(dst16 + orow * op.ld2 + n0 + q * 4) = o;
          }
        }
        break;
      }
      case AVLEN_CH_ATTN: {                            // 8 heads x D=32 inside groups of `seq` consecutive rows
        const int S = op.seq, g0 = (cr / S) * S;
        float qv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) qv[r] = op.res == 0 ? sav0[r] : sav1[r];
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (j < S) {
            bf16x4 kv = *reinterpret_cast<const bf16x4*>(&xs[(op.buf * RB + g0 + j) * XLD + n0 + q * 4]);
            float kf[4] = {(float)kv[0], (float)kv[1], (float)kv[2], (float)kv[3]};
            if (X3) {
              bf16x4 kl = *reinterpret_cast<const bf16x4*>(&xl[(op.buf * RB + g0 + j) * XLD + n0 + q * 4]);
#pragma unroll
              for (int r = 0; r < 4; r++) kf[r] += (float)kl[r];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) part[j] += qv[r] * kf[r];
          }
#pragma unroll
        for (int j = 0; j < 4; j++) { part[j] += __shfl_xor(part[j], 16, 64); part[j] += __shfl_xor(part[j], 32, 64); }
        if (q == 0 && c < RB) *reinterpret_cast<float4*>(&att[(wave * 16 + c) * 4]) = make_float4(part[0], part[1], part[2], part[3]);
        bar();
        // a head is 32 features = this wave and its neighbour
        const float4 pa = *reinterpret_cast<const float4*>(&att[((wave & ~1) * 16 + cr) * 4]);
        const float4 pb = *reinterpret_cast<const float4*>(&att[((wave | 1) * 16 + cr) * 4]);
        float sc[4] = {pa.x + pb.x, pa.y + pb.y, pa.z + pb.z, pa.w + pb.w};
        const float* km = (const float*)op.p0;
        const long grp = (long)(blk * RB + cr) / S;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          bool ok = j < S && (!km || !rok || km[grp * S + j] != 0.f);
          sc[j] = ok ? sc[j] * op.scale : -INFINITY;
          mx = fmaxf(mx, sc[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) { sc[j] = sc[j] == -INFINITY ? 0.f : __expf(sc[j] - mx); den += sc[j]; }
        const float inv = den > 0.f ? 1.f / den : 0.f;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (j < S) {
            bf16x4 vv = *reinterpret_cast<const bf16x4*>(&xs[(op.ld2 * RB + g0 + j) * XLD + n0 + q * 4]);
            float vf[4] = {(float)vv[0], (float)vv[1], (float)vv[2], (float)vv[3]};
            if (X3) {
              bf16x4 vl = *reinterpret_cast<const bf16x4*>(&xl[(op.ld2 * RB + g0 + j) * XLD + n0 + q * 4]);
#pragma unroll
              for (int r = 0; r < 4; r++) vf[r] += (float)vl[r];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] += sc[j] * inv * vf[r];
          }
#pragma unroll
        for (int r = 0; r < 4; r++) cur[r] = o[r];
        bar();                                         // every wave has read the K / V images
        publish(op.out_buf);
        bar();
        break;
      }
      case AVLEN_CH_ADD_PE: {                          // cur += table[step[row / div]]
        const float* tab = (const float*)op.p0; const float* step = (const float*)op.p1;
        int idx = rok ? (int)step[row / op.div] : 0;
        idx = idx < 0 ? 0 : (idx >= op.k ? op.k - 1 : idx);
        const float4 pv = *reinterpret_cast<const float4*>(tab + (long)idx * D + n0 + q * 4);
        cur[0] += pv.x; cur[1] += pv.y; cur[2] += pv.z; cur[3] += pv.w;
        bar();
        publish(op.out_buf);
        bar();
        break;
      }
      default: break;
    }
    op = nxt_op;
  }
#ifdef AVLEN_CHAIN_LAB
  if (tid == 0 && blk == 0) g_chain_stamps[n_ops] = clock64();
#endif
}

}  // namespace

static int g_chain_one_xcd = 1;
extern "C" void avlen_set_chain_one_xcd(int on) { g_chain_one_xcd = on ? 1 : 0; }

int avlen_chain_run(const avlen_chain* prog, int B, hipStream_t stream, int x3) {
  if (!prog || prog->n < 1 || prog->n > AVLEN_CHAIN_MAX_OPS || B <= 0) return AVLEN_ERR_ARG;
  DevProg dp;
  memset(&dp, 0, sizeof(dp));
  dp.n = prog->n;
  for (int i = 0; i < prog->n; i++) {
    const avlen_chain_op& o = prog->op[i];
    if (o.buf < 0 || o.buf >= NIMG || o.out_buf < 0 || o.out_buf >= NIMG) return AVLEN_ERR_ARG;
    DevOp& d = dp.op[i];
    d.kind = o.kind; d.k = o.k; d.ld = o.ld; d.ld2 = o.ld2; d.act = o.act; d.res = o.res; d.buf = o.buf; d.out_buf = o.out_buf;
    d.par = -1; d.p0 = o.p0; d.p1 = o.p1; d.div = o.div > 0 ? o.div : 1; d.seq = o.seq; d.scale = o.scale;
    if (o.kind == AVLEN_CH_ATTN && ((o.seq != 1 && o.seq != 2 && o.seq != 4) || o.ld2 < 0 || o.ld2 >= NIMG || (o.res != 0 && o.res != 1)))
      return AVLEN_ERR_ARG;
    if (o.kind == AVLEN_CH_ADD_PE && (!o.p0 || !o.p1 || o.k < 1)) return AVLEN_ERR_ARG;
    if (o.kind == AVLEN_CH_LINEAR) {
      if (o.k % 8 || o.k > (x3 ? KMAX : KMAX64) || o.k < 8 || o.ld < o.k || o.ld % 8 || !o.p0 || dp.n_lin >= MAX_LIN) return AVLEN_ERR_ARG;
      if (x3 && !o.p2) return AVLEN_ERR_ARG;
      dp.lin[dp.n_lin++] = DevLin{(const char*)o.p0, (const char*)o.p2, o.ld, x3 ? (o.k + 31) >> 5 : (o.k + 63) >> 6, o.k, 0};
      if (o.p1) {
        if (dp.n_par >= MAX_PAR || ((uintptr_t)o.p1 & 15)) return AVLEN_ERR_ARG;
        d.par = dp.n_par; dp.par_src[dp.n_par++] = (const float*)o.p1;
      }
    } else if (o.kind == AVLEN_CH_LAYERNORM) {
      if (!o.p0 || !o.p1 || dp.n_par + 2 > MAX_PAR || (((uintptr_t)o.p0 | (uintptr_t)o.p1) & 15)) return AVLEN_ERR_ARG;
      d.par = dp.n_par; dp.par_src[dp.n_par++] = (const float*)o.p0; dp.par_src[dp.n_par++] = (const float*)o.p1;
    } else if (o.kind == AVLEN_CH_LOAD_X16) {
      if (o.k % 8 || o.k > (x3 ? KMAX : KMAX64) || o.ld % 8 || (x3 && !o.p1)) return AVLEN_ERR_ARG;
    }
  }
  static unsigned long long done0 = 0, done1 = 0;
  if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&chain_kernel<false>), LDS_BYTES, &done0) != AVLEN_OK ||
      avlen_set_dyn_lds(reinterpret_cast<const void*>(&chain_kernel<true>), LDS_BYTES, &done1) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  // up to 32 working blocks fit one XCD (32 CUs, one 160 KB block each); larger batches use the plain grid
  const int nblk = ceil_div(B, x3 ? 8 : 16);
  const int xs = (g_chain_one_xcd && nblk > 1 && nblk <= 32) ? 3 : 0;
  if (x3) hipLaunchKernelGGL(chain_kernel<true>, dim3(nblk << xs), dim3(NTH), LDS_BYTES, stream, dp, B, xs);
  else hipLaunchKernelGGL(chain_kernel<false>, dim3(nblk << xs), dim3(NTH), LDS_BYTES, stream, dp, B, xs);
  return avlen_launch_status();
}
