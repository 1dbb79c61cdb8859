// CLIP text tower (third party: open_clip / clip.model.Transformer, 12 x ResidualAttentionBlock, width 512, 8 heads, QuickGELU;
// call site ss_baselines/savi/ppo/policy.py:847-849) as ONE launch, SEQUENCE-STATIONARY: one workgroup (512 threads) per dialog
// carries its <= 80 tokens through all 12 layers.
//
// Why: as a chain of launches (71 kernels) the tower is latency-bound -- 12-20 us per GEMM whatever the row count, 0.9 ms per step,
// on ALL 256 CUs (a 2.5 k-row GEMM cannot hide its prologue / epilogue) -- and it serialises with the visual towers, which also
// want every CU.  Here a dialog's residual stream never leaves the CU:
//   * residual x (80 x 512 fp32) lives in REGISTERS in MFMA accumulator layout (wave w owns columns [64 w, 64 w + 64): 80 VGPRs);
//     the out_proj / c_proj products accumulate straight into it,
//   * LayerNorm output, per-head-pair Q / K / V, the attention output and the MLP hidden chunk are fp16 (bf16) images in LDS,
//   * the weights (6.3 MB per layer) are STREAMED: every wave owns a private, fragment-ordered stream (packed once by
//     avlen_clip_pack_stream: [wave][layer][768 fragments][64 lanes][16 B] in exactly the order the wave consumes them) and keeps
//     CT_RING fragments (1 KiB each) in flight through a register ring -- no LDS staging, no descriptor set-up, no barrier on the weight path.
// A CU's ingest (~64 B/clk) bounds a layer at ~47 us, its matrix pipe at ~55 us for 80 tokens: ~0.7-0.8 ms for the tower on 64 CUs,
// leaving 192 CUs to the visual towers of the same step (tower_x3 / tower_head run beside it instead of before it).
// Arithmetic: 16-bit operands (fp16 for AVLEN_PREC_FP16, bf16 for AVLEN_PREC_BF16), fp32 accumulation, LayerNorm / softmax /
// QuickGELU in fp32 -- the same formats as the launch-per-GEMM path, with the LayerNorm applied explicitly (not folded).
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include "tower_util.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;

constexpr int CT_TH = 512, CT_ROWS = 80;
constexpr int CT_PAD = 16;                                  // padding fragments at the end of a wave's stream (>= the deepest ring)
constexpr int CT_FRAGS = 384;                               // weight fragments (1 KiB) per wave, layer and column half: 2 x (48 + 16) + 4 x (32 + 32)
// LDS map (bytes).  Every image has 16 bytes of padding per row: the 16 rows of an MFMA operand fragment then start in 16 distinct
// 16-byte bank slots (conflict-free ds_read_b128) with NO address arithmetic -- a k-step is an immediate offset.
constexpr int XN_ROW = 1024 + 16, QK_ROW = 128 + 16, HC_ROW = 512 + 16;
constexpr int XN_OFF = 0;                                   // LayerNorm output [80][512] 16-bit
constexpr int QO_OFF = XN_OFF + 80 * XN_ROW;                // Q, then (in place) the attention output: [2 heads][80][64]
constexpr int KS_OFF = QO_OFF + 2 * 80 * QK_ROW;            // K [2][80][64]
constexpr int VS_OFF = KS_OFF + 2 * 80 * QK_ROW;            // V [2][96][64]: rows 80 .. 95 stay zero (finite operands for masked keys)
constexpr int HC_OFF = QO_OFF;                              // MLP hidden chunk [80][256] 16-bit: aliases Q | K
constexpr int PART_OFF = VS_OFF + 2 * 96 * QK_ROW;          // LayerNorm partials [80][8 waves] float2
constexpr int CT_LDS = PART_OFF + 80 * 8 * 8;
static_assert(HC_OFF + 80 * HC_ROW <= VS_OFF && CT_LDS <= 160 * 1024, "CLIP tower LDS budget");

struct ClipLayerP { const float *ln1g, *ln1b, *ln2g, *ln2b, *b_in, *b_out, *b_fc, *b_proj; };
struct ClipArgs {
  const int64_t* tokens; const float* tok_emb; const float* pos_emb; const uint4* wstream; float* E;
  int ctx, vocab, layers; long frags_per_wave;              // stream stride of a wave (fragments)
  ClipLayerP L[12];
  long long* prof;                                          // lab builds (AVLEN_CT_PROF): per-workgroup phase cycle totals [2 B][8]
  unsigned* flags; char* xchg; int B;                       // K / V hand-off of the 5-tile dialogs: flag word per (dialog, column half), slots
  unsigned* xflags; char* xslots;                           // partial-sum exchange between the two column halves: flag per workgroup, 2 slots each
  // device-side work list (memoised tower, avlen_clip_text_cached_fwd): pair i < *count carries dialog row_idx[i]; null = all B rows
  const int* row_idx; const int* count;
};
// A hand-off that never arrives (partner workgroup lost): the dialog's output row is poisoned with NaN -- a stale or garbage
// embedding would flow into the rollout unnoticed (the GRU sequence kernels do the same, train_gru.hip)
#define CT_GIVE_UP() do { if (threadIdx.x < 128) *reinterpret_cast<float4*>(a.E + (long)b * 512 + 4 * threadIdx.x) = \
    make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")); return; } while (0)
#ifdef AVLEN_CT_PROF
#define CT_T0() long long ct_t = __builtin_amdgcn_s_memtime(); long long ct_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CT_PH(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); ct_acc[k] += n_ - ct_t; ct_t = n_; } while (0)
#define CT_DUMP() do { if (a.prof && tid == 0) for (int k_ = 0; k_ < 8; k_++) a.prof[(long)blockIdx.x * 8 + k_] = ct_acc[k_]; } while (0)
#else
#define CT_T0() do { } while (0)
#define CT_PH(k) do { } while (0)
#define CT_DUMP() do { } while (0)
#endif

template <bool F16> __device__ __forceinline__ f32x4 cmma(const uint4& w, const uint4& x, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, w), __builtin_bit_cast(h16x8, x), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
// four fp32 -> four 16-bit values (8 bytes)
template <bool F16> __device__ __forceinline__ uint2 cvt4(float a, float b, float c, float d) {
  if constexpr (F16) { const h16x4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d}; return __builtin_bit_cast(uint2, v); }
  else return make_uint2(pack2((f32x2){a, b}), pack2((f32x2){c, d}));
}
__device__ __forceinline__ uint4 lds16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

// One attention unit: head `ah` of the pair, query tile `mt`; scores transposed (lane (r16, q) holds query 16 mt + r16, keys
// 16 nt + 4 q + r) so P stays in registers; every key tile is computed and the causal mask does the rest (no branches: a masked
// tile costs two MFMAs, a branch would cut the block).  The output overwrites the unit's own Q rows.
template <bool F16, int CT_MT>
__device__ __forceinline__ void clip_attn_unit(char* lds, int ah, int mt, int r16, int q) {
  const char* Qb = lds + QO_OFF + ah * 80 * QK_ROW;
  const char* Kb = lds + KS_OFF + ah * 80 * QK_ROW;
  const char* Vb = lds + VS_OFF + ah * 96 * QK_ROW;
  const int qi = 16 * mt + r16;
  uint4 qf[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) qf[kk] = lds16(Qb + qi * QK_ROW + (4 * kk + q) * 16);
  f32x4 sacc[CT_MT];
#pragma unroll
  for (int nt = 0; nt < CT_MT; nt++) {
    sacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
      sacc[nt] = cmma<F16>(lds16(Kb + (16 * nt + r16) * QK_ROW + (4 * kk + q) * 16), qf[kk], sacc[nt]);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int nt = 0; nt < CT_MT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float v = (16 * nt + 4 * q + r <= qi) ? sacc[nt][r] : -INFINITY;
      sacc[nt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int nt = 0; nt < CT_MT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {                           // exp(-inf) = 0: the diagonal keeps mx finite
      const float pv = __expf(sacc[nt][r] - mx);
      sum += pv; sacc[nt][r] = pv;
    }
  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
  // O^T = V^T P^T over 32-key steps; V^T fragments by the transposing LDS read from the row-major V image
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; dt++) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < (CT_MT + 1) / 2; kk++) {
    const uint2 p0 = cvt4<F16>(sacc[2 * kk][0], sacc[2 * kk][1], sacc[2 * kk][2], sacc[2 * kk][3]);
    uint2 p1 = make_uint2(0u, 0u);
    if (2 * kk + 1 < CT_MT) p1 = cvt4<F16>(sacc[(2 * kk + 1) % CT_MT][0], sacc[(2 * kk + 1) % CT_MT][1], sacc[(2 * kk + 1) % CT_MT][2], sacc[(2 * kk + 1) % CT_MT][3]);
    const uint4 pf = make_uint4(p0.x, p0.y, p1.x, p1.y);
    const int vr = 32 * kk + 4 * q + (r16 >> 2);            // and vr + 16
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
      const int cb = (16 * dt + 4 * (r16 & 3)) * 2;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Vb + vr * QK_ROW + cb));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Vb + (vr + 16) * QK_ROW + cb));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      oacc[dt] = cmma<F16>(make_uint4(l2.x, l2.y, h2.x, h2.y), pf, oacc[dt]);
    }
  }
  const float inv = __builtin_amdgcn_rcpf(sum);             // sum >= 1 (the diagonal term is exp(0))
  char* Ob = lds + QO_OFF + ah * 80 * QK_ROW;
#pragma unroll
  for (int dt = 0; dt < 4; dt++) {
    const uint2 o = cvt4<F16>(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    *reinterpret_cast<uint2*>(Ob + qi * QK_ROW + (16 * dt + 4 * q) * 2) = o;
  }
}

// ---- K / V hand-off between the two workgroups of a 5-tile dialog (cdna_hip_programming.md, Guideline 16, form R1) ----
// slot (dialog, layer, head pair): [k | v][head of the pair][row 0 .. 47][64] 16-bit = 24,576 B.  Producer: write-through (agent-scope
// relaxed atomic) 8-byte stores, every storing wave drains, workgroup barrier, ONE lane stores the sequence number to the dialog's
// flag word.  Consumer: one lane polls the flag, ONE agent-scope acquire, barrier, then plain loads.  The flags are zeroed by a
// memset node in front of every launch; a slot is written once per launch, so the producer never has to wait for the consumer.
constexpr int CT_XROWS = 48, CT_SLOT = 2 * 2 * CT_XROWS * 128, CT_SLOTS = 24;      // slots per (dialog, column half): layer x its 2 head pairs
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ int clip_kv_lds(int idx16) {      // 16-byte chunk index of a slot -> byte offset of the chunk in LDS
  const int col = idx16 & 7, row = (idx16 >> 3) % CT_XROWS, head = (idx16 / (8 * CT_XROWS)) & 1, kv = idx16 / (16 * CT_XROWS);
  return (kv ? VS_OFF + head * 96 * QK_ROW : KS_OFF + head * 80 * QK_ROW) + row * QK_ROW + col * 16;
}
__device__ __forceinline__ void clip_publish_kv(const ClipArgs& a, const char* lds, int b, int seq0, int tid) {
  char* slot = a.xchg + ((long)b * CT_SLOTS + seq0) * CT_SLOT;
  unsigned lo16 = (unsigned)tid;
  asm volatile("" : "+v"(lo16));
#pragma unroll
  for (int k = 0; k < CT_SLOT / 16 / CT_TH; k++) {          // write-through (sc1) 16-byte stores, as the exchange below
    const int c = tid + k * CT_TH;
    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + clip_kv_lds(c));
    const f32x4* dstp = reinterpret_cast<const f32x4*>(slot + (long)k * CT_TH * 16) + lo16;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dstp), "v"(v) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its stores have left
  __syncthreads();
  if (tid == 0) __hip_atomic_store((gu32*)(a.flags + b), (unsigned)(seq0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool clip_fetch_kv(const ClipArgs& a, char* lds, int b, int seq0, int tid) {
  volatile int* ok = reinterpret_cast<volatile int*>(lds + PART_OFF);      // the LayerNorm partials are idle here
  if (tid == 0) {
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load((gu32*)(a.flags + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(seq0 + 1)) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) { good = 0; break; }        // seconds without the producer: give up (garbage output) rather than hang
    }
    *ok = good;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // sc1 form (see clip_exchange): every load of the slot is an sc1 load
  __syncthreads();
  if (*ok == 0) return false;
  const char* slot = a.xchg + ((long)b * CT_SLOTS + seq0) * CT_SLOT;
  unsigned lo16 = (unsigned)tid;
  asm volatile("" : "+v"(lo16));
  f32x4 v[CT_SLOT / 16 / CT_TH];
#pragma unroll
  for (int k = 0; k < CT_SLOT / 16 / CT_TH; k++) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slot + (long)k * CT_TH * 16) + lo16;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(src) : "memory");
  }
  static_assert(CT_SLOT / 16 / CT_TH == 3, "operands of the wait below");
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) :: "memory");
#pragma unroll
  for (int k = 0; k < CT_SLOT / 16 / CT_TH; k++) *reinterpret_cast<f32x4*>(lds + clip_kv_lds(tid + k * CT_TH)) = v[k];
  lds_barrier();
  return true;
}

// ---- partial-sum exchange between the two COLUMN halves of a dialog (same Guideline-16 form, symmetric) ----
// Every dialog (every row half of a 5-tile dialog) is carried by two workgroups that split the heads and the MLP's hidden units, i.e.
// the WEIGHT STREAM: a CU's ingest bounds one workgroup at ~47 us per 6.3 MB layer, so each streams half.  Twice per layer both hold
// a partial of the new residual stream over all 512 columns (half 0: x + its partial, half 1: its partial alone); each writes its
// registers to its own slot (two slots, ping-pong: the partner publishes exchange k + 1 only after it has read slot k), raises its
// flag, waits for the partner's and adds the partner's slot: both end with the same x (one commutative addition).
constexpr int CT_XSLOT = 4 * 4 * CT_TH * 16;                // up to 4 row tiles x 4 column tiles x 512 lanes x 16 B = 128 KB
template <int NT>
__device__ __forceinline__ bool clip_exchange(const ClipArgs& a, char* lds, f32x4 (&xr)[NT][4], int unit, unsigned seq, int tid) {
  static_assert(NT <= 4, "exchange slot size");
  // uniform slot bases (SGPRs) + ONE laundered 32-bit lane offset: the per-access addresses are invariant across the layer loop
  // and would otherwise be hoisted out of it and kept alive (96 address pairs: 600 spilled registers)
  char* mine = a.xslots + ((long)unit * 2 + (seq & 1)) * CT_XSLOT;
  const char* theirs = a.xslots + ((long)(unit ^ 1) * 2 + (seq & 1)) * CT_XSLOT;
  unsigned lo8 = (unsigned)tid * 2u, lo16 = (unsigned)tid;
  asm volatile("" : "+v"(lo8), "+v"(lo16));
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {                           // write-through (agent-scope, sc1) 16-byte stores: no release fence; one
      // fabric write per lane -- as 8-byte atomic stores (twice the writes) an exchange took 8 us
      const float4* dstp = reinterpret_cast<const float4*>(mine + (long)(i * 4 + j) * CT_TH * 16) + lo16;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dstp), "v"(xr[i][j]) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its stores have left
  __syncthreads();
  volatile int* ok = reinterpret_cast<volatile int*>(lds + PART_OFF);
  if (tid == 0) {
    __hip_atomic_store((gu32*)(a.xflags + unit), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load((gu32*)(a.xflags + (unit ^ 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 24)) { good = 0; break; }        // seconds without the partner: give up (garbage output) rather than hang
    }
    *ok = good;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // (no instruction: keeps the compiler from moving loads above the poll)
  __syncthreads();
  if (*ok == 0) return false;
  // Guideline 16, sc1 form: every byte of the slot was stored sc1 and drained before the flag, and EVERY load of it here is an sc1
  // load to registers (they bypass this CU's L1, which may hold the slot's lines of two exchanges ago) -- so the agent-scope acquire
  // (buffer_inv sc1, ~1.7 us) is not needed.  hipcc does not count the loads of an asm statement: the waits are explicit, and the
  // registers are operands of the wait so that no use can be scheduled in front of it.
  {
    f32x4 v[NT][4];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4* src = reinterpret_cast<const float4*>(theirs + (long)(i * 4 + j) * CT_TH * 16) + lo16;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i][j]) : "v"(src) : "memory");
      }
#pragma unroll
    for (int i = 0; i < NT; i++)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i][0]), "+v"(v[i][1]), "+v"(v[i][2]), "+v"(v[i][3]) :: "memory");
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) xr[i][j] += v[i][j];
  }
  __syncthreads();                                          // the flag word in LDS is free again
  return true;
}

// the whole tower for the 16-row tiles [I0, CT_MT) of one dialog (CT_MT = ceil(L / 16) live tiles: a shorter dialog skips the dead
// tiles' work -- one straight-line instance per tile range).  A dialog of 5 tiles is carried by TWO workgroups: rows 0 .. 47 (I0 = 0,
// CT_MT = 3, PUB: after every head pair's in_proj it publishes its K / V rows) and rows 48 .. 79 (I0 = 3, CT_MT = 5: it fetches those
// rows before its attention) -- the mask is causal, so the hand-off is one-directional and the first workgroup never waits.
template <bool F16, int I0, int CT_MT, bool PUB>
__device__ __forceinline__ void clip_tower_body(const ClipArgs& a, char* lds, const int64_t* __restrict__ tk, int L, int b, int h, int unit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int eot = L - 1;
  constexpr int NT = CT_MT - I0;                            // this workgroup's tiles: rows 16 (I0 + i) + r16
  // weight fragments in flight per wave (every phase's length is a multiple of it): the short-dialog instances have the registers
  // for a deeper ring
  constexpr int CT_RING = NT <= 3 ? 16 : 8;
  // the V images start zeroed: rows past the live tiles are read (with zero probabilities) by the 32-key steps of P V
  for (int i = tid; i < 2 * 96 * QK_ROW / 16; i += CT_TH) *reinterpret_cast<uint4*>(lds + VS_OFF + i * 16) = make_uint4(0u, 0u, 0u, 0u);
  // ---- residual stream: token + positional embedding; lane (r16, q) of wave w holds rows 16 i + r16, columns 64 w + 16 j + 4 q ..
  f32x4 xr[NT][4];
#pragma unroll
  for (int i = 0; i < NT; i++) {
    const int m = 16 * (I0 + i) + r16;
    long id = m < L ? tk[m] : 0;
    id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = 64 * wave + 16 * j + 4 * q;
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < L) {
        const float4 t = *reinterpret_cast<const float4*>(a.tok_emb + id * 512 + col);
        const float4 p = *reinterpret_cast<const float4*>(a.pos_emb + (long)m * 512 + col);
        e = make_float4(t.x + p.x, t.y + p.y, t.z + p.z, t.w + p.w);
      }
      xr[i][j] = (f32x4){e.x, e.y, e.z, e.w};
    }
  }
  // ---- weight ring: CT_RING fragments in flight; fragment f of this wave's stream is wp[f * 64] (64 lanes x 16 B, coalesced)
  // (uniform base in SGPRs + one 32-bit lane offset: the per-fragment offsets are scalar adds, not per-lane 64-bit addresses)
  const uint4* __restrict__ wp = a.wstream + (long)(h * 8 + __builtin_amdgcn_readfirstlane(wave)) * a.frags_per_wave * 64;
  const unsigned wl = (unsigned)lane;
  uint4 wq[CT_RING];
#pragma unroll
  for (int s = 0; s < CT_RING; s++) wq[s] = wp[s * 64 + wl];
  // TAKE(s): the fragment in ring slot s, and the slot's refill CT_RING fragments ahead (the stream ends with CT_RING fragments of padding)
#define CT_TAKE(dst, s) do { dst = wq[s]; wq[s] = wp[(CT_RING + (s)) * 64 + wl]; } while (0)
#define CT_STEP(f) do { if ((f) % CT_RING == CT_RING - 1) wp += CT_RING * 64; } while (0)

  float* part = reinterpret_cast<float*>(lds + PART_OFF);
  // LayerNorm of the residual stream -> XN (16-bit image)
  auto layer_norm = [&](const float* __restrict__ g, const float* __restrict__ be) {
    float4 gv[4], bv[4];                                    // requested first: their latency hides behind the statistics
#pragma unroll
    for (int j = 0; j < 4; j++) {
      gv[j] = *reinterpret_cast<const float4*>(g + 64 * wave + 16 * j + 4 * q);
      bv[j] = *reinterpret_cast<const float4*>(be + 64 * wave + 16 * j + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < NT; i++) {
      f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};                // two columns per instruction (v_pk_add_f32 / v_pk_fma_f32)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x2 lo = {xr[i][j][0], xr[i][j][1]}, hi = {xr[i][j][2], xr[i][j][3]};
        s1 += lo; s1 += hi;
        s2 = __builtin_elementwise_fma(lo, lo, s2); s2 = __builtin_elementwise_fma(hi, hi, s2);
      }
      float a1 = s1[0] + s1[1], a2 = s2[0] + s2[1];
      a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
      if (q == 0) *reinterpret_cast<float2*>(&part[((16 * (I0 + i) + r16) * 8 + wave) * 2]) = make_float2(a1, a2);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const int m = 16 * (I0 + i) + r16;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) {                         // fixed order over the 8 waves: bit-reproducible
        const float4 pp = *reinterpret_cast<const float4*>(&part[(m * 8 + 2 * k) * 2]);
        s1 += pp.x; s2 += pp.y; s1 += pp.z; s2 += pp.w;
      }
      const float mean = s1 * (1.f / 512.f);
      const float rstd = rsqrtf(fmaxf(s2 * (1.f / 512.f) - mean * mean, 0.f) + 1e-5f);
      const float nm = -mean * rstd;
#pragma unroll
      for (int j = 0; j < 4; j++) {                         // y = x (rstd g) + (b - mean rstd g)
        const f32x2 glo = {gv[j].x, gv[j].y}, ghi = {gv[j].z, gv[j].w}, blo = {bv[j].x, bv[j].y}, bhi = {bv[j].z, bv[j].w};
        const f32x2 alo = glo * rstd, ahi = ghi * rstd;
        const f32x2 clo = __builtin_elementwise_fma(glo, (f32x2){nm, nm}, blo), chi = __builtin_elementwise_fma(ghi, (f32x2){nm, nm}, bhi);
        const f32x2 ylo = __builtin_elementwise_fma((f32x2){xr[i][j][0], xr[i][j][1]}, alo, clo);
        const f32x2 yhi = __builtin_elementwise_fma((f32x2){xr[i][j][2], xr[i][j][3]}, ahi, chi);
        *reinterpret_cast<uint2*>(lds + XN_OFF + m * XN_ROW + (64 * wave + 16 * j + 4 * q) * 2) = cvt4<F16>(ylo[0], ylo[1], yhi[0], yhi[1]);
      }
    }
    lds_barrier();
  };

  CT_T0();
  for (int layer = 0; layer < a.layers; layer++) {
    const ClipLayerP& P = a.L[layer];
    CT_PH(7);
    layer_norm(P.ln1g, P.ln1b);
    CT_PH(0);
    // ================================================= attention, two heads at a time =================================================
    if (h) {                                                // half 1 carries its partial alone (the exchange adds half 0's x + partial)
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) xr[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int jp = 0; jp < 2; jp++) {
      const int hp = 2 * h + jp;                              // this column half's head pairs
      // ---- in_proj of the pair: 24 column tiles (head a: q 4 | k 4 | v 4), wave w takes tiles 3 w .. 3 w + 2; K = 512
      {
        f32x4 acc[NT][3];
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
          for (int t = 0; t < 3; t++) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the activation fragments of k-step ks + 1 are requested before the MFMAs of k-step ks (LDS latency off the critical path)
        uint4 xf[2][NT];
#pragma unroll
        for (int i = 0; i < NT; i++) xf[0][i] = lds16(lds + XN_OFF + (16 * (I0 + i) + r16) * XN_ROW + q * 16);
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
          if (ks + 1 < 16) {
#pragma unroll
            for (int i = 0; i < NT; i++) xf[(ks + 1) & 1][i] = lds16(lds + XN_OFF + (16 * (I0 + i) + r16) * XN_ROW + (4 * (ks + 1) + q) * 16);
          }
#pragma unroll
          for (int t = 0; t < 3; t++) {
            const int f = 3 * ks + t;                       // 0 .. 47: turns of the ring
            uint4 w;
            CT_TAKE(w, f % CT_RING);
            CT_STEP(f);
#pragma unroll
            for (int i = 0; i < NT; i++) acc[i][t] = cmma<F16>(w, xf[ks & 1][i], acc[i][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // bias, Q pre-scaled by 1 / sqrt(64) (exact), -> the pair's Q / K / V images
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const int gt = 3 * wave + t, ah = gt / 12, ty = (gt % 12) >> 2, sub = gt & 3;
          const int n = ty * 512 + (2 * hp + ah) * 64 + sub * 16 + 4 * q;
          const float4 bi = *reinterpret_cast<const float4*>(P.b_in + n);
          const float sc = ty == 0 ? 0.125f : 1.f;
          char* base = lds + (ty == 0 ? QO_OFF + ah * 80 * QK_ROW : ty == 1 ? KS_OFF + ah * 80 * QK_ROW : VS_OFF + ah * 96 * QK_ROW);
#pragma unroll
          for (int i = 0; i < NT; i++) {
            const int m = 16 * (I0 + i) + r16;
            const uint2 o = cvt4<F16>((acc[i][t][0] + bi.x) * sc, (acc[i][t][1] + bi.y) * sc, (acc[i][t][2] + bi.z) * sc, (acc[i][t][3] + bi.w) * sc);
            *reinterpret_cast<uint2*>(base + m * QK_ROW + (16 * sub + 4 * q) * 2) = o;
          }
        }
      }
      lds_barrier();
      if (PUB) clip_publish_kv(a, lds, 2 * b + h, layer * 2 + jp, tid);
      if (I0 > 0) { if (!clip_fetch_kv(a, lds, 2 * b + h, layer * 2 + jp, tid)) CT_GIVE_UP(); }
      CT_PH(1);
      // ---- causal attention: units (head of the pair, 16-query tile), one per wave and round (two units of a wave in one basic block
      // -- for the scheduler to interleave -- spilled at 5 tiles and ran slower)
      for (int u = wave; u < 2 * NT; u += 8) clip_attn_unit<F16, CT_MT>(lds, u / NT, I0 + u % NT, r16, q);
      lds_barrier();
      CT_PH(2);
      // ---- out_proj, the pair's 128 input columns: x += O_pair W_out[:, 128 hp ..]^T (wave w: its 64 output columns, 4 tiles)
      uint4 of[2][NT];
#pragma unroll
      for (int i = 0; i < NT; i++) of[0][i] = lds16(lds + QO_OFF + (16 * (I0 + i) + r16) * QK_ROW + q * 16);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        if (ks + 1 < 4) {
#pragma unroll
          for (int i = 0; i < NT; i++)
            of[(ks + 1) & 1][i] = lds16(lds + QO_OFF + ((ks + 1) >> 1) * 80 * QK_ROW + (16 * (I0 + i) + r16) * QK_ROW + (4 * ((ks + 1) & 1) + q) * 16);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int f = 4 * ks + t;
          uint4 w;
          CT_TAKE(w, f % CT_RING);
          CT_STEP(f);
#pragma unroll
          for (int i = 0; i < NT; i++) xr[i][t] = cmma<F16>(w, of[ks & 1][i], xr[i][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();                                          // the next pair's in_proj overwrites Q / K / V
      CT_PH(3);
    }
    if (h == 0) {
      float4 bo[4];
#pragma unroll
      for (int j = 0; j < 4; j++) bo[j] = *reinterpret_cast<const float4*>(P.b_out + 64 * wave + 16 * j + 4 * q);
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { xr[i][j][0] += bo[j].x; xr[i][j][1] += bo[j].y; xr[i][j][2] += bo[j].z; xr[i][j][3] += bo[j].w; }
    }
    if (!clip_exchange<NT>(a, lds, xr, unit, 2u * layer + 1u, tid)) CT_GIVE_UP();
    // ======================================================= MLP, 256 hidden units at a time =======================================================
    layer_norm(P.ln2g, P.ln2b);
    CT_PH(4);
    if (h) {
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) xr[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int jc = 0; jc < 4; jc++) {
      const int c = 4 * h + jc;                               // this column half's hidden-unit chunks
      {
        f32x4 acc[NT][2];
#pragma unroll
        for (int i = 0; i < NT; i++) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
        uint4 xf[2][NT];
#pragma unroll
        for (int i = 0; i < NT; i++) xf[0][i] = lds16(lds + XN_OFF + (16 * (I0 + i) + r16) * XN_ROW + q * 16);
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
          if (ks + 1 < 16) {
#pragma unroll
            for (int i = 0; i < NT; i++) xf[(ks + 1) & 1][i] = lds16(lds + XN_OFF + (16 * (I0 + i) + r16) * XN_ROW + (4 * (ks + 1) + q) * 16);
          }
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int f = 2 * ks + t;
            uint4 w;
            CT_TAKE(w, f % CT_RING);
            CT_STEP(f);
#pragma unroll
            for (int i = 0; i < NT; i++) acc[i][t] = cmma<F16>(w, xf[ks & 1][i], acc[i][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const float4 bf = *reinterpret_cast<const float4*>(P.b_fc + 256 * c + 32 * wave + 16 * t + 4 * q);
#pragma unroll
          for (int i = 0; i < NT; i++) {
            const int m = 16 * (I0 + i) + r16;
            float v[4] = {acc[i][t][0] + bf.x, acc[i][t][1] + bf.y, acc[i][t][2] + bf.z, acc[i][t][3] + bf.w};
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = v[r] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[r]));      // QuickGELU (1-ulp reciprocal)
            *reinterpret_cast<uint2*>(lds + HC_OFF + m * HC_ROW + (32 * wave + 16 * t + 4 * q) * 2) = cvt4<F16>(v[0], v[1], v[2], v[3]);
          }
        }
      }
      lds_barrier();
      CT_PH(5);
      uint4 hf[2][NT];
#pragma unroll
      for (int i = 0; i < NT; i++) hf[0][i] = lds16(lds + HC_OFF + (16 * (I0 + i) + r16) * HC_ROW + q * 16);
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int i = 0; i < NT; i++) hf[(ks + 1) & 1][i] = lds16(lds + HC_OFF + (16 * (I0 + i) + r16) * HC_ROW + (4 * (ks + 1) + q) * 16);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int f = 4 * ks + t;
          uint4 w;
          CT_TAKE(w, f % CT_RING);
          CT_STEP(f);
#pragma unroll
          for (int i = 0; i < NT; i++) xr[i][t] = cmma<F16>(w, hf[ks & 1][i], xr[i][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();                                          // the next chunk's hidden image overwrites this one
      CT_PH(6);
    }
    if (h == 0) {
      float4 bp[4];
#pragma unroll
      for (int j = 0; j < 4; j++) bp[j] = *reinterpret_cast<const float4*>(P.b_proj + 64 * wave + 16 * j + 4 * q);
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { xr[i][j][0] += bp[j].x; xr[i][j][1] += bp[j].y; xr[i][j][2] += bp[j].z; xr[i][j][3] += bp[j].w; }
    }
    if (!clip_exchange<NT>(a, lds, xr, unit, 2u * layer + 2u, tid)) CT_GIVE_UP();
  }
  // ---- the EOT row of the residual stream (ln_final and the projection follow as their own small launches)
#pragma unroll
  for (int i = 0; i < NT; i++)
    if (h == 0 && 16 * (I0 + i) + r16 == eot) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        *reinterpret_cast<float4*>(a.E + (long)b * 512 + 64 * wave + 16 * j + 4 * q) = make_float4(xr[i][j][0], xr[i][j][1], xr[i][j][2], xr[i][j][3]);
    }
  CT_DUMP();
#undef CT_TAKE
#undef CT_STEP
}

// Measured and rejected: the MLP in 512-unit chunks (four column tiles per activation fragment, half the barriers; the hidden image
// over Q | K | V) -- fully unrolled, an instance's layer body (~24 KB of code) grew to ~35 KB, two instances on the CUs that share a
// 64 KB instruction cache no longer fitted: 700 -> 2000 us; with the k-steps in groups of four (non-unrolled group loop) it fitted
// again but the ring spilled across the loop: 716 us against 698.  The unrolled bodies are sized for that cache.
// Measured and rejected: the same body on FOUR waves (one per SIMD, 512 registers each: residual stream + projection accumulators in
// the accumulator file, a 16-32 fragment ring, no spills at <= 3 tiles) -- 1060 / 1583 us at 40 / 72 tokens against 880 / 1200 on eight
// waves: a lone wave per SIMD cannot hide its own LDS / waitcnt / MFMA-issue latencies.
// Measured and rejected (tools/clip_lab.hip, 64 dialogs): eight extra "L2 warmer" workgroups (one per XCD) streaming the same bytes
// 0.75 MB ahead of the dialogs, paced by per-dialog progress words -- 925 -> 1236 us: a dialog's stream already runs at the CU's
// ingest ceiling (~43 of ~51 B/clk), not at the Infinity-Cache latency, and the progress stores cost more than the warm lines save.
template <bool F16>
__global__ __launch_bounds__(CT_TH) void clip_tower_kernel(ClipArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  // workgroup id -> (pair p, column half h): blocks of 16 ids hold 8 pairs, half 0 in ids 0 .. 7 and half 1 in ids 8 .. 15 of the block --
  // the two halves of a pair are dispatched within 16 ids of each other and share id % 8, i.e. (round-robin placement) an XCD.
  // Pairs [0, B) are the dialogs / the FIRST row halves of the 5-tile dialogs, pairs [B, 2 B) the second row halves: whatever a
  // workgroup waits for (its column partner, the first row half's K / V) has a smaller or neighbouring id and never waits for it.
  const int id = (int)blockIdx.x, h = (id >> 3) & 1, p = (id >> 4) * 8 + (id & 7);
  if (p >= 2 * a.B) return;
  const int row_half = p >= a.B ? 1 : 0, ctx = a.ctx;
  int b = p - row_half * a.B;
  if (a.row_idx) {                                          // memoised tower: only the rows whose tokens changed (uniform loads)
    if (b >= *a.count) return;
    b = a.row_idx[b];
  }
  const int unit = (b * 2 + row_half) * 2 + h;                // exchange partner: unit ^ 1
  const int64_t* __restrict__ tk = a.tokens + (long)b * ctx;
  // ---- live length: tokens up to the EOT (= first position of the largest id, as torch.argmax) -- nothing after it can reach
  // the output through the causal mask
  int L;
  {
    long best = -1; int bi = 0x7fffffff;
    for (int k = lane; k < ctx; k += 64) { const long v = tk[k]; if (v > best) { best = v; bi = k; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    L = __builtin_amdgcn_readfirstlane(bi) + 1;
  }
  const int mt = (L + 15) >> 4;
  if (row_half) {
    if (mt >= 5) clip_tower_body<F16, 3, 5, false>(a, lds, tk, L, b, h, unit);
    return;
  }
  switch (mt) {
    case 1: clip_tower_body<F16, 0, 1, false>(a, lds, tk, L, b, h, unit); break;
    case 2: clip_tower_body<F16, 0, 2, false>(a, lds, tk, L, b, h, unit); break;
    case 3: clip_tower_body<F16, 0, 3, false>(a, lds, tk, L, b, h, unit); break;
    case 4: clip_tower_body<F16, 0, 4, false>(a, lds, tk, L, b, h, unit); break;
    default: clip_tower_body<F16, 0, 3, true>(a, lds, tk, L, b, h, unit); break;
  }
}

struct ClipLayerWeights { const float* w_in[12]; const float* w_out[12]; const float* w_fc[12]; const float* w_proj[12]; };
// ---- weight stream packer: fragment f of wave w of layer l, lane (r16, q), element e = W[n][k] of the matrix / tile / k-step the
// kernel consumes at that point (see the kernel's loops)
__global__ void clip_pack_stream_kernel(ClipLayerWeights wts, uint4* __restrict__ dst, int layers, long frags_per_wave, int fmt) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;           // one thread per (layer, half, wave, fragment, lane)
  const long total = (long)layers * 16 * CT_FRAGS * 64;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  long rest = gid >> 6;
  const int fp = (int)(rest % CT_FRAGS); rest /= CT_FRAGS;
  const int wave = (int)(rest & 7), h = (int)((rest >> 3) & 1), layer = (int)(rest >> 4);
  const int r16 = lane & 15, q = lane >> 4;
  const float* W; int ld, n, k;
  if (fp < 128) {                                           // attention: this half's 2 pairs x (48 in_proj + 16 out_proj)
    const int hp = 2 * h + (fp >> 6), g = fp & 63;
    if (g < 48) {
      const int ks = g / 3, t = g % 3, gt = 3 * wave + t, ah = gt / 12, ty = (gt % 12) >> 2, sub = gt & 3;
      W = wts.w_in[layer]; ld = 512; n = ty * 512 + (2 * hp + ah) * 64 + sub * 16 + r16; k = 32 * ks + 8 * q;
    } else {
      const int ks = (g - 48) >> 2, t = (g - 48) & 3;
      W = wts.w_out[layer]; ld = 512; n = 64 * wave + 16 * t + r16; k = 128 * hp + 32 * ks + 8 * q;
    }
  } else {                                                  // MLP: this half's 4 chunks x (32 c_fc + 32 c_proj)
    const int c = 4 * h + ((fp - 128) >> 6), g = (fp - 128) & 63;
    if (g < 32) {
      const int ks = g >> 1, t = g & 1;
      W = wts.w_fc[layer]; ld = 512; n = 256 * c + 32 * wave + 16 * t + r16; k = 32 * ks + 8 * q;
    } else {
      const int ks = (g - 32) >> 2, t = (g - 32) & 3;
      W = wts.w_proj[layer]; ld = 2048; n = 64 * wave + 16 * t + r16; k = 256 * c + 32 * ks + 8 * q;
    }
  }
  const float4 v0 = *reinterpret_cast<const float4*>(W + (long)n * ld + k), v1 = *reinterpret_cast<const float4*>(W + (long)n * ld + k + 4);
  uint2 lo, hi;
  if (fmt == 1) { lo = cvt4<true>(v0.x, v0.y, v0.z, v0.w); hi = cvt4<true>(v1.x, v1.y, v1.z, v1.w); }
  else { lo = cvt4<false>(v0.x, v0.y, v0.z, v0.w); hi = cvt4<false>(v1.x, v1.y, v1.z, v1.w); }
  dst[((long)(h * 8 + wave) * frags_per_wave + (long)layer * CT_FRAGS + fp) * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

bool clip_stream_shape_ok(const avlen_clip_text* p) {
  if (!p || p->width != 512 || p->heads != 8 || p->layers < 1 || p->layers > 12 || p->ctx > CT_ROWS) return false;
  for (int l = 0; l < p->layers; l++) {
    const avlen_clip_block& b = p->block[l];
    if (b.attn.in_proj.out_f != 1536 || b.attn.in_proj.in_f != 512 || b.attn.out_proj.out_f != 512 || b.attn.out_proj.in_f != 512 ||
        b.fc.out_f != 2048 || b.fc.in_f != 512 || b.proj.out_f != 512 || b.proj.in_f != 2048) return false;
    if (!b.attn.in_proj.b || !b.attn.out_proj.b || !b.fc.b || !b.proj.b) return false;
  }
  return true;
}
inline long clip_frags_per_wave(int layers) { return (long)layers * CT_FRAGS + CT_PAD; }

}  // namespace

extern "C" size_t avlen_clip_stream_bytes(const avlen_clip_text* p) {
  return clip_stream_shape_ok(p) ? (size_t)16 * clip_frags_per_wave(p->layers) * 1024 : 0;
}

// Builds the per-wave weight stream of the one-launch tower from the fp32 weights (fmt 0: bf16, 1: fp16).  Derived data: call
// again whenever the weights change (CLIP is frozen in the reference: once).
extern "C" int avlen_clip_pack_stream(const avlen_clip_text* p, void* dst, int fmt, hipStream_t st) {
  if (!clip_stream_shape_ok(p) || !dst) return AVLEN_ERR_ARG;
  ClipLayerWeights w = {};
  for (int l = 0; l < p->layers; l++) {
    w.w_in[l] = p->block[l].attn.in_proj.w; w.w_out[l] = p->block[l].attn.out_proj.w;
    w.w_fc[l] = p->block[l].fc.w; w.w_proj[l] = p->block[l].proj.w;
  }
  const long fpw = clip_frags_per_wave(p->layers);
  if (avlen_zero_bytes(dst, (size_t)16 * fpw * 1024, st) != AVLEN_OK) return AVLEN_ERR_LAUNCH;      // the 16 padding fragments per wave
  const long total = (long)p->layers * 16 * CT_FRAGS * 64;
  hipLaunchKernelGGL(clip_pack_stream_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (uint4*)dst, p->layers, fpw, fmt);
  return avlen_launch_status();
}

// X rows of the residual stream at each dialog's EOT token (B x 512 fp32) through the 12 blocks in one launch
// flag block (16 KB: zeroed before every launch) | K / V slots (dialog, column half) | exchange slots (4 workgroups per dialog x 2)
size_t avlen_clip_tower_stream_ws_bytes(int B) { return 16384 + (size_t)B * 2 * CT_SLOTS * CT_SLOT + (size_t)B * 4 * 2 * CT_XSLOT; }

int avlen_clip_tower_stream_fwd(const avlen_clip_text* p, const int64_t* tokens, float* E, int B, int f16, void* ws, size_t ws_bytes,
                                hipStream_t st, const int* row_idx, const int* count) {
  if (!clip_stream_shape_ok(p) || !p->wstream || B <= 0 || B > 1024 || !ws || ws_bytes < avlen_clip_tower_stream_ws_bytes(B)) return AVLEN_ERR_ARG;
  if (B > 512) return AVLEN_ERR_ARG;
  if (avlen_zero_bytes(ws, 16384, st) != AVLEN_OK) return AVLEN_ERR_LAUNCH;      // the flag words (own block at the workspace's start)
  ClipArgs a = {};
  a.tokens = tokens; a.tok_emb = p->tok_emb; a.pos_emb = p->pos_emb; a.wstream = (const uint4*)p->wstream; a.E = E;
  a.ctx = p->ctx; a.vocab = p->vocab; a.layers = p->layers; a.frags_per_wave = clip_frags_per_wave(p->layers);
  a.flags = (unsigned*)ws; a.xflags = a.flags + 2 * B; a.xchg = (char*)ws + 16384; a.B = B;
  a.xslots = a.xchg + (size_t)B * 2 * CT_SLOTS * CT_SLOT;
  a.row_idx = row_idx; a.count = count;
  const int grid = ((2 * B + 7) / 8) * 16;
  for (int l = 0; l < p->layers; l++) {
    const avlen_clip_block& b = p->block[l];
    a.L[l] = ClipLayerP{b.ln1.g, b.ln1.b, b.ln2.g, b.ln2.b, b.attn.in_proj.b, b.attn.out_proj.b, b.fc.b, b.proj.b};
  }
  static unsigned long long done0 = 0, done1 = 0;
  if (f16) {
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&clip_tower_kernel<true>), CT_LDS, &done1) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(clip_tower_kernel<true>, dim3(grid), dim3(CT_TH), CT_LDS, st, a);
  } else {
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&clip_tower_kernel<false>), CT_LDS, &done0) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(clip_tower_kernel<false>, dim3(grid), dim3(CT_TH), CT_LDS, st, a);
  }
  return avlen_launch_status();
}
