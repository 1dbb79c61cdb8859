// CLIP text tower (third party: open_clip / clip.model.Transformer, 12 x ResidualAttentionBlock, width 512, 8 heads, QuickGELU;
// call site ss_baselines/savi/ppo/policy.py:847-849) as ONE launch, SEQUENCE-STATIONARY: a set of workgroups (512 threads each)
// carries a GROUP of up to four 16-row tiles -- whole dialogs packed by a device-side work list (clip_group_kernel) -- through
// all 12 layers.
//
// Why: as a chain of launches (71 kernels) the tower is latency-bound -- 12-20 us per GEMM whatever the row count, 0.9 ms per step.
// Here a group's residual stream never leaves its CUs:
//   * residual x (64 x 512 fp32) lives in REGISTERS in MFMA accumulator layout (wave w owns columns [64 w, 64 w + 64));
//     the out_proj / c_proj products accumulate straight into it,
//   * LayerNorm output, per-head-pair Q / K / V, the attention output and the MLP hidden chunk are fp16 (bf16) images in LDS,
//   * the weights (6.3 MB per layer) are STREAMED: every wave owns a private, fragment-ordered stream (packed once by
//     avlen_clip_pack_stream in exactly the order the wave consumes it) and keeps CT_RING fragments (1 KiB each) in flight through
//     a register ring -- no LDS staging, no descriptor set-up, no barrier on the weight path,
//   * a CU ingests ~43 of ~51 B/clk, which is THE bound: a group's stream is therefore split over FOUR workgroups (one head pair and
//     two MLP hidden chunks each; 2-way kept behind a knob), which twice per layer combine their partial residuals by a
//     reduce-scatter + all-gather through global memory (clip_exchange4), and a 5-tile dialog additionally splits its ROWS over
//     two such sets (one-directional K / V hand-off: the mask is causal).
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
  // work list built on the device by clip_group_kernel: groups[g], g < ngroups[0]; the launch whose column split matches
  // (ngroups * 4 <= max_wg4: the 4-way launch, else the 2-way one) runs, the other one's workgroups exit at once
  const struct ClipGroup* groups; const int* ngroups; int max_wg4;
  const uint4* wstream4; long frags_per_wave4;              // the 4-way column split's weight streams (32 of them)
  char* xslots2;                                            // 4-way exchange: finished column tiles (2 x 32 KB per workgroup)
};
// A workgroup set carries up to four 16-row tiles.  kind 0: whole dialogs PACKED into the tiles (dialog d[i], its tile lt[i], its
// live length L[i] for tile i; nt tiles) -- the attention of a tile sees the keys of its own dialog only; kind 1 / 2: the first
// (tiles 0 .. 2, publishes K / V) / second (tiles 3 .. 4, fetches them) row half of a 5-tile dialog.
struct ClipGroup { int nt, kind, d[4], lt[4], L[4], pad[2]; };
static_assert(sizeof(ClipGroup) == 64, "group table stride");
// A hand-off that never arrives (partner workgroup lost): the dialog's output row is poisoned with NaN -- a stale or garbage
// embedding would flow into the rollout unnoticed (the GRU sequence kernels do the same, train_gru.hip)
#define CT_GIVE_UP() do { if (threadIdx.x < 128) for (int i_ = 0; i_ < NT; i_++) \
    *reinterpret_cast<float4*>(a.E + (long)td[i_] * 512 + 4 * threadIdx.x) = \
    make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")); return; } while (0)
#ifdef AVLEN_CT_PROF
// lab builds: phase k's cycles are ADDED to prof[workgroup][k] as the phase ends (the lab zeroes the table and divides by its launch
// count).  Accumulators kept in registers for the whole body (the first version) cost the 4-way instance 1300 spilled registers and
// half its speed -- the profile no longer described the product's kernel.
#define CT_T0() long long ct_t = __builtin_amdgcn_s_memtime()
#define CT_PH(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); \
    if (a.prof && threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.prof) + (long)blockIdx.x * 8 + (k), (unsigned long long)(n_ - ct_t)); \
    ct_t = n_; } while (0)
#define CT_DUMP() do { } while (0)
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
__device__ __forceinline__ void clip_attn_unit(char* lds, int ah, int mt, int r16, int q, int kstart) {
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
      const int kidx = 16 * nt + 4 * q + r;               // causal, and (packed dialogs) only the query's own dialog: rows >= kstart
      const float v = (kidx <= qi && kidx >= kstart) ? sacc[nt][r] : -INFINITY;
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
// bslot: flag word of the (dialog, column part); slot: index of the 24 KB slot (dialog, column part, layer, head pair)
__device__ __forceinline__ void clip_publish_kv(const ClipArgs& a, const char* lds, int bslot, long slot_i, int seq0, int tid) {
  const int b = bslot;
  char* slot = a.xchg + slot_i * CT_SLOT;
  unsigned lo16 = (unsigned)tid;
  asm volatile("" : "+v"(lo16));
#pragma unroll
  for (int k = 0; k < CT_SLOT / 16 / CT_TH; k++) {          // write-through (sc1) 16-byte stores, as the exchange below
    const int c = tid + k * CT_TH;
    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + clip_kv_lds(c));
    const f32x4* dstp = reinterpret_cast<const f32x4*>(slot + (long)k * CT_TH * 16) + lo16;
    // the s_nop: a VALU write of a > 64-bit store's data registers needs one wait state after the store, and the compiler's hazard
    // recognizer does not look inside an asm statement (a lab build under register pressure reused the registers in the very next
    // instruction: run-to-run different exchange data; tools/asm_wait_check.py now walks the ISA for this)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(dstp), "v"(v) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its stores have left
  __syncthreads();
  if (tid == 0) __hip_atomic_store((gu32*)(a.flags + b), (unsigned)(seq0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool clip_fetch_kv(const ClipArgs& a, char* lds, int b, long slot_i, int seq0, int tid) {
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
  const char* slot = a.xchg + slot_i * CT_SLOT;
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
template <int NT, int SPLIT>
__device__ __forceinline__ bool clip_exchange_pair(const ClipArgs& a, char* lds, f32x4 (&xr)[NT][4], int unit, unsigned seq, int tid) {
  static_assert(NT <= 4, "exchange slot size");
  const int ubase = unit & ~(SPLIT - 1), me = unit & (SPLIT - 1);      // the SPLIT column parts of a group have consecutive units
  // uniform slot bases (SGPRs) + ONE laundered 32-bit lane offset: the per-access addresses are invariant across the layer loop
  // and would otherwise be hoisted out of it and kept alive (96 address pairs: 600 spilled registers)
  char* mine = a.xslots + ((long)unit * 2 + (seq & 1)) * CT_XSLOT;
  unsigned lo8 = (unsigned)tid * 2u, lo16 = (unsigned)tid;
  asm volatile("" : "+v"(lo8), "+v"(lo16));
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {                           // write-through (agent-scope, sc1) 16-byte stores: no release fence; one
      // fabric write per lane -- as 8-byte atomic stores (twice the writes) an exchange took 8 us
      const float4* dstp = reinterpret_cast<const float4*>(mine + (long)(i * 4 + j) * CT_TH * 16) + lo16;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(dstp), "v"(xr[i][j]) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its stores have left
  __syncthreads();
  volatile int* ok = reinterpret_cast<volatile int*>(lds + PART_OFF);
  if (tid == 0) {
    __hip_atomic_store((gu32*)(a.xflags + unit), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
#pragma unroll
    for (int k = 1; k < SPLIT; k++) {
      const int other = ubase + ((me + k) & (SPLIT - 1));
      while (good && __hip_atomic_load((gu32*)(a.xflags + other), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 24)) good = 0;                 // seconds without a partner: give up (NaN output) rather than hang
      }
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
#pragma unroll
  for (int k = 1; k < SPLIT; k++) {                          // the partners in a fixed order per workgroup: run-to-run bit-reproducible
    const char* theirs = a.xslots + ((long)(ubase + ((me + k) & (SPLIT - 1))) * 2 + (seq & 1)) * CT_XSLOT;
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


// ---- the same among FOUR column parts: reduce-scatter + all-gather ----
// A part would have to read three whole partials (3 x NT x 32 KB) one after the other -- measured 11 us per exchange.  Instead part k
// OWNS column tile j == k of every wave: (1) everybody writes its whole partial and raises its flag (sequence 2 s - 1), (2) the owner
// reads the three partners' copies of ITS column tile -- 3 NT loads per lane, all in flight -- adds them in a fixed order and writes
// the finished tile to its second slot, flag 2 s, (3) everybody reads the three finished tiles it does not own.  Two round trips with
// every load of a round in flight at once, and all four parts end with bit-identical residual rows (each column is summed once).
constexpr int CT_XSLOT2 = 4 * CT_TH * 16;                   // finished column tile: up to 4 row tiles x 512 lanes x 16 B = 32 KB
// Measured and rejected (tools/clip_lab.hip, 64 dialogs, groups on one XCD): the exchange at a narrower scope.  Workgroup-scope (sc0)
// loads hit the CU's own L1 and never see the partner's flag (every hand-off timed out); plain (write-back) stores with sc1 loads
// gave the right bits and 580-597 us against 595-601: the exchange is not bound by the stores' write-through.
template <int NT>
__device__ __forceinline__ bool clip_exchange4(const ClipArgs& a, char* lds, f32x4 (&xr)[NT][4], int unit, unsigned seq, int tid) {
  static_assert(NT <= 4, "exchange slot size");
  const int ubase = unit & ~3, me = unit & 3;
  char* mine = a.xslots + ((long)unit * 2 + (seq & 1)) * CT_XSLOT;
  char* mine2 = a.xslots2 + ((long)unit * 2 + (seq & 1)) * CT_XSLOT2;
  unsigned lo16 = (unsigned)tid;
  asm volatile("" : "+v"(lo16));
  volatile int* ok = reinterpret_cast<volatile int*>(lds + PART_OFF);
  auto wait_all = [&](unsigned want) {                       // lanes 0 .. 2 of wave 0 poll one partner each
    if (tid < 3) {
      const int other = ubase + ((me + 1 + tid) & 3);
      unsigned spins = 0;
      int good = 1;
      while (__hip_atomic_load((gu32*)(a.xflags + other), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) { good = 0; break; }
      }
      if (!good) *ok = 0;
    }
  };
  // ---- round 1: whole partials out
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j == me) continue;                                 // (uniform) the tile this part owns stays in its registers
      const float4* dstp = reinterpret_cast<const float4*>(mine + (long)(i * 4 + j) * CT_TH * 16) + lo16;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(dstp), "v"(xr[i][j]) : "memory");
    }
  if (tid == 0) *ok = 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store((gu32*)(a.xflags + unit), 2u * seq - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  wait_all(2u * seq - 1u);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __syncthreads();
  if (*ok == 0) return false;
  {
    f32x4 v[3][NT];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int other = ubase + ((me + 1 + k) & 3);
      const char* theirs = a.xslots + ((long)other * 2 + (seq & 1)) * CT_XSLOT;
#pragma unroll
      for (int i = 0; i < NT; i++) {
        // xr[i][me] lives at tile index i * 4 + me; `me` is uniform but not a compile-time constant: the offset is scalar arithmetic
        const float4* src = reinterpret_cast<const float4*>(theirs + (long)(i * 4 + me) * CT_TH * 16) + lo16;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k][i]) : "v"(src) : "memory");
      }
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0][i]), "+v"(v[1][i]), "+v"(v[2][i]) :: "memory");
    // own tile + partners me+1, me+2, me+3: a fixed order for this owner; nobody else sums this tile
#pragma unroll
    for (int i = 0; i < NT; i++) {
      f32x4 own;
      own = me == 0 ? xr[i][0] : me == 1 ? xr[i][1] : me == 2 ? xr[i][2] : xr[i][3];
      own += v[0][i]; own += v[1][i]; own += v[2][i];
      if (me == 0) xr[i][0] = own; else if (me == 1) xr[i][1] = own; else if (me == 2) xr[i][2] = own; else xr[i][3] = own;
      const float4* dstp = reinterpret_cast<const float4*>(mine2 + (long)i * CT_TH * 16) + lo16;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(dstp), "v"(own) : "memory");
    }
  }
  // ---- round 2: finished tiles
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store((gu32*)(a.xflags + unit), 2u * seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  wait_all(2u * seq);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __syncthreads();
  if (*ok == 0) return false;
  {
    f32x4 v[3][NT];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int other = ubase + ((me + 1 + k) & 3);
      const char* theirs2 = a.xslots2 + ((long)other * 2 + (seq & 1)) * CT_XSLOT2;
#pragma unroll
      for (int i = 0; i < NT; i++) {
        const float4* src = reinterpret_cast<const float4*>(theirs2 + (long)i * CT_TH * 16) + lo16;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k][i]) : "v"(src) : "memory");
      }
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0][i]), "+v"(v[1][i]), "+v"(v[2][i]) :: "memory");
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int oj = (me + 1 + k) & 3;                       // the column tile partner k owns
#pragma unroll
      for (int i = 0; i < NT; i++) {
        if (oj == 0) xr[i][0] = v[k][i]; else if (oj == 1) xr[i][1] = v[k][i]; else if (oj == 2) xr[i][2] = v[k][i]; else xr[i][3] = v[k][i];
      }
    }
  }
  __syncthreads();
  return true;
}
template <int NT, int SPLIT>
__device__ __forceinline__ bool clip_exchange(const ClipArgs& a, char* lds, f32x4 (&xr)[NT][4], int unit, unsigned seq, int tid) {
  if constexpr (SPLIT == 4) return clip_exchange4<NT>(a, lds, xr, unit, seq, tid);
  else return clip_exchange_pair<NT, SPLIT>(a, lds, xr, unit, seq, tid);
}

// the whole tower for the 16-row tiles [I0, CT_MT) of one dialog (CT_MT = ceil(L / 16) live tiles: a shorter dialog skips the dead
// tiles' work -- one straight-line instance per tile range).  A dialog of 5 tiles is carried by TWO workgroups: rows 0 .. 47 (I0 = 0,
// CT_MT = 3, PUB: after every head pair's in_proj it publishes its K / V rows) and rows 48 .. 79 (I0 = 3, CT_MT = 5: it fetches those
// rows before its attention) -- the mask is causal, so the hand-off is one-directional and the first workgroup never waits.
template <bool F16, int I0, int CT_MT, bool PUB, int SPLIT>
__device__ __forceinline__ void clip_tower_body(const ClipArgs& a, char* lds, const ClipGroup& grp, int h, int unit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  constexpr int NT = CT_MT - I0;                            // this workgroup's tiles: rows 16 (I0 + i) + r16
  // tile i: dialog td[i], token positions tp[i] .. tp[i] + 15 of it, its live length tL[i]; kspack: per tile, the first tile of its
  // dialog inside this workgroup's row space (the attention's key range starts there)
  int td[NT], tp[NT], tL[NT];
  unsigned kspack = 0;
#pragma unroll
  for (int i = 0; i < NT; i++) {
    td[i] = grp.d[i]; tp[i] = 16 * grp.lt[i]; tL[i] = grp.L[i];
    kspack |= (unsigned)((I0 + i) - grp.lt[i]) << (8 * i);
  }
  const int b = td[0];                                      // row halves of a 5-tile dialog: the dialog of the K / V hand-off
  // weight fragments in flight per wave (every phase's length is a multiple of it): the short-dialog instances have the registers
  // for a deeper ring
  constexpr int CT_RING = NT <= 3 ? 16 : 8;
  // the V images start zeroed: rows past the live tiles are read (with zero probabilities) by the 32-key steps of P V
  for (int i = tid; i < 2 * 96 * QK_ROW / 16; i += CT_TH) *reinterpret_cast<uint4*>(lds + VS_OFF + i * 16) = make_uint4(0u, 0u, 0u, 0u);
  // ---- residual stream: token + positional embedding; lane (r16, q) of wave w holds rows 16 i + r16, columns 64 w + 16 j + 4 q ..
  f32x4 xr[NT][4];
#pragma unroll
  for (int i = 0; i < NT; i++) {
    const int m = tp[i] + r16;                              // token position inside the tile's dialog
    const bool live = m < tL[i];
    long id = live ? a.tokens[(long)td[i] * a.ctx + m] : 0;
    id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = 64 * wave + 16 * j + 4 * q;
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        const float4 t = *reinterpret_cast<const float4*>(a.tok_emb + id * 512 + col);
        const float4 p = *reinterpret_cast<const float4*>(a.pos_emb + (long)m * 512 + col);
        e = make_float4(t.x + p.x, t.y + p.y, t.z + p.z, t.w + p.w);
      }
      xr[i][j] = (f32x4){e.x, e.y, e.z, e.w};
    }
  }
  // ---- weight ring: CT_RING fragments in flight; fragment f of this wave's stream is wp[f * 64] (64 lanes x 16 B, coalesced)
  // (uniform base in SGPRs + one 32-bit lane offset: the per-fragment offsets are scalar adds, not per-lane 64-bit addresses)
  const uint4* __restrict__ wp = (SPLIT == 4 ? a.wstream4 : a.wstream) +
                                 (long)(h * 8 + __builtin_amdgcn_readfirstlane(wave)) * (SPLIT == 4 ? a.frags_per_wave4 : a.frags_per_wave) * 64;
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
    for (int jp = 0; jp < 4 / SPLIT; jp++) {
      const int hp = (4 / SPLIT) * h + jp;                    // this column part's head pairs
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
      // K / V slots: (dialog, column part) x (layer, head pair of the part) -- SPLIT * (4 / SPLIT) * layers = 48 per dialog either way
      const int kv_seq = layer * (4 / SPLIT) + jp;
      const long kv_slot = ((long)(SPLIT * b + h) * (CT_SLOTS * 2 / SPLIT)) + kv_seq;
      if (PUB) clip_publish_kv(a, lds, SPLIT * b + h, kv_slot, kv_seq, tid);
      if (I0 > 0) { if (!clip_fetch_kv(a, lds, SPLIT * b + h, kv_slot, kv_seq, tid)) CT_GIVE_UP(); }
      CT_PH(1);
      // ---- causal attention: units (head of the pair, 16-query tile), one per wave and round (two units of a wave in one basic block
      // -- for the scheduler to interleave -- spilled at 5 tiles and ran slower)
      for (int u = wave; u < 2 * NT; u += 8)
        clip_attn_unit<F16, CT_MT>(lds, u / NT, I0 + u % NT, r16, q, 16 * (int)((kspack >> (8 * (u % NT))) & 0xffu));
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
    if (!clip_exchange<NT, SPLIT>(a, lds, xr, unit, 2u * layer + 1u, tid)) CT_GIVE_UP();
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
    for (int jc = 0; jc < 8 / SPLIT; jc++) {
      const int c = (8 / SPLIT) * h + jc;                     // this column part's hidden-unit chunks
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
    if (!clip_exchange<NT, SPLIT>(a, lds, xr, unit, 2u * layer + 2u, tid)) CT_GIVE_UP();
  }
  // ---- the EOT row of the residual stream (ln_final and the projection follow as their own small launches)
#pragma unroll
  for (int i = 0; i < NT; i++)
    if (h == 0 && tp[i] + r16 == tL[i] - 1) {                 // the EOT row of the tile's dialog
#pragma unroll
      for (int j = 0; j < 4; j++)
        *reinterpret_cast<float4*>(a.E + (long)td[i] * 512 + 64 * wave + 16 * j + 4 * q) = make_float4(xr[i][j][0], xr[i][j][1], xr[i][j][2], xr[i][j][3]);
    }
  CT_DUMP();
#undef CT_TAKE
#undef CT_STEP
}

// ---- work list (ONE launch in front of the tower): [memo: which rows changed] -> live lengths -> tile classes -> groups -> flags.
// One block of 16 waves.  (A) sixteen lanes per row: the EOT scan (first position of the largest id, as torch.argmax: nothing after it can
// reach the output through the causal mask) and, for the memoised tower (avlen_clip_text_cached_fwd), the comparison with the
// previous call's tokens, which are updated in place (row B of that buffer is the all-zero dialog).  (B) per tile class, the rows in
// row order: ranks by wave ballots + a prefix over the 64-row chunks.  (C) groups in CLOSED FORM, one thread per group -- whole
// dialogs packed into <= 4 tiles: [4] | [3 + 1] | [2 + 2], and a last odd [2 (+ 1 + 1)] | [1 + 1 + 1 + 1] | the first row halves of
// the 5-tile dialogs, then their second halves (whatever a workgroup waits for has a smaller id).  Multi-tile dialogs always start at
// an even tile of their group: their P V products pair key tiles exactly as they do alone.  (D) the flag words of the launch are
// zeroed here (no separate fill).  Until round 4 these were three launches (memo scan, fill, a one-thread packing loop: ~50 us on the
// step's critical path); this one takes ~10.
constexpr int CT_MAXB = 512;
using ClipMemo = avlen_clip_memo;                          // internal.h: {tokens_new, prev, hdr [valid, tower rows, zero rows], zidx}
__global__ __launch_bounds__(1024) void clip_worklist_kernel(const int64_t* __restrict__ tokens, ClipMemo memo, int B, int ctx,
                                                              ClipGroup* __restrict__ groups, int* __restrict__ ngroups,
                                                              unsigned* __restrict__ flags, int flag_words) {
  __shared__ int Ls[CT_MAXB], cls[CT_MAXB], lst[6][CT_MAXB], chunk_cnt[8][6], base[8][6], cnt[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool cached = memo.prev != nullptr;
  const int valid = cached ? memo.hdr[0] : 1;
  for (int i = tid; i < flag_words; i += 1024) flags[i] = 0u;
  // ---- (A) sixteen lanes per row, 64 rows per pass: all of a row's loads are in flight together (a wave per row took five
  // dependent passes of ~2 us for the benched 65 rows)
  const int sub = lane >> 4, l16 = lane & 15;
  for (int r0 = 0; r0 < B; r0 += 64) {
    const int r = r0 + 4 * wave + sub;
    long best = -1; int bi = 0x7fffffff;
    bool diff = false, nz = false;
    if (r < B) {
      for (int k = l16; k < ctx; k += 16) {
        long v;
        if (cached) {
          v = r < B - 1 ? memo.tokens_new[(long)r * ctx + k] : 0;            // the memo's buffer has one more row: all zero
          const long o = memo.prev[(long)r * ctx + k];
          diff = diff || v != o; nz = nz || v != 0;
          memo.prev[(long)r * ctx + k] = v;
        } else v = tokens[(long)r * ctx + k];
        if (v > best) { best = v; bi = k; }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const long ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const unsigned long long bd = __ballot(diff), bz = __ballot(nz);
    const bool any_diff = ((bd >> (16 * sub)) & 0xffffull) != 0, any_nz = ((bz >> (16 * sub)) & 0xffffull) != 0;
    if (l16 == 0 && r < B) {
      const int L = bi + 1;
      int t = (L + 15) >> 4; t = t > 5 ? 5 : t;
      int c = t;                                                             // 1 .. 5: tiles
      if (cached) c = !(any_diff || !valid) ? 0 : ((!any_nz && r < B - 1) ? 6 : t);   // 0: unchanged, 6: copies the shared zero embedding
      Ls[r] = L; cls[r] = c;
    }
  }
  __syncthreads();
  // ---- (B) rows per class in row order (classes 1 .. 5 -> lst[0 .. 4], class 6 -> lst[5])
  if (wave < 8) {
    const int r = 64 * wave + lane;
    const int c = r < B ? cls[r] : 0;
#pragma unroll
    for (int k = 1; k <= 6; k++) {
      const unsigned long long m = __ballot(c == k);
      if (lane == 0) chunk_cnt[wave][k - 1] = __popcll(m);
    }
  }
  __syncthreads();
  if (tid < 6) {
    int run = 0;
    for (int w = 0; w < 8; w++) { base[w][tid] = run; run += chunk_cnt[w][tid]; }
    cnt[tid] = run;
  }
  __syncthreads();
  if (wave < 8) {
    const int r = 64 * wave + lane;
    const int c = r < B ? cls[r] : 0;
#pragma unroll
    for (int k = 1; k <= 6; k++) {
      const unsigned long long m = __ballot(c == k);
      if (c == k) lst[k - 1][base[wave][k - 1] + __popcll(m & ((1ull << lane) - 1ull))] = r;
    }
  }
  __syncthreads();
  // ---- (C)
  const int n1 = cnt[0], n2 = cnt[1], n3 = cnt[2], n4 = cnt[3], n5 = cnt[4], nzr = cnt[5];
  const int u3 = n3 < n1 ? n3 : n1;                                          // 1-tile dialogs that ride with a 3-tile one
  const int G2 = (n2 + 1) >> 1;
  const int left = n1 - u3;
  const int u2 = (n2 & 1) ? (left < 2 ? left : 2) : 0;                       // ... with the odd last 2-tile one
  const int r1 = left - u2, G1 = (r1 + 3) >> 2;
  const int o3 = n4, o2 = o3 + n3, o1 = o2 + G2, o5a = o1 + G1, o5b = o5a + n5, ng = o5b + n5;
  for (int g = tid; g < ng; g += 1024) {
    int gd[4] = {0, 0, 0, 0}, glt[4] = {0, 0, 0, 0}, gL[4] = {0, 0, 0, 0}, t0 = 0, kind = 0;
    auto put = [&](int r) {
      const int L = Ls[r], t = (L + 15) >> 4;
      for (int k = 0; k < t && t0 < 4; k++) { gd[t0] = r; glt[t0] = k; gL[t0] = L; t0++; }
    };
    if (g < o3) put(lst[3][g]);
    else if (g < o2) { const int i = g - o3; put(lst[2][i]); if (i < u3) put(lst[0][i]); }
    else if (g < o1) {
      const int i = g - o2; put(lst[1][2 * i]);
      if (2 * i + 1 < n2) put(lst[1][2 * i + 1]);
      else for (int k = 0; k < u2; k++) put(lst[0][u3 + k]);
    } else if (g < o5a) {
      const int i = g - o1;
      for (int k = 0; k < 4 && 4 * i + k < r1; k++) put(lst[0][u3 + u2 + 4 * i + k]);
    } else {
      const int half = g >= o5b, r = lst[4][g - (half ? o5b : o5a)], L = Ls[r];
      kind = 1 + half; t0 = half ? 2 : 3;
      for (int k = 0; k < t0; k++) { gd[k] = r; glt[k] = (half ? 3 : 0) + k; gL[k] = L; }
    }
    int4* dst = reinterpret_cast<int4*>(groups + g);
    dst[0] = make_int4(t0, kind, gd[0], gd[1]);
    dst[1] = make_int4(gd[2], gd[3], glt[0], glt[1]);
    dst[2] = make_int4(glt[2], glt[3], gL[0], gL[1]);
    dst[3] = make_int4(gL[2], gL[3], 0, 0);
  }
  if (tid == 0) {
    ngroups[0] = ng;
    if (cached) { memo.hdr[1] = n1 + n2 + n3 + n4 + n5; memo.hdr[2] = nzr; memo.hdr[0] = 1; }
  }
  if (cached) for (int i = tid; i < nzr; i += 1024) memo.zidx[i] = lst[5][i];
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
template <bool F16, int SPLIT>
__global__ __launch_bounds__(CT_TH) void clip_tower_kernel(ClipArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // workgroup id -> (group g, column part h): blocks of 8 SPLIT ids hold 8 groups, part k in ids 8 k .. 8 k + 7 of the block -- the
  // parts of a group are dispatched within 8 SPLIT ids of each other and share id % 8, i.e. (round-robin placement) an XCD.
  const int id = (int)blockIdx.x, h = (id >> 3) & (SPLIT - 1), g = (id / (8 * SPLIT)) * 8 + (id & 7);
  const int ng = a.ngroups[0];
  // Which split serves the call is a property of the BUILD / knob, never of the batch: a dialog's embedding must not depend on how
  // many other dialogs share the call (the two splits round the fp16 operands of later layers at different points: 2e-3 apart) --
  // the memoised tower recomputes one changed row and must reproduce what the full batch gave.  Default: always 4-way (max_wg4 < 0).
  // Grids beyond the chip's residency are fine: the parts of a group lie within 32 consecutive ids and ids are dispatched in order,
  // so the resident set always holds complete groups that make progress.
  const bool use4 = a.max_wg4 < 0 || ng * 4 <= a.max_wg4;
  if ((SPLIT == 4) != use4 || g >= ng) return;
  const ClipGroup grp = a.groups[g];
  const int unit = g * SPLIT + h;
  if (grp.kind == 2) { clip_tower_body<F16, 3, 5, false, SPLIT>(a, lds, grp, h, unit); return; }
  if (grp.kind == 1) { clip_tower_body<F16, 0, 3, true, SPLIT>(a, lds, grp, h, unit); return; }
  switch (grp.nt) {
    case 1: clip_tower_body<F16, 0, 1, false, SPLIT>(a, lds, grp, h, unit); break;
    case 2: clip_tower_body<F16, 0, 2, false, SPLIT>(a, lds, grp, h, unit); break;
    case 3: clip_tower_body<F16, 0, 3, false, SPLIT>(a, lds, grp, h, unit); break;
    default: clip_tower_body<F16, 0, 4, false, SPLIT>(a, lds, grp, h, unit); break;
  }
}

struct ClipLayerWeights { const float* w_in[12]; const float* w_out[12]; const float* w_fc[12]; const float* w_proj[12]; };
// ---- weight stream packer: fragment f of wave w of layer l, lane (r16, q), element e = W[n][k] of the matrix / tile / k-step the
// kernel consumes at that point (see the kernel's loops)
__global__ void clip_pack_stream_kernel(ClipLayerWeights wts, uint4* __restrict__ dst, int layers, long frags_per_wave, int fmt, int split) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;           // one thread per (layer, column part, wave, fragment, lane)
  const int FR = 768 / split, ATT = 64 * (4 / split);                      // fragments per wave, layer and part; its attention share
  const long total = (long)layers * split * 8 * FR * 64;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  long rest = gid >> 6;
  const int fp = (int)(rest % FR); rest /= FR;
  const int wave = (int)(rest & 7), h = (int)((rest >> 3) % split), layer = (int)((rest >> 3) / split);
  const int r16 = lane & 15, q = lane >> 4;
  const float* W; int ld, n, k;
  if (fp < ATT) {                                           // attention: this part's head pairs x (48 in_proj + 16 out_proj)
    const int hp = (4 / split) * h + (fp >> 6), g = fp & 63;
    if (g < 48) {
      const int ks = g / 3, t = g % 3, gt = 3 * wave + t, ah = gt / 12, ty = (gt % 12) >> 2, sub = gt & 3;
      W = wts.w_in[layer]; ld = 512; n = ty * 512 + (2 * hp + ah) * 64 + sub * 16 + r16; k = 32 * ks + 8 * q;
    } else {
      const int ks = (g - 48) >> 2, t = (g - 48) & 3;
      W = wts.w_out[layer]; ld = 512; n = 64 * wave + 16 * t + r16; k = 128 * hp + 32 * ks + 8 * q;
    }
  } else {                                                  // MLP: this part's hidden chunks x (32 c_fc + 32 c_proj)
    const int c = (8 / split) * h + ((fp - ATT) >> 6), g = (fp - ATT) & 63;
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
  dst[((long)(h * 8 + wave) * frags_per_wave + (long)layer * FR + fp) * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
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
inline long clip_frags_per_wave(int layers, int split) { return (long)layers * (768 / split) + CT_PAD; }
inline size_t clip_stream_bytes_split(int layers, int split) { return (size_t)split * 8 * clip_frags_per_wave(layers, split) * 1024; }
constexpr size_t CT_FLAG_BYTES = 32768;                     // K / V flags (4 B words) + exchange flags (8 B words) of <= 512 dialogs

}  // namespace

// both column splits' streams, one behind the other: [2-way: 16 streams][4-way: 32 streams]
extern "C" size_t avlen_clip_stream_bytes(const avlen_clip_text* p) {
  return clip_stream_shape_ok(p) ? clip_stream_bytes_split(p->layers, 2) + clip_stream_bytes_split(p->layers, 4) : 0;
}

// Builds the per-wave weight streams of the one-launch tower from the fp32 weights (fmt 0: bf16, 1: fp16).  Derived data: call
// again whenever the weights change (CLIP is frozen in the reference: once).
extern "C" int avlen_clip_pack_stream(const avlen_clip_text* p, void* dst, int fmt, hipStream_t st) {
  if (!clip_stream_shape_ok(p) || !dst) return AVLEN_ERR_ARG;
  ClipLayerWeights w = {};
  for (int l = 0; l < p->layers; l++) {
    w.w_in[l] = p->block[l].attn.in_proj.w; w.w_out[l] = p->block[l].attn.out_proj.w;
    w.w_fc[l] = p->block[l].fc.w; w.w_proj[l] = p->block[l].proj.w;
  }
  if (avlen_zero_bytes(dst, avlen_clip_stream_bytes(p), st) != AVLEN_OK) return AVLEN_ERR_LAUNCH;      // the padding fragments per wave
  char* at = (char*)dst;
  for (int split = 2; split <= 4; split += 2) {
    const long fpw = clip_frags_per_wave(p->layers, split);
    const long total = (long)p->layers * split * 8 * (768 / split) * 64;
    hipLaunchKernelGGL(clip_pack_stream_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (uint4*)at, p->layers, fpw, fmt, split);
    at += clip_stream_bytes_split(p->layers, split);
  }
  return avlen_launch_status();
}

// Knob: -1 (default) = every call on the 4-way column split; 0 = every call on the 2-way split; n > 0 (lab) = 4-way when a call's
// groups x 4 workgroups <= n, else 2-way -- then a dialog's embedding depends (2e-3) on the size of the call it is part of.
static int g_split4_wgs = -1;
#ifdef AVLEN_CT_PROF
static long long* g_ct_prof = nullptr;                      // lab builds (tools/clip_lab.hip): per-workgroup phase totals [grid][8]
#endif
extern "C" void avlen_set_clip_tower_split4_wgs(int n) { g_split4_wgs = n; }

// X rows of the residual stream at each dialog's EOT token (B x 512 fp32) through the 12 blocks in one launch
// flag block (zeroed before every launch) | group table + count | K / V slots (dialog, column part) | exchange slots (2 per unit)
static inline int clip_max_units(int B) { const int u = 8 * B; return u > 256 ? u : 256; }    // <= 2 B groups x 4 column parts
size_t avlen_clip_tower_stream_ws_bytes(int B) {
  return CT_FLAG_BYTES + align_up((size_t)(2 * B + 1) * sizeof(ClipGroup) + 256, 256) + (size_t)B * 2 * CT_SLOTS * CT_SLOT +
         (size_t)clip_max_units(B) * 2 * (CT_XSLOT + CT_XSLOT2);
}

// memo (optional; avlen_clip_text_cached_fwd): {new tokens (B - 1 rows), previous tokens (B rows, updated in place; the tower reads
// THEM: pass them as `tokens`), header, zero-row list}: only rows that differ from the previous call run the tower
int avlen_clip_tower_stream_fwd(const avlen_clip_text* p, const int64_t* tokens, float* E, int B, int f16, void* ws, size_t ws_bytes,
                                hipStream_t st, const avlen_clip_memo* memo) {
  if (!clip_stream_shape_ok(p) || !p->wstream || B <= 0 || B > CT_MAXB || !ws || ws_bytes < avlen_clip_tower_stream_ws_bytes(B)) return AVLEN_ERR_ARG;
  ClipArgs a = {};
  a.tokens = tokens; a.tok_emb = p->tok_emb; a.pos_emb = p->pos_emb; a.wstream = (const uint4*)p->wstream; a.E = E;
  a.ctx = p->ctx; a.vocab = p->vocab; a.layers = p->layers; a.frags_per_wave = clip_frags_per_wave(p->layers, 2);
  a.wstream4 = (const uint4*)((const char*)p->wstream + clip_stream_bytes_split(p->layers, 2));
  a.frags_per_wave4 = clip_frags_per_wave(p->layers, 4);
  a.B = B;
#ifdef AVLEN_CT_PROF
  a.prof = g_ct_prof;
#endif
  // flags: K / V hand-off words [4 B] (indexed SPLIT * dialog + part), then the exchange words [8 B max] (indexed by unit)
  a.flags = (unsigned*)ws; a.xflags = a.flags + 4 * B;
  char* at = (char*)ws + CT_FLAG_BYTES;
  ClipGroup* groups = (ClipGroup*)at;
  int* ngroups = (int*)(at + (size_t)(2 * B + 1) * sizeof(ClipGroup));
  at += align_up((size_t)(2 * B + 1) * sizeof(ClipGroup) + 256, 256);
  a.groups = groups; a.ngroups = ngroups;
  a.xchg = at; a.xslots = a.xchg + (size_t)B * 2 * CT_SLOTS * CT_SLOT;
  a.xslots2 = a.xslots + (size_t)clip_max_units(B) * 2 * CT_XSLOT;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  a.max_wg4 = g_split4_wgs;                                // < 0: always the 4-way split (default); 0: always 2-way; n: lab (mixed)
  (void)n_cu;
  ClipMemo mm = {};
  if (memo) mm = *memo;
  hipLaunchKernelGGL(clip_worklist_kernel, dim3(1), dim3(1024), 0, st, tokens, mm, B, p->ctx, groups, ngroups, (unsigned*)ws,
                     (int)(CT_FLAG_BYTES / 4));
  for (int l = 0; l < p->layers; l++) {
    const avlen_clip_block& b = p->block[l];
    a.L[l] = ClipLayerP{b.ln1.g, b.ln1.b, b.ln2.g, b.ln2.b, b.attn.in_proj.b, b.attn.out_proj.b, b.fc.b, b.proj.b};
  }
  // the launch(es) the knob allows go out; with a mixed knob the work list decides on the device which one runs (see the kernel)
  const int ng4 = a.max_wg4 < 0 ? 2 * B : (a.max_wg4 / 4 < 2 * B ? a.max_wg4 / 4 : 2 * B);      // groups the 4-way launch may carry
  const int grid4 = ((ng4 + 7) / 8) * 32, grid2 = a.max_wg4 < 0 ? 0 : ((2 * B + 7) / 8) * 16;
  static unsigned long long d20 = 0, d21 = 0, d40 = 0, d41 = 0;
  if (f16) {
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&clip_tower_kernel<true, 4>), CT_LDS, &d41) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&clip_tower_kernel<true, 2>), CT_LDS, &d21) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    if (grid4 > 0) hipLaunchKernelGGL((clip_tower_kernel<true, 4>), dim3(grid4), dim3(CT_TH), CT_LDS, st, a);
    if (grid2 > 0) hipLaunchKernelGGL((clip_tower_kernel<true, 2>), dim3(grid2), dim3(CT_TH), CT_LDS, st, a);
  } else {
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&clip_tower_kernel<false, 4>), CT_LDS, &d40) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&clip_tower_kernel<false, 2>), CT_LDS, &d20) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    if (grid4 > 0) hipLaunchKernelGGL((clip_tower_kernel<false, 4>), dim3(grid4), dim3(CT_TH), CT_LDS, st, a);
    if (grid2 > 0) hipLaunchKernelGGL((clip_tower_kernel<false, 2>), dim3(grid2), dim3(CT_TH), CT_LDS, st, a);
  }
  return avlen_launch_status();
}
