// Training path of the GRU baseline policy (BASELINE configs[1]): forward with saved activations and backward of
//   AudioCNN / VisualCNN (audio_cnn.py:62-94,136-151; visual_cnn.py:82-107,165-190): conv+ReLU, conv+ReLU, conv, flatten, Linear+ReLU
//   RNNStateEncoder.seq_forward (av_nav/models/rnn_state_encoder.py:92-143): hidden * mask before every step, 1-layer GRU
//   AudioNavBaselineNet.forward (savi/ppo/policy.py:451-477): x = [audio | visual | category]
// driven by avlen_amd/av_nav.py:PPO.update, which mirrors ss_baselines/av_nav/ppo/ppo.py:60-151 (evaluate_actions ->
// clipped-surrogate / clipped-value / entropy loss -> backward -> clip-norm -> Adam).  The heads + loss backward is the same
// kernel pi_q uses (avlen_ppo_loss_heads_bwd, rl.hip) and runs between the two entry points below.
//
// Convolution backward: the weight gradient is dY^T * im2col(X) and the data gradient col2im(dY * W) -- both products run on the
// GEMM kernels of the training path (fp32-staged MFMA, or bf16 glds MFMA from 16384 rows on in bf16 mode); im2col writes the
// [kh][kw][c] K-order of the packed weights (a (kw, c) run is contiguous in NHWC, so rows are copied in 16-byte pieces) and
// col2im is a gather (no atomics, deterministic) that also applies the ReLU mask of the layer below.
// Gradients are produced in the PACKED layouts the forward uses ([O][KH][KW][I]; fc columns in NHWC flatten order) and
// re-laid into the canonical parameter layouts (OIHW; (O, C*H*W)) of the flat gradient buffer Adam steps over.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"
#include <math.h>

#define TRY(x) do { int _rc = (x); if (_rc != AVLEN_OK) return _rc; } while (0)
#include "train_kernels.h"

namespace {

constexpr size_t GEMM_SCRATCH = 96u << 20;
bool gru_seq_on() {             // AVLEN_GRU_SEQ=0 (lab builds): one launch per GRU step instead of the resident sequence kernels
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_GRU_SEQ", 1);
  return v != 0;
}
bool conv_dw_direct_on() {      // AVLEN_CONV_DW_DIRECT=0 (lab builds): the GEMM route for the conv weight gradients (A/B knob)
  static int v = -1;
  if (v < 0) v = (int)avlen_knob("AVLEN_CONV_DW_DIRECT", 1);
  return v != 0;
}
inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

struct Dims { int h[4], w[4], c[4]; };        // [0] = input, [i+1] = output of conv i
Dims cnn_dims(const avlen_cnn3* n, int H, int W) {
  Dims d; d.h[0] = H; d.w[0] = W; d.c[0] = n->conv[0].cin;
  for (int i = 0; i < 3; i++) {
    d.h[i + 1] = (d.h[i] - n->conv[i].kh) / n->conv[i].stride + 1;
    d.w[i + 1] = (d.w[i] - n->conv[i].kw) / n->conv[i].stride + 1;
    d.c[i + 1] = n->conv[i].cout;
  }
  return d;
}

// ---------------------------------------------------------------------------------------------------------------
// GRU (r | z | n gate order of nn.GRU): h' = (1 - z) n + z hm,  n = tanh(gi_n + r * gh_n),  hm = h_prev * mask
// One launch per time step.  The recurrence is N (<= 16 here) rows against W_hh (3H x H): a tile GEMM gives it a dozen
// workgroups and ~50 us; here ONE WAVE owns hidden unit j: it keeps rows j, H+j, 2H+j of W_hh in registers (3 x H/64 floats per
// lane, read once, coalesced), forms the three dot products with every batch row by wave reductions and applies the gate
// arithmetic in place -- H waves spread over the chip, W_hh streamed exactly once per step.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GRU_HP = 8;                    // floats per lane of one weight row: H <= 64 * GRU_HP
// Batch rows are processed MC at a time: all MC x 3 dot products are accumulated together (MC x GRU_HP independent loads in
// flight per lane), reduced across the wave, and then lane m finishes row m -- its gi loads, the gate arithmetic and the stores
// of the MC rows run in parallel lanes instead of as a serial tail in lane 0 (which made a step 29 us: ~8 dependent round trips).
template <int MC>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                           const float* __restrict__ gi, const float* __restrict__ hprev,
                                                           const float* __restrict__ mask, float* __restrict__ out,
                                                           float* __restrict__ hm_save, float* __restrict__ gh_save, int N, int H) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= H) return;
  float w[3][GRU_HP];
#pragma unroll
  for (int g = 0; g < 3; g++)
#pragma unroll
    for (int i = 0; i < GRU_HP; i++) { const int k = lane + 64 * i; w[g][i] = k < H ? w_hh[((long)g * H + j) * H + k] : 0.f; }
  const float b0 = b_hh[j], b1 = b_hh[H + j], b2 = b_hh[2 * H + j];
  for (int m0 = 0; m0 < N; m0 += MC) {
    float a0[MC], a1[MC], a2[MC];
#pragma unroll
    for (int mm = 0; mm < MC; mm++) {
      const int m = m0 + mm;
      a0[mm] = a1[mm] = a2[mm] = 0.f;
      if (m < N) {
#pragma unroll
        for (int i = 0; i < GRU_HP; i++) {
          const int k = lane + 64 * i;
          const float h = k < H ? hprev[(long)m * H + k] : 0.f;
          a0[mm] += h * w[0][i]; a1[mm] += h * w[1][i]; a2[mm] += h * w[2][i];
        }
      }
    }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int mm = 0; mm < MC; mm++) {
      const float s0 = wave_sum(a0[mm]), s1 = wave_sum(a1[mm]), s2 = wave_sum(a2[mm]);
      if (lane == mm) { r0 = s0; r1 = s1; r2 = s2; }
    }
    const int m = m0 + lane;
    if (lane < MC && m < N) {
      const float mk = mask[m];                       // the dot products are linear in the mask
      const float hj = hprev[(long)m * H + j] * mk;
      r0 = r0 * mk + b0; r1 = r1 * mk + b1; r2 = r2 * mk + b2;
      const float* a = gi + (long)m * 3 * H;
      const float r = 1.f / (1.f + expf(-(a[j] + r0)));
      const float z = 1.f / (1.f + expf(-(a[H + j] + r1)));
      const float nn = tanhf(a[2 * H + j] + r * r2);
      out[(long)m * H + j] = (1.f - z) * nn + z * hj;
      if (hm_save) {
        hm_save[(long)m * H + j] = hj;
        float* g = gh_save + (long)m * 3 * H;
        g[j] = r0; g[H + j] = r1; g[2 * H + j] = r2;
      }
    }
  }
}

// Backward of step t for hidden unit j (one wave), fused with the carry from step t+1:
//   dhm_{t+1}[m][j] = dz_{t+1}[m][j] + sum_n dGH_{t+1}[m][n] * W_hh[n][j]      (W_hh^T rows are contiguous in whhT: [H][3H])
//   dh = d_out[t][m][j] + dhm_{t+1}[m][j] * mask_{t+1}[m]
//   -> dGI[t], dGH[t] (rows m, columns j, H+j, 2H+j), dz_t = dh * z  (kept for the next launch)
// `dgh_next` null: last step (no carry).  Same MC-rows-at-a-time structure as the forward.
constexpr int GRU_KP = 24;                   // 3H <= 64 * GRU_KP
template <int MC>
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ whhT, const float* __restrict__ gi,
                                                           const float* __restrict__ gh, const float* __restrict__ hm,
                                                           const float* __restrict__ d_out, const float* __restrict__ dgh_next,
                                                           const float* __restrict__ dz_next, const float* __restrict__ mask_next,
                                                           float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dz,
                                                           int N, int H) {
  extern __shared__ float sd[];               // MC rows of dGH_{t+1}: shared by the block's four hidden units, staged once
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool valid = j < H;
  const int jj = valid ? j : H - 1;
  const int K = 3 * H;
  float w[GRU_KP];
  if (dgh_next) {
#pragma unroll
    for (int i = 0; i < GRU_KP; i++) { const int k = lane + 64 * i; w[i] = k < K ? whhT[(long)jj * K + min(k, K - 1)] : 0.f; }
  }
  for (int m0 = 0; m0 < N; m0 += MC) {
    float carry = 0.f;
    if (dgh_next) {
      const int rows = min(MC, N - m0);
      __syncthreads();
      {
        const float4* src = (const float4*)(dgh_next + (long)m0 * K);
        const int n4 = rows * K / 4;            // K = 3H, H % 4 == 0 (checked by the launcher)
        for (int i = threadIdx.x; i < n4; i += 256) ((float4*)sd)[i] = src[i];
      }
      __syncthreads();
      float acc[MC];
#pragma unroll
      for (int mm = 0; mm < MC; mm++) {
        acc[mm] = 0.f;
        if (mm < rows) {
#pragma unroll
          for (int i = 0; i < GRU_KP; i++) { const int k = lane + 64 * i; if (k < K) acc[mm] += sd[mm * K + k] * w[i]; }
        }
      }
#pragma unroll
      for (int mm = 0; mm < MC; mm++) { const float sv = wave_sum(acc[mm]); if (lane == mm) carry = sv; }
    }
    const int m = m0 + lane;
    if (valid && lane < MC && m < N) {
      if (dgh_next) carry = (carry + dz_next[(long)m * H + j]) * mask_next[m];
      const long o = (long)m * K;
      const float* a = gi + o; const float* b = gh + o;
      const float r = 1.f / (1.f + expf(-(a[j] + b[j])));
      const float z = 1.f / (1.f + expf(-(a[H + j] + b[H + j])));
      const float ghn = b[2 * H + j];
      const float nn = tanhf(a[2 * H + j] + r * ghn);
      const float dh = d_out[(long)m * H + j] + carry;
      const float dn = dh * (1.f - z);
      const float dzz = dh * (hm[(long)m * H + j] - nn);
      const float dpn = dn * (1.f - nn * nn);
      const float dpr = dpn * ghn * r * (1.f - r);
      const float dpz = dzz * z * (1.f - z);
      dgi[o + j] = dpr; dgi[o + H + j] = dpz; dgi[o + 2 * H + j] = dpn;
      dgh[o + j] = dpr; dgh[o + H + j] = dpz; dgh[o + 2 * H + j] = dpn * r;
      dz[(long)m * H + j] = dh * z;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole sequence in ONE launch (T steps of the minibatch's N <= 16 rows).  A per-step launch is bounded by what a step has to
// do before its first product: stream W_hh (3 MB) out of L2 again and start 128 workgroups -- 9.2 us forward, 14.5 us backward
// against ~1 us of arithmetic, 4,800 launches per update.  Here H/8 workgroups of 8 waves stay resident for all T steps: wave =
// hidden unit j with its three W_hh rows (forward) / its W_hh^T row (backward) in registers for the whole sequence, and the only
// per-step traffic is the exchange of the step's result between the workgroups:
//   * every value a step hands to the next one (h_t forward, dGH_t backward) is stored with an agent-scope store (write-through)
//     into a buffer the host pre-filled with a SENTINEL word (0xffffffff: a NaN pattern no arithmetic produces);
//   * the next step's workgroups stage that buffer into LDS with agent-scope loads and simply repeat the load of a piece until
//     none of its words is the sentinel -- the data is its own flag: one store and one load per hand-off, no barrier counter, no
//     fence, and a 32-bit word is either the sentinel or final;
//   * everything else (gi, masks, saved hm / gh, dGI) is plain traffic written before the launch or read after it.
// The per-element arithmetic and summation order are those of the per-step kernels above.
// A workgroup never waits for a LATER step of another workgroup, the grid (H/8 <= 64 workgroups) is co-resident, and the wait is
// (a payload word that equals the sentinel -- the one quiet-NaN pattern 0xffffffff, which no arithmetic here produces from finite
// inputs; NaN INPUTS propagate as the canonical 0x7fc00000 -- would read as "not ready" and end in the time-out, i.e. in NaN output too)
// bounded: after ~seconds without progress a workgroup raises *err and runs on -- the launch then poisons its result with NaN
// (visible in the losses) instead of hanging the device.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr unsigned GRU_SENT = 0xffffffffu;
#ifdef AVLEN_SEQ_PROF             // lab builds (tools/gru_seq_lab.hip): per-phase wall-clock totals of every workgroup
__device__ long long* g_seq_prof = nullptr;
#define SEQ_STAMP(k) do { if (g_seq_prof && threadIdx.x == 0) { const long long n_ = wall_clock64(); g_seq_prof[blockIdx.x * 8 + (k)] += n_ - seq_t_; seq_t_ = n_; } } while (0)
#define SEQ_STAMP_INIT long long seq_t_ = wall_clock64()
#else
#define SEQ_STAMP(k) do { } while (0)
#define SEQ_STAMP_INIT do { } while (0)
#endif
constexpr int SEQ_TH = 512;
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store((gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// stage `nw` words of `src` (a hand-off buffer) into LDS; KW words per thread at most.  One load instruction of a wave covers 256
// contiguous bytes: agent-scope loads are not cached, every 128-byte line a load touches is a request to the fabric (16-byte pieces
// per lane, fetched as four dword loads, asked for every line four times: 5.8 us for the backward's 48 KB).
template <int KW>
__device__ __forceinline__ void stage_polled(const float* src, float* dst, int nw, int tid, unsigned* err) {
  unsigned w[KW];
  // a launch whose hand-off already timed out once is lost (its result is poisoned below): the remaining steps read whatever is
  // there instead of spinning through the full bound again, T times over.  err == null: T = 1, nothing is ever waited for.
  unsigned spins = (err && __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? (1u << 21) : 0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < KW; k++) {                       // branch-free (words past the end repeat the last one): all loads in flight
      const int idx = tid + SEQ_TH * k < nw ? tid + SEQ_TH * k : nw - 1;
      w[k] = __hip_atomic_load((gu32*)src + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool again = false;
#pragma unroll
    for (int k = 0; k < KW; k++) again |= w[k] == GRU_SENT;
    if (!again) break;
    if (++spins > (1u << 21)) { if (err) *err = 1u; break; }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int k = 0; k < KW; k++) {
    const int idx = tid + SEQ_TH * k;
    if (idx < nw) dst[idx] = __uint_as_float(w[k]);
  }
}

// NV per-lane partial sums -> their 64-lane totals, value q in lane q (and lanes q + NVP, ...).  24 butterflies of 6 dependent
// ds_bpermute steps cost a step 3 us (profiles/r03_gru_seq_phases.txt); here the wave transposes through its own LDS patch instead:
// lane l writes row l ([lane][value], odd row stride: conflict-free), reader lane (part, q) adds the partials of 64 / LPV source lanes
// for value q (consecutive q: conflict-free), log2(LPV) shuffles join the parts.
template <int NV> struct RedGeom {
  static constexpr int NVP = NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
  static constexpr int LPV = 64 / NVP, PER = 64 / LPV, ROW = NV + 1;
  static constexpr int FLOATS = 64 * ROW;
};
template <int NV>
__device__ __forceinline__ float wave_reduce_many(const float (&val)[NV], float* red, int lane) {
  using G = RedGeom<NV>;
#pragma unroll
  for (int q = 0; q < NV; q++) red[lane * G::ROW + q] = val[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int q = lane % G::NVP, part = lane / G::NVP;
  const float* p = red + (part * G::PER) * G::ROW + (q < NV ? q : 0);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < G::PER; i += 2) { s0 += p[i * G::ROW]; s1 += p[(i + 1) * G::ROW]; }
  float s = s0 + s1;
#pragma unroll
  for (int o = G::LPV / 2; o > 0; o >>= 1) s += __shfl_xor(s, o * G::NVP, 64);
  __builtin_amdgcn_wave_barrier();               // the patch is rewritten by the wave's next call
  return s;
}

template <int MC>
__global__ __launch_bounds__(SEQ_TH) void gru_seq_fwd_kernel(const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                             const float* __restrict__ gi_all, const float* __restrict__ h0,
                                                             const float* __restrict__ masks, float* out, float* __restrict__ hm_save,
                                                             float* __restrict__ gh_save, int T, int N, int H, unsigned* err) {
  extern __shared__ __attribute__((aligned(16))) float sh[];      // N x H: the previous hidden state, then one reduction patch per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = blockIdx.x * 8 + (tid >> 6);                      // H % 8 == 0 (launcher)
  float* const red = sh + MC * 64 * GRU_HP + (tid >> 6) * RedGeom<3 * MC>::FLOATS;
  float w[3][GRU_HP];                                             // lane's columns: 4 * lane + 256 * (i / 4) + i % 4 (16-byte LDS reads)
#pragma unroll
  for (int g = 0; g < 3; g++)
#pragma unroll
    for (int i = 0; i < GRU_HP; i++) { const int k = 4 * lane + 256 * (i >> 2) + (i & 3); w[g][i] = k < H ? w_hh[((long)g * H + j) * H + k] : 0.f; }
  const float b0 = b_hh[j], b1 = b_hh[H + j], b2 = b_hh[2 * H + j];
  const int n4 = N * H / 4;
  constexpr int KW = MC * GRU_HP * 64 / SEQ_TH;
  // what a step needs besides the hand-off (its gi row, its mask) does not depend on the recurrence: lane m fetches step t + 1's
  // values while step t runs (loop-carried registers), so no memory round trip sits between the hand-off and the gate arithmetic
  const int ml = lane < N ? lane : 0;
  float p_mk = masks[ml], p_g0 = gi_all[(long)ml * 3 * H + j], p_g1 = gi_all[(long)ml * 3 * H + H + j], p_g2 = gi_all[(long)ml * 3 * H + 2 * H + j];
  SEQ_STAMP_INIT;
  for (int t = 0; t < T; t++) {
    if (t == 0) {
      for (int idx = tid; idx < n4; idx += SEQ_TH) reinterpret_cast<float4*>(sh)[idx] = reinterpret_cast<const float4*>(h0)[idx];
    } else {
      stage_polled<KW>(out + (size_t)(t - 1) * N * H, sh, n4 * 4, tid, err);
    }
    __syncthreads();
    SEQ_STAMP(0);
    float acc[3 * MC];                                              // [gate][row]
#pragma unroll
    for (int mm = 0; mm < MC; mm++) {                              // branch-free (every LDS read in flight): rows >= N repeat row 0 and are
      acc[mm] = acc[MC + mm] = acc[2 * MC + mm] = 0.f;             // dropped, columns >= H read column 0 against zero weights
      const float* hrow = sh + (mm < N ? mm : 0) * H;
#pragma unroll
      for (int i4 = 0; i4 < GRU_HP / 4; i4++) {
        const int k = 4 * lane + 256 * i4;                         // H % 4 == 0: a 16-byte piece is inside the row or past it
        const float4 hv = *reinterpret_cast<const float4*>(hrow + (k < H ? k : 0));
        const float h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int i = i4 * 4 + e;
          acc[mm] = fmaf(h4[e], w[0][i], acc[mm]); acc[MC + mm] = fmaf(h4[e], w[1][i], acc[MC + mm]);
          acc[2 * MC + mm] = fmaf(h4[e], w[2][i], acc[2 * MC + mm]);
        }
      }
    }
    const float tot = wave_reduce_many<3 * MC>(acc, red, lane);    // lane q = gate * MC + row
    float r0 = tot, r1 = __shfl(tot, MC + lane, 64), r2 = __shfl(tot, 2 * MC + lane, 64);
    SEQ_STAMP(1);
    if (lane < N) {
      const int m = lane;
      const long row = (long)t * N + m;
      const float mk = p_mk;
      const float hj = sh[m * H + j] * mk;
      r0 = r0 * mk + b0; r1 = r1 * mk + b1; r2 = r2 * mk + b2;
      const float r = 1.f / (1.f + expf(-(p_g0 + r0)));
      const float z = 1.f / (1.f + expf(-(p_g1 + r1)));
      const float nn = tanhf(p_g2 + r * r2);
      st_agent(out + row * H + j, (1.f - z) * nn + z * hj);
      if (hm_save) {                                     // training forward: what the backward reads
        hm_save[row * H + j] = hj;
        float* g = gh_save + row * 3 * H;
        g[j] = r0; g[H + j] = r1; g[2 * H + j] = r2;
      }
    }
    if (t + 1 < T) {
      const long nrow = (long)(t + 1) * N + ml;
      p_mk = masks[nrow]; p_g0 = gi_all[nrow * 3 * H + j]; p_g1 = gi_all[nrow * 3 * H + H + j]; p_g2 = gi_all[nrow * 3 * H + 2 * H + j];
    }
    SEQ_STAMP(2);
    __syncthreads();                                               // sh is rewritten by the next step
    SEQ_STAMP(3);
  }
  if (lane == 0 && err && __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) out[((size_t)(T - 1) * N) * H + j] = __builtin_nanf("");
}

// BPTT, steps T-1 .. 0 (see gru_step_bwd_kernel): the carry of dz stays in lane m's register, dGH_t is the hand-off
template <int MC>
__global__ __launch_bounds__(SEQ_TH) void gru_seq_bwd_kernel(const float* __restrict__ whhT, const float* __restrict__ gi,
                                                             const float* __restrict__ gh, const float* __restrict__ hm,
                                                             const float* __restrict__ d_out, const float* __restrict__ masks,
                                                             float* __restrict__ dgi, float* dgh, int T, int N, int H, unsigned* err) {
  extern __shared__ __attribute__((aligned(16))) float sd[];      // N x 3H: dGH of step t + 1, then one reduction patch per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = blockIdx.x * 8 + (tid >> 6);
  float* const red = sd + MC * 64 * GRU_KP + (tid >> 6) * RedGeom<MC>::FLOATS;
  const int K = 3 * H;
  float w[GRU_KP];                                                // lane's columns: 4 * lane + 256 * (i / 4) + i % 4
#pragma unroll
  for (int i = 0; i < GRU_KP; i++) { const int k = 4 * lane + 256 * (i >> 2) + (i & 3); w[i] = k < K ? whhT[(long)j * K + k] : 0.f; }
  constexpr int KW = MC * GRU_KP * 64 / SEQ_TH;
  float dz_keep = 0.f;
  // lane m's operands of step t (gi, gh, hm, d_out rows, the mask of step t + 1) are fetched one step ahead (see the forward)
  const int ml = lane < N ? lane : 0;
  float p_a0, p_a1, p_a2, p_b0, p_b1, p_b2, p_hm, p_do, p_mk;
  auto prefetch = [&](int t) {
    const long row = (long)t * N + ml, o = row * K;
    p_a0 = gi[o + j]; p_a1 = gi[o + H + j]; p_a2 = gi[o + 2 * H + j];
    p_b0 = gh[o + j]; p_b1 = gh[o + H + j]; p_b2 = gh[o + 2 * H + j];
    p_hm = hm[row * H + j]; p_do = d_out[row * H + j];
    p_mk = t < T - 1 ? masks[row + N] : 0.f;
  };
  prefetch(T - 1);
  SEQ_STAMP_INIT;
  for (int t = T - 1; t >= 0; t--) {
    float carry = 0.f;
    if (t < T - 1) {
      stage_polled<KW>(dgh + (size_t)(t + 1) * N * K, sd, N * K, tid, err);
      __syncthreads();
      SEQ_STAMP(0);
      float acc[MC];
#pragma unroll
      for (int mm = 0; mm < MC; mm++) {                            // branch-free, as in the forward
        acc[mm] = 0.f;
        const float* drow = sd + (mm < N ? mm : 0) * K;
#pragma unroll
        for (int i4 = 0; i4 < GRU_KP / 4; i4++) {
          const int k = 4 * lane + 256 * i4;
          const float4 v = *reinterpret_cast<const float4*>(drow + (k < K ? k : 0));
          acc[mm] = fmaf(v.x, w[i4 * 4], acc[mm]); acc[mm] = fmaf(v.y, w[i4 * 4 + 1], acc[mm]);
          acc[mm] = fmaf(v.z, w[i4 * 4 + 2], acc[mm]); acc[mm] = fmaf(v.w, w[i4 * 4 + 3], acc[mm]);
        }
      }
      carry = wave_reduce_many<MC>(acc, red, lane);                // lane m: row m
    }
    SEQ_STAMP(1);
    if (lane < N) {
      const int m = lane;
      const long row = (long)t * N + m;
      if (t < T - 1) carry = (carry + dz_keep) * p_mk;
      const long o = row * K;
      const float r = 1.f / (1.f + expf(-(p_a0 + p_b0)));
      const float z = 1.f / (1.f + expf(-(p_a1 + p_b1)));
      const float ghn = p_b2;
      const float nn = tanhf(p_a2 + r * ghn);
      const float dh = p_do + carry;
      const float dn = dh * (1.f - z);
      const float dzz = dh * (p_hm - nn);
      const float dpn = dn * (1.f - nn * nn);
      const float dpr = dpn * ghn * r * (1.f - r);
      const float dpz = dzz * z * (1.f - z);
      dgi[o + j] = dpr; dgi[o + H + j] = dpz; dgi[o + 2 * H + j] = dpn;
      st_agent(dgh + o + j, dpr); st_agent(dgh + o + H + j, dpz); st_agent(dgh + o + 2 * H + j, dpn * r);
      dz_keep = dh * z;
    }
    if (t > 0) prefetch(t - 1);
    SEQ_STAMP(2);
    __syncthreads();                                               // sd is rewritten by the next step
    SEQ_STAMP(3);
  }
  if (lane == 0 && err && __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dgi[j] = __builtin_nanf("");
}
// launchers: LDS = the staged hand-off (sized for MC rows of the widest H) + 8 reduction patches
template <int MC>
int launch_gru_seq_fwd_t(const float* w_hh, const float* b_hh, const float* gi, const float* h0, const float* masks, float* out, float* hm,
                         float* gh, int T, int N, int H, unsigned* err, hipStream_t st) {
  static unsigned long long done = 0;
  const int lds = (MC * 64 * GRU_HP + 8 * RedGeom<3 * MC>::FLOATS) * (int)sizeof(float);
  TRY(avlen_set_dyn_lds(reinterpret_cast<const void*>(&gru_seq_fwd_kernel<MC>), lds, &done));
  hipLaunchKernelGGL(gru_seq_fwd_kernel<MC>, dim3(H / 8), dim3(SEQ_TH), lds, st, w_hh, b_hh, gi, h0, masks, out, hm, gh, T, N, H, err);
  return avlen_launch_status();
}
int launch_gru_seq_fwd(const float* w_hh, const float* b_hh, const float* gi, const float* h0, const float* masks, float* out, float* hm,
                       float* gh, int T, int N, int H, unsigned* err, hipStream_t st) {
  return N <= 8 ? launch_gru_seq_fwd_t<8>(w_hh, b_hh, gi, h0, masks, out, hm, gh, T, N, H, err, st)
                : launch_gru_seq_fwd_t<16>(w_hh, b_hh, gi, h0, masks, out, hm, gh, T, N, H, err, st);
}
template <int MC>
int launch_gru_seq_bwd_t(const float* whhT, const float* gi, const float* gh, const float* hm, const float* d_out, const float* masks,
                         float* dgi, float* dgh, int T, int N, int H, unsigned* err, hipStream_t st) {
  static unsigned long long done = 0;
  const int lds = (MC * 64 * GRU_KP + 8 * RedGeom<MC>::FLOATS) * (int)sizeof(float);
  TRY(avlen_set_dyn_lds(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<MC>), lds, &done));
  hipLaunchKernelGGL(gru_seq_bwd_kernel<MC>, dim3(H / 8), dim3(SEQ_TH), lds, st, whhT, gi, gh, hm, d_out, masks, dgi, dgh, T, N, H, err);
  return avlen_launch_status();
}
int launch_gru_seq_bwd(const float* whhT, const float* gi, const float* gh, const float* hm, const float* d_out, const float* masks,
                       float* dgi, float* dgh, int T, int N, int H, unsigned* err, hipStream_t st) {
  return N <= 8 ? launch_gru_seq_bwd_t<8>(whhT, gi, gh, hm, d_out, masks, dgi, dgh, T, N, H, err, st)
                : launch_gru_seq_bwd_t<16>(whhT, gi, gh, hm, d_out, masks, dgi, dgh, T, N, H, err, st);
}
bool gru_seq_ok(int N, int H) { return N >= 1 && N <= 16 && H % 8 == 0 && H <= 64 * GRU_HP && gru_seq_on(); }

// [R][C] -> [C][R]
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float t[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const int r = r0 + ty + jj * 8, c = c0 + tx;
    t[ty + jj * 8][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const int c = c0 + ty + jj * 8, r = r0 + tx;
    if (c < C && r < R) dst[(long)c * R + r] = t[tx][ty + jj * 8];
  }
}


// ---------------------------------------------------------------------------------------------------------------
// workspace layout (identical in forward and backward)
// ---------------------------------------------------------------------------------------------------------------
struct CnnWs { float* a[3]; };
struct Ws {
  float *rgbd, *X, *GI, *GH, *HM, *hc;
  CnnWs aud, vis;
  // backward scratch
  float *dX, *dGI, *dGH, *dh[2], *dpre, *da, *db, *cols, *gpack, *whhT;
  unsigned* err;
  void* x16; void* a16[3];       // bf16 mode: 16-bit input / activations of the training forward (avlen_i_cnn3_fwd16_keep)
  void* gws; void* xs; size_t xs_bytes;
};
size_t cnn_cols_max(const avlen_cnn3* n, const Dims& d, long R) {
  size_t mx = 0;
  for (int i = 0; i < 3; i++)
    mx = zmax(mx, (size_t)R * d.h[i + 1] * d.w[i + 1] * n->conv[i].kh * n->conv[i].kw * n->conv[i].cin);
  return mx;
}
size_t cnn_act_max(const Dims& d, long R) {
  size_t mx = 0;
  for (int i = 1; i <= 3; i++) mx = zmax(mx, (size_t)R * d.h[i] * d.w[i] * d.c[i]);
  return mx;
}
size_t cnn_gpack_max(const avlen_cnn3* n) {
  size_t mx = (size_t)n->fc.out_f * n->fc.in_f;
  for (int i = 0; i < 3; i++) mx = zmax(mx, (size_t)n->conv[i].cout * n->conv[i].kh * n->conv[i].kw * n->conv[i].cin);
  return mx;
}
bool layout(WsBump& w, Ws& s, const avlen_cnn3* au, const avlen_cnn3* vi, const avlen_gru* g, int T, int N, int Ha, int Wa, int S,
            int prec) {
  const long R = (long)T * N;
  const int H = g->hidden, F = g->in_f;
  const Dims da = cnn_dims(au, Ha, Wa), dv = cnn_dims(vi, S, S);
  s.rgbd = w.take<float>((size_t)R * S * S * 4);
  s.X = w.take<float>((size_t)R * F);
  s.GI = w.take<float>((size_t)R * 3 * H); s.GH = w.take<float>((size_t)R * 3 * H); s.HM = w.take<float>((size_t)R * H);
  s.hc = w.take<float>((size_t)N * H);
  s.err = w.take<unsigned>(64);
  for (int i = 0; i < 3; i++) {
    s.aud.a[i] = w.take<float>((size_t)R * da.h[i + 1] * da.w[i + 1] * da.c[i + 1]);
    s.vis.a[i] = w.take<float>((size_t)R * dv.h[i + 1] * dv.w[i + 1] * dv.c[i + 1]);
  }
  s.dX = w.take<float>((size_t)R * F);
  s.dGI = w.take<float>((size_t)R * 3 * H); s.dGH = w.take<float>((size_t)R * 3 * H);
  s.dh[0] = w.take<float>((size_t)N * H); s.dh[1] = w.take<float>((size_t)N * H);
  s.whhT = w.take<float>((size_t)3 * H * H);
  s.dpre = w.take<float>((size_t)R * zmax(au->fc.out_f, vi->fc.out_f));
  const size_t am = zmax(cnn_act_max(da, R), cnn_act_max(dv, R));
  s.da = w.take<float>(am); s.db = w.take<float>(am);
  const size_t cm = zmax(cnn_cols_max(au, da, R), cnn_cols_max(vi, dv, R));
  s.cols = w.take<float>(cm);
  s.gpack = w.take<float>(zmax(cnn_gpack_max(au), cnn_gpack_max(vi)));
  s.gws = w.take<char>(GEMM_SCRATCH);
  // operand scratch of the large-M bf16 products (modules.hip: cast / transposed-cast copies of both operands): the largest
  // product is a conv's weight gradient, dY^T (cout x M) and cols^T (K x M)
  s.xs = nullptr; s.xs_bytes = 0;
  s.x16 = nullptr; s.a16[0] = s.a16[1] = s.a16[2] = nullptr;
  if (prec == AVLEN_PREC_BF16 || prec == AVLEN_PREC_BF16X3) {
    s.xs_bytes = (cm + am + (size_t)64 * 1024 * 1024) * 2 + (1u << 20);
    s.xs = w.take<char>(s.xs_bytes);
    const size_t px = zmax((size_t)R * Ha * Wa, (size_t)R * S * S);
    s.x16 = w.take<short>(px * 8);
    for (int i = 0; i < 3; i++)
      s.a16[i] = w.take<short>(zmax((size_t)R * da.h[i + 1] * da.w[i + 1] * da.c[i + 1], (size_t)R * dv.h[i + 1] * dv.w[i + 1] * dv.c[i + 1]));
  }
  return w.ok();
}

// acc: the accurate mode of the GRU baseline (AVLEN_PREC_BF16X3 at the ABI): the convolutions on compensated bf16 pairs (three
// MFMAs per product, the fp32-staged implicit GEMM -- fp16 operands measured 3e-3 on the hidden state against the reference's
// goldens, outside the 1e-3 this mode exists for), every Linear / GRU product in exact fp32 (c.prec)
int cnn_fwd(const avlen_ctx& c, const avlen_cnn3* n, const float* x, long R, int H, int W, CnnWs& a, float* out, int ld_out,
            void* x16 = nullptr, void* const* a16 = nullptr, bool acc = false) {
  const Dims d = cnn_dims(n, H, W);
  if (d.h[3] <= 0 || d.w[3] <= 0 || n->fc.in_f != d.h[3] * d.w[3] * d.c[3]) return AVLEN_ERR_ARG;
  if (c.prec == AVLEN_PREC_BF16 && !acc && x16 && conv_dw_direct_on()) {      // the 16-bit conv kernels, fp32 outputs kept for the backward
    const int rc = avlen_i_cnn3_fwd16_keep(n, x, (int)R, H, W, a.a, out, ld_out, x16, a16, c.gws, c.gws_bytes, c.st);
    if (rc != AVLEN_NOT_BIG) return rc;
  }
  const float* cur = x;
  for (int i = 0; i < 3; i++) {
    const avlen_conv& k = n->conv[i];
    TRY(avlen_conv2d_nhwc(cur, k.w, k.b, nullptr, a.a[i], (int)R, d.h[i], d.w[i], k.cin, k.cout, k.kh, k.kw, k.stride, 0,
                          i < 2 ? AVLEN_ACT_RELU : AVLEN_ACT_NONE, acc ? AVLEN_PREC_BF16X3 : c.prec, c.st));
    cur = a.a[i];
  }
  return avlen_i_linear(c, n->fc, cur, n->fc.in_f, out, ld_out, (int)R, AVLEN_ACT_RELU, nullptr, 0);
}

// d_out: gradient w.r.t. the CNN's (post-ReLU) output, rows `ld` apart inside dX; y: that output inside X
int cnn_bwd(const avlen_ctx& cl, Ws& s, const avlen_cnn3* n, const avlen_cnn3* g, const float* x, long R, int H, int W, CnnWs& a,
            const float* d_out, const float* y, int ld, bool conv16 = false) {
  // cl: the Linear products (fc); c: the conv gradients -- bf16 operands on the direct kernels in both fast modes
  avlen_ctx c = cl;
  if (conv16) c.prec = AVLEN_PREC_BF16;
  const Dims d = cnn_dims(n, H, W);
  hipStream_t st = c.st;
  const int O = n->fc.out_f, K = n->fc.in_f;
  // fc + ReLU
  {
    const long tot = R * O;
    hipLaunchKernelGGL(relu_mask_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_out, ld, y, ld, s.dpre, O, R, O);
    TRY(avlen_zero_bytes(s.gpack, (size_t)O * K * 4, st));
    avlen_linear G = n->fc; G.w = s.gpack; G.b = nullptr;
    TRY(avlen_i_linear_dw(cl, G, s.dpre, O, a.a[2], K, (int)R));
    const long nw = (long)O * K;
    hipLaunchKernelGGL(unpack_fc_grad_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, s.gpack, g->fc.w, O, d.c[3],
                       d.h[3] * d.w[3]);
    TRY(avlen_i_colsum_acc(cl, s.dpre, O, g->fc.b, (int)R, O));
    TRY(avlen_i_linear_dx(cl, n->fc, s.dpre, O, s.da, K, (int)R, nullptr, 0));           // d a[2]  (conv 2 has no ReLU)
  }
  float* dy = s.da; float* other = s.db;
  for (int i = 2; i >= 0; i--) {
    const avlen_conv& k = n->conv[i];
    const long M = R * d.h[i + 1] * d.w[i + 1];
    const int Kc = k.kh * k.kw * k.cin;
    const float* in = i == 0 ? x : a.a[i - 1];
    if (M > 0x7fffffffL) return AVLEN_ERR_ARG;
    // weight / bias gradient: the direct kernel (conv_bwd.hip: nothing materialised, bias sums included) where it applies
    int rc = conv_dw_direct_on() ? avlen_i_conv_dw_direct(c, s.gpack, g->conv[i].b, k.cout, dy, in, R, d.h[i], d.w[i], k.cin, d.h[i + 1],
                                                          d.w[i + 1], k.kh, k.kw, k.stride)
                                 : AVLEN_NOT_BIG;
    if (rc == AVLEN_NOT_BIG) {
      TRY(avlen_zero_bytes(s.gpack, (size_t)k.cout * Kc * 4, st));
      avlen_linear G{s.gpack, nullptr, k.cout, Kc, nullptr, 0};
      rc = avlen_i_conv_dw16(c, G, dy, k.cout, in, R, d.h[i], d.w[i], k.cin, d.h[i + 1], d.w[i + 1], k.kh, k.kw, k.stride, 0);
      if (rc == AVLEN_NOT_BIG) {
        TRY(im2col(st, in, s.cols, R, d.h[i], d.w[i], k.cin, d.h[i + 1], d.w[i + 1], k.kh, k.kw, k.stride, 0));
        TRY(avlen_i_linear_dw(c, G, dy, k.cout, s.cols, Kc, (int)M));
      } else TRY(rc);
      TRY(avlen_i_colsum_acc(c, dy, k.cout, g->conv[i].b, (int)M, k.cout));
    } else TRY(rc);
    const long nw = (long)k.cout * Kc;
    hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, s.gpack, g->conv[i].w, k.cout,
                       k.cin, k.kh, k.kw);
    if (i == 0) break;
    // data gradient, masked by the ReLU of the layer below: the direct kernel (conv_bwd.hip) where it applies, else dcols = dY * Wp
    // gathered back onto the input pixels
    rc = conv_dw_direct_on() ? avlen_i_conv_dx_direct(c, k.w, dy, a.a[i - 1], other, R, d.h[i], d.w[i], k.cin, d.h[i + 1], d.w[i + 1], k.cout,
                                                      k.kh, k.kw, k.stride)
                             : AVLEN_NOT_BIG;
    if (rc == AVLEN_NOT_BIG) {
      avlen_linear Wl{k.w, nullptr, k.cout, Kc, nullptr, 0};
      TRY(avlen_i_linear_dx(c, Wl, dy, k.cout, s.cols, Kc, (int)M, nullptr, 0));
      TRY(col2im_relu(st, s.cols, a.a[i - 1], other, R, d.h[i], d.w[i], k.cin, d.h[i + 1], d.w[i + 1], k.kh, k.kw, k.stride, 0));
    } else TRY(rc);
    float* t = dy; dy = other; other = t;
  }
  return avlen_launch_status();
}

}  // namespace

// out[m][n] = sum_k x[m][k] W[n][k] + b[n] for a handful of rows (M <= 64) against a tall weight matrix whose K is not
// 8-aligned (the GRU's input projection: 16 x 1045 x 1536 per rollout step): one wave per output column, W row in registers,
// 16 rows at a time.  The tile GEMM needs 56 us for it (a dozen workgroups, fp32 staging); this is latency only.
namespace {
constexpr int SK_KP = 20;                    // K <= 64 * SK_KP
__global__ __launch_bounds__(256) void skinny_linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                            const float* __restrict__ b, float* __restrict__ out, int ldo, int M,
                                                            int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float w[SK_KP];
#pragma unroll
  for (int i = 0; i < SK_KP; i++) { const int k = lane + 64 * i; w[i] = k < K ? W[(long)n * K + k] : 0.f; }
  const float bn = b ? b[n] : 0.f;
  for (int m0 = 0; m0 < M; m0 += 16) {
    float acc[16];
#pragma unroll
    for (int mm = 0; mm < 16; mm++) {
      acc[mm] = 0.f;
      if (m0 + mm < M) {
#pragma unroll
        for (int i = 0; i < SK_KP; i++) { const int k = lane + 64 * i; acc[mm] += (k < K ? x[(long)(m0 + mm) * ldx + k] : 0.f) * w[i]; }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int mm = 0; mm < 16; mm++) { const float sv = wave_sum(acc[mm]); if (lane == mm) r = sv; }
    if (lane < 16 && m0 + lane < M) out[(long)(m0 + lane) * ldo + n] = r + bn;
  }
}
}  // namespace
bool avlen_i_skinny_linear_ok(int M, int K) { return M <= 64 && K <= 64 * SK_KP; }
int avlen_i_skinny_linear(const float* x, int ldx, const float* W, const float* b, float* out, int ldo, int M, int N, int K,
                          hipStream_t st) {
  hipLaunchKernelGGL(skinny_linear_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, st, x, ldx, W, b, out, ldo, M, N, K);
  return avlen_launch_status();
}

// one GRU step for up to a few dozen rows (rollout `act`, modules.hip:avlen_gru_fwd): see gru_step_fwd_kernel
bool avlen_i_gru_step_ok(int N, int H) { return N <= 64 && H <= 64 * GRU_HP; }
int avlen_i_gru_step_fwd(const avlen_gru* p, const float* gi, const float* hprev, const float* mask, float* out, int N,
                         hipStream_t st) {
  const int H = p->hidden;
  if (gru_seq_ok(N, H)) {       // the sequence kernel with T = 1 (no hand-off): LDS-transposed reductions, fma products: 15 -> 5 us at N = 16
    // T = 1: no hand-off, nothing is waited for -> no error word (the kernel guards every use of it)
    return launch_gru_seq_fwd(p->w_hh, p->b_hh, gi, hprev, mask, out, nullptr, nullptr, 1, N, H, nullptr, st);
  }
  auto kern = N <= 8 ? gru_step_fwd_kernel<8> : gru_step_fwd_kernel<16>;
  hipLaunchKernelGGL(kern, dim3(ceil_div(H, 4)), dim3(256), 0, st, p->w_hh, p->b_hh, gi, hprev, mask, out,
                     (float*)nullptr, (float*)nullptr, N, H);
  return avlen_launch_status();
}

extern "C" size_t avlen_baseline_train_workspace_bytes(const avlen_cnn3* audio, const avlen_cnn3* visual, const avlen_gru* gru,
                                                       int T, int N, int Ha, int Wa, int S, int prec) {
  WsBump w(nullptr, 0);
  Ws s;
  layout(w, s, audio, visual, gru, T, N, Ha, Wa, S, prec);
  return w.off + 4096;
}

extern "C" int avlen_baseline_train_fwd(const avlen_cnn3* audio, const avlen_cnn3* visual, const avlen_gru* gru,
                                        const float* spec, const void* rgb, int rgb_u8, const float* depth, const float* category, int ncat,
                                        const float* h0, const float* masks, float* out, float* h_out, int T, int N, int Ha, int Wa,
                                        int S, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!audio || !visual || !gru || T <= 0 || N <= 0) return AVLEN_ERR_ARG;
  const int H = gru->hidden, F = gru->in_f;
  if (audio->fc.out_f + visual->fc.out_f + ncat != F) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes);
  Ws s;
  if (!ws || !layout(w, s, audio, visual, gru, T, N, Ha, Wa, S, prec)) return AVLEN_ERR_WS;
  const long R = (long)T * N;
  const bool acc = prec == AVLEN_PREC_BF16X3;            // accurate mode: compensated convs, exact fp32 for every other product
  avlen_ctx c{st, acc ? AVLEN_PREC_FP32 : prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  TRY(avlen_rgbd_concat(rgb, rgb_u8, depth, s.rgbd, (int)R, S * S, st));
  TRY(cnn_fwd(c, audio, spec, R, Ha, Wa, s.aud, s.X, F, s.x16, s.a16, acc));
  TRY(cnn_fwd(c, visual, s.rgbd, R, S, S, s.vis, s.X + audio->fc.out_f, F, s.x16, s.a16, acc));
  if (ncat) TRY(avlen_copy_rows(category, ncat, s.X + audio->fc.out_f + visual->fc.out_f, F, (int)R, ncat, st));
  avlen_linear ih{gru->w_ih, gru->b_ih, 3 * H, F, nullptr, 0};
  TRY(avlen_i_linear(c, ih, s.X, F, s.GI, 3 * H, (int)R, 0, nullptr, 0));
  if (H > 64 * GRU_HP) return AVLEN_ERR_ARG;
  const float* hprev = h0;
  if (gru_seq_ok(N, H)) {                    // the whole sequence in one launch: `out` is the hand-off buffer (sentinel-filled)
    if (hipMemsetAsync(out, 0xff, (size_t)R * H * sizeof(float), st) != hipSuccess) return AVLEN_ERR_LAUNCH;
    TRY(avlen_zero_bytes(s.err, 64 * sizeof(unsigned), st));
    TRY(launch_gru_seq_fwd(gru->w_hh, gru->b_hh, s.GI, h0, masks, out, s.HM, s.GH, T, N, H, s.err, st));
    hprev = out + (size_t)(T - 1) * N * H;
  } else {
    for (int t = 0; t < T; t++) {
      auto kern = N <= 8 ? gru_step_fwd_kernel<8> : gru_step_fwd_kernel<16>;
      hipLaunchKernelGGL(kern, dim3(ceil_div(H, 4)), dim3(256), 0, st, gru->w_hh, gru->b_hh,
                         s.GI + (size_t)t * N * 3 * H, hprev, masks + (size_t)t * N, out + (size_t)t * N * H,
                         s.HM + (size_t)t * N * H, s.GH + (size_t)t * N * 3 * H, N, H);
      hprev = out + (size_t)t * N * H;
    }
  }
  TRY(avlen_launch_status());
  if (h_out) TRY(avlen_copy_rows(hprev, H, h_out, H, N, H, st));
  return AVLEN_OK;
}

extern "C" int avlen_baseline_train_bwd(const avlen_cnn3* audio, const avlen_cnn3* visual, const avlen_gru* gru,
                                        const avlen_cnn3* g_audio, const avlen_cnn3* g_visual, const avlen_gru* g_gru,
                                        const float* spec, const float* masks, const float* d_out, int T, int N, int Ha, int Wa, int S,
                                        int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!audio || !visual || !gru || !g_audio || !g_visual || !g_gru || T <= 0 || N <= 0) return AVLEN_ERR_ARG;
  const int H = gru->hidden, F = gru->in_f;
  WsBump w(ws, ws_bytes);
  Ws s;
  if (!ws || !layout(w, s, audio, visual, gru, T, N, Ha, Wa, S, prec)) return AVLEN_ERR_WS;
  const long R = (long)T * N;
  const bool acc = prec == AVLEN_PREC_BF16X3;
  avlen_ctx c{st, acc ? AVLEN_PREC_FP32 : prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  // ---- BPTT through the masked GRU: one launch per step (see gru_step_bwd_kernel)
  if (3 * H > 64 * GRU_KP || H % 4) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(3 * H, 32), ceil_div(H, 32)), dim3(32, 8), 0, st, gru->w_hh, s.whhT, 3 * H, H);
  if (gru_seq_ok(N, H)) {                    // one resident launch for all T steps: dGH is the hand-off buffer (sentinel-filled)
    if (hipMemsetAsync(s.dGH, 0xff, (size_t)R * 3 * H * sizeof(float), st) != hipSuccess) return AVLEN_ERR_LAUNCH;
    TRY(avlen_zero_bytes(s.err, 64 * sizeof(unsigned), st));
    TRY(launch_gru_seq_bwd(s.whhT, s.GI, s.GH, s.HM, d_out, masks, s.dGI, s.dGH, T, N, H, s.err, st));
  } else {
    for (int t = T - 1; t >= 0; t--) {
      const bool last = t == T - 1;
      // 8 rows per pass: 8 x 3H floats of LDS (48 KB at H = 512)
      hipLaunchKernelGGL(gru_step_bwd_kernel<8>, dim3(ceil_div(H, 4)), dim3(256), (size_t)8 * 3 * H * sizeof(float), st, s.whhT, s.GI + (size_t)t * N * 3 * H,
                         s.GH + (size_t)t * N * 3 * H, s.HM + (size_t)t * N * H, d_out + (size_t)t * N * H,
                         last ? nullptr : s.dGH + (size_t)(t + 1) * N * 3 * H, last ? nullptr : s.dh[(t + 1) & 1],
                         last ? nullptr : masks + (size_t)(t + 1) * N, s.dGI + (size_t)t * N * 3 * H, s.dGH + (size_t)t * N * 3 * H,
                         s.dh[t & 1], N, H);
    }
  }
  TRY(avlen_launch_status());
  avlen_linear Gih{g_gru->w_ih, nullptr, 3 * H, F, nullptr, 0}, Ghh{g_gru->w_hh, nullptr, 3 * H, H, nullptr, 0};
  TRY(avlen_i_linear_dw(c, Gih, s.dGI, 3 * H, s.X, F, (int)R));
  TRY(avlen_i_linear_dw(c, Ghh, s.dGH, 3 * H, s.HM, H, (int)R));
  TRY(avlen_i_colsum_acc(c, s.dGI, 3 * H, g_gru->b_ih, (int)R, 3 * H));
  TRY(avlen_i_colsum_acc(c, s.dGH, 3 * H, g_gru->b_hh, (int)R, 3 * H));
  avlen_linear ih{gru->w_ih, gru->b_ih, 3 * H, F, nullptr, 0};
  TRY(avlen_i_linear_dx(c, ih, s.dGI, 3 * H, s.dX, F, (int)R, nullptr, 0));
  // ---- the two CNNs (the category columns of x are data)
  TRY(cnn_bwd(c, s, audio, g_audio, spec, R, Ha, Wa, s.aud, s.dX, s.X, F, acc));
  TRY(cnn_bwd(c, s, visual, g_visual, s.rgbd, R, S, S, s.vis, s.dX + audio->fc.out_f, s.X + audio->fc.out_f, F, acc));
  return avlen_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// AudioCNN alone (pi_l's goal encoder under PPO.update_dialog, ppo.py:99-154): the same forward / backward as above with its
// own workspace.  out rows are ld_out apart (the CNN's 128 columns inside the policy's feature rows).
// ---------------------------------------------------------------------------------------------------------------
namespace {
bool cnn_layout(WsBump& w, Ws& s, const avlen_cnn3* n, long R, int H, int W, int prec) {
  const Dims d = cnn_dims(n, H, W);
  for (int i = 0; i < 3; i++) s.aud.a[i] = w.take<float>((size_t)R * d.h[i + 1] * d.w[i + 1] * d.c[i + 1]);
  s.dpre = w.take<float>((size_t)R * n->fc.out_f);
  const size_t am = cnn_act_max(d, R), cm = cnn_cols_max(n, d, R);
  s.da = w.take<float>(am); s.db = w.take<float>(am);
  s.cols = w.take<float>(cm);
  s.gpack = w.take<float>(cnn_gpack_max(n));
  s.gws = w.take<char>(GEMM_SCRATCH);
  s.xs = nullptr; s.xs_bytes = 0;
  if (prec == AVLEN_PREC_BF16) {
    s.xs_bytes = (cm + am + (size_t)64 * 1024 * 1024) * 2 + (1u << 20);
    s.xs = w.take<char>(s.xs_bytes);
  }
  return w.ok();
}
}  // namespace

extern "C" size_t avlen_cnn3_train_workspace_bytes(const avlen_cnn3* net, int B, int H, int W, int prec) {
  WsBump w(nullptr, 0); Ws s;
  cnn_layout(w, s, net, B, H, W, prec);
  return w.off + 4096;
}
extern "C" int avlen_cnn3_train_fwd(const avlen_cnn3* net, const float* x, int B, int H, int W, float* out, int ld_out, int prec,
                                    void* ws, size_t ws_bytes, hipStream_t st) {
  if (!net || !x || !out || B <= 0) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes); Ws s;
  if (!ws || !cnn_layout(w, s, net, B, H, W, prec)) return AVLEN_ERR_WS;
  avlen_ctx c{st, prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  return cnn_fwd(c, net, x, B, H, W, s.aud, out, ld_out);
}
// y = the forward's output rows (post-ReLU), d_out = their gradient; both ld apart
extern "C" int avlen_cnn3_train_bwd(const avlen_cnn3* net, const avlen_cnn3* grads, const float* x, const float* y, const float* d_out,
                                    int ld, int B, int H, int W, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!net || !grads || !x || !y || !d_out || B <= 0) return AVLEN_ERR_ARG;
  WsBump w(ws, ws_bytes); Ws s;
  if (!ws || !cnn_layout(w, s, net, B, H, W, prec)) return AVLEN_ERR_WS;
  avlen_ctx c{st, prec, s.gws, GEMM_SCRATCH};
  c.xs = s.xs; c.xs_bytes = s.xs_bytes;
  return cnn_bwd(c, s, net, grads, x, B, H, W, s.aud, d_out, y, ld);
}
