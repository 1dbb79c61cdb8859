// Multi-head attention core (softmax(QK^T)V) for short sequences (S <= ~512, head dim 32/64), gfx950.
//
// The sequences on this path are tiny by GPU standards (scene memory 301 tokens, CLIP text 77,
// dialog memory 4), so one wavefront owns 64 query rows of one (sample, head): a lane keeps its q row
// and output accumulator in registers, keys/values stream through LDS in chunks of 64 with the
// PADDED KEYS COMPACTED AWAY while staging (wave ballot + prefix popcount), so an external memory that
// is half empty costs half the work.  Softmax is the online (running max / sum) form, rescaled once
// per 8 keys.  fp32 throughout (VALU): this is <20 % of the SMT block's FLOPs.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {

constexpr int CH = 64;      // keys per LDS chunk (= wave width: lane j stages key j)

template <int D>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                      int ldk, const float* __restrict__ V, int ldv,
                                                      float* __restrict__ O, int ldo, __bf16* __restrict__ O16, int ldo16,
                                                      const float* __restrict__ key_mask, float* __restrict__ lse, int H,
                                                      int Sq, int Sk, int causal, float scale) {
  __shared__ __attribute__((aligned(16))) float ks[CH * D];
  __shared__ __attribute__((aligned(16))) float vs[CH * D];
  __shared__ int kidx[CH];
  const int lane = threadIdx.x;
  const int b = blockIdx.z, h = blockIdx.y;
  const int i = blockIdx.x * 64 + lane;                 // this lane's query row
  const bool qok = i < Sq;
  float q[D], o[D];
#pragma unroll
  for (int d = 0; d < D; d++) { q[d] = 0.f; o[d] = 0.f; }
  if (qok) {
    const float4* qp = reinterpret_cast<const float4*>(Q + ((long)b * Sq + i) * ldq + h * D);
#pragma unroll
    for (int d4 = 0; d4 < D / 4; d4++) {
      float4 t = qp[d4];
      q[d4 * 4] = t.x * scale; q[d4 * 4 + 1] = t.y * scale; q[d4 * 4 + 2] = t.z * scale; q[d4 * 4 + 3] = t.w * scale;
    }
  }
  float m = -INFINITY, l = 0.f;
  // causal: keys beyond the last query row of this block are never needed
  const int sk_end = causal ? min(Sk, blockIdx.x * 64 + 64) : Sk;
  for (int j0 = 0; j0 < sk_end; j0 += CH) {
    const int j = j0 + lane;
    bool valid = j < sk_end && (key_mask == nullptr || key_mask[(long)b * Sk + j] != 0.f);
    unsigned long long bal = __ballot(valid);
    int nvalid = __popcll(bal);
    int pos = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                                   // previous chunk fully consumed
    if (valid) {
      const float4* kp = reinterpret_cast<const float4*>(K + ((long)b * Sk + j) * ldk + h * D);
      const float4* vp = reinterpret_cast<const float4*>(V + ((long)b * Sk + j) * ldv + h * D);
      float4* kd = reinterpret_cast<float4*>(&ks[pos * D]);
      float4* vd = reinterpret_cast<float4*>(&vs[pos * D]);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++) { kd[d4] = kp[d4]; vd[d4] = vp[d4]; }
      kidx[pos] = j;
    }
    __syncthreads();
    for (int g0 = 0; g0 < nvalid; g0 += 8) {
      float s[8];
      float mg = -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int jj = g0 + u;
        float acc = -INFINITY;
        if (jj < nvalid) {                              // wave-uniform
          const float4* kr = reinterpret_cast<const float4*>(&ks[jj * D]);
          float a = 0.f;
#pragma unroll
          for (int d4 = 0; d4 < D / 4; d4++) {
            float4 t = kr[d4];
            a += q[d4 * 4] * t.x + q[d4 * 4 + 1] * t.y + q[d4 * 4 + 2] * t.z + q[d4 * 4 + 3] * t.w;
          }
          acc = (causal && kidx[jj] > i) ? -INFINITY : a;
        }
        s[u] = acc;
        mg = fmaxf(mg, acc);
      }
      float mn = fmaxf(m, mg);
      float corr = (m == -INFINITY) ? 0.f : __expf(m - mn);
      l *= corr;
#pragma unroll
      for (int d = 0; d < D; d++) o[d] *= corr;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int jj = g0 + u;
        if (jj < nvalid) {
          float p = (s[u] == -INFINITY) ? 0.f : __expf(s[u] - mn);
          l += p;
          const float4* vr = reinterpret_cast<const float4*>(&vs[jj * D]);
#pragma unroll
          for (int d4 = 0; d4 < D / 4; d4++) {
            float4 t = vr[d4];
            o[d4 * 4] += p * t.x; o[d4 * 4 + 1] += p * t.y; o[d4 * 4 + 2] += p * t.z; o[d4 * 4 + 3] += p * t.w;
          }
        }
      }
      m = mn;
    }
  }
  if (qok) {
    float inv = l > 0.f ? 1.f / l : 0.f;
    if (O) {
      float4* op = reinterpret_cast<float4*>(O + ((long)b * Sq + i) * ldo + h * D);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++)
        op[d4] = make_float4(o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv);
    }
    if (O16) {
      __bf16* oh = O16 + ((long)b * Sq + i) * ldo16 + h * D;
#pragma unroll
      for (int d = 0; d < D; d++) oh[d] = (__bf16)(o[d] * inv);
    }
    if (lse) lse[((long)b * H + h) * Sq + i] = m + __logf(l);
  }
}

// Unmasked / causal variant (CLIP text: S = 77, D = 64): FOUR lanes per query, each owning D/4 of the head dim, so a
// 32-query block is 128 threads and the dot products / PV updates are 4x shorter per lane (2 shuffles per key).
template <int D>
__global__ __launch_bounds__(128) void attn_fwd4_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                        int ldk, const float* __restrict__ V, int ldv,
                                                        float* __restrict__ O, int ldo, __bf16* __restrict__ O16, int ldo16,
                                                        float* __restrict__ lse, int H, int Sq, int Sk, int causal,
                                                        float scale, const int* __restrict__ seg_off) {
  // seg_off (optional, ragged batches): sample b owns rows seg_off[b] .. seg_off[b+1]-1 of Q/K/V/O (Sq = Sk = its length)
  long row0 = (long)blockIdx.z * Sq;
  if (seg_off) {
    row0 = seg_off[blockIdx.z];
    Sq = Sk = seg_off[blockIdx.z + 1] - seg_off[blockIdx.z];
    if ((int)blockIdx.x * 32 >= Sq) return;
  }
  const long krow0 = seg_off ? row0 : (long)blockIdx.z * Sk;
  constexpr int DP = D / 4;                       // dims per lane
  constexpr int KC = 96;                          // keys per LDS chunk (CLIP's 77 keys land in one pass)
  __shared__ __attribute__((aligned(16))) float ks[KC * D];
  __shared__ __attribute__((aligned(16))) float vs[KC * D];
  const int tid = threadIdx.x, part = tid & 3, ql = tid >> 2;
  const int b = blockIdx.z, h = blockIdx.y;
  const int i = blockIdx.x * 32 + ql;
  const bool qok = i < Sq;
  float q[DP], o[DP];
#pragma unroll
  for (int d = 0; d < DP; d++) { q[d] = 0.f; o[d] = 0.f; }
  if (qok) {
    const float* qp = Q + (row0 + i) * ldq + h * D + part * DP;
#pragma unroll
    for (int d = 0; d < DP; d++) q[d] = qp[d] * scale;
  }
  float m = -INFINITY, l = 0.f;
  const int sk_end = causal ? min(Sk, blockIdx.x * 32 + 32) : Sk;
  for (int j0 = 0; j0 < sk_end; j0 += KC) {
    __syncthreads();
    for (int jl = ql; jl < KC; jl += 32) {         // thread (key = jl, part) stages D/4 floats of K and V
      const int j = j0 + jl;
      if (j < sk_end) {
        const float4* kp = reinterpret_cast<const float4*>(K + (krow0 + j) * ldk + h * D + part * DP);
        const float4* vp = reinterpret_cast<const float4*>(V + (krow0 + j) * ldv + h * D + part * DP);
        float4* kd = reinterpret_cast<float4*>(&ks[jl * D + part * DP]);
        float4* vd = reinterpret_cast<float4*>(&vs[jl * D + part * DP]);
#pragma unroll
        for (int d4 = 0; d4 < DP / 4; d4++) { kd[d4] = kp[d4]; vd[d4] = vp[d4]; }
      }
    }
    __syncthreads();
    const int nk = min(KC, sk_end - j0);
    for (int g0 = 0; g0 < nk; g0 += 4) {
      float s[4];
      float mg = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int jj = g0 + u;
        float a = 0.f;
        if (jj < nk) {
          const float* kr = &ks[jj * D + part * DP];
#pragma unroll
          for (int d = 0; d < DP; d++) a += q[d] * kr[d];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (jj >= nk || (causal && j0 + jj > i)) a = -INFINITY;
        s[u] = a;
        mg = fmaxf(mg, a);
      }
      const float mn = fmaxf(m, mg);
      if (mn == -INFINITY) continue;               // nothing visible yet for this query
      const float corr = (m == -INFINITY) ? 0.f : __expf(m - mn);
      l *= corr;
#pragma unroll
      for (int d = 0; d < DP; d++) o[d] *= corr;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int jj = g0 + u;
        if (jj < nk) {
          const float pw = (s[u] == -INFINITY) ? 0.f : __expf(s[u] - mn);
          l += pw;
          const float* vr = &vs[jj * D + part * DP];
#pragma unroll
          for (int d = 0; d < DP; d++) o[d] += pw * vr[d];
        }
      }
      m = mn;
    }
  }
  if (qok) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (O) {
      float* op = O + (row0 + i) * ldo + h * D + part * DP;
#pragma unroll
      for (int d = 0; d < DP; d++) op[d] = o[d] * inv;
    }
    if (O16) {
      __bf16* oh = O16 + (row0 + i) * ldo16 + h * D + part * DP;
#pragma unroll
      for (int d = 0; d < DP; d++) oh[d] = (__bf16)(o[d] * inv);
    }
    if (lse && part == 0) lse[((long)b * H + h) * Sq + i] = m + __logf(l);
  }
}

// MFMA attention for short unmasked / causal sequences (CLIP text: Sk <= 96, D = 64), bf16 operands, fp32 softmax.
// One block (4 waves) per (sample, head): Q, K and V^T are staged once in LDS as bf16; wave w owns the 16-query tiles
// w and w+4: S = Q K^T (12 MFMA per key tile), row softmax with 4 shuffles, P -> LDS (transposes the accumulator layout
// into an A operand), O = P V (12 MFMA per 16 output dims).  Ragged batches via seg_off (see attn_fwd4_kernel).
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(4))) float af32x4;

__global__ __launch_bounds__(256) void attn_mfma64_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                          int ldk, const float* __restrict__ V, int ldv,
                                                          __bf16* __restrict__ O16, int ldo16, int Sq, int Sk, int causal,
                                                          float scale, const int* __restrict__ seg_off) {
  constexpr int D = 64, SKP = 96, KR = D + 8, VR = SKP + 8;     // padded LDS rows (bytes: 144 / 208) -> fewer bank conflicts
  __shared__ __attribute__((aligned(16))) __bf16 qs[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 ks[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 vt[D * VR];      // V transposed: [d][key]
  __shared__ __attribute__((aligned(16))) __bf16 ps[4][16 * VR];  // per wave: P tile [query 16][key 96]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  long row0 = (long)b * Sq, krow0 = (long)b * Sk;
  if (seg_off) { row0 = krow0 = seg_off[b]; Sq = Sk = seg_off[b + 1] - seg_off[b]; }
  // ---- stage (fp32 -> bf16); rows >= Sk / Sq are zero
  for (int i = tid; i < SKP * 16; i += 256) {
    int r = i >> 4, c4 = (i & 15) * 4;
    float4 kv = make_float4(0, 0, 0, 0), vv = kv, qv = kv;
    if (r < Sk) {
      kv = *reinterpret_cast<const float4*>(K + (krow0 + r) * ldk + h * D + c4);
      vv = *reinterpret_cast<const float4*>(V + (krow0 + r) * ldv + h * D + c4);
    }
    if (r < Sq) qv = *reinterpret_cast<const float4*>(Q + (row0 + r) * ldq + h * D + c4);
    __bf16* kd = &ks[r * KR + c4]; __bf16* qd = &qs[r * KR + c4];
    kd[0] = (__bf16)kv.x; kd[1] = (__bf16)kv.y; kd[2] = (__bf16)kv.z; kd[3] = (__bf16)kv.w;
    qd[0] = (__bf16)(qv.x * scale); qd[1] = (__bf16)(qv.y * scale); qd[2] = (__bf16)(qv.z * scale); qd[3] = (__bf16)(qv.w * scale);
    vt[(c4 + 0) * VR + r] = (__bf16)vv.x; vt[(c4 + 1) * VR + r] = (__bf16)vv.y;
    vt[(c4 + 2) * VR + r] = (__bf16)vv.z; vt[(c4 + 3) * VR + r] = (__bf16)vv.w;
  }
  __syncthreads();
  const int n_kt = (Sk + 15) >> 4;                 // key tiles in use (<= 6)
  __bf16* pw = ps[wave];
  for (int mt = wave; mt * 16 < Sq; mt += 4) {
    // ---- S = Q K^T for 16 queries x up to 96 keys: lane holds S[query = q4*4 + r][key = nt*16 + r16]
    af32x4 sacc[6];
#pragma unroll
    for (int nt = 0; nt < 6; nt++) sacc[nt] = (af32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      abf16x8 qf = *reinterpret_cast<const abf16x8*>(&qs[(mt * 16 + r16) * KR + kk * 32 + q4 * 8]);
#pragma unroll
      for (int nt = 0; nt < 6; nt++) {
        if (nt < n_kt) {
          abf16x8 kf = *reinterpret_cast<const abf16x8*>(&ks[(nt * 16 + r16) * KR + kk * 32 + q4 * 8]);
          sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, sacc[nt], 0, 0, 0);
        }
      }
    }
    // ---- masked row softmax (rows live on 16 lanes with equal q4; 4 rows per lane)
    float rmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 6; nt++) {
      const int key = nt * 16 + r16;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int qi = mt * 16 + q4 * 4 + r;
        bool ok = nt < n_kt && key < Sk && !(causal && key > qi);
        float v = ok ? sacc[nt][r] : -INFINITY;
        sacc[nt][r] = v;
        rmax[r] = fmaxf(rmax[r], v);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) rmax[r] = fmaxf(rmax[r], __shfl_xor(rmax[r], o, 64));
#pragma unroll
    for (int nt = 0; nt < 6; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float pv = (sacc[nt][r] == -INFINITY) ? 0.f : __expf(sacc[nt][r] - rmax[r]);
        rsum[r] += pv;
        pw[(q4 * 4 + r) * VR + nt * 16 + r16] = (__bf16)pv;       // P[query][key]
      }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) rsum[r] += __shfl_xor(rsum[r], o, 64);
    // the P tile is private to this wave; make its LDS writes visible to the wave's own reads
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // ---- O = P V : lane holds O[query = q4*4 + r][d = dt*16 + r16]
    af32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; dt++) oacc[dt] = (af32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      if (kk * 32 < Sk) {
        abf16x8 pf = *reinterpret_cast<const abf16x8*>(&pw[r16 * VR + kk * 32 + q4 * 8]);
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
          abf16x8 vf = *reinterpret_cast<const abf16x8*>(&vt[(dt * 16 + r16) * VR + kk * 32 + q4 * 8]);
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, oacc[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int qi = mt * 16 + q4 * 4 + r;
      if (qi < Sq) {
        const float inv = rsum[r] > 0.f ? 1.f / rsum[r] : 0.f;
        __bf16* op = O16 + (row0 + qi) * ldo16 + h * D + r16;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) op[dt * 16] = (__bf16)(oacc[dt][r] * inv);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Self-attention straight from the packed bf16 projection [R][q | k | v] (D = 64, <= 96 tokens per sequence: CLIP text).
// Compared with attn_mfma64_kernel: the tiles are staged with 16-byte copies (no conversion, no transposing scatter), the
// scores are computed TRANSPOSED (S^T = K Q^T), so each lane ends up holding, for one query, four consecutive keys of every
// key tile -- which is already the B-operand layout of the second MFMA under the key order {4q..4q+3, 16+4q..16+4q+3} per
// 32-key step: P never leaves registers.  The matching V^T fragment is read from the row-major V tile with the hardware
// transposing read ds_read_b64_tr_b16 (one 4-key x 16-dim block per 16-lane group).  O^T = V^T P^T comes out with four
// consecutive output dims per lane -> 8-byte stores.  The softmax reductions are in-lane + 2 cross-lane moves.
typedef __attribute__((ext_vector_type(4))) __bf16 abf16x4;

typedef __attribute__((ext_vector_type(8))) _Float16 af16x8;
typedef __attribute__((ext_vector_type(4))) short ai16x4;
template <bool F16> __device__ __forceinline__ af32x4 amfma16(const abf16x8& a, const abf16x8& b, const af32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(af16x8, a), __builtin_bit_cast(af16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ __bf16 ato16(float v) {
  if constexpr (F16) return __builtin_bit_cast(__bf16, (_Float16)v);
  else return (__bf16)v;
}

// F16: the packed projection and the output hold IEEE half values (same byte layout; the transposing LDS read moves 16-bit words)
template <bool F16>
__global__ __launch_bounds__(256) void attn_qkv16_kernel(const __bf16* __restrict__ QKV, int ld, int koff, int voff,
                                                         __bf16* __restrict__ O16, int ldo16, int S, int causal, float scale,
                                                         const int* __restrict__ seg_off) {
  constexpr int D = 64, SKP = 96, KR = 80;          // 160-byte LDS rows: conflict-free for both read kinds
  __shared__ __attribute__((aligned(16))) __bf16 qs[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 ks[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 vs[SKP * KR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  long row0 = (long)b * S;
  if (seg_off) { row0 = seg_off[b]; S = seg_off[b + 1] - seg_off[b]; }
  const int S32 = (S + 31) & ~31;                   // rows S .. S32-1 are zero (finite operands for the masked keys)
  for (int i = tid; i < S32 * 8; i += 256) {
    const int r = i >> 3, c = i & 7;
    uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, vv = qv;
    if (r < S) {
      const __bf16* base = QKV + (row0 + r) * ld + h * D + c * 8;
      qv = *reinterpret_cast<const uint4*>(base);
      kv = *reinterpret_cast<const uint4*>(base + koff);
      vv = *reinterpret_cast<const uint4*>(base + voff);
    }
    *reinterpret_cast<uint4*>(&qs[r * KR + c * 8]) = qv;
    *reinterpret_cast<uint4*>(&ks[r * KR + c * 8]) = kv;
    *reinterpret_cast<uint4*>(&vs[r * KR + c * 8]) = vv;
  }
  __syncthreads();
  const int n_kt = (S + 15) >> 4, n_kk = (S + 31) >> 5;
  for (int mt = wave; mt * 16 < S; mt += 4) {
    // ---- S^T: lane (r16, q4) holds S[query = mt*16 + r16][key = nt*16 + q4*4 + r]
    af32x4 sacc[6];
#pragma unroll
    for (int nt = 0; nt < 6; nt++) sacc[nt] = (af32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      abf16x8 qf = *reinterpret_cast<const abf16x8*>(&qs[(mt * 16 + r16) * KR + kk * 32 + q4 * 8]);
#pragma unroll
      for (int nt = 0; nt < 6; nt++)
        if (nt < n_kt) {
          abf16x8 kf = *reinterpret_cast<const abf16x8*>(&ks[(nt * 16 + r16) * KR + kk * 32 + q4 * 8]);
          sacc[nt] = amfma16<F16>(kf, qf, sacc[nt]);
        }
    }
    const int qi = mt * 16 + r16;
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 6; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int key = nt * 16 + q4 * 4 + r;
        const bool ok = nt < n_kt && key < S && !(causal && key > qi);
        const float v = ok ? sacc[nt][r] * scale : -INFINITY;
        sacc[nt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 6; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float pv = (sacc[nt][r] == -INFINITY) ? 0.f : __expf(sacc[nt][r] - mx);
        sum += pv; sacc[nt][r] = pv;
      }
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    // ---- O^T = V^T P^T: lane holds O[query = mt*16 + r16][d = dt*16 + q4*4 + r]
    af32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; dt++) oacc[dt] = (af32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; kk++)
      if (kk < n_kk) {
        abf16x8 pf;
#pragma unroll
        for (int e = 0; e < 4; e++) { pf[e] = ato16<F16>(sacc[2 * kk][e]); pf[4 + e] = ato16<F16>(sacc[2 * kk + 1][e]); }
        const __bf16* vb = &vs[(kk * 32 + q4 * 4 + (r16 >> 2)) * KR + 4 * (r16 & 3)];
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
          abf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(vb + dt * 16));
          abf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(vb + 16 * KR + dt * 16));
          abf16x8 vf;
#pragma unroll
          for (int e = 0; e < 4; e++) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
          oacc[dt] = amfma16<F16>(vf, pf, oacc[dt]);
        }
      }
    if (qi < S) {
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
      __bf16* op = O16 + (row0 + qi) * ldo16 + h * D + q4 * 4;
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        abf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = ato16<F16>(oacc[dt][r] * inv);
        *reinterpret_cast<abf16x4*>(op + dt * 16) = o;
      }
    }
  }
}

// Scene-memory self-attention straight from the packed bf16 projection [R][q | k | v] (D = 32, <= 160 tokens per sample,
// key-padding mask): same scheme as attn_qkv16_kernel -- transposed scores so P stays in registers, V^T fragments by
// ds_read_b64_tr_b16 from the row-major V tile -- with one MFMA k-step per score tile (K = D = 32) and up to 10 key tiles.
template <int SKP, bool X3, int NTH = 256>   // max tokens per sample (160 or 320); X3: compensated bf16 pairs (low planes qkv_lo / o_lo elements
// behind); NTH threads: the 320-token compensated instance holds 121 KB of LDS -- one block per CU -- and runs 8 waves so that two
// waves per SIMD hide each other's LDS / MFMA / softmax latencies (4 waves: 3.9 ms for the 2nd-stage update's 2400 x 8 heads, slower
// than the fp32 VALU kernel)
__global__ __launch_bounds__(NTH) void attn_smt16_kernel(const __bf16* __restrict__ QKV, int ld, int koff, int voff,
                                                         __bf16* __restrict__ O16, int ldo16, int S, float scale,
                                                         const float* __restrict__ key_mask, const int* __restrict__ seg_off,
                                                         long qkv_lo, long o_lo, float* __restrict__ O32, int ldo32,
                                                         float* __restrict__ lse) {
  constexpr int D = 32, KR = 48, NKT = SKP / 16;    // 96-byte LDS rows: conflict-free for both read kinds
  constexpr int NP = X3 ? 2 : 1;                    // planes: hi (, lo)
  __shared__ __attribute__((aligned(16))) __bf16 qs[X3 ? 1 : SKP * KR];     // X3: the Q fragments come straight from global memory
  __shared__ __attribute__((aligned(16))) __bf16 ks[NP][SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 vs[NP][SKP * KR];
  __shared__ float km[SKP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  long row0 = (long)b * S;
  if (seg_off) { row0 = seg_off[b]; S = min(seg_off[b + 1] - seg_off[b], SKP); key_mask = nullptr; }   // ragged: all tokens live
  const int S32 = (S + 31) & ~31;
  for (int i = tid; i < S32 * 4; i += NTH) {                   // 4 x 16-byte chunks per 32-wide row
    const int r = i >> 2, c = i & 3;
#pragma unroll
    for (int pl = 0; pl < NP; pl++) {
      uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, vv = qv;
      if (r < S) {
        const __bf16* base = QKV + (pl ? qkv_lo : 0) + (row0 + r) * ld + h * D + c * 8;
        qv = *reinterpret_cast<const uint4*>(base);
        kv = *reinterpret_cast<const uint4*>(base + koff);
        vv = *reinterpret_cast<const uint4*>(base + voff);
      }
      if (!X3) *reinterpret_cast<uint4*>(&qs[r * KR + c * 8]) = qv;
      *reinterpret_cast<uint4*>(&ks[pl][r * KR + c * 8]) = kv;
      *reinterpret_cast<uint4*>(&vs[pl][r * KR + c * 8]) = vv;
    }
  }
  for (int i = tid; i < SKP; i += NTH) km[i] = (i < S && (!key_mask || key_mask[(long)b * S + i] != 0.f)) ? 1.f : 0.f;
  __syncthreads();
  const int n_kt = (S + 15) >> 4, n_kk = (S + 31) >> 5;
  for (int mt = wave; mt * 16 < S; mt += NTH / 64) {
    af32x4 sacc[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; nt++) sacc[nt] = (af32x4){0.f, 0.f, 0.f, 0.f};
    abf16x8 qf, ql;
    if (X3) {
      uint4 a = make_uint4(0, 0, 0, 0), l = a;
      if (mt * 16 + r16 < S) {
        const __bf16* qp = QKV + (row0 + mt * 16 + r16) * ld + h * D + q4 * 8;
        a = *reinterpret_cast<const uint4*>(qp); l = *reinterpret_cast<const uint4*>(qp + qkv_lo);
      }
      qf = __builtin_bit_cast(abf16x8, a); ql = __builtin_bit_cast(abf16x8, l);
    } else {
      qf = *reinterpret_cast<const abf16x8*>(&qs[(mt * 16 + r16) * KR + q4 * 8]);
      ql = qf;
    }
#pragma unroll
    for (int nt = 0; nt < NKT; nt++)
      if (nt < n_kt) {
        abf16x8 kf = *reinterpret_cast<const abf16x8*>(&ks[0][(nt * 16 + r16) * KR + q4 * 8]);
        if (X3) {
          abf16x8 kl = *reinterpret_cast<const abf16x8*>(&ks[NP - 1][(nt * 16 + r16) * KR + q4 * 8]);
          sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qf, sacc[nt], 0, 0, 0);
          sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql, sacc[nt], 0, 0, 0);
        }
        sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sacc[nt], 0, 0, 0);      // S^T: rows = keys, cols = queries
      }
    const int qi = mt * 16 + r16;
    // the softmax in the base-2 domain: scores scaled by scale * log2(e) once, p = exp2(s - max) is ONE v_exp_f32 per element (a
    // masked key's -inf gives 0 by itself: the maximum is finite, the current token is always a valid key) -- this loop is what
    // bounds the kernel at 301 keys (VALU, not MFMA)
    const float scale2 = scale * 1.44269504088896340736f;
    float mx = -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < NKT; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int key = nt * 16 + q4 * 4 + r;
        const bool ok = nt < n_kt && km[key] != 0.f;
        const float v = ok ? sacc[nt][r] * scale2 : -INFINITY;
        sacc[nt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NKT; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float pv = __builtin_amdgcn_exp2f(sacc[nt][r] - mx);
        sum += pv; sacc[nt][r] = pv;
      }
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    af32x4 oacc[2];
    oacc[0] = (af32x4){0.f, 0.f, 0.f, 0.f}; oacc[1] = oacc[0];
#pragma unroll
    for (int kk = 0; kk < NKT / 2; kk++)
      if (kk < n_kk) {
        abf16x8 pf, pl;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          pf[e] = (__bf16)sacc[2 * kk][e]; pf[4 + e] = (__bf16)sacc[2 * kk + 1][e];
          if (X3) { pl[e] = (__bf16)(sacc[2 * kk][e] - (float)pf[e]); pl[4 + e] = (__bf16)(sacc[2 * kk + 1][e] - (float)pf[4 + e]); }
        }
        const int vo = (kk * 32 + q4 * 4 + (r16 >> 2)) * KR + 4 * (r16 & 3);
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          abf16x8 vf[NP];
#pragma unroll
          for (int p2 = 0; p2 < NP; p2++) {
            const __bf16* vb = &vs[p2][vo];
            abf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(vb + dt * 16));
            abf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(vb + 16 * KR + dt * 16));
#pragma unroll
            for (int e = 0; e < 4; e++) { vf[p2][e] = lo[e]; vf[p2][4 + e] = hi[e]; }
          }
          if (X3) {
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pl, oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[NP - 1], pf, oacc[dt], 0, 0, 0);
          }
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pf, oacc[dt], 0, 0, 0);
        }
      }
    if (qi < S) {
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
      __bf16* op = O16 + (row0 + qi) * ldo16 + h * D + q4 * 4;
#pragma unroll
      for (int dt = 0; dt < 2; dt++) {
        abf16x4 o, ol;
#pragma unroll
        for (int r = 0; r < 4; r++) { const float v = oacc[dt][r] * inv; o[r] = (__bf16)v; ol[r] = (__bf16)(v - (float)o[r]); }
        *reinterpret_cast<abf16x4*>(op + dt * 16) = o;
        if (X3) *reinterpret_cast<abf16x4*>(op + o_lo + dt * 16) = ol;
        // training forward: the fp32 output and the row's log-sum-exp for the backward (same definitions as attn_fwd_kernel)
        if (O32) *reinterpret_cast<float4*>(O32 + (row0 + qi) * ldo32 + h * D + q4 * 4 + dt * 16) =
            make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
      }
      if (lse && q4 == 0) lse[((long)b * gridDim.x + h) * S + qi] = (mx + __log2f(sum)) * 0.69314718055994530942f;   // natural units
    }
  }
}

// One query per (sample, head) over <= 192 masked keys (the decoder's cross attention on the memory tokens), D = 32:
// one wave per (sample, head); keys across lanes for the scores and the softmax, then output dims across lanes.
template <int T>          // keys per lane: up to 64*T keys
__global__ __launch_bounds__(64) void attn_q1_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                     const float* __restrict__ V, int ldv, __bf16* __restrict__ O16, int ldo16,
                                                     int Sk, float scale, const float* __restrict__ key_mask,
                                                     const int* __restrict__ seg_off, long o_lo) {
  constexpr int D = 32;
  __shared__ float p[64 * T];
  const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  long krow0 = (long)b * Sk;
  if (seg_off) { krow0 = seg_off[b]; Sk = min(seg_off[b + 1] - seg_off[b], 64 * T); key_mask = nullptr; }
  const float* q = Q + (long)b * ldq + h * D;
  float qv[D];
#pragma unroll
  for (int i = 0; i < D / 4; i++) {
    float4 t = *reinterpret_cast<const float4*>(q + i * 4);
    qv[4 * i] = t.x; qv[4 * i + 1] = t.y; qv[4 * i + 2] = t.z; qv[4 * i + 3] = t.w;
  }
  float sc[T]; float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < T; t++) {
    const int j = lane + 64 * t;
    sc[t] = -INFINITY;
    if (j < Sk && (!key_mask || key_mask[(long)b * Sk + j] != 0.f)) {
      const float* kr = K + (krow0 + j) * ldk + h * D;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < D / 4; i++) {
        float4 kv = *reinterpret_cast<const float4*>(kr + i * 4);
        a += qv[4 * i] * kv.x + qv[4 * i + 1] * kv.y + qv[4 * i + 2] * kv.z + qv[4 * i + 3] * kv.w;
      }
      sc[t] = a * scale;
    }
    mx = fmaxf(mx, sc[t]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < T; t++) { sc[t] = sc[t] == -INFINITY ? 0.f : __expf(sc[t] - mx); sum += sc[t]; }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
  for (int t = 0; t < T; t++) p[lane + 64 * t] = sc[t] * inv;
  __syncthreads();
  // lanes 0-31: even keys, lanes 32-63: odd keys; lane & 31 = output dim
  const int d = lane & 31, par = lane >> 5;
  float o = 0.f;
  for (int j = par; j < Sk; j += 2) o += p[j] * V[(krow0 + j) * ldv + h * D + d];
  o += __shfl_xor(o, 32, 64);
  if (lane < 32) {
    const __bf16 hv = (__bf16)o;
    O16[(long)b * ldo16 + h * D + d] = hv;
    if (o_lo) O16[o_lo + (long)b * ldo16 + h * D + d] = (__bf16)(o - (float)hv);
  }
}

// dQ: lane per query (same streaming structure as forward).  Also writes delta = rowsum(dO * O).
template <int D>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                         int ldk, const float* __restrict__ V, int ldv,
                                                         const float* __restrict__ O, int ldo,
                                                         const float* __restrict__ dO, int lddo,
                                                         const float* __restrict__ key_mask, const float* __restrict__ lse,
                                                         float* __restrict__ delta, float* __restrict__ dQ, int lddq,
                                                         int H, int Sq, int Sk, int causal, float scale) {
  __shared__ __attribute__((aligned(16))) float ks[CH * D];
  __shared__ __attribute__((aligned(16))) float vs[CH * D];
  __shared__ int kidx[CH];
  const int lane = threadIdx.x, b = blockIdx.z, h = blockIdx.y;
  const int i = blockIdx.x * 64 + lane;
  const bool qok = i < Sq;
  float q[D], go[D], dq[D];
  float dl = 0.f, ls = 0.f;
#pragma unroll
  for (int d = 0; d < D; d++) { q[d] = 0.f; go[d] = 0.f; dq[d] = 0.f; }
  if (qok) {
    const float* qp = Q + ((long)b * Sq + i) * ldq + h * D;
    const float* gp = dO + ((long)b * Sq + i) * lddo + h * D;
    const float* op = O + ((long)b * Sq + i) * ldo + h * D;
#pragma unroll
    for (int d = 0; d < D; d++) { q[d] = qp[d] * scale; go[d] = gp[d]; dl += gp[d] * op[d]; }
    ls = lse[((long)b * H + h) * Sq + i];
    delta[((long)b * H + h) * Sq + i] = dl;
  }
  const int sk_end = causal ? min(Sk, blockIdx.x * 64 + 64) : Sk;
  for (int j0 = 0; j0 < sk_end; j0 += CH) {
    const int j = j0 + lane;
    bool valid = j < sk_end && (key_mask == nullptr || key_mask[(long)b * Sk + j] != 0.f);
    unsigned long long bal = __ballot(valid);
    int nvalid = __popcll(bal);
    int pos = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (valid) {
      const float4* kp = reinterpret_cast<const float4*>(K + ((long)b * Sk + j) * ldk + h * D);
      const float4* vp = reinterpret_cast<const float4*>(V + ((long)b * Sk + j) * ldv + h * D);
      float4* kd = reinterpret_cast<float4*>(&ks[pos * D]);
      float4* vd = reinterpret_cast<float4*>(&vs[pos * D]);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++) { kd[d4] = kp[d4]; vd[d4] = vp[d4]; }
      kidx[pos] = j;
    }
    __syncthreads();
    for (int jj = 0; jj < nvalid; jj++) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; d++) { s += q[d] * ks[jj * D + d]; dp += go[d] * vs[jj * D + d]; }
      float p = (causal && kidx[jj] > i) ? 0.f : __expf(s - ls);
      float ds = p * (dp - dl) * scale;
#pragma unroll
      for (int d = 0; d < D; d++) dq[d] += ds * ks[jj * D + d];
    }
  }
  if (qok) {
    float* o = dQ + ((long)b * Sq + i) * lddq + h * D;
#pragma unroll
    for (int d = 0; d < D; d++) o[d] = dq[d];
  }
}

// dK/dV: lane per key; queries (q, dO, lse, delta) stream through LDS in chunks of 64.
template <int D>
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K,
                                                          int ldk, const float* __restrict__ V, int ldv,
                                                          const float* __restrict__ dO, int lddo,
                                                          const float* __restrict__ key_mask, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, float* __restrict__ dK, int lddk,
                                                          float* __restrict__ dV, int lddv, int H, int Sq, int Sk,
                                                          int causal, float scale) {
  __shared__ __attribute__((aligned(16))) float qs[CH * D];
  __shared__ __attribute__((aligned(16))) float gs[CH * D];
  __shared__ float s_lse[CH], s_del[CH];
  const int lane = threadIdx.x, b = blockIdx.z, h = blockIdx.y;
  const int j = blockIdx.x * 64 + lane;
  const bool kok = j < Sk;
  const bool kvalid = kok && (key_mask == nullptr || key_mask[(long)b * Sk + j] != 0.f);
  float k[D], v[D], dk[D], dv[D];
#pragma unroll
  for (int d = 0; d < D; d++) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
  if (kok) {
    const float* kp = K + ((long)b * Sk + j) * ldk + h * D;
    const float* vp = V + ((long)b * Sk + j) * ldv + h * D;
#pragma unroll
    for (int d = 0; d < D; d++) { k[d] = kp[d]; v[d] = vp[d]; }
  }
  const int i_beg = causal ? blockIdx.x * 64 : 0;       // queries before this block's first key see none of its keys
  for (int i0 = i_beg; i0 < Sq; i0 += CH) {
    const int i = i0 + lane;
    __syncthreads();
    if (i < Sq) {
      const float4* qp = reinterpret_cast<const float4*>(Q + ((long)b * Sq + i) * ldq + h * D);
      const float4* gp = reinterpret_cast<const float4*>(dO + ((long)b * Sq + i) * lddo + h * D);
      float4* qd = reinterpret_cast<float4*>(&qs[lane * D]);
      float4* gd = reinterpret_cast<float4*>(&gs[lane * D]);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; d4++) { qd[d4] = qp[d4]; gd[d4] = gp[d4]; }
      s_lse[lane] = lse[((long)b * H + h) * Sq + i];
      s_del[lane] = delta[((long)b * H + h) * Sq + i];
    }
    __syncthreads();
    const int nq = min(CH, Sq - i0);
    for (int ii = 0; ii < nq; ii++) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; d++) { s += qs[ii * D + d] * k[d]; dp += gs[ii * D + d] * v[d]; }
      float p = (!kvalid || (causal && j > i0 + ii)) ? 0.f : __expf(s * scale - s_lse[ii]);
      float ds = p * (dp - s_del[ii]) * scale;
#pragma unroll
      for (int d = 0; d < D; d++) { dv[d] += p * gs[ii * D + d]; dk[d] += ds * qs[ii * D + d]; }
    }
  }
  if (kok) {
    float* ok = dK + ((long)b * Sk + j) * lddk + h * D;
    float* ov = dV + ((long)b * Sk + j) * lddv + h * D;
#pragma unroll
    for (int d = 0; d < D; d++) { ok[d] = dk[d]; ov[d] = dv[d]; }
  }
}

}  // namespace

int avlen_attention_fwd16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                          void* O16, int ldo16, const float* key_mask, float* lse, int B, int H, int Sq, int Sk, int D,
                          int causal, float scale, hipStream_t stream) {
  return avlen_attention_fwd16_seg(Q, ldq, K, ldk, V, ldv, O, ldo, O16, ldo16, key_mask, lse, B, H, Sq, Sk, D, causal, scale,
                                   nullptr, stream);
}

// SMT self-attention from the packed bf16 projection (D = 32): q | k | v at columns 0 | H*32 | 2*H*32; key_mask [B][S], or
// seg_off [B+1] for a ragged batch of live tokens (S = upper bound of tokens per sample).
int avlen_attention_smt16(const void* QKV16, int ld, void* O16, int ldo16, int B, int H, int S, float scale,
                          const float* key_mask, const int* seg_off, hipStream_t stream, long qkv_lo, long o_lo, float* O32, int ldo32,
                          float* lse) {
  if (!QKV16 || !O16 || B <= 0 || H <= 0 || S <= 0 || S > 320 || (ld & 7) || (ldo16 & 3)) return AVLEN_ERR_ARG;
  if ((O32 && ((ldo32 & 3) || ((uintptr_t)O32 & 15))) || ((O32 || lse) && seg_off)) return AVLEN_ERR_ARG;
  if (qkv_lo || o_lo) {                 // compensated pairs: both planes of k and v in LDS, q fragments from global memory
    if (!qkv_lo || !o_lo) return AVLEN_ERR_ARG;
    if (S <= 160)
      hipLaunchKernelGGL((attn_smt16_kernel<160, true>), dim3(H, B), dim3(256), 0, stream, (const __bf16*)QKV16, ld, H * 32, 2 * H * 32,
                         (__bf16*)O16, ldo16, S, scale, key_mask, seg_off, qkv_lo, o_lo, O32, ldo32, lse);
    else
      hipLaunchKernelGGL((attn_smt16_kernel<320, true, 512>), dim3(H, B), dim3(512), 0, stream, (const __bf16*)QKV16, ld, H * 32, 2 * H * 32,
                         (__bf16*)O16, ldo16, S, scale, key_mask, seg_off, qkv_lo, o_lo, O32, ldo32, lse);
    return avlen_launch_status();
  }
  if (S <= 160)
    hipLaunchKernelGGL((attn_smt16_kernel<160, false>), dim3(H, B), dim3(256), 0, stream, (const __bf16*)QKV16, ld, H * 32, 2 * H * 32,
                       (__bf16*)O16, ldo16, S, scale, key_mask, seg_off, 0L, 0L, O32, ldo32, lse);
  else {
    static unsigned long long attr_done = 0;             // per-device bit mask (a process may drive several devices)
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&attn_smt16_kernel<320, false>), 0, &attr_done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_smt16_kernel<320, false>), dim3(H, B), dim3(256), 0, stream, (const __bf16*)QKV16, ld, H * 32, 2 * H * 32,
                       (__bf16*)O16, ldo16, S, scale, key_mask, seg_off, 0L, 0L, O32, ldo32, lse);
  }
  return avlen_launch_status();
}
// One query per sample (D = 32, fp32 q / k / v): the decoder's cross attention (<= 320 keys per sample).
int avlen_attention_q1(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* O16, int ldo16, int B,
                       int H, int Sk, float scale, const float* key_mask, const int* seg_off, hipStream_t stream, long o_lo) {
  if (!Q || !K || !V || !O16 || B <= 0 || H <= 0 || Sk <= 0 || Sk > 320 || ((ldq | ldk | ldv) & 3)) return AVLEN_ERR_ARG;
  if (Sk <= 192)
    hipLaunchKernelGGL(attn_q1_kernel<3>, dim3(H, B), dim3(64), 0, stream, Q, ldq, K, ldk, V, ldv, (__bf16*)O16, ldo16, Sk, scale,
                       key_mask, seg_off, o_lo);
  else
    hipLaunchKernelGGL(attn_q1_kernel<5>, dim3(H, B), dim3(64), 0, stream, Q, ldq, K, ldk, V, ldv, (__bf16*)O16, ldo16, Sk, scale,
                       key_mask, seg_off, o_lo);
  return avlen_launch_status();
}

// Packed bf16 projection [R][ld] with q | k | v at columns 0 | H*64 | 2*H*64 (head h at +64h) -> O16 [R][ldo16].
int avlen_attention_qkv16(const void* QKV16, int ld, void* O16, int ldo16, int B, int H, int S, int causal, float scale,
                          const int* seg_off, hipStream_t stream, int f16) {
  if (!QKV16 || !O16 || B <= 0 || H <= 0 || S <= 0 || S > 96 || (ld & 7) || (ldo16 & 3)) return AVLEN_ERR_ARG;
  if (f16)
    hipLaunchKernelGGL(attn_qkv16_kernel<true>, dim3(H, B), dim3(256), 0, stream, (const __bf16*)QKV16, ld, H * 64, 2 * H * 64,
                       (__bf16*)O16, ldo16, S, causal, scale, seg_off);
  else
    hipLaunchKernelGGL(attn_qkv16_kernel<false>, dim3(H, B), dim3(256), 0, stream, (const __bf16*)QKV16, ld, H * 64, 2 * H * 64,
                       (__bf16*)O16, ldo16, S, causal, scale, seg_off);
  return avlen_launch_status();
}

int avlen_attention_fwd16_seg(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                              void* O16, int ldo16, const float* key_mask, float* lse, int B, int H, int Sq, int Sk, int D,
                              int causal, float scale, const int* seg_off, hipStream_t stream) {
  if (seg_off && (key_mask || lse)) return AVLEN_ERR_ARG;
  if (!key_mask && !lse && !O && O16 && D == 64 && Sk <= 96 && Sq <= 96) {       // CLIP text: bf16 MFMA attention
    hipLaunchKernelGGL(attn_mfma64_kernel, dim3(H, B), dim3(256), 0, stream, Q, ldq, K, ldk, V, ldv, (__bf16*)O16, ldo16, Sq,
                       Sk, causal, scale, seg_off);
    return avlen_launch_status();
  }
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || (ldq | ldk | ldv | ldo) % 4) return AVLEN_ERR_ARG;
  __bf16* oh = (__bf16*)O16;
  if (!key_mask && (Sq >= 16 || seg_off) && (D == 32 || D == 64)) {       // no padding mask: 4-lanes-per-query kernel
    dim3 g4(ceil_div(Sq, 32), H, B), b4(128);
    if (D == 32)
      hipLaunchKernelGGL((attn_fwd4_kernel<32>), g4, b4, 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, oh, ldo16, lse, H, Sq, Sk, causal, scale, seg_off);
    else
      hipLaunchKernelGGL((attn_fwd4_kernel<64>), g4, b4, 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, oh, ldo16, lse, H, Sq, Sk, causal, scale, seg_off);
    return avlen_launch_status();
  }
  if (seg_off) return AVLEN_ERR_ARG;
  dim3 grid(ceil_div(Sq, 64), H, B), block(64);
  if (D == 32)
    hipLaunchKernelGGL((attn_fwd_kernel<32>), grid, block, 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, oh, ldo16, key_mask, lse, H, Sq, Sk, causal, scale);
  else if (D == 64)
    hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, oh, ldo16, key_mask, lse, H, Sq, Sk, causal, scale);
  else return AVLEN_ERR_ARG;
  return avlen_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-attention BACKWARD on the matrix cores (bf16 operands, fp32 accumulate): one 256-thread block per (sample, head) holds
// Q, K, V and dO of the head (S <= SKP tokens, D = 32) in LDS as bf16 and makes two passes over the S x S score matrix with the
// forward kernel's idioms (transposed score tiles so that P / dS stay in registers as the B operand of the second MFMA,
// hardware-transposing LDS reads for the other operand):
//   pass 1, query-major : lane = (query r16, 8 keys)   ->  delta' = sum_k P dP, then dQ^T += K^T dS^T
//   pass 2, key-major   : lane = (key r16, 8 queries)  ->  dV^T += dO^T P,  dK^T += Q^T dS
// with P = exp(scale q.k - lse), dS = P (dO.v - delta') scale; dS enters its MFMAs as a bf16 hi + lo pair.  The fp32
// rowsum(dO * O) is still written to `delta` for the caller.
// Replaces the two fp32 VALU kernels above on the 2nd-stage training path (722 k token rows per minibatch: 5.2 ms -> per launch).
template <int SKP, int NTH>      // NTH threads = NTH/64 waves share the head's tiles: one block per CU (LDS), so the waves of ONE block hide each other's latency
__global__ __launch_bounds__(NTH) void attn_bwd16_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                         const float* __restrict__ V, int ldv, const float* __restrict__ O, int ldo,
                                                         const float* __restrict__ dO, int lddo,
                                                         const float* __restrict__ key_mask, const float* __restrict__ lse,
                                                         float* __restrict__ delta, float* __restrict__ dQ, int lddq,
                                                         float* __restrict__ dK, int lddk, float* __restrict__ dV, int lddv, int H,
                                                         int S, float scale, const __bf16* __restrict__ QKV16, int ld16,
                                                         __bf16* __restrict__ dQKV16, int ldd16, float* __restrict__ colsum) {
  // dQKV16 (optional): the gradients ALSO as the packed bf16 rows [R][ldd16] (dq | dk | dv at columns 0 | H*32 | 2*H*32) -- the row-major
  // operand of the in-projection's backward -- and their column sums added to colsum[3*H*32] (its bias gradient); dQ / dK / dV may
  // then be null.  Saves the fp32 round trip of the 3d-wide gradient and the cast pass over it.
  // QKV16 (optional): q | k | v as the packed bf16 projection [R][ld16] (columns 0 | H*32 | 2*H*32) the forward kept -- the same
  // values the fp32 -> bf16 staging below produces, without the fp32 copy (Q, K, V may then be null)
  constexpr int D = 32, KR = 48;                    // 96-byte LDS rows: conflict-free for both read kinds (as attn_smt16_kernel)
  __shared__ __attribute__((aligned(16))) __bf16 qs[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 ks[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 vs[SKP * KR];
  __shared__ __attribute__((aligned(16))) __bf16 gs[SKP * KR];
  __shared__ float km[SKP], s_lse[SKP], s_del[SKP];
  __shared__ float cred[NTH / 64][96];             // per-wave column sums of dq | dk | dv (this head's 32 dims each)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const long row0 = (long)b * S;
  const int S32 = (S + 31) & ~31;                   // rows S .. S32-1 are zero
  af32x4 csq[2], csk[2], csv[2];
#pragma unroll
  for (int dt = 0; dt < 2; dt++) { csq[dt] = (af32x4){0.f, 0.f, 0.f, 0.f}; csk[dt] = csq[dt]; csv[dt] = csq[dt]; }
  // ---- stage: 4 x 32-byte chunks (8 floats) per row and tensor; delta from the fp32 dO and O of the same chunk
  for (int i = tid; i < S32 * 4; i += NTH) {
    const int r = i >> 2, c = i & 3;
    abf16x8 q8, k8, v8, g8;
#pragma unroll
    for (int e = 0; e < 8; e++) { q8[e] = (__bf16)0.f; k8[e] = q8[e]; v8[e] = q8[e]; g8[e] = q8[e]; }
    float dl = 0.f;
    if (r < S) {
      const float4* gp = reinterpret_cast<const float4*>(dO + (row0 + r) * lddo + h * D + c * 8);
      const float4* op = reinterpret_cast<const float4*>(O + (row0 + r) * ldo + h * D + c * 8);
      if (QKV16) {
        const __bf16* base = QKV16 + (row0 + r) * ld16 + h * D + c * 8;
        q8 = *reinterpret_cast<const abf16x8*>(base);
        k8 = *reinterpret_cast<const abf16x8*>(base + H * D);
        v8 = *reinterpret_cast<const abf16x8*>(base + 2 * H * D);
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const float4 g = gp[u], o = op[u];
          g8[4 * u] = (__bf16)g.x; g8[4 * u + 1] = (__bf16)g.y; g8[4 * u + 2] = (__bf16)g.z; g8[4 * u + 3] = (__bf16)g.w;
          dl += g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
        }
      } else {
        const float4* qp = reinterpret_cast<const float4*>(Q + (row0 + r) * ldq + h * D + c * 8);
        const float4* kp = reinterpret_cast<const float4*>(K + (row0 + r) * ldk + h * D + c * 8);
        const float4* vp = reinterpret_cast<const float4*>(V + (row0 + r) * ldv + h * D + c * 8);
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const float4 a = qp[u], bb = kp[u], cc = vp[u], g = gp[u], o = op[u];
          q8[4 * u] = (__bf16)a.x; q8[4 * u + 1] = (__bf16)a.y; q8[4 * u + 2] = (__bf16)a.z; q8[4 * u + 3] = (__bf16)a.w;
          k8[4 * u] = (__bf16)bb.x; k8[4 * u + 1] = (__bf16)bb.y; k8[4 * u + 2] = (__bf16)bb.z; k8[4 * u + 3] = (__bf16)bb.w;
          v8[4 * u] = (__bf16)cc.x; v8[4 * u + 1] = (__bf16)cc.y; v8[4 * u + 2] = (__bf16)cc.z; v8[4 * u + 3] = (__bf16)cc.w;
          g8[4 * u] = (__bf16)g.x; g8[4 * u + 1] = (__bf16)g.y; g8[4 * u + 2] = (__bf16)g.z; g8[4 * u + 3] = (__bf16)g.w;
          dl += g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
        }
      }
    }
    *reinterpret_cast<abf16x8*>(&qs[r * KR + c * 8]) = q8;
    *reinterpret_cast<abf16x8*>(&ks[r * KR + c * 8]) = k8;
    *reinterpret_cast<abf16x8*>(&vs[r * KR + c * 8]) = v8;
    *reinterpret_cast<abf16x8*>(&gs[r * KR + c * 8]) = g8;
    dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64);      // the 4 chunk-threads of a row are adjacent lanes
    if (c == 0) {
      s_del[r] = dl;
      if (r < S) delta[((long)b * H + h) * S + r] = dl;
    }
  }
  for (int i = tid; i < SKP; i += NTH) {
    km[i] = (i < S && (!key_mask || key_mask[(long)b * S + i] != 0.f)) ? 1.f : 0.f;
    s_lse[i] = i < S ? lse[((long)b * H + h) * S + i] : 0.f;
    if (i >= S32) s_del[i] = 0.f;
  }
  __syncthreads();
  const int n_t = (S + 15) >> 4, n_kk = (S + 31) >> 5;
  // ---- pass 1: per 16 queries and wave: delta', then dQ (two sweeps over the keys)
  for (int mt = wave; mt < n_t; mt += NTH / 64) {
    const int qi = mt * 16 + r16;
    const abf16x8 qf = *reinterpret_cast<const abf16x8*>(&qs[qi * KR + q4 * 8]);
    const abf16x8 gf = *reinterpret_cast<const abf16x8*>(&gs[qi * KR + q4 * 8]);
    const float ls0 = s_lse[qi];
    const bool qok = qi < S;
    // sweep A: the softmax normaliser and delta' = sum_k P_k dP_k from the SAME bf16 products the gradients use.  (The forward's
    // fp32 lse and rowsum(dO * O) are the same numbers mathematically, but P sums to 1 and dP - delta cancels -- exactly, when one
    // key holds all the mass -- and bf16-rounded scores / dP against fp32 lse / delta leave 1e-1-sized ghosts in dK / dQ.)
    float dl = 0.f, zs = 0.f;
    for (int kk = 0; kk < n_kk; kk++) {
      const af32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int krow = kk * 32 + t * 16 + r16;
        const abf16x8 kf = *reinterpret_cast<const abf16x8*>(&ks[krow * KR + q4 * 8]);
        const abf16x8 vf = *reinterpret_cast<const abf16x8*>(&vs[krow * KR + q4 * 8]);
        const af32x4 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);
        const af32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kj = kk * 32 + t * 16 + q4 * 4 + r;
          if (qok && km[kj] != 0.f) { const float e = __expf(sc[r] * scale - ls0); zs += e; dl += e * dp[r]; }
        }
      }
    }
    zs += __shfl_xor(zs, 16, 64); zs += __shfl_xor(zs, 32, 64);
    dl += __shfl_xor(dl, 16, 64); dl += __shfl_xor(dl, 32, 64);
    // the row's softmax of the bf16-rounded scores: lse' = lse + log Z (Z = 1 +- 4e-3), delta' = sum_k P'_k dP_k
    const float ls = (zs > 0.f) ? ls0 + __logf(zs) : ls0;
    dl = (zs > 0.f) ? dl / zs : 0.f;
    if (q4 == 0) { s_del[qi] = dl; s_lse[qi] = ls; }  // pass 2 (key-major) reads both after the barrier
    af32x4 dqa[2];
    dqa[0] = (af32x4){0.f, 0.f, 0.f, 0.f}; dqa[1] = dqa[0];
    for (int kk = 0; kk < n_kk; kk++) {
      const af32x4 z = {0.f, 0.f, 0.f, 0.f};
      abf16x8 dsf, dsl;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int krow = kk * 32 + t * 16 + r16;
        const abf16x8 kf = *reinterpret_cast<const abf16x8*>(&ks[krow * KR + q4 * 8]);
        const abf16x8 vf = *reinterpret_cast<const abf16x8*>(&vs[krow * KR + q4 * 8]);
        const af32x4 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);   // [query r16][key kk*32 + t*16 + q4*4 + r]
        const af32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kj = kk * 32 + t * 16 + q4 * 4 + r;
          const float p = (qok && km[kj] != 0.f) ? __expf(sc[r] * scale - ls) : 0.f;
          const float ds = p * (dp[r] - dl) * scale;
          const __bf16 hi_ = (__bf16)ds;
          dsf[t * 4 + r] = hi_;
          dsl[t * 4 + r] = (__bf16)(ds - (float)hi_);
        }
      }
      const __bf16* kb = &ks[(kk * 32 + q4 * 4 + (r16 >> 2)) * KR + 4 * (r16 & 3)];
#pragma unroll
      for (int dt = 0; dt < 2; dt++) {
        abf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(kb + dt * 16));
        abf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(kb + 16 * KR + dt * 16));
        abf16x8 tf;
#pragma unroll
        for (int e = 0; e < 4; e++) { tf[e] = lo[e]; tf[4 + e] = hi[e]; }
        dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, dsf, dqa[dt], 0, 0, 0);
        dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, dsl, dqa[dt], 0, 0, 0);
      }
    }
    if (qok) {
      if (dQ) {
        float* oq = dQ + (row0 + qi) * lddq + h * D + q4 * 4;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
          *reinterpret_cast<float4*>(oq + dt * 16) = make_float4(dqa[dt][0], dqa[dt][1], dqa[dt][2], dqa[dt][3]);
      }
      if (dQKV16) {
        __bf16* o16 = dQKV16 + (row0 + qi) * ldd16 + h * D + q4 * 4;
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          abf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = (__bf16)dqa[dt][r];
          *reinterpret_cast<abf16x4*>(o16 + dt * 16) = o;
          csq[dt] += dqa[dt];
        }
      }
    }
  }
  __syncthreads();
  // ---- pass 2: dK, dV for 16 keys per wave iteration
  for (int kt = wave; kt < n_t; kt += NTH / 64) {
    const int key = kt * 16 + r16;
    const abf16x8 kf = *reinterpret_cast<const abf16x8*>(&ks[key * KR + q4 * 8]);
    const abf16x8 vf = *reinterpret_cast<const abf16x8*>(&vs[key * KR + q4 * 8]);
    const bool kvalid = km[key] != 0.f;
    af32x4 dva[2], dka[2];
    dva[0] = (af32x4){0.f, 0.f, 0.f, 0.f}; dva[1] = dva[0]; dka[0] = dva[0]; dka[1] = dva[0];
    for (int kk = 0; kk < n_kk; kk++) {
      const af32x4 z = {0.f, 0.f, 0.f, 0.f};
      af32x4 sc[2], dp[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int qrow = kk * 32 + t * 16 + r16;
        const abf16x8 qf = *reinterpret_cast<const abf16x8*>(&qs[qrow * KR + q4 * 8]);
        const abf16x8 gf = *reinterpret_cast<const abf16x8*>(&gs[qrow * KR + q4 * 8]);
        sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);       // [key r16][query kk*32 + t*16 + q4*4 + r]
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf, z, 0, 0, 0);
      }
      // dS sums to zero over the keys of a row, so dK / dQ are cancelling sums: dS goes in as a bf16 hi + lo pair (two MFMAs)
      abf16x8 pf, dsf, dsl;
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int qi = kk * 32 + t * 16 + q4 * 4 + r;
          const float p = (kvalid && qi < S) ? __expf(sc[t][r] * scale - s_lse[qi]) : 0.f;
          pf[t * 4 + r] = (__bf16)p;
          const float ds = p * (dp[t][r] - s_del[qi]) * scale;
          const __bf16 hi_ = (__bf16)ds;
          dsf[t * 4 + r] = hi_;
          dsl[t * 4 + r] = (__bf16)(ds - (float)hi_);
        }
      const __bf16* gb = &gs[(kk * 32 + q4 * 4 + (r16 >> 2)) * KR + 4 * (r16 & 3)];
      const __bf16* qb = &qs[(kk * 32 + q4 * 4 + (r16 >> 2)) * KR + 4 * (r16 & 3)];
#pragma unroll
      for (int dt = 0; dt < 2; dt++) {
        abf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(gb + dt * 16));
        abf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(gb + 16 * KR + dt * 16));
        abf16x8 tf;
#pragma unroll
        for (int e = 0; e < 4; e++) { tf[e] = lo[e]; tf[4 + e] = hi[e]; }
        dva[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, pf, dva[dt], 0, 0, 0);
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(qb + dt * 16));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) abf16x4*)(qb + 16 * KR + dt * 16));
#pragma unroll
        for (int e = 0; e < 4; e++) { tf[e] = lo[e]; tf[4 + e] = hi[e]; }
        dka[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, dsf, dka[dt], 0, 0, 0);
        dka[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, dsl, dka[dt], 0, 0, 0);
      }
    }
    if (key < S) {
      if (dV) {
        float* ov = dV + (row0 + key) * lddv + h * D + q4 * 4;
        float* ok = dK + (row0 + key) * lddk + h * D + q4 * 4;
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          *reinterpret_cast<float4*>(ov + dt * 16) = make_float4(dva[dt][0], dva[dt][1], dva[dt][2], dva[dt][3]);
          *reinterpret_cast<float4*>(ok + dt * 16) = make_float4(dka[dt][0], dka[dt][1], dka[dt][2], dka[dt][3]);
        }
      }
      if (dQKV16) {
        __bf16* o16 = dQKV16 + (row0 + key) * ldd16 + h * D + q4 * 4;
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          abf16x4 ok16, ov16;
#pragma unroll
          for (int r = 0; r < 4; r++) { ok16[r] = (__bf16)dka[dt][r]; ov16[r] = (__bf16)dva[dt][r]; }
          *reinterpret_cast<abf16x4*>(o16 + H * D + dt * 16) = ok16;
          *reinterpret_cast<abf16x4*>(o16 + 2 * H * D + dt * 16) = ov16;
          csk[dt] += dka[dt]; csv[dt] += dva[dt];
        }
      }
    }
  }
  if (colsum) {      // (uniform) column sums: over the 16 rows of a lane group, the waves, then one atomic per column and block
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float a = csq[dt][r], bb = csk[dt][r], cc = csv[dt][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); bb += __shfl_xor(bb, o, 64); cc += __shfl_xor(cc, o, 64); }
        if (r16 == 0) {
          const int dim = dt * 16 + q4 * 4 + r;
          cred[wave][dim] = a; cred[wave][32 + dim] = bb; cred[wave][64 + dim] = cc;
        }
      }
    __syncthreads();
    if (tid < 96) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NTH / 64; w++) v += cred[w][tid];
      atomicAdd(&colsum[(tid >> 5) * H * D + h * D + (tid & 31)], v);
    }
  }
}


extern "C" int avlen_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                                   int ldo, const float* key_mask, float* lse, int B, int H, int Sq, int Sk, int D,
                                   int causal, float scale, hipStream_t stream) {
  return avlen_attention_fwd16(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, 0, key_mask, lse, B, H, Sq, Sk, D, causal, scale, stream);
}

// bf16-operand self-attention backward (Sq == Sk == S <= 320, D = 32, key mask, no causal mask); same arguments as
// avlen_attention_bwd.  AVLEN_ERR_ARG when the shape is outside that envelope (the caller then takes the fp32 kernels).
extern "C" int avlen_attention_bwd_bf16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                        const float* O, int ldo, const float* dO, int lddo, const float* key_mask,
                                        const float* lse, float* delta, float* dQ, int lddq, float* dK, int lddk, float* dV,
                                        int lddv, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                                        hipStream_t stream) {
  return avlen_attention_bwd_p16(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, key_mask, lse, delta, dQ, lddq, dK, lddk, dV, lddv, B, H, Sq, Sk,
                                 D, causal, scale, stream, nullptr, 0, nullptr, 0, nullptr);
}
int avlen_attention_bwd_p16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                            const float* O, int ldo, const float* dO, int lddo, const float* key_mask,
                            const float* lse, float* delta, float* dQ, int lddq, float* dK, int lddk, float* dV,
                            int lddv, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                            hipStream_t stream, const void* QKV16_, int ld16, void* dQKV16_, int ldd16, float* colsum) {
  const __bf16* QKV16 = (const __bf16*)QKV16_;
  __bf16* dQKV16 = (__bf16*)dQKV16_;
  if (B <= 0 || H <= 0 || Sq != Sk || Sq <= 0 || Sq > 320 || D != 32 || causal || ((ldo | lddo) & 3))
    return AVLEN_ERR_ARG;
  if (dQKV16 ? ((ldd16 & 3) || ((uintptr_t)dQKV16 & 7) || !colsum) : (!dQ || !dK || !dV || colsum)) return AVLEN_ERR_ARG;
  if (dQ && ((lddq | lddk | lddv) & 3)) return AVLEN_ERR_ARG;
  if ((dQ != nullptr) != (dV != nullptr) || (dQ != nullptr) != (dK != nullptr)) return AVLEN_ERR_ARG;
  if (QKV16 ? ((ld16 & 7) || ((uintptr_t)QKV16 & 15)) : ((ldq | ldk | ldv) & 3) != 0) return AVLEN_ERR_ARG;
  if (Sq <= 160)
    hipLaunchKernelGGL((attn_bwd16_kernel<160, 256>), dim3(H, B), dim3(256), 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, key_mask, lse,
                       delta, dQ, lddq, dK, lddk, dV, lddv, H, Sq, scale, QKV16, ld16, dQKV16, ldd16, colsum);
  else {
    static int nth = -1;                           // AVLEN_ATTN_BWD16_THREADS=256|512|1024 (A/B knob)
    if (nth < 0) nth = (int)avlen_knob("AVLEN_ATTN_BWD16_THREADS", 1024);
    if (nth == 256)
      hipLaunchKernelGGL((attn_bwd16_kernel<320, 256>), dim3(H, B), dim3(256), 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, key_mask,
                         lse, delta, dQ, lddq, dK, lddk, dV, lddv, H, Sq, scale, QKV16, ld16, dQKV16, ldd16, colsum);
    else if (nth == 512)
      hipLaunchKernelGGL((attn_bwd16_kernel<320, 512>), dim3(H, B), dim3(512), 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, key_mask,
                         lse, delta, dQ, lddq, dK, lddk, dV, lddv, H, Sq, scale, QKV16, ld16, dQKV16, ldd16, colsum);
    else      // 16 waves: measured 9.2 / 5.6 / 4.0 ms for 256 / 512 / 1024 threads at 2400 x 8 heads x 301 tokens (fp32 kernels: 17 ms)
      hipLaunchKernelGGL((attn_bwd16_kernel<320, 1024>), dim3(H, B), dim3(1024), 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo,
                         key_mask, lse, delta, dQ, lddq, dK, lddk, dV, lddv, H, Sq, scale, QKV16, ld16, dQKV16, ldd16, colsum);
  }
  return avlen_launch_status();
}

extern "C" int avlen_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                   const float* O, int ldo, const float* dO, int lddo, const float* key_mask,
                                   const float* lse, float* delta, float* dQ, int lddq, float* dK, int lddk, float* dV,
                                   int lddv, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                                   hipStream_t stream) {
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || (ldq | ldk | ldv | lddo) % 4) return AVLEN_ERR_ARG;
  dim3 gq(ceil_div(Sq, 64), H, B), gk(ceil_div(Sk, 64), H, B), block(64);
  if (D == 32) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<32>), gq, block, 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, key_mask, lse, delta, dQ, lddq, H, Sq, Sk, causal, scale);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<32>), gk, block, 0, stream, Q, ldq, K, ldk, V, ldv, dO, lddo, key_mask, lse, delta, dK, lddk, dV, lddv, H, Sq, Sk, causal, scale);
  } else if (D == 64) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), gq, block, 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, key_mask, lse, delta, dQ, lddq, H, Sq, Sk, causal, scale);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<64>), gk, block, 0, stream, Q, ldq, K, ldk, V, ldv, dO, lddo, key_mask, lse, delta, dK, lddk, dV, lddv, H, Sq, Sk, causal, scale);
  } else return AVLEN_ERR_ARG;
  return avlen_launch_status();
}
