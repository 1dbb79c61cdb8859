// Implicit-GEMM convolution / linear layer on MFMA for gfx950 (CDNA4).
//
//   C[m, n] = epilogue( sum_k A(m, k) * Wt(n, k) )
//
// * linear mode : A(m,k) = A[m*lda + k]   (or A[k*lda + m] when transA)
// * conv mode   : m -> (b, oy, ox), k -> (ky, kx, ci) over an NHWC activation tensor; out-of-image
//                 taps read as zero (padding), so no im2col buffer ever exists in HBM.
// * Wt(n,k) = B[n*ldb + k]  (row per output feature, K contiguous; conv weights are pre-packed to
//   [Cout][kh][kw][Cin])  or B[k*ldb + n] when transB (used by the dX / dW backward products).
//
// Tiling: 128 x BN x 32 block tile, 256 threads = 4 wavefronts of 64 lanes, each wave owns a
// 32 x BN strip (2 x BN/16 MFMA 16x16 tiles).  Operands are staged global -> registers -> LDS with
// the next tile's global loads issued before the current tile's MFMAs (register prefetch).  Two
// arithmetic modes share the structure:
//   BF16: operands rounded to bf16 while staging, v_mfma_f32_16x16x32_bf16, fp32 accumulate
//   FP32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) -- the parity mode
//   X3  : compensated bf16 ("bf16x3"): every fp32 operand is split while staging into hi = bf16(x) and lo = bf16(x - hi), and
//         a product is three bf16 MFMAs (lo*hi + hi*lo + hi*hi) into the same fp32 accumulator: ~2^-17 relative error per
//         product instead of 2^-9, at 3/16 of the fp32 MFMA's cycles
// Epilogue: + bias, ReLU / QuickGELU, + residual, fp32 store; or split-K slabs reduced by
// avlen_splitk_reduce (deterministic, no atomics).
#include "common.h"
#include "../../include/avlen_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int NTHREADS = 256;

struct IgemmParams {
  const float* A; const float* B; float* C; const float* bias; const float* residual;
  int M, N, K;
  int lda, ldb, ldc, ldr;
  int transA, transB;
  int conv, H, W, Cin, OH, OW, KH, KW, stride, pad;
  int act;
  int splitk, kper;        // kper: K elements per split (multiple of BK)
  int to_slab;             // write raw accumulators to slab [z][M][N] (split-K and/or accumulate)
  int a_vec, b_vec;        // 16-byte vector loads allowed along the contiguous dim
};

constexpr int MODE_FP32 = 0, MODE_BF16 = 1, MODE_X3 = 2;
template <int MODE> struct Elem { using T = __bf16; static constexpr int LDS = 40; };
template <> struct Elem<MODE_FP32> { using T = float; static constexpr int LDS = 36; };

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v / (1.f + __expf(-1.702f * v));       // QuickGELU x*sigmoid(1.702x)
  return v;
}

template <int BN, int MODE>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(IgemmParams p) {
  using T = typename Elem<MODE>::T;
  constexpr int LDS = Elem<MODE>::LDS;
  constexpr bool BF16 = MODE == MODE_BF16, X3 = MODE == MODE_X3;
  constexpr int NI = BN / 16;
  constexpr int A_IT = BM * 8 / NTHREADS;                      // float4 per thread for the A tile (=4)
  constexpr int B_IT = (BN * 8 + NTHREADS - 1) / NTHREADS;     // >= 1

  __shared__ __attribute__((aligned(16))) T As[BM * LDS];
  __shared__ __attribute__((aligned(16))) T Bs[BN * LDS];
  __shared__ __attribute__((aligned(16))) T Al[X3 ? BM * LDS : 8];       // X3: the low halves of the split operands
  __shared__ __attribute__((aligned(16))) T Bl[X3 ? BN * LDS : 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_tiles = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / n_tiles) * BM;
  const int n0 = (blockIdx.x % n_tiles) * BN;
  const int kbeg = blockIdx.z * p.kper;
  const int kend = min(p.K, kbeg + p.kper);

  // ---- per-thread A row descriptors (conv mode: image coordinates of the output pixel) ----
  long a_base[A_IT]; int a_iy0[A_IT], a_ix0[A_IT]; bool a_ok[A_IT];
  if (!p.transA) {
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
      int row = (tid >> 3) + i * 32;
      int m = m0 + row;
      a_ok[i] = m < p.M;
      if (p.conv) {
        int ohw = p.OH * p.OW;
        int b = m / ohw, r = m - b * ohw;
        int oy = r / p.OW, ox = r - oy * p.OW;
        a_base[i] = (long)b * p.H * p.W * p.Cin;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
      } else {
        a_base[i] = (long)m * p.lda; a_iy0[i] = 0; a_ix0[i] = 0;
      }
    }
  }

  float4 a_reg[A_IT], b_reg[B_IT];

  auto load_a = [&](int k0) {
    if (p.transA) {                      // A[k*lda + m], float4 along m
#pragma unroll
      for (int i = 0; i < A_IT; i++) {
        int f = tid + i * NTHREADS;
        int kk = f / (BM / 4), mq = f % (BM / 4);
        int k = k0 + kk, m = m0 + mq * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) {
          const float* src = p.A + (long)k * p.lda + m;
          if (p.a_vec && m + 3 < p.M) v = *reinterpret_cast<const float4*>(src);
          else {
            if (m < p.M) v.x = src[0];
            if (m + 1 < p.M) v.y = src[1];
            if (m + 2 < p.M) v.z = src[2];
            if (m + 3 < p.M) v.w = src[3];
          }
        }
        a_reg[i] = v;
      }
      return;
    }
    const int k = k0 + (tid & 7) * 4;
    if (p.conv) {
      if (p.a_vec) {                     // Cin % 4 == 0: the 4 k's share one tap and are contiguous
        int tap = k / p.Cin, ci = k - tap * p.Cin;
        int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
          if (a_ok[i] && k < kend && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
            v = *reinterpret_cast<const float4*>(p.A + a_base[i] + ((long)iy * p.W + ix) * p.Cin + ci);
          a_reg[i] = v;
        }
      } else {
        int kyv[4], kxv[4], civ[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          int kj = k + j;
          int tap = kj / p.Cin; civ[j] = kj - tap * p.Cin;
          kyv[j] = tap / p.KW; kxv[j] = tap - kyv[j] * p.KW;
        }
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            int iy = a_iy0[i] + kyv[j], ix = a_ix0[i] + kxv[j];
            bool ok = a_ok[i] && (k + j) < kend && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            t[j] = ok ? p.A[a_base[i] + ((long)iy * p.W + ix) * p.Cin + civ[j]] : 0.f;
          }
          a_reg[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; i++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ok[i]) {
          const float* src = p.A + a_base[i] + k;
          if (p.a_vec && k + 3 < kend) v = *reinterpret_cast<const float4*>(src);
          else {
            if (k < kend) v.x = src[0];
            if (k + 1 < kend) v.y = src[1];
            if (k + 2 < kend) v.z = src[2];
            if (k + 3 < kend) v.w = src[3];
          }
        }
        a_reg[i] = v;
      }
    }
  };

  auto load_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
      int f = tid + i * NTHREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < BN * 8) {
        if (p.transB) {                  // B[k*ldb + n], float4 along n
          int kk = f / (BN / 4), nq = f % (BN / 4);
          int k = k0 + kk, n = n0 + nq * 4;
          if (k < kend) {
            const float* src = p.B + (long)k * p.ldb + n;
            if (p.b_vec && n + 3 < p.N) v = *reinterpret_cast<const float4*>(src);
            else {
              if (n < p.N) v.x = src[0];
              if (n + 1 < p.N) v.y = src[1];
              if (n + 2 < p.N) v.z = src[2];
              if (n + 3 < p.N) v.w = src[3];
            }
          }
        } else {
          int row = f >> 3, k = k0 + (f & 7) * 4, n = n0 + row;
          if (n < p.N) {
            const float* src = p.B + (long)n * p.ldb + k;
            if (p.b_vec && k + 3 < kend) v = *reinterpret_cast<const float4*>(src);
            else {
              if (k < kend) v.x = src[0];
              if (k + 1 < kend) v.y = src[1];
              if (k + 2 < kend) v.z = src[2];
              if (k + 3 < kend) v.w = src[3];
            }
          }
        }
      }
      b_reg[i] = v;
    }
  };

  auto put4 = [&](T* dst, T* dst_lo, float4 v) {          // 4 consecutive k of one LDS row
    if constexpr (BF16 || X3) {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      bf16x4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
      *reinterpret_cast<bf16x4*>(dst) = o;
      if constexpr (X3) {
        bf16x4 l; l[0] = (__bf16)(v.x - (float)o[0]); l[1] = (__bf16)(v.y - (float)o[1]);
        l[2] = (__bf16)(v.z - (float)o[2]); l[3] = (__bf16)(v.w - (float)o[3]);
        *reinterpret_cast<bf16x4*>(dst_lo) = l;
      }
    } else {
      *reinterpret_cast<float4*>(dst) = v;
    }
  };
  auto put1 = [&](T* dst, T* dst_lo, float v) {
    const T h = (T)v;
    *dst = h;
    if constexpr (X3) *dst_lo = (T)(v - (float)h);
  };

  auto store_a = [&]() {
    if (p.transA) {
#pragma unroll
      for (int i = 0; i < A_IT; i++) {
        int f = tid + i * NTHREADS;
        int kk = f / (BM / 4), mq = f % (BM / 4);
        float t[4] = {a_reg[i].x, a_reg[i].y, a_reg[i].z, a_reg[i].w};
#pragma unroll
        for (int j = 0; j < 4; j++) put1(&As[(mq * 4 + j) * LDS + kk], &Al[X3 ? (mq * 4 + j) * LDS + kk : 0], t[j]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; i++) {
        const int o = ((tid >> 3) + i * 32) * LDS + (tid & 7) * 4;
        put4(&As[o], &Al[X3 ? o : 0], a_reg[i]);
      }
    }
  };
  auto store_b = [&]() {
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
      int f = tid + i * NTHREADS;
      if (f < BN * 8) {
        if (p.transB) {
          int kk = f / (BN / 4), nq = f % (BN / 4);
          float t[4] = {b_reg[i].x, b_reg[i].y, b_reg[i].z, b_reg[i].w};
#pragma unroll
          for (int j = 0; j < 4; j++) put1(&Bs[(nq * 4 + j) * LDS + kk], &Bl[X3 ? (nq * 4 + j) * LDS + kk : 0], t[j]);
        } else {
          const int o = (f >> 3) * LDS + (f & 7) * 4;
          put4(&Bs[o], &Bl[X3 ? o : 0], b_reg[i]);
        }
      }
    }
  };

  f32x4 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (kbeg < kend) { load_a(kbeg); load_b(kbeg); }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();                       // previous tile's LDS reads are done
    store_a(); store_b();
    __syncthreads();
    if (k0 + BK < kend) { load_a(k0 + BK); load_b(k0 + BK); }     // prefetch under the MFMAs

    const int r16 = lane & 15, q = lane >> 4;
    if constexpr (BF16) {
      bf16x8 af[2], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < 2; mi++)
        af[mi] = *reinterpret_cast<const bf16x8*>(&As[(wave * 32 + mi * 16 + r16) * LDS + q * 8]);
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
        bfr[ni] = *reinterpret_cast<const bf16x8*>(&Bs[(ni * 16 + r16) * LDS + q * 8]);
#pragma unroll
      for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    } else if constexpr (X3) {
      bf16x8 ah[2], al[2], bh[NI], bl[NI];
#pragma unroll
      for (int mi = 0; mi < 2; mi++) {
        ah[mi] = *reinterpret_cast<const bf16x8*>(&As[(wave * 32 + mi * 16 + r16) * LDS + q * 8]);
        al[mi] = *reinterpret_cast<const bf16x8*>(&Al[(wave * 32 + mi * 16 + r16) * LDS + q * 8]);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ni++) {
        bh[ni] = *reinterpret_cast<const bf16x8*>(&Bs[(ni * 16 + r16) * LDS + q * 8]);
        bl[ni] = *reinterpret_cast<const bf16x8*>(&Bl[(ni * 16 + r16) * LDS + q * 8]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {                      // small terms first, then the leading product
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; kk++) {
        float af[2], bfr[NI];
#pragma unroll
        for (int mi = 0; mi < 2; mi++) af[mi] = As[(wave * 32 + mi * 16 + r16) * LDS + kk * 4 + q];
#pragma unroll
        for (int ni = 0; ni < NI; ni++) bfr[ni] = Bs[(ni * 16 + r16) * LDS + kk * 4 + q];
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg.  Bias / residual loads are all issued before the
  // first store so their latencies overlap (a per-tile load->use->store chain costs one round trip per tile).
  const int col_l = lane & 15, rq = lane >> 4;
  float bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    int col = n0 + ni * 16 + col_l;
    bv[ni] = (p.bias && !p.to_slab && col < p.N) ? p.bias[col] : 0.f;
  }
  if (p.to_slab) {
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++) {
        int col = n0 + ni * 16 + col_l;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int row = m0 + wave * 32 + mi * 16 + rq * 4 + r;
          if (col < p.N && row < p.M) p.C[((long)blockIdx.z * p.M + row) * p.N + col] = acc[mi][ni][r];
        }
      }
    return;
  }
  if (p.residual) {
    float rv[2][NI][4];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++) {
        int col = n0 + ni * 16 + col_l;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int row = m0 + wave * 32 + mi * 16 + rq * 4 + r;
          rv[mi][ni][r] = (col < p.N && row < p.M) ? p.residual[(long)row * p.ldr + col] : 0.f;
        }
      }
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float v = act_apply(acc[mi][ni][r] + bv[ni], p.act) + rv[mi][ni][r];
          acc[mi][ni][r] = p.act == AVLEN_ACT_RELU_POST ? fmaxf(v, 0.f) : v;      // ReLU after the residual add
        }
  } else {
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[mi][ni][r] = act_apply(acc[mi][ni][r] + bv[ni], p.act);
  }
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      int col = n0 + ni * 16 + col_l;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int row = m0 + wave * 32 + mi * 16 + rq * 4 + r;
        if (col < p.N && row < p.M) p.C[(long)row * p.ldc + col] = acc[mi][ni][r];
      }
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, const float* bias,
                                     const float* residual, int M, int N, int ldc, int ldr, int splits, int act,
                                     float beta) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)M * N;
  if (i >= tot) return;
  int row = (int)(i / N), col = (int)(i - (long)row * N);
  float s = 0.f;
  int z = 0;
  for (; z + 8 <= splits; z += 8) {                       // eight loads in flight, added in split order (same sum as one by one)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = part[(long)(z + u) * tot + i];
#pragma unroll
    for (int u = 0; u < 8; u++) s += v[u];
  }
  for (; z < splits; z++) s += part[(long)z * tot + i];
  if (bias) s += bias[col];
  s = act_apply(s, act);
  if (residual) s += residual[(long)row * ldr + col];
  if (act == AVLEN_ACT_RELU_POST) s = fmaxf(s, 0.f);
  float* o = out + (long)row * ldc + col;
  *o = (beta != 0.f) ? beta * (*o) + s : s;
}

template <int BF16>
int launch_igemm(const IgemmParams& p, hipStream_t st) {
  int bn = p.N <= 16 ? 16 : p.N <= 32 ? 32 : p.N <= 64 ? 64 : 128;
  int n_tiles = ceil_div(p.N, bn), m_tiles = ceil_div(p.M, BM);
  dim3 grid(m_tiles * n_tiles, 1, p.splitk);
  switch (bn) {
    case 16: hipLaunchKernelGGL((igemm_kernel<16, BF16>), grid, dim3(NTHREADS), 0, st, p); break;
    case 32: hipLaunchKernelGGL((igemm_kernel<32, BF16>), grid, dim3(NTHREADS), 0, st, p); break;
    case 64: hipLaunchKernelGGL((igemm_kernel<64, BF16>), grid, dim3(NTHREADS), 0, st, p); break;
    default: hipLaunchKernelGGL((igemm_kernel<128, BF16>), grid, dim3(NTHREADS), 0, st, p); break;
  }
  return avlen_launch_status();
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------

extern "C" size_t avlen_gemm_workspace_bytes(int M, int N, int K, int splitk) {
  (void)K;
  return (size_t)(splitk < 1 ? 1 : splitk) * M * N * sizeof(float) + 256;   // slab(s) for split-K / accumulate
}

// Heuristic: how many K-splits a (M,N,K) product wants so that a small-M / huge-K product still
// fills the 256 CUs.  Returns 1 when the tile grid already covers the chip.
extern "C" int avlen_gemm_pick_splitk(int M, int N, int K) {
  int bn = N <= 16 ? 16 : N <= 32 ? 32 : N <= 64 ? 64 : 128;
  long tiles = (long)ceil_div(M, BM) * ceil_div(N, bn);
  if (tiles >= 128 || K < 512) return 1;
  long want = 256 / tiles;
  long maxs = K / 128;                        // keep >= 128 K elements per split
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

static int run_igemm(IgemmParams p, int prec, float beta, void* ws, size_t ws_bytes, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return AVLEN_ERR_ARG;
  float* out = p.C; const float* bias = p.bias; const float* res = p.residual; int act = p.act;
  int splits = p.splitk < 1 ? 1 : p.splitk;
  int per = ceil_div(ceil_div(p.K, splits), BK) * BK;
  splits = ceil_div(p.K, per);
  p.kper = per; p.splitk = splits;
  p.to_slab = (splits > 1 || beta != 0.f) ? 1 : 0;
  if (p.to_slab) {
    if (!ws || ws_bytes < avlen_gemm_workspace_bytes(p.M, p.N, p.K, splits)) return AVLEN_ERR_WS;
    p.C = (float*)ws;
  }
  int rc = prec == AVLEN_PREC_BF16 ? launch_igemm<MODE_BF16>(p, st)
         : prec == AVLEN_PREC_BF16X3 ? launch_igemm<MODE_X3>(p, st) : launch_igemm<MODE_FP32>(p, st);
  if (rc != AVLEN_OK) return rc;
  if (p.to_slab) {
    long tot = (long)p.M * p.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, out, bias, res, p.M, p.N, p.ldc, p.ldr, splits, act, beta);
    return avlen_launch_status();
  }
  return AVLEN_OK;
}

extern "C" int avlen_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C,
                          int ldc, const float* bias, const float* residual, int ldr, int M, int N, int K,
                          int act, int prec, int splitk, float beta, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
  IgemmParams p = {};
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.transA = transA; p.transB = transB; p.conv = 0; p.act = act; p.splitk = splitk < 1 ? 1 : splitk;
  p.a_vec = (lda % 4 == 0) && al16(A);
  p.b_vec = (ldb % 4 == 0) && al16(B);
  return run_igemm(p, prec, beta, ws, ws_bytes, stream);
}

extern "C" int avlen_conv2d_nhwc(const float* X, const float* Wp, const float* bias, const float* residual,
                                 float* Y, int Bn, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                 int pad, int act, int prec, hipStream_t stream) {
  IgemmParams p = {};
  int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  if (OH <= 0 || OW <= 0) return AVLEN_ERR_ARG;
  p.A = X; p.B = Wp; p.C = Y; p.bias = bias; p.residual = residual;
  p.M = Bn * OH * OW; p.N = Cout; p.K = KH * KW * Cin;
  p.lda = 0; p.ldb = p.K; p.ldc = Cout; p.ldr = Cout;
  p.conv = 1; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.act = act; p.splitk = 1;
  p.a_vec = (Cin % 4 == 0) && al16(X);
  p.b_vec = (p.K % 4 == 0) && al16(Wp);
  return run_igemm(p, prec, 0.f, nullptr, 0, stream);
}
