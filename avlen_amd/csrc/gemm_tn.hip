// Weight gradient of a Linear over MANY rows: C[N1][N2] (+)= A^T B with A [M][N1] and B [M][N2] both ROW-major 16-bit (bf16),
// i.e. G.w += dY^T X for dY [M][out_f], X [M][in_f] (ss_baselines/savi/ppo/ppo.py:207-270 -- loss.backward() of the 2nd-stage
// update: M = 722 k token rows per minibatch, out_f / in_f = 256 .. 768).
//
// Until round 5 this product ran on the row-times-row MFMA GEMM (igemm2.hip), which wants the contraction index contiguous: both
// operands were first written TRANSPOSED ([N][Mp], a 0.4-0.5 ms pass each) and the 2 x 370 MB were then read at 1.4 TB/s.  Here the
// operands stay row-major -- the same dY16 rows also feed dX = dY W -- and the transpose happens in the LDS read:
//   * a workgroup (512 threads) owns a 256 x 256 tile of C and a CHUNK of rows; per step it stages 32 rows x 256 columns of A and of
//     B (2 x 16 KB) through registers into LDS (rows 544 B apart: 8 consecutive rows start 8 banks apart),
//   * every wave reads its operand fragments with ds_read_b64_tr_b16 (the k index of the MFMA is the ROW of the tile: a lane gets
//     rows {4 q .. 4 q + 3} and {16 + 4 q .. 16 + 4 q + 3} of one column, for A and B alike, so the contraction pairs up),
//   * 8 waves = 4 (64 rows of C) x 2 (128 columns): 32 MFMA 16x16x32 per wave and step against 24 transposed reads,
//   * global loads run two steps ahead of the MFMAs (two register sets), one barrier per step,
//   * partial tiles go to the split scratch and are summed in chunk order by gemm_tn_reduce_kernel (deterministic; no atomics).
// Bound: HBM -- 2 x 2 x M x 256 bytes per 256 x 256 tile, 131 flop per byte: the roofline at 8 TB/s is 1 PFLOP/s.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TN_TH = 512, TN_STEP = 32, TN_TILE = 256;
constexpr int TN_ROW = 512 + 32;                       // LDS row stride (bytes)
constexpr int TN_IMG = TN_STEP * TN_ROW;                // one operand, one buffer
constexpr int TN_LDS = 4 * TN_IMG;                      // A | B x two buffers = 69,632 B

struct TnArgs {
  const bf16* A; const bf16* B; long lda, ldb, M, rows_per_chunk;
  int N1p, N2p;                                         // operand columns (multiples of 8; the pad columns hold zeros)
  float* part;                                          // [chunks][gy][gz][256][256]
};

__device__ __forceinline__ uint4 tn_load(const bf16* p, bool ok) {
  return ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(TN_TH) void gemm_tn_kernel(TnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
  const int wa = wave & 3, wb = wave >> 2;
  const long m_lo = (long)blockIdx.x * a.rows_per_chunk;
  const long m_hi = m_lo + a.rows_per_chunk < a.M ? m_lo + a.rows_per_chunk : a.M;
  const int n1_0 = blockIdx.y * TN_TILE, n2_0 = blockIdx.z * TN_TILE;
  // staging map: thread -> (row, 16-byte chunk) x 2 per operand
  const int srow = tid >> 5, sch = tid & 31;              // second piece: row + 16
  const bool a_ok = n1_0 + 8 * sch < a.N1p, b_ok = n2_0 + 8 * sch < a.N2p;
  const bf16* ap = a.A + n1_0 + 8 * sch;
  const bf16* bp = a.B + n2_0 + 8 * sch;
  const int st_off = srow * TN_ROW + sch * 16;
  // transposed-read map: 16-lane group q reads the 4 x 16 block of rows 4 q .. 4 q + 3 (+ 16); lane 4 q' + p gives row q', columns 4 p ..
  const int rd_off = (4 * q + (r16 >> 2)) * TN_ROW + 8 * (r16 & 3);
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 ra[2][2], rb[2][2];
  auto fetch = [&](int set, long m0) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const long r = m0 + srow + 16 * j;
      const bool in = r < m_hi;
      ra[set][j] = tn_load(ap + r * a.lda, in && a_ok);
      rb[set][j] = tn_load(bp + r * a.ldb, in && b_ok);
    }
  };
  auto stash = [&](int set, int buf) {
    char* A_ = lds + buf * 2 * TN_IMG; char* B_ = A_ + TN_IMG;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      *reinterpret_cast<uint4*>(A_ + st_off + 16 * j * TN_ROW) = ra[set][j];
      *reinterpret_cast<uint4*>(B_ + st_off + 16 * j * TN_ROW) = rb[set][j];
    }
  };
  auto mma = [&](int buf) {
    const char* A_ = lds + buf * 2 * TN_IMG + rd_off + wa * 128; const char* B_ = lds + buf * 2 * TN_IMG + TN_IMG + rd_off + wb * 256;
    bf16x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(A_ + i * 32));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(A_ + i * 32 + 16 * TN_ROW));
#pragma unroll
      for (int e = 0; e < 4; e++) { af[i][e] = lo[e]; af[i][4 + e] = hi[e]; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(B_ + j * 32));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(B_ + j * 32 + 16 * TN_ROW));
      bf16x8 bf;
#pragma unroll
      for (int e = 0; e < 4; e++) { bf[e] = lo[e]; bf[4 + e] = hi[e]; }
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
    }
  };
  const long steps = (m_hi - m_lo + TN_STEP - 1) / TN_STEP;       // uniform per workgroup; every thread runs every barrier
  if (steps > 0) fetch(0, m_lo);
  if (steps > 1) fetch(1, m_lo + TN_STEP);
  for (long s = 0; s < steps; s += 2) {
    stash(0, 0);
    if (s + 2 < steps) fetch(0, m_lo + (s + 2) * TN_STEP);
    __syncthreads();
    mma(0);
    if (s + 1 < steps) {
      stash(1, 1);
      if (s + 3 < steps) fetch(1, m_lo + (s + 3) * TN_STEP);
      __syncthreads();
      mma(1);
    }
  }
  // partial tile: acc[i][j][r] = C[64 wa + 16 i + 4 q + r][128 wb + 16 j + r16]
  float* out = a.part + (((long)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (TN_TILE * TN_TILE);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        out[(64 * wa + 16 * i + 4 * q + r) * TN_TILE + 128 * wb + 16 * j + r16] = acc[i][j][r];
}

// C[n1][n2] = beta * C + sum over chunks (in chunk order) of the partial tiles
__global__ void gemm_tn_reduce_kernel(const float* __restrict__ part, int chunks, int gy, int gz, float* __restrict__ C, int ldc, int N1, int N2,
                                      float beta) {
  const int n2 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, n1 = blockIdx.y;
  if (n2 >= N2) return;
  const int ty = n1 / TN_TILE, tz = n2 / TN_TILE;
  const float* p = part + ((long)ty * gz + tz) * (TN_TILE * TN_TILE) + (long)(n1 % TN_TILE) * TN_TILE + n2 % TN_TILE;
  const long stride = (long)gy * gz * TN_TILE * TN_TILE;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(p + (long)(c + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; c < chunks; c++) {
    const float4 v = *reinterpret_cast<const float4*>(p + (long)c * stride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = C + (long)n1 * ldc + n2;
  const float t[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int e = 0; e < 4; e++)
    if (n2 + e < N2) o[e] = (beta != 0.f ? beta * o[e] : 0.f) + t[e];
}

}  // namespace

size_t avlen_i_gemm_tn_workspace_bytes(long M, int N1, int N2) {
  const int gy = ceil_div(N1, TN_TILE), gz = ceil_div(N2, TN_TILE);
  long chunks = 256 / (gy * gz); if (chunks < 1) chunks = 1;
  const long steps = (M + TN_STEP - 1) / TN_STEP;
  if (chunks > steps) chunks = steps > 0 ? steps : 1;
  return (size_t)chunks * gy * gz * TN_TILE * TN_TILE * sizeof(float) + 256;
}

// C [N1][N2] (row stride ldc) = beta * C + A^T B; A [M][lda >= pad8(N1)], B [M][ldb >= pad8(N2)] bf16 whose columns N .. pad8(N) - 1
// are zero.  AVLEN_ERR_WS when `ws` is smaller than avlen_i_gemm_tn_workspace_bytes.
int avlen_i_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, long M, int N1, int N2, float* C, int ldc, float beta,
                         void* ws, size_t ws_bytes, hipStream_t st) {
  if (!A || !B || !C || M <= 0 || N1 <= 0 || N2 <= 0 || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return AVLEN_ERR_ARG;
  const int N1p = (N1 + 7) & ~7, N2p = (N2 + 7) & ~7;
  if (lda < N1p || ldb < N2p) return AVLEN_ERR_ARG;
  if (ws_bytes < avlen_i_gemm_tn_workspace_bytes(M, N1, N2)) return AVLEN_ERR_WS;
  const int gy = ceil_div(N1, TN_TILE), gz = ceil_div(N2, TN_TILE);
  long chunks = 256 / (gy * gz); if (chunks < 1) chunks = 1;
  const long steps = (M + TN_STEP - 1) / TN_STEP;
  if (chunks > steps) chunks = steps;
  const long spc = (steps + chunks - 1) / chunks;           // steps per chunk
  chunks = (steps + spc - 1) / spc;                         // no empty chunk
  static unsigned long long done = 0;
  if (avlen_set_dyn_lds((const void*)gemm_tn_kernel, TN_LDS, &done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  TnArgs a{(const bf16*)A, (const bf16*)B, lda, ldb, M, spc * TN_STEP, N1p, N2p, (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255)};
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)chunks, gy, gz), dim3(TN_TH), TN_LDS, st, a);
  if (avlen_launch_status() != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(ceil_div(ceil_div(N2, 4), 64), N1), dim3(64), 0, st, a.part, (int)chunks, gy, gz, C, ldc, N1, N2,
                     beta);
  return avlen_launch_status();
}

extern "C" size_t avlen_gemm_tn_bf16_workspace_bytes(long M, int N1, int N2) { return avlen_i_gemm_tn_workspace_bytes(M, N1, N2); }
extern "C" int avlen_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, long M, int N1, int N2, float* C, int ldc, float beta,
                                  void* ws, size_t ws_bytes, hipStream_t st) {
  return avlen_i_gemm_tn_bf16(A, lda, B, ldb, M, N1, N2, C, ldc, beta, ws, ws_bytes, st);
}
