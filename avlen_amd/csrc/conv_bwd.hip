// Direct convolution weight gradient for the 3-conv CNNs of the GRU baseline (BASELINE configs[1]; audio_cnn.py:62-94,
// visual_cnn.py:82-107: valid convolutions, NHWC activations kept in fp32 by the training forward):
//   dW[co][kh][kw][ci] = sum_{b, oh, ow} dY[b][oh][ow][co] * X[b][oh*s + kh][ow*s + kw][ci]        db[co] = sum dY[b][oh][ow][co]
// The contraction runs over the output positions, so the MFMA's k index must walk positions while each lane keeps one channel --
// the transposed layout of both operands.  The GEMM route (modules.hip:avlen_i_conv_dw16) materialises that layout in HBM: a
// transposed bf16 im2col of X (K x M, up to 1.9 GB for one conv) and a transposed cast of dY, both read back by the GEMM; 14.8 % of
// the cfg2 cycle was the im2col pass alone (profiles/r03_rocprof_gru.md).  Here nothing is materialised:
//   * the k index of one MFMA is (4 output positions of one output row) x (8 IMAGES): a workgroup owns 8 images at a time and LDS
//     holds every element as a 16-byte group of its 8 images' bf16 values, so an operand fragment of either matrix is ONE aligned
//     ds_read_b128 whatever the stride, kernel size or width (no padded runs, no alignment cases);
//   * a workgroup walks down the output rows of its 8 images with the KH input rows it needs in an LDS ring: moving one row down
//     brings in `stride` new input rows and one dY row, fetched (fp32, coalesced along the NHWC rows, 8 images per lane) into
//     registers while the MFMAs of the current row run, packed to bf16 and written as conflict-free 16-byte stores;
//   * 8 waves split the K = KH*KW*C columns in 16-wide tiles, every wave holds all cout rows of its tiles in accumulators for the
//     whole launch; one partial per workgroup, summed in fixed order by a second small kernel (deterministic), the bias gradient
//     comes out of the fp32 dY values on their way into LDS.
// X and dY are each read once: the kernel is bound by that traffic (fp32 activations), not by the products.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int DW_TH = 512;      // 8 waves
constexpr int DW_XE = 8;        // 16-byte elements of new input rows per thread and row step   (stride * W * C <= DW_XE * DW_TH; template: 4 or 8)
constexpr int DW_YE = 4;        // ... of one dY row                                             (OW * cout     <= DW_YE * DW_TH)

struct DwArgs {
  const float* X; const float* dY; float* part;
  int R, H, Wc, C, OH, OW, KH, KW, s, cout, K, Kp;
  int octets, TL;               // groups of 8 images; octets * OH row steps in all
  unsigned xring_bytes, y_off, lds_bytes;
};

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8], int nv) {        // images nv .. 7 of the group do not exist: zero
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (bf16)(e < nv ? v[e] : 0.f);
  return o;
}

template <int MT, int NTW, int XE>
__global__ __launch_bounds__(DW_TH) void conv_dw_kernel(const DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int Wc = a.Wc, KH = a.KH, s = a.s, OW = a.OW, cout = a.cout;
  // all of LDS finite: operand fragments of padded positions / columns read it
  for (unsigned o = tid * 16u; o < a.lds_bytes; o += DW_TH * 16u) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0, 0, 0, 0);
  char* const ybase = lds + a.y_off;

  // this wave's K columns: tiles wv, wv + 8, ...; lane li is column n (clamped: columns >= K are computed and dropped)
  unsigned col_off[NTW]; int col_kh[NTW];
#pragma unroll
  for (int q = 0; q < NTW; q++) {
    int n = (wv + 8 * q) * 16 + li;
    n = n < a.K ? n : a.K - 1;
    const int per = a.KW * a.C;
    col_kh[q] = n / per;
    col_off[q] = (unsigned)(n - col_kh[q] * per) * 16u + (unsigned)(lg * s * a.C) * 16u;
  }
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < NTW; q++) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;                                    // bias gradient of channel tid % cout (DW_TH % cout == 0)

  float xr[XE][8], yr[DW_YE][8];
  auto load_x = [&](int o, int ih0, int nr) {
    const int total = nr * Wc;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int b = o * 8 + e;
      const float* p = a.X + ((long)(b < a.R ? b : 0) * a.H + ih0) * Wc;          // images past R: image 0, zeroed on the way into LDS
#pragma unroll
      for (int j = 0; j < XE; j++) { const int idx = tid + DW_TH * j; xr[j][e] = p[idx < total ? idx : 0]; }      // branch-free: all loads in flight
    }
  };
  auto store_x = [&](int o, int ih0, int nr) {
    const int total = nr * Wc, nv = a.R - o * 8;
#pragma unroll
    for (int j = 0; j < XE; j++) {
      const int idx = tid + DW_TH * j;
      if (idx < total) {
        const int r = idx / Wc, col = idx - r * Wc;
        const int slot = (ih0 + r) % KH;
        *reinterpret_cast<bf16x8*>(lds + ((unsigned)(slot * Wc + col)) * 16u) = pack8(xr[j], nv);
      }
    }
  };
  auto load_y = [&](int o, int oh) {
    const int total = OW * cout;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int b = o * 8 + e;
      const float* p = a.dY + ((long)(b < a.R ? b : 0) * a.OH + oh) * total;
#pragma unroll
      for (int j = 0; j < DW_YE; j++) { const int idx = tid + DW_TH * j; yr[j][e] = p[idx < total ? idx : 0]; }
    }
  };
  auto store_y = [&](int o) {
    const int total = OW * cout, nv = a.R - o * 8;
#pragma unroll
    for (int j = 0; j < DW_YE; j++) {
      const int idx = tid + DW_TH * j;
      if (idx < total) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) t += e < nv ? yr[j][e] : 0.f;
        bsum += t;
        *reinterpret_cast<bf16x8*>(ybase + (unsigned)idx * 16u) = pack8(yr[j], nv);
      }
    }
  };

  const int L0 = (int)((long)blockIdx.x * a.TL / gridDim.x), L1 = (int)((long)(blockIdx.x + 1) * a.TL / gridDim.x);
  const int nks = (OW + 3) >> 2;
  const unsigned a_step = 4u * cout * 16u, b_step = 4u * s * a.C * 16u;
  for (int L = L0; L < L1; L++) {
    const int o = L / a.OH, oh = L - o * a.OH;
    if (L == L0 || oh == 0) {                          // (re)fill the ring: rows oh*s .. oh*s + KH - 1, `s` at a time
      __syncthreads();
      for (int r0 = 0; r0 < KH; r0 += s) {
        const int nr = KH - r0 < s ? KH - r0 : s;
        load_x(o, oh * s + r0, nr);
        store_x(o, oh * s + r0, nr);
      }
      load_y(o, oh);
      store_y(o);
      __syncthreads();
    }
    const bool pf = L + 1 < L1 && oh + 1 < a.OH;      // next row step: same images
    if (pf) { load_x(o, oh * s + KH, s); load_y(o, oh + 1); }
    // ---- the products of output row oh
    unsigned ba[NTW];
#pragma unroll
    for (int q = 0; q < NTW; q++) ba[q] = (unsigned)(((oh * s + col_kh[q]) % KH) * Wc) * 16u + col_off[q];
    unsigned aa = (unsigned)(lg * cout + li) * 16u;
    bf16x8 af[2][MT], bfr[2][NTW];
    auto fetch = [&](int buf) {
#pragma unroll
      for (int m = 0; m < MT; m++) af[buf][m] = *reinterpret_cast<const bf16x8*>(ybase + aa + m * 256u);
#pragma unroll
      for (int q = 0; q < NTW; q++) bfr[buf][q] = *reinterpret_cast<const bf16x8*>(lds + ba[q]);
      aa += a_step;
#pragma unroll
      for (int q = 0; q < NTW; q++) ba[q] += b_step;
    };
    auto mma = [&](int buf) {
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < NTW; q++) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[buf][m], bfr[buf][q], acc[m][q], 0, 0, 0);
    };
    fetch(0);
    int ks = 0;
    for (; ks + 2 <= nks; ks += 2) {
      fetch(1); mma(0);
      if (ks + 2 < nks) fetch(0);
      mma(1);
    }
    if (ks < nks) mma(0);
    if (pf) {
      __syncthreads();
      store_x(o, oh * s + KH, s);
      store_y(o);
      __syncthreads();
    }
  }

  // ---- this workgroup's partial: [cout][Kp] then [cout] bias sums
  float* const P = a.part + (size_t)blockIdx.x * ((size_t)cout * a.Kp + cout);
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < NTW; q++) {
      const int nt = wv + 8 * q;
      if (nt * 16 < a.Kp) {
#pragma unroll
        for (int r = 0; r < 4; r++) P[(size_t)(m * 16 + lg * 4 + r) * a.Kp + nt * 16 + li] = acc[m][q][r];
      }
    }
  __syncthreads();
  float* const sh = reinterpret_cast<float*>(lds);
  sh[tid] = bsum;
  __syncthreads();
  if (tid < cout) {
    float t = 0.f;
    for (int j = tid; j < DW_TH; j += cout) t += sh[j];
    P[(size_t)cout * a.Kp + tid] = t;
  }
}

// gw[co][k] = sum over the workgroups' partials; gb[co] += their bias sums.  64 outputs per block, the partials dealt to 4 waves
// (wave y: partials y, y + 4, ...; 8 independent chains each), combined in fixed order: deterministic.
__global__ __launch_bounds__(256) void conv_dw_reduce_kernel(const float* __restrict__ part, int nparts, int cout, int K, int Kp,
                                                             float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float sh[4][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + x;
  const size_t stride = (size_t)cout * Kp + cout;
  const bool isw = i < cout * K, isb = !isw && i < cout * K + cout;
  size_t off = 0;
  if (isw) { const int co = i / K; off = (size_t)co * Kp + (i - co * K); }
  else if (isb) off = (size_t)cout * Kp + (i - cout * K);
  float t[8];
#pragma unroll
  for (int u = 0; u < 8; u++) t[u] = 0.f;
  if (isw || isb) {
    const float* p = part + off;
    int w = y;
    for (; w + 28 < nparts; w += 32) {
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] += p[(size_t)(w + 4 * u) * stride];
    }
    for (; w < nparts; w += 4) t[0] += p[(size_t)w * stride];
  }
  sh[y][x] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  if (y == 0) {
    const float v = (sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x]);
    if (isw) gw[i] = v;
    else if (isb && gb) gb[i - cout * K] += v;
  }
}

int n_cus() {
  static int n[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!n[dev]) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; n[dev] = v; }
  return n[dev];
}

template <int MT, int NTW, int XE>
int launch_dw_x(const DwArgs& a, int grid, hipStream_t st) {
  static unsigned long long done = 0;
  if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&conv_dw_kernel<MT, NTW, XE>), 160 * 1024, &done) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
  hipLaunchKernelGGL((conv_dw_kernel<MT, NTW, XE>), dim3(grid), dim3(DW_TH), a.lds_bytes, st, a);
  return avlen_launch_status();
}
template <int MT, int NTW>
int launch_dw(const DwArgs& a, int grid, hipStream_t st) {
  return (long)a.s * a.Wc <= 4 * DW_TH ? launch_dw_x<MT, NTW, 4>(a, grid, st) : launch_dw_x<MT, NTW, DW_XE>(a, grid, st);
}
template <int MT>
int launch_dw_n(const DwArgs& a, int ntw, int grid, hipStream_t st) {
  switch (ntw) {
    case 1: return launch_dw<MT, 1>(a, grid, st);
    case 2: return launch_dw<MT, 2>(a, grid, st);
    case 3: return launch_dw<MT, 3>(a, grid, st);
    case 4: return launch_dw<MT, 4>(a, grid, st);
    case 5: return launch_dw<MT, 5>(a, grid, st);
  }
  return AVLEN_NOT_BIG;
}
}  // namespace

// ========================================================================================================================
// Direct data gradient (bf16 mode):  dX[b][ih][iw][ci] = [act > 0] * sum_{kh, kw, co} dY[b][(ih-kh)/s][(iw-kw)/s][co] * W[co][kh][kw][ci]
// over the taps whose division is exact and lands inside the output.  The GEMM route writes dY * W as an fp32 im2col-shaped buffer
// (M x K, 0.7 GB for one conv of a minibatch) and gathers it back (`col2im`); here a workgroup keeps the transposed weights
// ([tap][ci][co] bf16, packed once per call) and ONE image's dY (bf16, zero halo) in LDS and forms 16-pixel x 16-channel output tiles
// directly: the contraction index (tap, co) is channel-last in dY, so both operand fragments are aligned 16-byte LDS reads; with a
// stride s the input pixels split into s*s parity classes, each with its own tap subset, and a tile is 16 pixels of one class along a
// row (their dY positions are consecutive).  dY is read once, dX written once, the ReLU mask of the layer below applied on the way.
// ========================================================================================================================
namespace {
constexpr int DX_TH = 512;
constexpr int DX_PAD = 8;                        // bf16 elements of padding per LDS row of cout values (conflict-free 16-byte reads)
struct DxArgs {
  const float* dY; const float* act; float* dX; const bf16* wt;
  int R, H, W, cin, OH, OW, KH, KW, s, cout, halo;
  int yrows, ycols;                              // padded dY image in LDS: (OH + 2 halo) x (OW + 2 halo) pixels
  unsigned wt_bytes, y_off, lds_bytes;
};

// W packed [co][kh][kw][ci] fp32 -> wt [tap][ci][cout + DX_PAD] bf16 (the LDS image of conv_dx_kernel)
__global__ void conv_dx_pack_kernel(const float* __restrict__ w, bf16* __restrict__ wt, int cout, int taps, int cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = cout + DX_PAD;
  if (i >= taps * cin * row) return;
  const int co = i % row, tc = i / row;          // tc = tap * cin + ci
  wt[i] = co < cout ? (bf16)w[(long)co * taps * cin + tc] : (bf16)0.f;
}

constexpr int DX_UMAX = 6;                       // 8-channel pieces of one dY image per thread (OH * OW * cout / 8 <= DX_UMAX * DX_TH)
template <int NT>                                // cin / 16
__global__ __launch_bounds__(DX_TH) void conv_dx_kernel(const DxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int cout = a.cout, rowb = (cout + DX_PAD) * 2, s = a.s, halo = a.halo;
  for (unsigned o = tid * 16u; o < a.wt_bytes; o += DX_TH * 16u)
    *reinterpret_cast<uint4*>(lds + o) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.wt) + o);
  for (unsigned o = a.y_off + tid * 16u; o < a.lds_bytes; o += DX_TH * 16u) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0, 0, 0, 0);
  char* const ybase = lds + a.y_off;
  const int c8 = cout >> 3, units = a.OH * a.OW * c8;            // 8-channel pieces of one dY image
  // tiles of an image: (parity class, input row of the class, 16 pixels of the class along the row)
  const int per_row_max = (a.W + s - 1) / s, jt = (per_row_max + 15) >> 4;
  const int rows_max = (a.H + s - 1) / s;
  const int ntiles = s * s * rows_max * jt;
  // the next image's dY travels in registers while this image's tiles run (branch-free loads: all in flight)
  float4 yv[DX_UMAX][2];
  auto fetch_y = [&](int b) {
    const float* src = a.dY + (long)b * a.OH * a.OW * cout;
#pragma unroll
    for (int k = 0; k < DX_UMAX; k++) {
      const int u = tid + DX_TH * k < units ? tid + DX_TH * k : units - 1;
      yv[k][0] = *reinterpret_cast<const float4*>(src + (long)u * 8); yv[k][1] = *reinterpret_cast<const float4*>(src + (long)u * 8 + 4);
    }
  };
  auto store_y = [&]() {
#pragma unroll
    for (int k = 0; k < DX_UMAX; k++) {
      const int u = tid + DX_TH * k;
      if (u < units) {
        const int pix = u / c8, cc = u - pix * c8, oh = pix / a.OW, ow = pix - oh * a.OW;
        bf16x8 o;
        o[0] = (bf16)yv[k][0].x; o[1] = (bf16)yv[k][0].y; o[2] = (bf16)yv[k][0].z; o[3] = (bf16)yv[k][0].w;
        o[4] = (bf16)yv[k][1].x; o[5] = (bf16)yv[k][1].y; o[6] = (bf16)yv[k][1].z; o[7] = (bf16)yv[k][1].w;
        *reinterpret_cast<bf16x8*>(ybase + ((oh + halo) * a.ycols + ow + halo) * rowb + cc * 16) = o;
      }
    }
  };
  // a tile's geometry; `ok` false: no such tile (past the image)
  struct Tile { int py, px, ih, j0; bool ok; };
  auto tile_of = [&](int t) {
    Tile q;
    const int cls = t / (rows_max * jt), rem = t - cls * rows_max * jt, r = rem / jt;
    q.py = cls / s; q.px = cls - q.py * s; q.ih = q.py + r * s; q.j0 = (rem - r * jt) * 16;
    q.ok = t < ntiles && q.ih < a.H && q.px + q.j0 * s < a.W;
    return q;
  };
  // the ReLU mask of a tile's outputs (act of the layer below), fetched one tile ahead: a load under `if (pixel exists)` is a
  // dependent round trip per row of the tile (first version: 297 us for a conv whose traffic is worth 120 us)
  float mk[4][NT];
  auto fetch_mask = [&](int b, const Tile& q) {
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      int iw = q.px + (q.j0 + lg * 4 + r4) * s;
      iw = iw < a.W ? iw : a.W - 1;
      const long o = (((long)b * a.H + (q.ok ? q.ih : 0)) * a.W + iw) * a.cin + li;
#pragma unroll
      for (int n = 0; n < NT; n++) mk[r4][n] = a.act ? a.act[o + n * 16] : 1.f;
    }
  };
  if ((int)blockIdx.x < a.R) fetch_y(blockIdx.x);
  for (int b = blockIdx.x; b < a.R; b += gridDim.x) {
    __syncthreads();                                             // the previous image's tiles are done with the dY image
    store_y();
    __syncthreads();
    if (b + (int)gridDim.x < a.R) fetch_y(b + gridDim.x);
    Tile q = tile_of(wv);
    fetch_mask(b, q);
    for (int t = wv; t < ntiles; t += 8) {
      const Tile cur = q;
      float mc[4][NT];
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
        for (int n = 0; n < NT; n++) mc[r4][n] = mk[r4][n];
      q = tile_of(t + 8);
      fetch_mask(b, q);                                          // next tile's mask in flight during this tile's products
      if (!cur.ok) continue;                                     // wave-uniform
      const int py = cur.py, px = cur.px, ih = cur.ih, j0 = cur.j0;
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kh = py; kh < a.KH; kh += s) {                    // ih - kh divisible by s  <=>  kh = py (mod s)
        const int oh = (ih - kh) / s;                            // exact; negative / >= OH: the zero halo
        for (int kw = px; kw < a.KW; kw += s) {
          const int ow0 = (px - kw) / s + j0;                    // exact: px - kw is a multiple of s
          const int tap = kh * a.KW + kw;
          const char* ya = ybase + ((oh + halo) * a.ycols + ow0 + li + halo) * rowb + lg * 16;
          const char* wb = lds + ((tap * a.cin + li) * rowb) + lg * 16;
          for (int kk = 0; kk < cout; kk += 32) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(ya + kk * 2);
#pragma unroll
            for (int n = 0; n < NT; n++) {
              const bf16x8 bfv = *reinterpret_cast<const bf16x8*>(wb + n * 16 * rowb + kk * 2);
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc[n], 0, 0, 0);
            }
          }
        }
      }
      // D: row (pixel) lg * 4 + r, column (channel) li
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const int iw = px + (j0 + lg * 4 + r4) * s;
        if (iw < a.W) {
          const long o = (((long)b * a.H + ih) * a.W + iw) * a.cin + li;
#pragma unroll
          for (int n = 0; n < NT; n++) a.dX[o + n * 16] = mc[r4][n] > 0.f ? acc[n][r4] : 0.f;
        }
      }
    }
  }
}

}  // namespace

// dX (B, H, W, cin) = [act > 0] * conv_transpose(dY, W); W packed [cout][KH][KW][cin] fp32.  AVLEN_NOT_BIG: not applicable.
// Scratch: c.gws (the transposed bf16 weights).
int avlen_i_conv_dx_direct(const avlen_ctx& c, const float* w, const float* dY, const float* act, float* dX, long R, int H, int W, int cin,
                           int OH, int OW, int cout, int KH, int KW, int s) {
  if (c.prec != AVLEN_PREC_BF16 || R <= 0 || R > (1L << 28) || (cin != 32 && cin != 64) || (cout & 31) || cout > 128) return AVLEN_NOT_BIG;
  if (s < 1 || s > 2 || KH < s || KW < s || (OH - 1) * s + KH > H || (OW - 1) * s + KW > W) return AVLEN_NOT_BIG;
  DxArgs a;
  a.dY = dY; a.act = act; a.dX = dX;
  a.R = (int)R; a.H = H; a.W = W; a.cin = cin; a.OH = OH; a.OW = OW; a.KH = KH; a.KW = KW; a.s = s; a.cout = cout;
  const int hk = (KH > KW ? KH : KW);
  a.halo = (hk - 1 + s - 1) / s;                 // positions (ih - kh) / s below 0 / past the end: zero
  // A 16-pixel tile reads up to 15 positions past the last valid one of its row: they wrap into the next row of the LDS image (finite
  // values, and only in rows of the product whose pixel does not exist); the last row needs 16 more positions behind the image.
  a.yrows = OH + 2 * a.halo; a.ycols = OW + 2 * a.halo;
  const int rowb = (cout + DX_PAD) * 2;
  a.wt_bytes = (unsigned)(KH * KW * cin * rowb);
  a.y_off = (a.wt_bytes + 15u) & ~15u;
  a.lds_bytes = a.y_off + (unsigned)((a.yrows * a.ycols + 16) * rowb);
  a.lds_bytes = (a.lds_bytes + 15u) & ~15u;
  if (a.lds_bytes > 160u * 1024u || !c.gws || a.wt_bytes > c.gws_bytes || (long)OH * OW * (cout / 8) > DX_UMAX * DX_TH) return AVLEN_NOT_BIG;
  a.wt = (const bf16*)c.gws;
  const int nw = KH * KW * cin * (cout + DX_PAD);
  hipLaunchKernelGGL(conv_dx_pack_kernel, dim3((nw + 255) / 256), dim3(256), 0, c.st, w, (bf16*)c.gws, cout, KH * KW, cin);
  const int grid = (int)(R < n_cus() ? R : n_cus());
  static unsigned long long done2 = 0, done4 = 0;
  if (cin == 32) {
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&conv_dx_kernel<2>), 160 * 1024, &done2) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_dx_kernel<2>, dim3(grid), dim3(DX_TH), a.lds_bytes, c.st, a);
  } else {
    if (avlen_set_dyn_lds(reinterpret_cast<const void*>(&conv_dx_kernel<4>), 160 * 1024, &done4) != AVLEN_OK) return AVLEN_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_dx_kernel<4>, dim3(grid), dim3(DX_TH), a.lds_bytes, c.st, a);
  }
  return avlen_launch_status();
}

// gw [cout][KH*KW*C] (packed K order) = dY^T im2col(X), gb[cout] += column sums of dY (gb may be null).  bf16 operands, fp32
// accumulation, valid convolution (no padding).  AVLEN_NOT_BIG: shape outside the kernel's limits (caller takes the GEMM route).
// Scratch: c.gws (one partial per workgroup).
int avlen_i_conv_dw_direct(const avlen_ctx& c, float* gw, float* gb, int cout, const float* dY, const float* X, long R, int H,
                           int W, int C, int OH, int OW, int KH, int KW, int s) {
  if (c.prec != AVLEN_PREC_BF16 || R <= 0 || R > (1L << 28) || (cout != 32 && cout != 64)) return AVLEN_NOT_BIG;
  if (s < 1 || KH < s || (OH - 1) * s + KH > H || (OW - 1) * s + KW > W || OH < 1 || OW < 1) return AVLEN_NOT_BIG;
  DwArgs a;
  a.X = X; a.dY = dY; a.part = (float*)c.gws;
  a.R = (int)R; a.H = H; a.Wc = W * C; a.C = C; a.OH = OH; a.OW = OW; a.KH = KH; a.KW = KW; a.s = s; a.cout = cout;
  a.K = KH * KW * C; a.Kp = (a.K + 15) / 16 * 16;
  const int ntw = (a.Kp / 16 + 7) / 8;
  if (ntw > 5 || (long)s * a.Wc > DW_XE * DW_TH || (long)OW * cout > DW_YE * DW_TH) return AVLEN_NOT_BIG;
  const int OWp = (OW + 3) & ~3;
  const long over = ((long)(OWp - 1) * s + KW) * C - a.Wc;             // fragments of the padded positions read past the last ring row
  a.xring_bytes = (unsigned)((long)KH * a.Wc * 16);
  a.y_off = a.xring_bytes + (unsigned)((over > 0 ? over : 0) * 16);
  a.lds_bytes = a.y_off + (unsigned)OWp * cout * 16u;
  if (a.lds_bytes < DW_TH * 4u) a.lds_bytes = DW_TH * 4u;
  if (a.lds_bytes > 160u * 1024u) return AVLEN_NOT_BIG;
  a.octets = (int)((R + 7) / 8);
  const long TL = (long)a.octets * OH;
  if (TL > 0x7fffffffL) return AVLEN_NOT_BIG;
  a.TL = (int)TL;
  const int grid = (int)(TL < n_cus() ? TL : n_cus());
  const size_t per = (size_t)cout * a.Kp + cout;
  if (!c.gws || per * grid * 4 > c.gws_bytes) return AVLEN_NOT_BIG;
  int rc = cout == 32 ? launch_dw_n<2>(a, ntw, grid, c.st) : launch_dw_n<4>(a, ntw, grid, c.st);
  if (rc != AVLEN_OK) return rc;
  const int tot = cout * a.K + cout;
  hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((tot + 63) / 64), dim3(256), 0, c.st, a.part, grid, cout, a.K, a.Kp, gw, gb);
  return avlen_launch_status();
}
