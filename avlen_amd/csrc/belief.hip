// BeliefPredictor (ss_baselines/savi/models/belief_predictor.py:56-206) for gfx950: the two spectrogram networks and the
// per-environment belief filter that runs between `RolloutStorage.insert` and the next `act` of every rollout step.
//
//   avlen_resnet18_any_fwd   <- predictor = custom_resnet18(2 | 23 ch), fc 4608 -> 2   (belief_predictor.py:66-72, 126-137;
//                               smt_resnet.py:56-149 at the spectrogram's own size: no resize, GroupNorm(16))
//   avlen_resnet18_tv_fwd    <- classifier = torchvision resnet18 (conv1 2 -> 64), eval-mode BatchNorm folded into the convs
//                               by the host (belief_predictor.py:79-81, 179); third-party architecture, parity unpinned
//   avlen_belief_input       <- cnn_forward's channel concat for the distractor variant (belief_predictor.py:129-134)
//   avlen_belief_update      <- the python loop of update() (belief_predictor.py:139-206) + base_to_odom / odom_to_base
//                               (:213-230), one thread per environment, filter state resident on the device (the reference
//                               pulls the network outputs and every pose to the host and loops in numpy)
//
// Two flows per network.  prec = FP32 (parity mode): fp32-staged implicit-GEMM conv (any H, W, Cin) + the generic GroupNorm.
// prec = BF16 (rollout mode): every conv on the glds / 8-wave MFMA implicit GEMM of igemm2.hip (bf16 operands, split-K for the
// long-K / few-row convs of the classifier's late stages), GroupNorm from the fp32 raw output straight to the next conv's bf16
// operand, the classifier's residual stream carried in fp32 beside its bf16 operand copy.  The spectrogram is 65 x 26, so
// none of the towers' 64 x 64 specialisations (direct conv, LDS-resident tail) apply.
#include "common.h"
#include "../../include/avlen_hip.h"
#include "internal.h"

namespace {

#define TRY(x) do { int _rc = (x); if (_rc != AVLEN_OK) return _rc; } while (0)

inline int conv_out(int h, const avlen_conv& k) { return (h + 2 * k.pad - k.kh) / k.stride + 1; }

__global__ void maxpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int OH,
                                    int OW, int k, int s, int p) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)B * OH * OW * C;
  if (i >= tot) return;
  int c = (int)(i % C); long r = i / C;
  int ox = (int)(r % OW); r /= OW;
  int oy = (int)(r % OH); int b = (int)(r / OH);
  float m = -INFINITY;
  for (int dy = 0; dy < k; dy++) {
    int iy = oy * s - p + dy;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < k; dx++) {
      int ix = ox * s - p + dx;
      if (ix < 0 || ix >= W) continue;
      m = fmaxf(m, x[(((long)b * H + iy) * W + ix) * C + c]);
    }
  }
  y[i] = m;
}

// adaptive average pool to 1x1: one block per sample, threads over channels
__global__ void avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < HW; i++) s += x[((long)b * HW + i) * C + c];
    y[(long)b * C + c] = s / (float)HW;
  }
}

__global__ void belief_input_kernel(const float* __restrict__ spec, const float* __restrict__ cat, float* __restrict__ out,
                                    int B, int HW, int Cs, int Cc) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = Cs + Cc;
  if (i >= (long)B * HW * C) return;
  int c = (int)(i % C); long px = i / C; int b = (int)(px / HW);
  out[i] = c < Cs ? spec[px * Cs + c] : cat[(long)b * Cc + (c - Cs)];
}

// per-environment sum of the spectrogram ("is the sound still playing", belief_predictor.py:157,187)
__global__ void spec_sum_kernel(const float* __restrict__ spec, float* __restrict__ out, long n) {
  __shared__ float sh[16];
  const float* s = spec + (long)blockIdx.x * n;
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) a += s[i];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

// belief_predictor.py:213-230 in float32, as numpy evaluates them on float32 operands
__device__ inline void base_to_odom(float bx, float by, const float* pose, float* ox, float* oy) {
  const float angle = -pose[2];
  const float d = sqrtf(bx * bx + by * by);
  const float theta = atan2f(by, bx);
  *ox = pose[0] + d * cosf(theta + angle);
  *oy = pose[1] + d * sinf(theta + angle);
}
__device__ inline void odom_to_base(float ox, float oy, const float* pose, float* bx, float* by) {
  const float angle = -pose[2];
  const float dx = ox - pose[0], dy = oy - pose[1];
  const float dth = atan2f(dy, dx) - angle;
  const float d = sqrtf(dx * dx + dy * dy);
  *bx = d * cosf(dth); *by = d * sinf(dth);
}

__global__ void belief_update_kernel(const float* __restrict__ pointgoals, int ld_pg, const float* __restrict__ labels, int ld_lab,
                                     const float* __restrict__ pose, int ld_pose, const float* __restrict__ spec_sum,
                                     const unsigned char* __restrict__ dones, float* __restrict__ last_pg,
                                     int* __restrict__ has_pg, float* __restrict__ last_label, int* __restrict__ has_label,
                                     float* __restrict__ location_belief, float* __restrict__ category_belief, int B, int n_label,
                                     float w, int current_pred_only) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const bool done = dones && dones[i];
  const bool sounding = spec_sum[i] != 0.f;
  if (pointgoals) {                                    // :146-172
    const float* ps = pose + (long)i * ld_pose;
    if (done) has_pg[i] = 0;
    float ax, ay;
    if (sounding) {
      const float bx = -pointgoals[(long)i * ld_pg + 1], by = pointgoals[(long)i * ld_pg];
      if (!has_pg[i] || current_pred_only) { ax = bx; ay = by; }
      else {
        float lx, ly;
        odom_to_base(last_pg[i * 2], last_pg[i * 2 + 1], ps, &lx, &ly);
        ax = (1.f - w) * bx + w * lx; ay = (1.f - w) * by + w * ly;
      }
      base_to_odom(ax, ay, ps, &last_pg[i * 2], &last_pg[i * 2 + 1]);
      has_pg[i] = 1;
    } else if (!has_pg[i]) { ax = 10.f; ay = 10.f; }
    else odom_to_base(last_pg[i * 2], last_pg[i * 2 + 1], ps, &ax, &ay);
    location_belief[i * 2] = ax; location_belief[i * 2 + 1] = ay;
  }
  if (labels) {                                        // :175-206
    if (done) has_label[i] = 0;
    float* ll = last_label + (long)i * n_label;
    float* out = category_belief + (long)i * n_label;
    const float* lab = labels + (long)i * ld_lab;
    if (sounding) {
      const bool fresh = !has_label[i] || current_pred_only;
      for (int c = 0; c < n_label; c++) {
        const float v = fresh ? lab[c] : (1.f - w) * lab[c] + w * ll[c];
        ll[c] = v; out[c] = v;
      }
      has_label[i] = 1;
    } else if (!has_label[i]) {
      const float u = (float)(1.0 / (double)n_label);
      for (int c = 0; c < n_label; c++) out[c] = u;
    } else {
      for (int c = 0; c < n_label; c++) out[c] = ll[c];
    }
  }
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8b;

// 16-bit storage of the fast flows: bf16, or -- f16 -- IEEE half in the same slots (AVLEN_PREC_FP16: the bf16x3 policies' mode for
// these auxiliary networks; weights packed with fmt 1)
__device__ __forceinline__ __bf16 h16_from(float v, int f16) { return f16 ? __builtin_bit_cast(__bf16, (_Float16)v) : (__bf16)v; }
__device__ __forceinline__ float h16_to(__bf16 v, int f16) { return f16 ? (float)__builtin_bit_cast(_Float16, v) : (float)v; }

// GroupNorm(G) for the bf16 flow: x = raw conv output in fp32 (B,HW,C), one block per sample (the activations of these
// networks are a few tens of KB per sample and stay in L2 between the two passes); y (bf16) = [relu](xhat*g + b [+ res]).
__global__ __launch_bounds__(1024) void gn_any16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const __bf16* __restrict__ res,
                                                       __bf16* __restrict__ y, int HW, int C, int G, int relu, float eps, int f16) {
  __shared__ float s_sum[128], s_sq[128], s_scale[128], s_shift[128];
  __shared__ float part[2][16][128];                   // per-wave channel sums; every reduction below runs in a FIXED order (bit-reproducible)
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (long)b * HW * C;
  const long n4 = (long)HW * C / 4;
  const int c0 = (tid * 4) % C;                       // 4096 % C == 0: a thread keeps its 4 channels
  float a[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (long f = tid; f < n4; f += 1024) {
    float4 v = *reinterpret_cast<const float4*>(xb + f * 4);
    a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
  }
  // lanes l and l + C/4 (+ 2C/4 ...) of a wave own the same 4 channels (64 % (C/4) == 0 for C <= 128, C % 16 == 0): butterfly
  // over those lane bits, then one row of C sums per wave -- a serial walk over 1024 per-thread partials per channel cost
  // 10-17 us per launch for a tensor worth 2 us (20 launches per step of the BeliefPredictor)
  const int lane = tid & 63, wv = tid >> 6, per = C >> 2;          // per: lanes that cover all channels once (4 .. 32)
#pragma unroll
  for (int o = 32; o >= 4; o >>= 1) {
    if (o >= per) {
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] += __shfl_xor(a[i], o, 64); q[i] += __shfl_xor(q[i], o, 64); }
    }
  }
  if (lane < per) {
#pragma unroll
    for (int i = 0; i < 4; i++) { part[0][wv][lane * 4 + i] = a[i]; part[1][wv][lane * 4 + i] = q[i]; }
  }
  __syncthreads();
  if (tid < C) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) { s1 += part[0][w][tid]; s2 += part[1][w][tid]; }
    s_sum[tid] = s1; s_sq[tid] = s2;
  }
  __syncthreads();
  const int cg = C / G;
  if (tid < G) {
    double sum = 0.0, sq = 0.0;
    for (int c = tid * cg; c < (tid + 1) * cg; c++) { sum += s_sum[c]; sq += s_sq[c]; }
    double n = (double)HW * cg, mean = sum / n, var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = tid * cg; c < (tid + 1) * cg; c++) {
      float sc = gamma[c] * rstd;
      s_scale[c] = sc; s_shift[c] = beta[c] - (float)mean * sc;
    }
  }
  __syncthreads();
  float sc[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { sc[i] = s_scale[c0 + i]; sh[i] = s_shift[c0 + i]; }
  const __bf16* rb = res ? res + (long)b * HW * C : nullptr;
  __bf16* yb = y + (long)b * HW * C;
  for (long f = tid; f < n4; f += 1024) {
    float4 v = *reinterpret_cast<const float4*>(xb + f * 4);
    float o[4] = {v.x * sc[0] + sh[0], v.y * sc[1] + sh[1], v.z * sc[2] + sh[2], v.w * sc[3] + sh[3]};
    if (rb) {
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] += h16_to(rb[f * 4 + i], f16);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) yb[f * 4 + i] = h16_from(relu ? fmaxf(o[i], 0.f) : o[i], f16);
  }
}

int gn_any16(const float* x, const avlen_affine& n, const __bf16* res, __bf16* y, int B, int HW, int C, int relu, hipStream_t st,
             int f16 = 0) {
  if (C > 128 || (1024 % C) || (C % 16)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(gn_any16_kernel, dim3(B), dim3(1024), 0, st, x, n.g, n.b, res, y, HW, C, 16, relu, 1e-5f, f16);
  return avlen_launch_status();
}

// maxpool for the bf16 flow: fp32 in, fp32 (the next block's residual) and bf16 (the next conv's operand) out
__global__ void maxpool_nhwc2_kernel(const float* __restrict__ x, float* __restrict__ y32, __bf16* __restrict__ y16, int B, int H,
                                     int W, int C, int OH, int OW, int k, int s, int p, int f16) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)B * OH * OW * C;
  if (i >= tot) return;
  int c = (int)(i % C); long r = i / C;
  int ox = (int)(r % OW); r /= OW;
  int oy = (int)(r % OH); int b = (int)(r / OH);
  float m = -INFINITY;
  for (int dy = 0; dy < k; dy++) {
    int iy = oy * s - p + dy;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < k; dx++) {
      int ix = ox * s - p + dx;
      if (ix < 0 || ix >= W) continue;
      m = fmaxf(m, x[(((long)b * H + iy) * W + ix) * C + c]);
    }
  }
  y32[i] = m; y16[i] = h16_from(m, f16);
}

constexpr size_t CONV_SCRATCH = (size_t)32 << 20;       // split-K slabs of the bf16 convs (run_g2 falls back to fewer splits)

size_t any_ws(int B, int H, int W) {
  size_t px = (size_t)B * H * W;
  size_t fp32_flow = 4 * (px * 16 * sizeof(float) + 256);
  size_t bf16_flow = px * 32 * 2 + 3 * (px * 16 * sizeof(float) + 256) + 4 * (px * 16 * 2 + 256) + CONV_SCRATCH;
  return (fp32_flow > bf16_flow ? fp32_flow : bf16_flow) + 8192;
}

bool conv16_ok(const avlen_conv& k) { return k.w16 && k.cin16 >= 8 && !(k.cin16 & (k.cin16 - 1)) && k.cin16 >= k.cin; }

// bf16 flow of the GroupNorm ResNet: every conv on the glds/MFMA implicit GEMM (fp32 raw output), GroupNorm reads the fp32
// raw tensor and writes the bf16 activation the next conv consumes.
int any_fwd_bf16(const avlen_resnet18* net, const float* x, int B, int H, int W, int C, float* out, int ld_out, void* ws,
                 size_t ws_bytes, hipStream_t st, int f16 = 0) {
  WsBump w(ws, ws_bytes);
  const size_t px = (size_t)B * H * W;
  const avlen_conv& c1 = net->conv1;
  __bf16* x16 = w.take<__bf16>(px * c1.cin16);
  float* raw[3]; __bf16* act[4];
  for (int i = 0; i < 3; i++) raw[i] = w.take<float>(px * 16);
  for (int i = 0; i < 4; i++) act[i] = w.take<__bf16>(px * 16);
  void* gws = w.take<char>(CONV_SCRATCH);
  if (!w.ok() || c1.cin16 > 32) return AVLEN_ERR_WS;
  auto conv = [&](const avlen_conv& k, const __bf16* in, float* o32, int h, int wd) {
    static int split = -1;                             // AVLEN_BELIEF_SPLITK=0: no split-K on the predictor's convs (A/B knob)
    if (split < 0) split = (int)avlen_knob("AVLEN_BELIEF_SPLITK", 1);
    avlen_g2_opts go; go.f16 = f16;
    return avlen_conv2d_nhwc_h16(in, k.w16, nullptr, nullptr, o32, nullptr, nullptr, B, h, wd, k.cin16, k.cout, k.kh, k.kw, k.stride,
                                 k.pad, 0, split ? gws : nullptr, split ? CONV_SCRATCH : 0, st, &go);
  };
  TRY(avlen_cast_h16(x, C, x16, c1.cin16, (long)px, C, f16 ? 1 : 0, st));
  const int h1 = conv_out(H, c1), w1 = conv_out(W, c1);
  TRY(conv(c1, x16, raw[0], H, W));
  TRY(gn_any16(raw[0], net->bn1, nullptr, act[0], B, h1 * w1, c1.cout, 1, st, f16));
  __bf16* cur = act[0]; __bf16* a1 = act[1]; __bf16* idt = act[2]; __bf16* nxt = act[3];
  int h = h1, wd = w1, ch = c1.cout;
  for (int i = 0; i < 8; i++) {
    const avlen_resblock& k = net->block[i];
    if (k.conv1.cin != ch || k.conv1.cin16 != ch) return AVLEN_ERR_ARG;
    const int oh = conv_out(h, k.conv1), ow = conv_out(wd, k.conv1), co = k.conv1.cout;
    if (oh <= 0 || ow <= 0 || (size_t)B * oh * ow * co > px * 16) return AVLEN_ERR_ARG;
    TRY(conv(k.conv1, cur, raw[0], h, wd));
    TRY(gn_any16(raw[0], k.bn1, nullptr, a1, B, oh * ow, co, 1, st, f16));
    TRY(conv(k.conv2, a1, raw[1], oh, ow));
    const __bf16* identity = cur;
    if (k.has_down) {
      TRY(conv(k.down, cur, raw[2], h, wd));
      TRY(gn_any16(raw[2], k.bnd, nullptr, idt, B, oh * ow, co, 0, st, f16));
      identity = idt;
    }
    TRY(gn_any16(raw[1], k.bn2, identity, nxt, B, oh * ow, co, 1, st, f16));
    __bf16* o = cur; cur = nxt; nxt = o;
    h = oh; wd = ow; ch = co;
  }
  if (net->fc.in_f != h * wd * ch || !net->fc.w16) return AVLEN_ERR_ARG;
  return avlen_gemm_h16(cur, net->fc.in_f, net->fc.w16, net->fc.ld16, out, ld_out, nullptr, 0, net->fc.b, nullptr, 0, B,
                        net->fc.out_f, net->fc.in_f, 0, f16 ? 1 : 0, gws, CONV_SCRATCH, st);
}

}  // namespace

extern "C" size_t avlen_resnet18_any_workspace_bytes(int B, int H, int W) { return any_ws(B, H, W); }

extern "C" int avlen_resnet18_any_fwd(const avlen_resnet18* net, const float* x, int B, int H, int W, int C, float* out,
                                      int ld_out, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!net || !x || !out || B <= 0 || C != net->conv1.cin || ws_bytes < any_ws(B, H, W)) return AVLEN_ERR_WS;
  // widest activation: conv1 / layer1 at (h1, w1, 16); every later stage halves the extent and doubles the channels
  const int h1 = conv_out(H, net->conv1), w1 = conv_out(W, net->conv1);
  if (h1 <= 0 || w1 <= 0 || h1 > H || w1 > W || net->conv1.cout > 16) return AVLEN_ERR_ARG;
  if (prec == AVLEN_PREC_BF16 || prec == AVLEN_PREC_FP16) {        // the caller packed the 16-bit weights in the matching format
    bool ok = conv16_ok(net->conv1) && net->fc.w16;
    for (int i = 0; i < 8 && ok; i++)
      ok = conv16_ok(net->block[i].conv1) && conv16_ok(net->block[i].conv2) && (!net->block[i].has_down || conv16_ok(net->block[i].down));
    if (ok) return any_fwd_bf16(net, x, B, H, W, C, out, ld_out, ws, ws_bytes, st, prec == AVLEN_PREC_FP16);
    if (prec == AVLEN_PREC_FP16) return AVLEN_ERR_ARG;             // no fp16 fallback
  }
  WsBump w(ws, ws_bytes);
  const size_t act = (size_t)B * H * W * 16;
  float* buf[4];
  for (int i = 0; i < 4; i++) buf[i] = w.take<float>(act);
  const avlen_conv& c1 = net->conv1;
  TRY(avlen_conv2d_nhwc(x, c1.w, nullptr, nullptr, buf[0], B, H, W, c1.cin, c1.cout, c1.kh, c1.kw, c1.stride, c1.pad, 0, prec, st));
  TRY(avlen_groupnorm_nhwc(buf[0], net->bn1.g, net->bn1.b, nullptr, buf[1], B, h1 * w1, c1.cout, 16, 1, 1e-5f, st));
  float* cur = buf[1]; float* t1 = buf[0]; float* t2 = buf[2]; float* t3 = buf[3];
  int h = h1, wd = w1, ch = c1.cout;
  for (int i = 0; i < 8; i++) {
    const avlen_resblock& k = net->block[i];
    if (k.conv1.cin != ch) return AVLEN_ERR_ARG;
    const int oh = conv_out(h, k.conv1), ow = conv_out(wd, k.conv1), co = k.conv1.cout;
    if (oh <= 0 || ow <= 0 || (size_t)B * oh * ow * co > act) return AVLEN_ERR_ARG;
    TRY(avlen_conv2d_nhwc(cur, k.conv1.w, nullptr, nullptr, t1, B, h, wd, ch, co, k.conv1.kh, k.conv1.kw, k.conv1.stride,
                          k.conv1.pad, 0, prec, st));
    TRY(avlen_groupnorm_nhwc(t1, k.bn1.g, k.bn1.b, nullptr, t1, B, oh * ow, co, 16, 1, 1e-5f, st));
    TRY(avlen_conv2d_nhwc(t1, k.conv2.w, nullptr, nullptr, t2, B, oh, ow, co, co, k.conv2.kh, k.conv2.kw, k.conv2.stride,
                          k.conv2.pad, 0, prec, st));
    const float* identity = cur;
    if (k.has_down) {
      TRY(avlen_conv2d_nhwc(cur, k.down.w, nullptr, nullptr, t3, B, h, wd, ch, co, k.down.kh, k.down.kw, k.down.stride,
                            k.down.pad, 0, prec, st));
      TRY(avlen_groupnorm_nhwc(t3, k.bnd.g, k.bnd.b, nullptr, t3, B, oh * ow, co, 16, 0, 1e-5f, st));
      identity = t3;
    }
    TRY(avlen_groupnorm_nhwc(t2, k.bn2.g, k.bn2.b, identity, t2, B, oh * ow, co, 16, 1, 1e-5f, st));
    float* o = cur; cur = t2; t2 = o;
    h = oh; wd = ow; ch = co;
  }
  if (net->fc.in_f != h * wd * ch) return AVLEN_ERR_ARG;
  return avlen_gemm(cur, net->fc.in_f, 0, net->fc.w, net->fc.in_f, 0, out, ld_out, net->fc.b, nullptr, 0, B, net->fc.out_f,
                    net->fc.in_f, 0, prec, 1, 0.f, nullptr, 0, st);
}

// torchvision resnet18 after the host folded every eval-mode BatchNorm into its conv (conv.w scaled per output channel,
// conv.b = beta - mean * scale): conv7x7 s2 + ReLU, maxpool 3x3 s2 p1, 8 BasicBlocks (ReLU after the residual add),
// global average pool, fc.  Activation extents: conv1 output is the widest tensor (64 channels at ~H/2 x W/2).
extern "C" size_t avlen_resnet18_tv_workspace_bytes(int B, int H, int W) {
  size_t act = (size_t)B * ((H + 1) / 2 + 1) * ((W + 1) / 2 + 1) * 64;
  return 4 * (act * sizeof(float) + 256) + 3 * (act * 2 + 256) + (size_t)B * H * W * 8 * 2 + (size_t)B * 512 * sizeof(float) +
         CONV_SCRATCH + 8192;
}

namespace {
// bf16 flow of the BatchNorm-folded torchvision ResNet: operands bf16 (x16 / t16), residual stream kept in fp32 beside it
int tv_fwd_bf16(const avlen_resnet18* net, const float* x, int B, int H, int W, int C, float* out, int ld_out, void* ws,
                size_t ws_bytes, hipStream_t st, int f16 = 0) {
  WsBump w(ws, ws_bytes);
  const size_t act = (size_t)B * ((H + 1) / 2 + 1) * ((W + 1) / 2 + 1) * 64;
  const avlen_conv& c1 = net->conv1;
  float* f32[4]; __bf16* h16[3];
  for (int i = 0; i < 4; i++) f32[i] = w.take<float>(act);
  for (int i = 0; i < 3; i++) h16[i] = w.take<__bf16>(act);
  __bf16* x16 = w.take<__bf16>((size_t)B * H * W * 8);
  float* pooled = w.take<float>((size_t)B * 512);
  void* gws = w.take<char>(CONV_SCRATCH);
  if (!w.ok() || c1.cin16 != 8) return AVLEN_ERR_WS;
  auto conv = [&](const avlen_conv& k, const __bf16* in, const float* res, float* o32, __bf16* o16, int h, int wd, int act_) {
    avlen_g2_opts go; go.f16 = f16;
    return avlen_conv2d_nhwc_h16(in, k.w16, k.b, res, o32, o16, nullptr, B, h, wd, k.cin16, k.cout, k.kh, k.kw, k.stride, k.pad,
                                 act_, gws, CONV_SCRATCH, st, &go);
  };
  TRY(avlen_cast_h16(x, C, x16, 8, (long)B * H * W, C, f16 ? 1 : 0, st));
  const int h1 = conv_out(H, c1), w1 = conv_out(W, c1);
  if (h1 <= 0 || w1 <= 0 || (size_t)B * h1 * w1 * c1.cout > act) return AVLEN_ERR_ARG;
  TRY(conv(c1, x16, nullptr, f32[0], nullptr, H, W, AVLEN_ACT_RELU));
  int h = (h1 + 2 - 3) / 2 + 1, wd = (w1 + 2 - 3) / 2 + 1, ch = c1.cout;
  float* cur32 = f32[1]; __bf16* cur16 = h16[0];
  {
    long tot = (long)B * h * wd * ch;
    hipLaunchKernelGGL(maxpool_nhwc2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, f32[0], cur32, cur16, B, h1, w1,
                       ch, h, wd, 3, 2, 1, f16);
    TRY(avlen_launch_status());
  }
  float* nxt32 = f32[0]; float* down32 = f32[2]; __bf16* t16 = h16[1]; __bf16* nxt16 = h16[2];
  for (int i = 0; i < 8; i++) {
    const avlen_resblock& k = net->block[i];
    if (k.conv1.cin != ch || k.conv1.cin16 != ch || !k.conv1.b || !k.conv2.b) return AVLEN_ERR_ARG;
    const int oh = conv_out(h, k.conv1), ow = conv_out(wd, k.conv1), co = k.conv1.cout;
    if (oh <= 0 || ow <= 0 || (size_t)B * oh * ow * co > act || co > 512) return AVLEN_ERR_ARG;
    TRY(conv(k.conv1, cur16, nullptr, nullptr, t16, h, wd, AVLEN_ACT_RELU));
    const float* identity = cur32;
    if (k.has_down) {
      TRY(conv(k.down, cur16, nullptr, down32, nullptr, h, wd, 0));
      identity = down32;
    }
    TRY(conv(k.conv2, t16, identity, nxt32, nxt16, oh, ow, AVLEN_ACT_RELU_POST));
    float* o32 = cur32; cur32 = nxt32; nxt32 = o32;
    __bf16* o16 = cur16; cur16 = nxt16; nxt16 = o16;
    h = oh; wd = ow; ch = co;
  }
  if (net->fc.in_f != ch) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(B), dim3(256), 0, st, cur32, pooled, h * wd, ch);
  TRY(avlen_launch_status());
  return avlen_gemm(pooled, ch, 0, net->fc.w, ch, 0, out, ld_out, net->fc.b, nullptr, 0, B, net->fc.out_f, ch, 0,
                    f16 ? AVLEN_PREC_BF16X3 : AVLEN_PREC_BF16, 1, 0.f, nullptr, 0, st);
}
}  // namespace

extern "C" int avlen_resnet18_tv_fwd(const avlen_resnet18* net, const float* x, int B, int H, int W, int C, float* out,
                                     int ld_out, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!net || !x || !out || B <= 0 || C != net->conv1.cin || !net->conv1.b || ws_bytes < avlen_resnet18_tv_workspace_bytes(B, H, W))
    return AVLEN_ERR_WS;
  if (prec == AVLEN_PREC_BF16 || prec == AVLEN_PREC_FP16) {
    bool ok = conv16_ok(net->conv1);
    for (int i = 0; i < 8 && ok; i++)
      ok = conv16_ok(net->block[i].conv1) && conv16_ok(net->block[i].conv2) && (!net->block[i].has_down || conv16_ok(net->block[i].down));
    if (ok) return tv_fwd_bf16(net, x, B, H, W, C, out, ld_out, ws, ws_bytes, st, prec == AVLEN_PREC_FP16);
    if (prec == AVLEN_PREC_FP16) return AVLEN_ERR_ARG;
  }
  WsBump w(ws, ws_bytes);
  const size_t act = (size_t)B * ((H + 1) / 2 + 1) * ((W + 1) / 2 + 1) * 64;
  float* buf[4];
  for (int i = 0; i < 4; i++) buf[i] = w.take<float>(act);
  float* pooled = w.take<float>((size_t)B * 512);
  const avlen_conv& c1 = net->conv1;
  const int h1 = conv_out(H, c1), w1 = conv_out(W, c1);
  if (h1 <= 0 || w1 <= 0 || (size_t)B * h1 * w1 * c1.cout > act) return AVLEN_ERR_ARG;
  TRY(avlen_conv2d_nhwc(x, c1.w, c1.b, nullptr, buf[0], B, H, W, c1.cin, c1.cout, c1.kh, c1.kw, c1.stride, c1.pad, AVLEN_ACT_RELU,
                        prec, st));
  int h = (h1 + 2 - 3) / 2 + 1, wd = (w1 + 2 - 3) / 2 + 1, ch = c1.cout;
  {
    long tot = (long)B * h * wd * ch;
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, buf[0], buf[1], B, h1, w1, ch, h,
                       wd, 3, 2, 1);
    TRY(avlen_launch_status());
  }
  float* cur = buf[1]; float* t1 = buf[0]; float* t2 = buf[2]; float* t3 = buf[3];
  for (int i = 0; i < 8; i++) {
    const avlen_resblock& k = net->block[i];
    if (k.conv1.cin != ch || !k.conv1.b || !k.conv2.b) return AVLEN_ERR_ARG;
    const int oh = conv_out(h, k.conv1), ow = conv_out(wd, k.conv1), co = k.conv1.cout;
    if (oh <= 0 || ow <= 0 || (size_t)B * oh * ow * co > act || co > 512) return AVLEN_ERR_ARG;
    TRY(avlen_conv2d_nhwc(cur, k.conv1.w, k.conv1.b, nullptr, t1, B, h, wd, ch, co, k.conv1.kh, k.conv1.kw, k.conv1.stride,
                          k.conv1.pad, AVLEN_ACT_RELU, prec, st));
    const float* identity = cur;
    if (k.has_down) {
      TRY(avlen_conv2d_nhwc(cur, k.down.w, k.down.b, nullptr, t3, B, h, wd, ch, co, k.down.kh, k.down.kw, k.down.stride,
                            k.down.pad, 0, prec, st));
      identity = t3;
    }
    TRY(avlen_conv2d_nhwc(t1, k.conv2.w, k.conv2.b, identity, t2, B, oh, ow, co, co, k.conv2.kh, k.conv2.kw, k.conv2.stride,
                          k.conv2.pad, AVLEN_ACT_RELU_POST, prec, st));
    float* o = cur; cur = t2; t2 = o;
    h = oh; wd = ow; ch = co;
  }
  if (net->fc.in_f != ch) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(B), dim3(256), 0, st, cur, pooled, h * wd, ch);
  TRY(avlen_launch_status());
  return avlen_gemm(pooled, ch, 0, net->fc.w, ch, 0, out, ld_out, net->fc.b, nullptr, 0, B, net->fc.out_f, ch, 0, prec, 1, 0.f,
                    nullptr, 0, st);
}

extern "C" int avlen_belief_input(const float* spec, const float* category, float* out, int B, int HW, int Cs, int Cc,
                                  hipStream_t st) {
  if (!spec || !category || !out || B <= 0) return AVLEN_ERR_ARG;
  long tot = (long)B * HW * (Cs + Cc);
  hipLaunchKernelGGL(belief_input_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, spec, category, out, B, HW, Cs, Cc);
  return avlen_launch_status();
}

extern "C" int avlen_belief_update(const float* pointgoals, int ld_pg, const float* labels, int ld_lab, const float* pose,
                                   int ld_pose, const float* spectrogram, long spec_elems, const unsigned char* dones,
                                   float* last_pointgoal, int* has_pointgoal, float* last_label, int* has_label,
                                   float* location_belief, float* category_belief, float* spec_sum, int B, int n_label,
                                   float weighting_factor, int current_pred_only, hipStream_t st) {
  if (B <= 0 || !spectrogram || !spec_sum || (!pointgoals && !labels)) return AVLEN_ERR_ARG;
  if (pointgoals && (!pose || !last_pointgoal || !has_pointgoal || !location_belief)) return AVLEN_ERR_ARG;
  if (labels && (!last_label || !has_label || !category_belief)) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(spec_sum_kernel, dim3(B), dim3(256), 0, st, spectrogram, spec_sum, spec_elems);
  hipLaunchKernelGGL(belief_update_kernel, dim3((B + 63) / 64), dim3(64), 0, st, pointgoals, ld_pg, labels, ld_lab, pose, ld_pose,
                     (const float*)spec_sum, dones, last_pointgoal, has_pointgoal, last_label, has_label, location_belief,
                     category_belief, B, n_label, weighting_factor, current_pred_only);
  return avlen_launch_status();
}
