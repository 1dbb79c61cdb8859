// Policy heads, PPO loss (+ its backward through the heads), GAE, clip-norm + Adam, rollout-storage
// kernels.  All tiny, latency/HBM-bound: one wavefront per sample row, wave(64) shuffles for the
// 256-wide dot products and for the loss reductions.
#include "common.h"
#include "../../include/avlen_hip.h"

namespace {

constexpr int MAXA = 8;      // max actions per categorical head

// dot(feats[row], W[j]) for j < nout, each over d (d % 64 == 0, d <= 1024): lane-strided, wave reduce
__device__ __forceinline__ float row_dot(const float* __restrict__ f, const float* __restrict__ w, int d, int lane) {
  float s = 0.f;
  for (int i = lane; i < d; i += 64) s += f[i] * w[i];
  return wave_sum(s);
}

__global__ __launch_bounds__(256) void heads_fwd_kernel(avlen_heads h, const float* __restrict__ feats, int d, int A,
                                                        float* __restrict__ logits, float* __restrict__ probs,
                                                        float* __restrict__ value, float* __restrict__ unct,
                                                        const int64_t* __restrict__ actions, float* __restrict__ log_prob,
                                                        float* __restrict__ entropy, int B,
                                                        const float* __restrict__ noise = nullptr, int64_t* __restrict__ action_out = nullptr,
                                                        int64_t* __restrict__ action_host = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* f = feats + (long)row * d;
  // all MAXA + 3 dot products share one pass over the feature row: every weight load is issued before the first
  // reduction (independent accumulators), then the wave reductions run back to back
  constexpr int NO = MAXA + 3;
  const float* wp[NO];
#pragma unroll
  for (int a = 0; a < MAXA; a++) wp[a] = h.action.w + (long)(a < A ? a : 0) * d;
  wp[MAXA] = h.critic.w;
  wp[MAXA + 1] = h.has_unct ? h.unct.w : h.critic.w;
  wp[MAXA + 2] = h.has_unct ? h.unct.w + d : h.critic.w;
  float acc[NO];
#pragma unroll
  for (int o = 0; o < NO; o++) acc[o] = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float x = f[i];
#pragma unroll
    for (int o = 0; o < NO; o++) acc[o] += x * wp[o][i];
  }
#pragma unroll
  for (int o = 0; o < NO; o++) acc[o] = wave_sum(acc[o]);
  float z[MAXA];
  float mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < MAXA; a++) {
    z[a] = -INFINITY;
    if (a < A) { z[a] = acc[a] + h.action.b[a]; mx = fmaxf(mx, z[a]); }
  }
  float se = 0.f;
#pragma unroll
  for (int a = 0; a < MAXA; a++) if (a < A) se += expf(z[a] - mx);
  float lse = mx + logf(se);
  float v = acc[MAXA] + h.critic.b[0];
  float u0 = 0.f, u1 = 0.f;
  if (h.has_unct) { u0 = acc[MAXA + 1] + h.unct.b[0]; u1 = acc[MAXA + 2] + h.unct.b[1]; }
  if (lane == 0) {
    float ent = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; a++) {
      if (a < A) {
        float lp = z[a] - lse, p = expf(lp);
        ent -= p * lp;
        if (logits) logits[(long)row * A + a] = z[a];
        if (probs) probs[(long)row * A + a] = p;
      }
    }
    if (value) value[row] = v;
    if (unct && h.has_unct) { unct[(long)row * 2] = u0; unct[(long)row * 2 + 1] = u1; }
    if (entropy) entropy[row] = ent;
    long sampled = -1;
    if (noise && action_out) {
      // the exponential race of CustomFixedCategorical.sample on host-drawn noise (see avlen_sample_race: same quotients, same
      // first-maximum rule, on the probabilities just written) -- fused here so that a rollout forward needs no second pass
      float best = 0.f; int bi = 0;
#pragma unroll
      for (int a = 0; a < MAXA; a++)
        if (a < A) {
          const float v_ = __fdiv_rn(expf(z[a] - lse), noise[(long)row * A + a]);
          if (a == 0 || v_ > best || (v_ != v_ && best == best)) { best = v_; bi = a; }
        }
      sampled = bi;
      action_out[row] = bi;
      if (action_host) action_host[row] = bi;          // pinned host memory, mapped: the simulator's copy needs no second launch
    }
    if ((actions || sampled >= 0) && log_prob) {
      long a = sampled >= 0 ? sampled : actions[row];
      float za = 0.f;
#pragma unroll
      for (int k = 0; k < MAXA; k++) if (k == a) za = z[k];
      log_prob[row] = za - lse;
    }
  }
}

// PPO loss + backward through the heads.  One wave per row.  Head-parameter gradients are reduced over
// the block's 4 rows in LDS, then one atomicAdd per element per block.
__global__ __launch_bounds__(256) void ppo_loss_kernel(avlen_heads h, avlen_heads g, const float* __restrict__ feats, int d,
                                                       int A, const int64_t* __restrict__ actions,
                                                       const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                       const int64_t* __restrict__ rl_masks,
                                                       const float* __restrict__ value_preds,
                                                       const float* __restrict__ returns, const int64_t* __restrict__ unct_gt,
                                                       const float* __restrict__ norm, float clip, float vc, float ec,
                                                       float uc, float* __restrict__ loss_sums, float* __restrict__ d_feats,
                                                       int B, int rows_per_block) {
  __shared__ float s_dz[4][MAXA + 4];          // per-wave: dz[A], dv, du0, du1
  __shared__ float s_loss[4][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float inv_rl = norm[0], inv_R = norm[1];
  const int nout = A + 1 + (h.has_unct ? 2 : 0);
  float lacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int r0 = blockIdx.x * rows_per_block, r1 = min(B, r0 + rows_per_block);
  // per-thread accumulators for dW: thread (w,lane) owns columns lane + 64*i of every head row; we loop
  // head rows inside, so keep [nout][d/64] partials in registers only for d <= 256 -> use LDS staging instead
  for (int base = r0; base < r1; base += 4) {
    const int row = base + w;
    const bool ok = row < r1;
    float dz[MAXA]; float dv = 0.f, du0 = 0.f, du1 = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; a++) dz[a] = 0.f;
    if (ok) {
      const float* f = feats + (long)row * d;
      float z[MAXA]; float mx = -INFINITY;
#pragma unroll
      for (int a = 0; a < MAXA; a++) {
        z[a] = -INFINITY;
        if (a < A) { z[a] = row_dot(f, h.action.w + (long)a * d, d, lane) + h.action.b[a]; mx = fmaxf(mx, z[a]); }
      }
      float se = 0.f;
#pragma unroll
      for (int a = 0; a < MAXA; a++) if (a < A) se += expf(z[a] - mx);
      const float lse = mx + logf(se);
      const float v = row_dot(f, h.critic.w, d, lane) + h.critic.b[0];
      float u0 = 0.f, u1 = 0.f;
      if (h.has_unct) {
        u0 = row_dot(f, h.unct.w, d, lane) + h.unct.b[0];
        u1 = row_dot(f, h.unct.w + d, d, lane) + h.unct.b[1];
      }
      // ---- scalar loss algebra (replicated in every lane; cheap)
      const long act = actions[row];
      float p[MAXA], lp[MAXA]; float ent = 0.f, lpa = 0.f;
#pragma unroll
      for (int a = 0; a < MAXA; a++) {
        p[a] = 0.f; lp[a] = 0.f;
        if (a < A) { lp[a] = z[a] - lse; p[a] = expf(lp[a]); ent -= p[a] * lp[a]; if (a == act) lpa = lp[a]; }
      }
      const float m = (float)rl_masks[row];
      const float ad = adv[row];
      const float ratio = expf(lpa - old_lp[row]);
      const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
      const float surr1 = ratio * ad * m, surr2 = rc * ad * m;
      const float action_term = -fminf(surr1, surr2) * inv_rl;
      // d(min)/d ratio with torch's tie rule (half / half) and clamp's inclusive pass-through
      const float w1 = surr1 < surr2 ? 1.f : (surr1 == surr2 ? 0.5f : 0.f);
      const float w2 = (surr2 < surr1 ? 1.f : (surr1 == surr2 ? 0.5f : 0.f)) *
                       ((ratio >= 1.f - clip && ratio <= 1.f + clip) ? 1.f : 0.f);
      const float dlpa = -(w1 + w2) * ad * m * ratio * inv_rl;       // d total / d logp[action]
      const float vp = value_preds[row], ret = returns[row];
      const float dvp = v - vp;
      const float vclip = vp + fminf(fmaxf(dvp, -clip), clip);
      const float l1 = (v - ret) * (v - ret), l2 = (vclip - ret) * (vclip - ret);
      const float value_term = 0.5f * fmaxf(l1, l2) * inv_R;
      const float t1 = l1 > l2 ? 1.f : (l1 == l2 ? 0.5f : 0.f);
      const float t2 = (l2 > l1 ? 1.f : (l1 == l2 ? 0.5f : 0.f)) * ((dvp >= -clip && dvp <= clip) ? 1.f : 0.f);
      dv = vc * 0.5f * inv_R * (t1 * 2.f * (v - ret) + t2 * 2.f * (vclip - ret));
      float unct_term = 0.f;
      if (h.has_unct) {
        const float um = fmaxf(u0, u1);
        const float ul = um + logf(expf(u0 - um) + expf(u1 - um));
        const long gt = unct_gt[row];
        const float q0 = expf(u0 - ul), q1 = expf(u1 - ul);
        unct_term = -((gt == 0 ? u0 : u1) - ul) * inv_R;
        du0 = uc * inv_R * (q0 - (gt == 0 ? 1.f : 0.f));
        du1 = uc * inv_R * (q1 - (gt == 1 ? 1.f : 0.f));
      }
      // total = vc*Lv + La - ec*H + uc*Lu ;  H = mean_rows(ent)
#pragma unroll
      for (int a = 0; a < MAXA; a++) {
        if (a < A) {
          float dH = -p[a] * (lp[a] + ent) * inv_R;             // d(mean entropy)/dz_a
          float dl = dlpa * ((a == act ? 1.f : 0.f) - p[a]);    // through logp[action]
          dz[a] = dl - ec * dH;
        }
      }
      if (lane == 0) {
        lacc[0] += value_term; lacc[1] += action_term; lacc[2] += ent * inv_R;
        lacc[3] += v * inv_R; lacc[4] += ret * inv_R; lacc[5] += unct_term;
      }
      // ---- d_feats[row] = sum_j dout_j * W_j
      float* df = d_feats + (long)row * d;
      for (int i = lane; i < d; i += 64) {
        float s = dv * h.critic.w[i];
#pragma unroll
        for (int a = 0; a < MAXA; a++) if (a < A) s += dz[a] * h.action.w[(long)a * d + i];
        if (h.has_unct) s += du0 * h.unct.w[i] + du1 * h.unct.w[d + i];
        df[i] = s;
      }
    }
    // ---- head parameter gradients for these (up to) 4 rows
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < MAXA; a++) s_dz[w][a] = dz[a];
      s_dz[w][MAXA] = dv; s_dz[w][MAXA + 1] = du0; s_dz[w][MAXA + 2] = du1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nout * d; e += 256) {
      int j = e / d, i = e - j * d;
      float s = 0.f;
      for (int ww = 0; ww < 4; ww++) {
        int rr = base + ww;
        if (rr < r1) {
          float dj = j < A ? s_dz[ww][j] : (j == A ? s_dz[ww][MAXA] : s_dz[ww][MAXA + 1 + (j - A - 1)]);
          s += dj * feats[(long)rr * d + i];
        }
      }
      float* dst = j < A ? g.action.w + (long)j * d + i : (j == A ? g.critic.w + i : g.unct.w + (long)(j - A - 1) * d + i);
      atomicAdd(dst, s);
    }
    if (threadIdx.x < nout) {
      int j = threadIdx.x;
      float s = 0.f;
      for (int ww = 0; ww < 4; ww++)
        if (base + ww < r1) s += j < A ? s_dz[ww][j] : (j == A ? s_dz[ww][MAXA] : s_dz[ww][MAXA + 1 + (j - A - 1)]);
      float* dst = j < A ? g.action.b + j : (j == A ? g.critic.b : g.unct.b + (j - A - 1));
      atomicAdd(dst, s);
    }
    __syncthreads();
  }
  if (lane == 0) for (int k = 0; k < 6; k++) s_loss[w][k] = lacc[k];
  __syncthreads();
  if (threadIdx.x < 6) {
    int k = threadIdx.x;
    atomicAdd(&loss_sums[k], s_loss[0][k] + s_loss[1][k] + s_loss[2][k] + s_loss[3][k]);
  }
}

__global__ void rl_mask_norm_kernel(const int64_t* __restrict__ m, int R, float* __restrict__ norm) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < R; i += blockDim.x) s += (float)m[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) { norm[0] = 1.f / s; norm[1] = 1.f / (float)R; }
}

// One lane per environment: the 150-step backward recurrence is sequential in t, parallel in envs.
__global__ void gae_kernel(const float* __restrict__ rewards, float* __restrict__ values, const float* __restrict__ masks,
                           const float* __restrict__ next_value, float* __restrict__ returns, float* __restrict__ adv, int T,
                           int N, float gamma, float tau) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  values[(long)T * N + n] = next_value[n];
  float gae = 0.f;
  float vnext = next_value[n];
  for (int t = T - 1; t >= 0; t--) {
    float v = values[(long)t * N + n], mk = masks[(long)(t + 1) * N + n];
    float delta = rewards[(long)t * N + n] + gamma * vnext * mk - v;
    gae = delta + gamma * tau * mk * gae;
    float r = gae + v;
    returns[(long)t * N + n] = r;
    if (adv) adv[(long)t * N + n] = r - v;
    vnext = v;
  }
}

// use_gae=False branch (rollout_storage.py:406-412): returns[T] = next_value; returns[t] = returns[t+1] * gamma * masks[t+1] + r[t]
__global__ void returns_kernel(const float* __restrict__ rewards, const float* __restrict__ masks, const float* __restrict__ next_value,
                               float* __restrict__ returns, int T, int N, float gamma) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float r = next_value[n];
  returns[(long)T * N + n] = r;
  for (int t = T - 1; t >= 0; t--) {
    r = r * gamma * masks[(long)(t + 1) * N + n] + rewards[(long)t * N + n];
    returns[(long)t * N + n] = r;
  }
}

__global__ void sumsq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, (double)s);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float max_norm,
                            const double* __restrict__ norm_sq) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = 1.f;
  if (norm_sq) {
    float tot = (float)sqrt(*norm_sq);
    coef = fminf(max_norm / (tot + 1e-6f), 1.f);
  }
  float gi = g[i] * coef;
  float mi = m[i] * b1 + gi * (1.f - b1);
  float vi = v[i] * b2 + gi * gi * (1.f - b2);
  m[i] = mi; v[i] = vi;
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

__global__ void extmem_insert_kernel(float* __restrict__ memory, float* __restrict__ masks, const float* __restrict__ feats,
                                     int ld_feats, const float* __restrict__ not_done, float* __restrict__ masks_out, int idx,
                                     int total, int capacity, int N, int dim) {
  const int n = blockIdx.x;
  __shared__ float sh[16];
  __shared__ int s_over;
  for (int i = threadIdx.x; i < dim; i += blockDim.x) memory[((long)idx * N + n) * dim + i] = feats[(long)n * ld_feats + i];
  float s = 0.f;
  for (int i = threadIdx.x; i < total; i += blockDim.x) s += masks[(long)n * total + i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) s_over = (s == (float)capacity);
  __syncthreads();
  const int drop = (idx - capacity + total) % total;
  const float nd = not_done[n];
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    float mv = masks[(long)n * total + i];
    if (s_over && i == drop) mv = 0.f;
    if (i == idx) mv = 1.f;
    mv *= nd;
    masks[(long)n * total + i] = mv;
    if (masks_out) masks_out[(long)n * total + i] = mv;
  }
}

// All of a step's external-memory rings in one launch (blockIdx.y = ring)
struct ExtMemOps { avlen_extmem_op op[4]; };
__global__ void extmem_insert_multi_kernel(ExtMemOps ops) {
  const avlen_extmem_op o = ops.op[blockIdx.y];
  const int n = blockIdx.x;
  if (n >= o.N) return;
  __shared__ float sh[16];
  __shared__ int s_over;
  for (int i = threadIdx.x; i < o.dim; i += blockDim.x) o.memory[((long)o.idx * o.N + n) * o.dim + i] = o.feats[(long)n * o.ld_feats + i];
  float s = 0.f;
  for (int i = threadIdx.x; i < o.total; i += blockDim.x) s += o.masks[(long)n * o.total + i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) s_over = (s == (float)o.capacity);
  __syncthreads();
  const int drop = (o.idx - o.capacity + o.total) % o.total;
  const float nd = o.not_done[n];
  for (int i = threadIdx.x; i < o.total; i += blockDim.x) {
    float mv = o.masks[(long)n * o.total + i];
    if (s_over && i == drop) mv = 0.f;
    if (i == o.idx) mv = 1.f;
    mv *= nd;
    o.masks[(long)n * o.total + i] = mv;
    if (o.masks_out) o.masks_out[(long)n * o.total + i] = mv;
  }
}

template <typename T>
__global__ void gather_kernel(const T* __restrict__ src, T* __restrict__ dst, const int64_t* __restrict__ env, int N, int n_mb,
                              size_t D, size_t tot) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tot) return;
  size_t r = i / D, c = i - r * D;
  size_t t = r / n_mb, j = r - t * n_mb;
  dst[i] = src[(t * N + (size_t)env[j]) * D + c];
}

}  // namespace

extern "C" int avlen_heads_fwd(const avlen_heads* h, const float* feats, int d, int A, float* logits, float* probs,
                               float* value, float* unct, const int64_t* actions, float* log_prob, float* entropy, int B,
                               hipStream_t stream) {
  if (!h || B <= 0 || A > MAXA || d % 64) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, stream, *h, feats, d, A, logits, probs, value,
                     unct, actions, log_prob, entropy, B);
  return avlen_launch_status();
}

// avlen_heads_fwd with the sampling fused in: action_out[b] = the race's winner on `noise` (B x A, host-drawn Exp(1)), log_prob /
// entropy of that action; everything else as avlen_heads_fwd.
extern "C" int avlen_heads_act_host_fwd(const avlen_heads* h, const float* feats, int d, int A, float* logits, float* probs, float* value,
                                        float* unct, const float* noise, int64_t* action_out, int64_t* action_host, float* log_prob,
                                        float* entropy, int B, hipStream_t stream) {
  if (!h || !noise || !action_out || B <= 0 || A > MAXA || d % 64) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, stream, *h, feats, d, A, logits, probs, value, unct,
                     (const int64_t*)nullptr, log_prob, entropy, B, noise, action_out, action_host);
  return avlen_launch_status();
}
extern "C" int avlen_heads_act_fwd(const avlen_heads* h, const float* feats, int d, int A, float* logits, float* probs, float* value,
                                   float* unct, const float* noise, int64_t* action_out, float* log_prob, float* entropy, int B,
                                   hipStream_t stream) {
  return avlen_heads_act_host_fwd(h, feats, d, A, logits, probs, value, unct, noise, action_out, nullptr, log_prob, entropy, B, stream);
}

// CustomFixedCategorical.sample (common/utils.py:48-49 -> torch.multinomial, one draw per row) is the exponential race
// argmax_a(p[a] / q[a]) with q ~ Exp(1) drawn from the HOST generator.  The noise does not depend on the
// probabilities: the host draws it in the reference's order and uploads it, the race runs here -- IEEE fp32 division (-ffp-contract
// off, no fast-math: the same quotient bits as the host's `probs / q`), first maximum wins (torch.argmax on the host) -- so the
// action is the reference's for the same generator state, without the probabilities ever leaving the device.
namespace {
__global__ void sample_race_kernel(const float* __restrict__ probs, const float* __restrict__ noise, int64_t* __restrict__ action,
                                   int B, int A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float best = __fdiv_rn(probs[(long)b * A], noise[(long)b * A]);
  int bi = 0;
  for (int a = 1; a < A; a++) {
    const float v = __fdiv_rn(probs[(long)b * A + a], noise[(long)b * A + a]);
    if (v > best || (v != v && best == best)) { best = v; bi = a; }      // NaN ranks highest, as in torch.argmax
  }
  action[b] = bi;
}
}  // namespace
extern "C" int avlen_sample_race(const float* probs, const float* noise, int64_t* action, int B, int A, hipStream_t stream) {
  if (!probs || !noise || !action || B <= 0 || A <= 0) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(sample_race_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, stream, probs, noise, action, B, A);
  return avlen_launch_status();
}

extern "C" int avlen_ppo_loss_heads_bwd(const avlen_heads* h, const avlen_heads* g, const float* feats, int d, int A,
                                        const int64_t* actions, const float* old_log_probs, const float* adv,
                                        const int64_t* rl_masks, const float* value_preds, const float* returns,
                                        const int64_t* unct_gt, const float* norm, float clip, float value_coef,
                                        float entropy_coef, float unct_coef, float* loss_sums, float* d_feats, int B,
                                        hipStream_t stream) {
  if (!h || !g || B <= 0 || A > MAXA || d % 64) return AVLEN_ERR_ARG;
  int rpb = B >= 16384 ? 64 : B >= 1024 ? 16 : 4;
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(ceil_div(B, rpb)), dim3(256), 0, stream, *h, *g, feats, d, A, actions,
                     old_log_probs, adv, rl_masks, value_preds, returns, unct_gt, norm, clip, value_coef, entropy_coef,
                     unct_coef, loss_sums, d_feats, B, rpb);
  return avlen_launch_status();
}

// ---- PPO.update_dialog's loss (ppo.py:139-145): CrossEntropyLoss(weight=w) between the vln action logits of the rows with
// o_masks != 0 and their oracle actions:  loss = sum_i w[y_i] * (-log p_i[y_i]) / sum_i w[y_i].
// Pass 1: norm[0] = sum_i m_i w[y_i].  Pass 2: logits = feats * W^T + b, loss, d logits = m_i w[y_i] (p - onehot) / norm,
// head gradients (accumulated with atomics), d_feats = d logits * W.
namespace {
__global__ void dialog_norm_kernel(const float* __restrict__ o_actions, const int64_t* __restrict__ o_masks, const float* __restrict__ wcls,
                                   int A, float* __restrict__ norm, int R) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < R; i += blockDim.x)
    if (o_masks[i] != 0) { const int y = (int)o_actions[i]; if (y >= 0 && y < A) s += wcls[y]; }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) norm[0] = s;
}
__global__ __launch_bounds__(256) void dialog_loss_kernel(avlen_heads h, avlen_heads g, const float* __restrict__ feats, int d, int A,
                                                          const float* __restrict__ o_actions, const int64_t* __restrict__ o_masks,
                                                          const float* __restrict__ wcls, const float* __restrict__ norm,
                                                          float* __restrict__ loss, float* __restrict__ d_feats, int R) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= R) return;
  const float* f = feats + (long)row * d;
  float z[MAXA], mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < MAXA; a++) {
    z[a] = -INFINITY;
    if (a < A) { z[a] = row_dot(f, h.action.w + (long)a * d, d, lane) + h.action.b[a]; mx = fmaxf(mx, z[a]); }
  }
  float se = 0.f;
#pragma unroll
  for (int a = 0; a < MAXA; a++) if (a < A) se += expf(z[a] - mx);
  const float lse = mx + logf(se);
  const bool on = o_masks[row] != 0;
  const int y = (int)o_actions[row];
  const float wy = (on && y >= 0 && y < A) ? wcls[y] : 0.f;
  const float inv = norm[0] > 0.f ? 1.f / norm[0] : 0.f;
  float dz[MAXA];
#pragma unroll
  for (int a = 0; a < MAXA; a++) dz[a] = a < A ? wy * inv * (expf(z[a] - lse) - (a == y ? 1.f : 0.f)) : 0.f;
  float zy = 0.f;
#pragma unroll
  for (int a = 0; a < MAXA; a++) if (a == y) zy = z[a];
  if (lane == 0 && wy != 0.f) atomicAdd(loss, -wy * inv * (zy - lse));
  // head gradients and d_feats: lanes stride the feature columns
  for (int k = lane; k < d; k += 64) {
    const float fk = f[k];
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; a++)
      if (a < A) {
        acc += dz[a] * h.action.w[(long)a * d + k];
        if (wy != 0.f) atomicAdd(&g.action.w[(long)a * d + k], dz[a] * fk);
      }
    d_feats[(long)row * d + k] = acc;
  }
  if (wy != 0.f) {
#pragma unroll
    for (int a = 0; a < MAXA; a++) if (a < A && lane == a) atomicAdd(&g.action.b[a], dz[a]);
  }
}
}  // namespace

extern "C" int avlen_dialog_loss_heads_bwd(const avlen_heads* h, const avlen_heads* g, const float* feats, int d, int A,
                                           const float* o_actions, const int64_t* o_masks, const float* class_weights, float* norm,
                                           float* loss, float* d_feats, int R, hipStream_t stream) {
  if (!h || !g || !feats || !o_actions || !o_masks || !class_weights || !norm || !loss || !d_feats || R <= 0 || A > MAXA) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(dialog_norm_kernel, dim3(1), dim3(1024), 0, stream, o_actions, o_masks, class_weights, A, norm, R);
  hipLaunchKernelGGL(dialog_loss_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, stream, *h, *g, feats, d, A, o_actions, o_masks,
                     class_weights, norm, loss, d_feats, R);
  return avlen_launch_status();
}

extern "C" int avlen_rl_mask_norm(const int64_t* rl_masks, int R, float* norm, hipStream_t stream) {
  hipLaunchKernelGGL(rl_mask_norm_kernel, dim3(1), dim3(1024), 0, stream, rl_masks, R, norm);
  return avlen_launch_status();
}

extern "C" int avlen_gae_scan(const float* rewards, float* values, const float* masks, const float* next_value,
                              float* returns, float* advantages, int T_used, int N, float gamma, float tau,
                              hipStream_t stream) {
  if (T_used <= 0 || N <= 0) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(gae_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, stream, rewards, values, masks, next_value, returns,
                     advantages, T_used, N, gamma, tau);
  return avlen_launch_status();
}

extern "C" int avlen_discounted_returns(const float* rewards, const float* masks, const float* next_value, float* returns, int T_used,
                                        int N, float gamma, hipStream_t stream) {
  if (T_used <= 0 || N <= 0) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(returns_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, stream, rewards, masks, next_value, returns, T_used, N, gamma);
  return avlen_launch_status();
}

extern "C" int avlen_grad_sumsq(const float* grad, size_t n, double* norm_sq, hipStream_t stream) {
  if (!n) return AVLEN_OK;
  int blocks = (int)((n + 255) / 256); if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, grad, n, norm_sq);
  return avlen_launch_status();
}

extern "C" int avlen_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                               float beta1, float beta2, float eps, int step, float max_grad_norm, const double* norm_sq,
                               hipStream_t stream) {
  if (!n) return AVLEN_OK;
  float bc1 = 1.f - powf(beta1, (float)step), bc2 = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, bc1, bc2, max_grad_norm, norm_sq);
  return avlen_launch_status();
}

extern "C" int avlen_extmem_insert(float* memory, float* masks, const float* feats, int ld_feats, const float* not_done,
                                   float* masks_out, int idx, int total, int capacity, int N, int dim, hipStream_t stream) {
  if (idx < 0 || idx >= total || N <= 0) return AVLEN_ERR_ARG;
  hipLaunchKernelGGL(extmem_insert_kernel, dim3(N), dim3(256), 0, stream, memory, masks, feats, ld_feats, not_done, masks_out,
                     idx, total, capacity, N, dim);
  return avlen_launch_status();
}

extern "C" int avlen_extmem_insert_multi(const avlen_extmem_op* ops, int n, hipStream_t stream) {
  if (!ops || n < 1 || n > 4) return AVLEN_ERR_ARG;
  ExtMemOps o = {};
  int maxn = 0;
  for (int i = 0; i < n; i++) {
    if (ops[i].idx < 0 || ops[i].idx >= ops[i].total || ops[i].N <= 0) return AVLEN_ERR_ARG;
    o.op[i] = ops[i];
    if (ops[i].N > maxn) maxn = ops[i].N;
  }
  hipLaunchKernelGGL(extmem_insert_multi_kernel, dim3(maxn, n), dim3(256), 0, stream, o);
  return avlen_launch_status();
}

extern "C" int avlen_minibatch_gather(const void* src, void* dst, const int64_t* env, int T, int N, int n_mb, size_t D,
                                      int elem_bytes, hipStream_t stream) {
  size_t tot = (size_t)T * n_mb * D;
  if (!tot) return AVLEN_OK;
  if (elem_bytes == 4 && D % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    size_t t4 = tot / 4;
    hipLaunchKernelGGL((gather_kernel<float4>), dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, stream, (const float4*)src,
                       (float4*)dst, env, N, n_mb, D / 4, t4);
  } else if (elem_bytes == 4) {
    hipLaunchKernelGGL((gather_kernel<float>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, (const float*)src,
                       (float*)dst, env, N, n_mb, D, tot);
  } else if (elem_bytes == 8) {
    hipLaunchKernelGGL((gather_kernel<int64_t>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                       (const int64_t*)src, (int64_t*)dst, env, N, n_mb, D, tot);
  } else return AVLEN_ERR_ARG;
  return avlen_launch_status();
}
