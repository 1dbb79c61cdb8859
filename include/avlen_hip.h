/* avlen_hip.h -- C ABI of libavlen_hip.so: the MI355X (gfx950) implementation of the SAVi/AVLEN
 * PPO rollout-and-update hot path.
 *
 * The reference (merlresearch/avlen) has no FFI on this path: its boundary is the Python class API
 * of ss_baselines/savi/ppo/policy.py (Policy.act* / evaluate_actions*), ss_baselines/savi/ppo/ppo.py
 * (PPO.update) and ss_baselines/savi/models/rollout_storage.py.  Each entry point below names the
 * reference code whose arithmetic it replaces (paths relative to the reference root).  The Python
 * classes in avlen_amd/ keep the reference's names/signatures and call these through ctypes.
 *
 * Conventions: every pointer is a DEVICE pointer unless named host_*; fp32 storage; images are NHWC;
 * sequences are batch-major [B][S][d]; functions enqueue work on `stream` and return immediately
 * (never allocate, never synchronise; graph-capturable); return AVLEN_OK (0) or an AVLEN_ERR_* code.
 * `prec` selects the MFMA operand type of the dense products: AVLEN_PREC_FP32 (exact fp32 MFMA,
 * the parity mode), AVLEN_PREC_BF16 (bf16 operands, fp32 accumulate) or AVLEN_PREC_BF16X3 (every operand split into
 * bf16 hi + lo, three bf16 MFMAs per product: fp32-grade results at matrix-core speed).  Scratch comes from the
 * caller: ask the matching *_workspace_bytes() first.
 */
#ifndef AVLEN_HIP_H
#define AVLEN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* avlen_stream_t;        /* == hipStream_t */

#define AVLEN_PREC_FP32 0
#define AVLEN_PREC_BF16 1
#define AVLEN_PREC_BF16X3 2   /* compensated bf16: operands split hi + lo, three bf16 MFMAs per product, fp32 accumulate */
#define AVLEN_PREC_FP16 3     /* IEEE half operands, fp32 accumulate: the frozen CLIP text tower (what the reference's CUDA path runs
                                 it in, clip.load -> fp16) and the AudioCNN of the bf16x3 mode; needs fp16 weight shadows */
#define AVLEN_ACT_NONE 0
#define AVLEN_ACT_RELU 1
#define AVLEN_ACT_QUICKGELU 2
#define AVLEN_ACT_RELU_POST 4   /* fp32-staged conv/GEMM only: ReLU AFTER the residual add (torchvision BasicBlock) */

/* ------------------------------------------------------------------ parameter views ---------- */
/* w[out_f][in_f] fp32 (canonical).  w16: optional bf16 shadow of the same matrix, row stride ld16 (multiple of 8,
 * padding zero) used by the bf16 fast path; NULL -> that layer runs on the fp32-staged kernel. */
/* w16lo: optional LOW plane of the compensated bf16 pair, bf16(w - bf16(w)), same layout as w16 (AVLEN_PREC_BF16X3 fast paths). */
typedef struct { float* w; float* b; int out_f; int in_f; void* w16; int ld16; void* w16lo; } avlen_linear;
/* w packed [cout][kh][kw][cin] fp32; w16: bf16 [cout][kh][kw][cin16] with cin16 = max(8, cin) zero-padded. */
/* w16c: optional compact bf16 copy [cout][kh][kw][cin] WITHOUT channel padding, for the "super-pixel" form of a conv whose
 * cin * stride == 8 and kw % stride == 0 (AudioCNN conv0 on the 257x101 spectrogram, audio_cnn.py:62-83): `stride`
 * horizontally adjacent pixels x cin channels form one 8-channel pixel, the conv becomes (kh, kw/stride), stride (stride, 1),
 * K = kh*kw*cin instead of kh*kw*8. */
/* w16f: optional copy of w16 in MFMA-fragment order for kernels that keep their weights in registers (tower_tail.hip): with
 * K = kh*kw*cin16 (a multiple of 32) and cout a multiple of 16, [cout/16][K/32][lane 0..63][8] bf16 where lane = 16 q + r holds
 * w16[16 t + r][32 i + 8 q .. + 7] -- one 16-byte load per lane, 1 KiB contiguous per wave and k-step. */
/* w16lo / w16flo: optional LOW planes of the compensated bf16 pair (bf16(w - bf16(w))) in the layouts of w16 / w16f. */
typedef struct { float* w; float* b; int cin, cout, kh, kw, stride, pad; void* w16; int cin16; void* w16c; void* w16f;
                 void* w16lo; void* w16flo; } avlen_conv;
typedef struct { float* g; float* b; } avlen_affine;                               /* norm scale / shift */
typedef struct { avlen_conv conv1, conv2, down; avlen_affine bn1, bn2, bnd; int has_down; } avlen_resblock;
/* CustomResNet (smt_resnet.py:56-149): conv7x7 + GroupNorm(16) + 8 basic blocks + fc(8192->64).
 * fc.w is packed so that its columns follow the NHWC flatten order (h,w,c). */
typedef struct { avlen_conv conv1; avlen_affine bn1; avlen_resblock block[8]; avlen_linear fc; } avlen_resnet18;
/* AudioCNN / VisualCNN (audio_cnn.py:62-94, visual_cnn.py:82-107): 3 convs (+bias, ReLU after the
 * first two) + fc + ReLU.  fc.w packed to the NHWC flatten order. */
/* half_fmt: format of the 16-bit weight shadows (w16, w16c, fc.w16): 0 = bf16, 1 = fp16 (AVLEN_PREC_FP16 calls). */
typedef struct { avlen_conv conv[3]; avlen_linear fc; int half_fmt; } avlen_cnn3;
typedef struct { avlen_linear in_proj, out_proj; } avlen_mha;                      /* packed q|k|v rows */
typedef struct { avlen_mha self_attn; avlen_linear lin1, lin2; avlen_affine norm1, norm2; } avlen_enc_layer;
typedef struct { avlen_mha self_attn, cross_attn; avlen_linear lin1, lin2; avlen_affine norm1, norm2, norm3; } avlen_dec_layer;
/* torch.nn.Transformer(d, nhead, 1 enc, 1 dec, ff, relu, post-norm) + final enc/dec LayerNorms
 * (smt_state_encoder.py:88-96). */
typedef struct { avlen_enc_layer enc; avlen_affine enc_norm; avlen_dec_layer dec; avlen_affine dec_norm; int d, nhead; } avlen_transformer;
/* SMTStateEncoder (smt_state_encoder.py:23-96): pose Linear(5,16), fusion MLP, transformer. */
typedef struct { avlen_linear pose, fus0, fus2; avlen_transformer tr; } avlen_smt;
/* DialogStateEncoder (dialog_state_encoder.py:42-99): fusion MLP(512->256->256), transformer, pe[100][256]. */
typedef struct { avlen_linear fus0, fus2; avlen_transformer tr; float* pe; int pe_len; } avlen_dialog;
/* LayerNorm folded into the following Linear (derived data, bf16 fast path): LN(x) W^T + b == rstd * (x W'^T - mean * s) + c
 * with W' = W * gamma (per input column), s[n] = sum_k W'[n][k], c[n] = b[n] + sum_k beta[k] W[n][k]; built by
 * avlen_ln_fold_weights.  w16f NULL -> the LayerNorm runs as its own kernel. */
typedef struct { void* w16f; float* s; float* c; } avlen_ln_fold;
typedef struct { avlen_affine ln1, ln2; avlen_mha attn; avlen_linear fc, proj; avlen_ln_fold attn_fold, fc_fold; } avlen_clip_block;
/* CLIP ViT-B/32 text tower (third party; call site policy.py:847-849). text_proj is [width][out]. */
/* half_fmt: format of every 16-bit weight shadow of the tower (w16, w16f): 0 = bf16, 1 = fp16 (AVLEN_PREC_FP16 calls). */
/* wstream: per-wave fragment-ordered copy of the 12 blocks' Linear weights in half_fmt (avlen_clip_pack_stream; NULL = absent):
 * with it the 16-bit forward runs the blocks as ONE sequence-stationary launch (csrc/clip_tower.hip). */
typedef struct { float* tok_emb; float* pos_emb; avlen_clip_block block[12]; avlen_affine ln_final;
                 float* text_proj; int vocab, ctx, width, heads, layers, out_dim, half_fmt; void* wstream;
                 float* text_proj_t; } avlen_clip_text;     /* text_proj_t: optional [out][width] copy (few-row projection: one wave per column).
                                                               text_proj NULL (one-launch tower only): out = ln_final(EOT rows), (B, width) --
                                                               for a caller that folded the projection into the Linear behind it */
/* nn.GRU(in, H, 1 layer) (av_nav/models/rnn_state_encoder.py:36-40): w_ih[3H][in], w_hh[3H][H], r|z|n. */
typedef struct { float* w_ih; float* w_hh; float* b_ih; float* b_hh; int in_f, hidden; } avlen_gru;
/* CategoricalNet + CriticHead (+ CriticHead2) of one policy (policy.py:46-61, 279-297). */
typedef struct { avlen_linear action, critic, unct; int has_unct; } avlen_heads;

/* ------------------------------------------------------------------ dense primitives ---------- */
/* C[M][N] = act(A * Wt^T + bias) + residual.  A(m,k)=A[m*lda+k] (transA: A[k*lda+m]); Wt(n,k)=B[n*ldb+k]
 * (transB: B[k*ldb+n]).  splitk>1 or beta!=0 route through slabs in `ws` (C = beta*C + ...).
 * Replaces torch.nn.functional.linear and its two backward products. */
int avlen_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
               const float* bias, const float* residual, int ldr, int M, int N, int K, int act, int prec,
               int splitk, float beta, void* ws, size_t ws_bytes, avlen_stream_t stream);
size_t avlen_gemm_workspace_bytes(int M, int N, int K, int splitk);
int avlen_gemm_pick_splitk(int M, int N, int K);
/* NHWC convolution as implicit GEMM; Wp packed [Cout][KH][KW][Cin].  Replaces nn.Conv2d forward
 * (smt_resnet.py:29-34,77-78,116-119; audio_cnn.py:62-83; visual_cnn.py:82-103). */
int avlen_conv2d_nhwc(const float* X, const float* Wp, const float* bias, const float* residual, float* Y,
                      int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int act,
                      int prec, avlen_stream_t stream);
/* OIHW -> [O][KH][KW][I] (conv) and (O, C*H*W) -> (O, H*W*C) (fc after an NCHW flatten). */
int avlen_pack_conv_weight(const float* w_oihw, float* w_packed, int O, int I, int KH, int KW, avlen_stream_t stream);
int avlen_pack_fc_after_flatten(const float* w, float* w_packed, int O, int C, int HW, avlen_stream_t stream);

/* ---- bf16-operand fast path (rollout "perf mode"): A, W already bf16 in HBM (leading dims multiples of 8,
 * padding zero-filled), tiles staged HBM->LDS with global_load_lds, BK=64, double-buffered; outputs fp32
 * (C32) and/or bf16 (C16), either may be NULL.  Same math as avlen_gemm / avlen_conv2d_nhwc with prec=BF16. */
int avlen_gemm_bf16(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16, int ldc16,
                    const float* bias, const float* residual, int ldr, int M, int N, int K, int act, void* ws,
                    size_t ws_bytes, avlen_stream_t stream);
size_t avlen_gemm_bf16_workspace_bytes(int M, int N);
/* X NHWC bf16 with Cin a power of two >= 8 (conv1's 3/1 input channels are zero-padded to 8); Wp bf16
 * [Cout][KH][KW][Cin].  gn_stats (optional, pre-zeroed, [B][2][Cout]): the epilogue adds each sample's
 * per-channel sum and sum of squares of the raw output, so GroupNorm needs no separate statistics pass. */
int avlen_conv2d_nhwc_bf16(const void* X, const void* Wp, const float* bias, const float* residual, float* Y32,
                           void* Y16, float* gn_stats, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, int act, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* Builds an avlen_ln_fold from a Linear (W fp32 [N][K], bias or NULL) and the LayerNorm (gamma, beta) in front of it;
 * w16f rows are ld16 apart (>= K, multiple of 8, padding zero).  CLIP residual blocks: ln_1 -> attn.in_proj,
 * ln_2 -> mlp.c_fc (third party CLIP, call site policy.py:847-849). */
int avlen_ln_fold_weights(const float* W, const float* bias, const float* gamma, const float* beta, void* w16f, int ld16,
                          float* s, float* c, int N, int K, avlen_stream_t stream);
/* Direct stride-1 "same" convolution for the small-channel tower stages (LDS halo tile, register-resident weights):
 * (Cin,Cout,W,K) in {(16,16,64,3), (32,32,32,3), (8,16,64,7)}; X/Y NHWC bf16; optional fused GroupNorm statistics. */
int avlen_conv_direct_bf16(const void* X, const void* Wp, void* Y16, float* gn_stats, int B, int W, int Cin, int Cout,
                           int K, avlen_stream_t stream);
int avlen_cast_bf16(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, avlen_stream_t stream);
/* The same cast into another 16-bit format: fmt 0 = bf16, 1 = fp16 (IEEE half), 2 = the LOW plane of the compensated bf16 pair,
 * bf16(x - bf16(x)) (AVLEN_PREC_BF16X3 operands are hi + lo planes of identical layout). */
int avlen_cast_h16(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, int fmt, avlen_stream_t stream);
/* Single-query cross attention in "memory space" (csrc/cross1.hip; the decoder layer of the scene-memory transformer, one target token
 * per sample: ss_baselines/savi/models/smt_state_encoder.py:152-166), d = 256, 8 heads of 32, S <= 320 keys.  The K | V projections
 * of the S memory rows are absorbed into the query side, so the training path at scale forms neither K | V nor their gradients:
 *   expand : out[b][h][c] = sum_j X[b][32 h + j] W[32 h + j][c]          (A = q_h W_k[h];  dm = dout_h W_v[h])          W: rows of a [d][d] slice
 *   fwd    : P [B][8][S] = masked softmax_key(scale * A[b][h] . mem[b][key]),  Mo[b][h] = sum_key P mem[b][key]            (mem: bf16 [B*S][256],
 *            low plane `lo` elements behind the high one, 0 = none; maskx [B][S], 1 = valid)
 *   reduce : Y[b][32 h + j] = sum_c Z[b][h][c] W[32 h + j][c] (+ bias)   (out = W_v[h] Mo + b_v;  dq = dA W_k[h]^T)
 *   bwd    : dA[b][h] = sum_key g mem[key],  dMEM[b*S + key] = sum_h P dm_h + g A_h,   g = scale * P (dm . mem[key] - sum_key P dm . mem)
 *   dw     : dW[32 h + j][c] += sum_b X[b][32 h + j] Z[b][h][c]          (dW_v += dout^T Mo;  dW_k += q^T dA) */
int avlen_cross1_expand(const float* X, int ldx, const float* W, int ldw, float* out, int B, avlen_stream_t stream);
int avlen_cross1_reduce(const float* Z, const float* W, int ldw, const float* bias, float* Y, int ldy, int B, avlen_stream_t stream);
int avlen_cross1_dw(const float* X, int ldx, const float* Z, float* dW, int ldw, int B, avlen_stream_t stream);
int avlen_cross1_fwd(const float* A, const void* MEM16, long lo, const float* maskx, float* P, float* Mo, int B, int S, float scale,
                     avlen_stream_t stream);
int avlen_cross1_bwd(const float* P, const float* DM, const float* A, const void* MEM16, long lo, float* dA, float* dMEM, int B, int S,
                     float scale, avlen_stream_t stream);
/* Weight gradient of a Linear over many rows without transposed operand copies (replaces the dW term of loss.backward(),
 * ss_baselines/savi/ppo/ppo.py:207-270): C [N1][N2] (row stride ldc) = beta * C + A^T B with A [M][lda], B [M][ldb] ROW-major bf16
 * (lda, ldb multiples of 8, >= the column count rounded up to 8; the pad columns must hold zeros).  Partial tiles are summed in a
 * fixed order (bit-reproducible).  ws: avlen_gemm_tn_bf16_workspace_bytes(M, N1, N2). */
size_t avlen_gemm_tn_bf16_workspace_bytes(long M, int N1, int N2);
int avlen_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, long M, int N1, int N2, float* C, int ldc, float beta,
                       void* ws, size_t ws_bytes, avlen_stream_t stream);
/* avlen_gemm_bf16 on 16-bit operands of format fmt (0 = bf16, 1 = fp16; C16 is written in the same format). */
int avlen_gemm_h16(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16, int ldc16,
                   const float* bias, const float* residual, int ldr, int M, int N, int K, int act, int fmt, void* ws,
                   size_t ws_bytes, avlen_stream_t stream);
/* avlen_ln_fold_weights with the folded weights stored in format fmt (0 = bf16, 1 = fp16). */
int avlen_ln_fold_weights_h16(const float* W, const float* bias, const float* gamma, const float* beta, void* w16f, int ld16,
                              float* s, float* c, int N, int K, int fmt, avlen_stream_t stream);
int avlen_pack_conv_weight_bf16(const float* w_oihw, void* w_packed, int O, int I, int KH, int KW, int Cpad,
                                avlen_stream_t stream);
int avlen_pack_fc_after_flatten_bf16(const float* w, void* w_packed, int O, int C, int HW, avlen_stream_t stream);
/* w16 [cout][K] -> w16f (avlen_conv::w16f), cout % 16 == 0, K % 32 == 0 */
int avlen_pack_conv_weight_frag(const void* w16, void* w16f, int cout, int K, avlen_stream_t stream);

/* ------------------------------------------------------------------ normalisation ------------- */
/* y = [relu]( GroupNorm_G(x)*g + b [+ residual] ), x NHWC (B,HW,C).  smt_resnet.py:30-33,40-51,79. */
int avlen_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                         int B, int HW, int C, int G, int relu, float eps, avlen_stream_t stream);
/* y = LayerNorm(x [+ residual]) over the last dim d (d % 64 == 0, d <= 1024); saves mean/rstd when non-NULL. */
int avlen_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, int rows, int d, float eps, avlen_stream_t stream);
/* dx = LN backward (also the gradient of the residual input); dgamma/dbeta += (atomic fp32). xhat is
 * recomputed from y: xhat = (y - beta)/gamma is avoided -- pass the saved pre-norm sum `xsum` = x+residual. */
int avlen_layernorm_bwd(const float* dy, const float* xsum, const float* gamma, const float* mean,
                        const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int d,
                        avlen_stream_t stream);

/* ------------------------------------------------------------------ attention ------------------ */
/* Multi-head attention core, batch-major.  Q[b][i] at Q + (b*Sq+i)*ldq + h*D, same for K,V (Sk rows),
 * O (Sq rows).  key_mask[b][j] (1 = valid, 0 = padded) may be NULL; causal masks j>i.  scale multiplies
 * q.k.  lse[b][h][i] (log-sum-exp of the scaled scores) is saved when non-NULL.  D in {32, 64}.
 * Replaces the softmax(QK^T)V core of nn.MultiheadAttention (smt_state_encoder.py:160-166). */
int avlen_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                        int ldo, const float* key_mask, float* lse, int B, int H, int Sq, int Sk, int D,
                        int causal, float scale, avlen_stream_t stream);
/* Gradients of the same; dQ/dK/dV are overwritten.  delta is a (B*H*Sq) scratch. */
int avlen_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                        const float* O, int ldo, const float* dO, int lddo, const float* key_mask,
                        const float* lse, float* delta, float* dQ, int lddq, float* dK, int lddk, float* dV,
                        int lddv, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                        avlen_stream_t stream);
/* The same gradients on the matrix cores (bf16 operands, fp32 accumulate; P and dS stay in registers) for the scene-memory
 * encoder's self-attention: Sq == Sk <= 320, D == 32, no causal mask -- AVLEN_ERR_ARG outside that envelope.  Used by
 * avlen_smt_bwd in bf16 mode. */
int avlen_attention_bwd_bf16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                             const float* O, int ldo, const float* dO, int lddo, const float* key_mask,
                             const float* lse, float* delta, float* dQ, int lddq, float* dK, int lddk, float* dV,
                             int lddv, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                             avlen_stream_t stream);

/* ------------------------------------------------------------------ small fused kernels -------- */
/* (B,S,S,C) NHWC -> (x / divisor) -> kxk block mean -> (B,64,64,C).  K1+K2: smt_cnn.py:83-93 +
 * common/utils.py:467-557 (area resize to 64, identity crop).  divisor = 255 for rgb, 1 for depth.
 * Image arguments of this section and of the tower entry points below come with a `*_u8` flag: 0 = fp32 pixels (what
 * common/utils.py:129-156 `batch_obs` hands over), 1 = uint8 pixels as the simulator produces them (RGB stays
 * uint8 from the sensor through the rollout storage to this prologue; converted to float first, so bit-identical). */
int avlen_preprocess_image(const void* x, int x_u8, float* y, int B, int S, int C, float divisor, avlen_stream_t stream);
/* VisualCNN input (visual_cnn.py:165-183): y[b,h,w,:] = [rgb/255 (3), depth (1)]. */
int avlen_rgbd_concat(const void* rgb, int rgb_u8, const float* depth, float* y, int B, int HW, avlen_stream_t stream);
/* Non-CNN feature columns (policy.py:662-674, 1035-1036, 1062-1063): for each row b
 *   feats[b, col_action .. +16)   = action_encoder.w[:, prev_action[b]] + b      (one-hot x Linear)
 *   feats[b, col_cat .. +21)      = category[b]              (only if category != NULL)
 *   feats[b, col_pose .. +4)      = pose[b]
 *   feats[b, col_extra .. +n_extra) = extra[b]               (query_state; only if extra != NULL)
 * and the belief/goal vector (policy.py:605-618): goal[b] = [category_belief(21) | location_belief(2) | 0...].
 * vis / aud (optional): encoder feature rows computed into other buffers (the shared towers / AudioCNNs of several policies):
 * feats[b, 0 .. n_vis) = vis[b], feats[b, col_aud .. +n_aud) = aud[b] -- policy.py:662-668's torch.cat as part of this launch. */
int avlen_feature_assemble(float* feats, int ldf, const avlen_linear* action_encoder, const int64_t* prev_actions,
                           int col_action, const float* category, int col_cat, const float* pose, int col_pose,
                           const float* extra, int n_extra, int col_extra, const float* category_belief,
                           const float* location_belief, float* goal, int d_goal, int B, const float* vis, int ld_vis, int n_vis,
                           const float* aud, int ld_aud, int n_aud, int col_aud, avlen_stream_t stream);
/* rows[b] = [feats[b, 0:n_keep) | tail[b]]: the external-memory row of pi_q (policy.py:1062-1063). */
int avlen_concat_rows(const float* a, int lda, int na, const float* b, int ldb, int nb, float* out, int ldo, int B,
                      avlen_stream_t stream);

/* ------------------------------------------------------------------ encoders (module level) ---- */
size_t avlen_resnet18_workspace_bytes(int B);
/* SMTCNN tower (smt_cnn.py:78-115 + smt_resnet.py:132-146) on one modality: img (B,S,S,C) raw sensor,
 * divisor = 255 for rgb, 1 for depth; writes 64 features to out[b*ld_out + 0..63]. */
/* The same towers in compensated bf16 (AVLEN_PREC_BF16X3; needs the w16lo / w16flo planes): `groups` towers of identical
 * shape, image b of the batch = image row_index[b] of imgs[g] when row_index is given; outs[g] (B, 64) fp32 with row stride ld_out. */
size_t avlen_resnet18_group_x3_workspace_bytes(int groups, int B);
int avlen_resnet18_group_fwd_x3(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                                const float* divisors, float* const* outs, int ld_out, int groups, int B, int S,
                                const int32_t* row_index, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* The same in two calls on ONE workspace: phase 1 = the towers (layer-4 outputs stay in `ws`), 2 = the fc on what phase 1 left
 * there, 3 = both.  Lets a captured forward keep the fc off the path between the towers and the AudioCNNs. */
int avlen_resnet18_group_fwd_x3_phase(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                                      const float* divisors, float* const* outs, int ld_out, int groups, int B, int S,
                                      const int32_t* row_index, int phase, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* avlen_pack_conv_weight_bf16 into 16-bit format fmt (0 = bf16, 1 = fp16, 2 = low plane of the compensated bf16 pair) */
int avlen_pack_conv_weight_h16(const float* w_oihw, void* w_packed, int O, int I, int KH, int KW, int Cpad, int fmt,
                               avlen_stream_t stream);
int avlen_pack_fc_after_flatten_h16(const float* w, void* w_packed, int O, int C, int HW, int fmt, avlen_stream_t stream);
int avlen_resnet18_fwd(const avlen_resnet18* net, const void* img, int img_u8, int B, int S, int C, float divisor, float* out,
                       int ld_out, int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* `groups` (<= 8) towers of identical shape in lock-step on the bf16 fast path: every conv / GroupNorm / fc is ONE
 * grouped launch (blockIdx.y = tower).  Used to run rgb+depth of a policy -- or all six towers of pi_q/pi_g/pi_l,
 * which see the same observation -- as single launches.  Arrays are host arrays of length `groups`. */
size_t avlen_resnet18_group_workspace_bytes(int groups, int B);
int avlen_resnet18_group_fwd(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                             const float* divisors, float* const* outs, int ld_out, int groups, int B, int S, void* ws,
                             size_t ws_bytes, avlen_stream_t stream);
/* ---- BeliefPredictor (belief_predictor.py:56-206), SURVEY 8(f) rank 1 ---- */
/* predictor = custom_resnet18 at the input's own extent (no resize; belief_predictor.py:66-72,126-137 over
 * smt_resnet.py:132-146): x NHWC (B,H,W,C) fp32 -> out[b*ld_out + 0..fc.out_f).  conv weights in `w` (packed fp32),
 * fc.w packed to the NHWC flatten order of the last stage. */
size_t avlen_resnet18_any_workspace_bytes(int B, int H, int W);
int avlen_resnet18_any_fwd(const avlen_resnet18* net, const float* x, int B, int H, int W, int C, float* out, int ld_out,
                           int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* classifier = torchvision resnet18 with conv1 (C -> 64) (belief_predictor.py:79-81,179; third-party architecture):
 * eval-mode BatchNorms already folded into the convs by the host (conv.b = beta - mean*scale, bn* unused);
 * conv7x7 s2 + ReLU, maxpool 3x3 s2 p1, 8 BasicBlocks, global average pool, fc. */
size_t avlen_resnet18_tv_workspace_bytes(int B, int H, int W);
int avlen_resnet18_tv_fwd(const avlen_resnet18* net, const float* x, int B, int H, int W, int C, float* out, int ld_out,
                          int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* cnn_forward's input for the distractor variant (belief_predictor.py:129-134): out (B,HW,Cs+Cc) = [spec | category]. */
int avlen_belief_input(const float* spec, const float* category, float* out, int B, int HW, int Cs, int Cc,
                       avlen_stream_t stream);
/* The per-environment filter of BeliefPredictor.update (belief_predictor.py:146-206, 213-230), float32 like the numpy
 * loop it replaces.  pointgoals (B,>=2) / labels (B,>=n_label): network outputs, either may be NULL (that half is
 * skipped); pose (B,ld_pose) = x, y, heading, t; spectrogram (B,spec_elems) decides "sounding" (sum != 0); dones (B) bytes
 * or NULL.  Filter state (last_pointgoal (B,2) in the odometry frame, last_label (B,n_label), has_* (B) flags) lives
 * on the device and is updated in place; results are written into the observation tensors location_belief (B,2) and
 * category_belief (B,n_label).  spec_sum: (B) scratch. */
int avlen_belief_update(const float* pointgoals, int ld_pg, const float* labels, int ld_lab, const float* pose, int ld_pose,
                        const float* spectrogram, long spec_elems, const unsigned char* dones, float* last_pointgoal,
                        int* has_pointgoal, float* last_label, int* has_label, float* location_belief,
                        float* category_belief, float* spec_sum, int B, int n_label, float weighting_factor,
                        int current_pred_only, avlen_stream_t stream);
/* The same with a row index (bf16 fast path): image / spectrogram b of the batch is item row_index[b] of the tensor the
 * pointer addresses -- the PPO minibatch (rollout_storage.py:591-810) reads the (T+1, N, ...) observation storage in place
 * instead of gathering 470 KB per stored step first. */
int avlen_resnet18_group_fwd_indexed(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                                     const float* divisors, float* const* outs, int ld_out, int groups, int B, int S,
                                     const int32_t* row_index, void* ws, size_t ws_bytes, avlen_stream_t stream);
int avlen_cnn3_fwd_indexed(const avlen_cnn3* net, const float* x, const int32_t* row_index, int B, int H, int W, float* out,
                           int ld_out, void* ws, size_t ws_bytes, avlen_stream_t stream);
size_t avlen_cnn3_workspace_bytes(const avlen_cnn3* net, int B, int H, int W);
/* AudioCNN.forward (audio_cnn.py:136-151) / VisualCNN.cnn: x NHWC (B,H,W,conv[0].cin) -> out[b*ld_out + 0..fc.out_f). */
int avlen_cnn3_fwd(const avlen_cnn3* net, const float* x, int B, int H, int W, float* out, int ld_out, int prec,
                   void* ws, size_t ws_bytes, avlen_stream_t stream);
/* `groups` (<= 8) AudioCNNs of identical architecture on the SAME input x (the audio encoders of pi_q / pi_g / pi_l all
 * read the step's spectrogram, ppo_trainer.py:449-636): one bf16 cast of x, one grouped launch per layer (bf16 fast
 * path only).  outs[g][b*ld_out + 0..fc.out_f). */
size_t avlen_cnn3_group_workspace_bytes(const avlen_cnn3* net, int groups, int B, int H, int W);
int avlen_cnn3_group_fwd(const avlen_cnn3* const* nets, const float* x, int groups, int B, int H, int W,
                         float* const* outs, int ld_out, void* ws, size_t ws_bytes, avlen_stream_t stream);

/* ------------------------------------------------------------------ SMT / dialog / CLIP -------- */
/* SMTStateEncoder.single_forward (smt_state_encoder.py:109-188).
 *   x      (B, F)         current features [.. | pose(4) at pose_col | ..]
 *   memory (M, NC, F)     external memory rows, reference layout (slot-major); sample b reads column
 *                         mem_index[b] (int32), or column b when mem_index == NULL (then NC == B).  The
 *                         index form lets a (T*N_mb)-row PPO minibatch read the N-column ring in place
 *                         instead of materialising the reference's (300, T*N_mb, F) copy (K21).
 *   masks  (B, M)         1 = valid slot
 *   goal   (B, d)         decoder target (belief vector)
 *   out    (B, d)
 * current_token_only=1 is the `pretraining=True` configuration (:126-129): every memory key is masked,
 * so only the current token is computed (bit-for-bit the same function, 200x fewer FLOPs).
 * save_for_backward != 0: the workspace keeps what avlen_smt_bwd needs (fp32-staged kernels); == 0 with
 * prec = BF16 and full memory takes the bf16 fast path (inference only). */
size_t avlen_smt_workspace_bytes(const avlen_smt* p, int B, int M, int F, int current_token_only);
int avlen_smt_fwd(const avlen_smt* p, const float* x, const float* memory, const int32_t* mem_index, int NC,
                  const float* masks, const float* goal, float* out, int B, int M, int F, int pose_col,
                  int current_token_only, int save_for_backward, int prec, void* ws, size_t ws_bytes,
                  avlen_stream_t stream);
/* Backward of avlen_smt_fwd w.r.t. the parameters only (x, memory and goal carry no gradient on this
 * path: policy.py:1035-1036).  `ws` must be the forward's workspace, untouched.  Gradients are
 * ACCUMULATED into `g` (same layout as p). */
int avlen_smt_bwd(const avlen_smt* p, const avlen_smt* g, const float* goal, const float* d_out, int B, int M, int F,
                  int pose_col, int current_token_only, int prec, float* d_x, int ld_dx, void* ws, size_t ws_bytes,
                  avlen_stream_t stream);
/* Tuning knob: row count from which the training path's Linear products (avlen_smt_fwd with save_for_backward,
 * avlen_smt_bwd, bf16 mode) cast their fp32 operands to bf16 once and run the glds/MFMA GEMM (default 4096 rows). */
void avlen_set_big_m(long rows);
/* bf16x3 training at scale: from `rows` token rows (B x (M + 1); default 65536, 0 = never, < 0 = default) avlen_smt_bwd runs the
 * backward's products and attention on plain bf16 operands with fp32 accumulation (mixed-precision training); the forward -- the
 * logits, the PPO ratio, the losses -- stays compensated at every size. */
void avlen_set_x3_mixed_backward_rows(long rows);
/* The training forward at scale (rows >= avlen_set_big_m; bf16 mode, and bf16x3 where its backward is mixed) keeps its activations
 * as 16-bit operand planes emitted by their producers and runs the self attention on the matrix cores (default on).  0: every
 * product casts its fp32 input (the layout the small-batch path uses); 3: planes on, but the decoder's single-query cross attention
 * through the K | V projection of the memory rows instead of in memory space (csrc/cross1.hip); tests compare the three. */
void avlen_set_big16(int on);
/* The AudioCNN's three convolutions (audio_cnn.py: 8x8 s4, 4x4 s2, 3x3 s1 on a 2-channel spectrogram) run as ONE launch with the
 * activations in LDS when the geometry fits (default on; csrc/audio3.hip).  0: a cast and one implicit-GEMM launch per conv. */
void avlen_set_audio3(int on);
/* Scheduling knob of the bf16x3 tower group (one persistent work-queue launch, one workgroup per CU): CUs it leaves free for the
 * other streams of the step (default 0 = every CU). */
void avlen_set_tower_x3_reserved_cus(int n);
/* Measurement (bench.py's roofline record): mean duration in microseconds of the persistent bf16x3 tower launches since the last
 * reset, as they ran -- inside the rollout step, beside the other streams -- and how many there were; stamped by the kernel itself
 * on the device's constant-rate wall clock.  Synchronises the device.  reset != 0 clears the counters. */
int avlen_tower_x3_timing(double* mean_us, long long* launches, int reset);
/* DialogStateEncoder.single_forward (dialog_state_encoder.py:114-155): x_att (B,d), memory_state (M,B,d),
 * masks (B,M), d_emb (B,d) or NULL, agent_step (B) float, goal (B,d) -> out (B,d). */
size_t avlen_dialog_workspace_bytes(const avlen_dialog* p, int B, int M);
int avlen_dialog_fwd(const avlen_dialog* p, const float* x_att, const float* memory_state, const float* masks,
                     const float* d_emb, const float* agent_step, const float* goal, float* out, int B, int M,
                     int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* CLIP.encode_text (frozen): tokens (B,ctx) int64 -> out (B,out_dim). */
size_t avlen_clip_text_workspace_bytes(const avlen_clip_text* p, int B);
int avlen_clip_text_fwd(const avlen_clip_text* p, const int64_t* tokens, float* out, int B, int prec, void* ws,
                        size_t ws_bytes, avlen_stream_t stream);
/* The same function with a per-row memo of the frozen tower (ppo_trainer.py:347,582-586: a dialog is constant for NUM_DIALOG_STEPS
 * steps and all-zero for an env without a query): only rows whose 77 tokens differ from the previous call's run the 12 blocks,
 * all-zero rows share one embedding.  `state`: caller-owned device block of avlen_clip_text_cache_bytes(p, B) bytes, one per batch
 * size; zero-filled = empty (do that after the tower's weights change).  Workspace: avlen_clip_text_workspace_bytes(p, B + 1).
 * Without the one-launch tower (fp32, no weight stream, B > 511) every row is computed (== avlen_clip_text_fwd). */
size_t avlen_clip_text_cache_bytes(const avlen_clip_text* p, int B);
int avlen_clip_text_cached_fwd(const avlen_clip_text* p, const int64_t* tokens, void* state, size_t state_bytes, float* out, int B,
                               int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* dialog_layer(CLIP.encode_text(tokens)) of the rollout step (policy.py:844-851) on the memoised tower above with its tail as ONE
 * launch (memo update of the all-zero rows + ln_final + 16-bit cast + the product): p = the tower with text_proj == NULL, fold =
 * dialog_layer with text_projection folded into its weight ([out_f][width], 16-bit shadow in the tower's format; out_f % 16 == 0,
 * <= 256).  out (B, fold->out_f).  16-bit modes, width 512, B + 1 <= 512 only: AVLEN_ERR_ARG otherwise.
 * warm_ptrs / warm_bytes (n_warm <= 4, optional): byte ranges (16-byte aligned) the tail launch's spare workgroups pull into the L2s
 * while it runs -- the weights of the dialog state encoder's chain that follows (as avlen_prefetch_l2, without a launch of its own). */
int avlen_clip_text_dialog_fwd(const avlen_clip_text* p, const avlen_linear* fold, const int64_t* tokens, void* state, size_t state_bytes,
                               float* out, int B, int prec, void* ws, size_t ws_bytes, const void* const* warm_ptrs,
                               const int64_t* warm_bytes, int n_warm, avlen_stream_t stream);
/* The one-launch tower's weight stream (csrc/clip_tower.hip): bytes for `p` (0: shape not supported -- width 512, 8 heads,
 * ctx <= 80, 4x MLP, biases present) and the packer (fmt 0 bf16, 1 fp16; from the fp32 weights; derived data). */
size_t avlen_clip_stream_bytes(const avlen_clip_text* p);
/* Knob of the one-launch tower (it packs whole dialogs into groups of <= 4 row tiles and splits a group's weight stream over 4
 * workgroups): n = -1 (default) every call 4-way; 0 every call 2-way; n > 0 (lab) 4-way when groups x 4 <= n, else 2-way -- the two
 * splits round later layers' fp16 operands at different points (2e-3 apart), so only the first two keep a dialog's embedding
 * independent of the other dialogs of the call. */
void avlen_set_clip_tower_split4_wgs(int n);
/* Fused row-batch chains (csrc/chain.hip: the single-token decoder / collapsed encoder / per-step tails of pi_q, pi_g, pi_l): 1
 * (default) = batches of up to 32 blocks are launched with only every 8th block of the grid working, so that the working blocks
 * share ONE XCD's L2 for the weights they all stream (speed only); 0 = plain grid. */
void avlen_set_chain_one_xcd(int on);
int avlen_clip_pack_stream(const avlen_clip_text* p, void* dst, int fmt, avlen_stream_t stream);

/* ------------------------------------------------------------------ GRU ------------------------ */
/* RNNStateEncoder (rnn_state_encoder.py:80-143).  T==1: single_forward; T>1: seq_forward with x (T*N,in)
 * T-major, masks (T*N).  h0 (N,H) -> out (T*N,H), h_out (N,H).  h is multiplied by masks[t] before step t. */
size_t avlen_gru_workspace_bytes(const avlen_gru* p, int T, int N);
int avlen_gru_fwd(const avlen_gru* p, const float* x, const float* h0, const float* masks, float* out, float* h_out,
                  int T, int N, int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);

/* ------------------------------------------------------------------ heads / PPO ---------------- */
/* CategoricalNet + critics forward (common/utils.py:44-72, policy.py:86-96): feats (B,d) ->
 * logits/probs/logp (B,A), value (B), unct (B,2) (if has_unct), and, when actions != NULL,
 * log_prob[b] = logp[b, actions[b]] and entropy[b] (per row). Any output pointer may be NULL. */
int avlen_heads_fwd(const avlen_heads* h, const float* feats, int d, int A, float* logits, float* probs,
                    float* value, float* unct, const int64_t* actions, float* log_prob, float* entropy, int B,
                    avlen_stream_t stream);
/* avlen_heads_fwd + avlen_sample_race + the chosen action's log-prob / entropy in one launch (a rollout forward's tail). */
int avlen_heads_act_fwd(const avlen_heads* h, const float* feats, int d, int A, float* logits, float* probs, float* value,
                        float* unct, const float* noise, int64_t* action_out, float* log_prob, float* entropy, int B,
                        avlen_stream_t stream);
/* ... and a second copy of the sampled actions stored straight into `action_host` (optional): PINNED host memory mapped into the
 * device's address space (hipHostMalloc / torch pin_memory) -- what the trainer's query loop (ppo_trainer.py:463) and envs.step
 * (:864) read, valid once the stream has passed the launch; no device-to-host copy launch. */
int avlen_heads_act_host_fwd(const avlen_heads* h, const float* feats, int d, int A, float* logits, float* probs, float* value,
                             float* unct, const float* noise, int64_t* action_out, int64_t* action_host, float* log_prob,
                             float* entropy, int B, avlen_stream_t stream);
/* CustomFixedCategorical.sample (common/utils.py:48-49; torch.multinomial's exponential race) with HOST-drawn noise: action[b] =
 * first argmax_a probs[b,a] / noise[b,a] (IEEE fp32 division) -- the reference's action for the same host generator state. */
int avlen_sample_race(const float* probs, const float* noise, int64_t* action, int B, int A, avlen_stream_t stream);
/* PPO loss (ppo.py:219-262) fused with its backward through the heads.  Row-wise inputs as in the
 * reference minibatch; norm = {1/sum(rl_masks), 1/R} read from device (2 floats, see avlen_rl_mask_norm).
 * Outputs: loss_sums[6] += {value_loss, action_loss, entropy, values_mean, returns_mean, unct_loss}
 * contributions (already normalised); d_feats (B,d) overwritten; head gradients accumulated into g. */
int avlen_ppo_loss_heads_bwd(const avlen_heads* h, const avlen_heads* g, const float* feats, int d, int A,
                             const int64_t* actions, const float* old_log_probs, const float* adv,
                             const int64_t* rl_masks, const float* value_preds, const float* returns,
                             const int64_t* unct_gt, const float* norm, float clip, float value_coef,
                             float entropy_coef, float unct_coef, float* loss_sums, float* d_feats, int B,
                             avlen_stream_t stream);
/* norm[0] = 1/sum(rl_masks[0..R)), norm[1] = 1/R. */
int avlen_rl_mask_norm(const int64_t* rl_masks, int R, float* norm, avlen_stream_t stream);
/* GAE backward scan (rollout_storage.py:394-405): values[T_used] <- next_value first.  rewards (T,N),
 * values (T+1,N), masks (T+1,N), returns (T+1,N) out, advantages (T,N) out (= returns - values). */
int avlen_gae_scan(const float* rewards, float* values, const float* masks, const float* next_value, float* returns,
                   float* advantages, int T_used, int N, float gamma, float tau, avlen_stream_t stream);
/* use_gae=False branch of compute_returns (rollout_storage.py:406-412; common/rollout_storage.py:129-135). */
int avlen_discounted_returns(const float* rewards, const float* masks, const float* next_value, float* returns, int T_used,
                             int N, float gamma, avlen_stream_t stream);
/* clip_grad_norm_ + Adam (ppo.py:297-300, torch.optim.Adam): norm_sq is a device double accumulated by
 * avlen_grad_sumsq over every trained segment, then each segment is stepped. */
int avlen_grad_sumsq(const float* grad, size_t n, double* norm_sq, avlen_stream_t stream);
int avlen_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                    float beta1, float beta2, float eps, int step, float max_grad_norm, const double* norm_sq,
                    avlen_stream_t stream);

/* ---- GRU baseline training (BASELINE configs[1]): AudioNavBaselineNet forward with saved activations and its backward
 * (savi/ppo/policy.py:451-477; audio_cnn.py:62-94; visual_cnn.py:82-107; av_nav/models/rnn_state_encoder.py:92-143), driven by
 * the av_nav PPO update (av_nav/ppo/ppo.py:60-151) between the two calls with avlen_ppo_loss_heads_bwd.  Rows are T-major
 * (row = t*N + n).  rgb (R,S,S,3) 0..255, depth (R,S,S,1), spec (R,Ha,Wa,C); out (R,H).  g_*: gradient views with the SAME
 * struct types whose w / b pointers address the CANONICAL gradient tensors (conv OIHW, fc (out, C*H*W), GRU as nn.GRU);
 * gradients are accumulated into them.  The workspace carries the saved activations from _fwd to _bwd. */
size_t avlen_baseline_train_workspace_bytes(const avlen_cnn3* audio, const avlen_cnn3* visual, const avlen_gru* gru, int T, int N,
                                            int Ha, int Wa, int S, int prec);
int avlen_baseline_train_fwd(const avlen_cnn3* audio, const avlen_cnn3* visual, const avlen_gru* gru, const float* spec,
                             const void* rgb, int rgb_u8, const float* depth, const float* category, int ncat, const float* h0,
                             const float* masks, float* out, float* h_out, int T, int N, int Ha, int Wa, int S, int prec, void* ws,
                             size_t ws_bytes, avlen_stream_t stream);
int avlen_baseline_train_bwd(const avlen_cnn3* audio, const avlen_cnn3* visual, const avlen_gru* gru, const avlen_cnn3* g_audio,
                             const avlen_cnn3* g_visual, const avlen_gru* g_gru, const float* spec, const float* masks,
                             const float* d_out, int T, int N, int Ha, int Wa, int S, int prec, void* ws, size_t ws_bytes,
                             avlen_stream_t stream);

/* ---- PPO.update_dialog (ppo.py:99-154): the pieces of pi_l's backward that are not shared with pi_q's update.
 * avlen_smt_bwd's d_x (B, F; optional) = gradient w.r.t. the current observation's feature row (pose columns zero).
 * Dialog state encoder (dialog_state_encoder.py:114-155) in training form: _train_fwd keeps every activation in the workspace,
 * _bwd accumulates parameter gradients into `g` and returns d_x_att (B,d) and, when the forward had a dialog embedding, d_demb. */
size_t avlen_dialog_train_workspace_bytes(const avlen_dialog* p, int B, int M);
int avlen_dialog_train_fwd(const avlen_dialog* p, const float* x_att, const float* memory_state, const float* masks,
                           const float* d_emb, const float* agent_step, const float* goal, float* out, int B, int M, int prec,
                           void* ws, size_t ws_bytes, avlen_stream_t stream);
int avlen_dialog_bwd(const avlen_dialog* p, const avlen_dialog* g, const float* goal, const float* d_out, int has_dialog,
                     float* d_x_att, float* d_demb, int B, int M, int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* One Linear between modules (dialog_layer, policy.py:849): G.w += dY^T X, G.b += colsum dY, dX = dY W (dX optional). */
size_t avlen_linear_bwd_workspace_bytes(void);
int avlen_linear_bwd(const avlen_linear* L, const avlen_linear* G, const float* X, int ldx, const float* dY, int ldy, float* dX,
                     int lddx, int M, int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);
/* action_encoder = Linear(one_hot(prev_action)) (policy.py:662-667): G.w[j][a] += sum_{b: a_b = a} d[b][j], G.b[j] += sum_b d[b][j]. */
int avlen_action_encoder_bwd(const float* d_feats, int ld, const int64_t* prev_actions, const avlen_linear* G, int B,
                             avlen_stream_t stream);
/* CrossEntropyLoss(weight) of the vln action logits over rows with o_masks != 0 against o_actions (ppo.py:139-145), its head
 * gradients (accumulated) and d_feats (R,d).  norm: 1 float scratch; loss: 1 float, accumulated. */
int avlen_dialog_loss_heads_bwd(const avlen_heads* h, const avlen_heads* g, const float* feats, int d, int A,
                                const float* o_actions, const int64_t* o_masks, const float* class_weights, float* norm,
                                float* loss, float* d_feats, int R, avlen_stream_t stream);
/* AudioCNN alone in training form (pi_l's goal encoder under update_dialog): see avlen_baseline_train_*. */
size_t avlen_cnn3_train_workspace_bytes(const avlen_cnn3* net, int B, int H, int W, int prec);
int avlen_cnn3_train_fwd(const avlen_cnn3* net, const float* x, int B, int H, int W, float* out, int ld_out, int prec, void* ws,
                         size_t ws_bytes, avlen_stream_t stream);
int avlen_cnn3_train_bwd(const avlen_cnn3* net, const avlen_cnn3* grads, const float* x, const float* y, const float* d_out,
                         int ld, int B, int H, int W, int prec, void* ws, size_t ws_bytes, avlen_stream_t stream);

/* ---- GroupNorm ResNet-18 training (CustomResNet, smt_resnet.py:37-149): forward with saved activations + backward.
 * Users: BeliefPredictor's online regression (ppo_trainer.py:959-1030) and pi_l's towers under PPO.update_dialog (ppo.py:99-154).
 * x (B,H,W,C) NHWC fp32 is the network input proper (towers: the preprocessed 64x64 image); out (B, fc.out_f).  `grads`: the same
 * struct type whose conv .w / GroupNorm .g,.b / fc .w,.b pointers address CANONICAL gradient tensors (conv OIHW, fc
 * (out, C*H*W)); accumulated into.  d_x: optional input gradient.  The workspace carries the activations from _fwd to _bwd. */
size_t avlen_resnet18_train_workspace_bytes(const avlen_resnet18* net, int B, int H, int W, int prec);
int avlen_resnet18_train_fwd(const avlen_resnet18* net, const float* x, int B, int H, int W, float* out, int ld_out, int prec,
                             void* ws, size_t ws_bytes, avlen_stream_t stream);
int avlen_resnet18_train_bwd(const avlen_resnet18* net, const avlen_resnet18* grads, const float* x, const float* d_out,
                             int ld_dout, int B, int H, int W, float* d_x, int prec, void* ws, size_t ws_bytes,
                             avlen_stream_t stream);
/* Loss of train_belief_predictor (ppo_trainer.py:1000-1008, 1017-1022): masked MSE between preds (R,2) and the transformed
 * goal (gt[1], -gt[0]); mask = the row's spectrogram (spec_elems values) is not all-zero.  d_preds (R,2) out;
 * acc[0] += loss, acc[1] += correct rows (rounded prediction equals the goal), acc[2] += masked rows. */
int avlen_belief_regression_loss(const float* preds, const float* spec, long spec_elems, const float* gts, int ld_gt,
                                 float* d_preds, float* acc, int R, avlen_stream_t stream);

/* ---- binaural spectrogram on the device (soundspaces/tasks/nav.py:88-101 SpectrogramSensor.compute_spectrogram):
 * audio (B, 2, L) fp32 -> log1p(pool x pool block mean of |STFT|) -> out (B, ceil(NB/pool), ceil(F/pool), 2), NB = nfft/2 + 1,
 * F = 1 + L/hop centred frames.  window (nfft) = the analysis window already zero-padded to nfft; basis (2*NB, nfft) = rows
 * cos(2 pi k n / nfft) for k < NB, then -sin(...).  reflect: 1 = reflect padding of the signal (librosa < 0.10), 0 = zeros.
 * Third-party definitions (librosa.stft, skimage block_reduce) restated: parity unpinned. */
size_t avlen_spectrogram_workspace_bytes(int B, int L, int nfft, int hop);
int avlen_spectrogram(const float* audio, int B, int L, const float* window, const float* basis, int nfft, int hop, int pool,
                      int reflect, float* out, void* ws, size_t ws_bytes, avlen_stream_t stream);

/* ------------------------------------------------------------------ rollout storage ------------ */
/* ExternalMemory.insert (rollout_storage.py:930-941) on ONE copy of the ring: memory (total,N,dim),
 * masks (N,total).  Also snapshots the new masks to masks_out (N,total) when non-NULL. */
int avlen_extmem_insert(float* memory, float* masks, const float* feats, int ld_feats, const float* not_done,
                        float* masks_out, int idx, int total, int capacity, int N, int dim, avlen_stream_t stream);
/* The same for up to 4 rings in ONE launch (a rollout step writes the goal, option, vln and dialog rings). */
typedef struct { float* memory; float* masks; const float* feats; int ld_feats; const float* not_done; float* masks_out;
                 int idx, total, capacity, N, dim; } avlen_extmem_op;
int avlen_extmem_insert_multi(const avlen_extmem_op* ops, int n, avlen_stream_t stream);
/* recurrent_generator's gather (rollout_storage.py:649-782): dst[(t*n_mb + j), :] = src[t, env[j], :]
 * for t < T; src is (T_alloc, N, D) fp32 (elem_bytes=4) or int64 (elem_bytes=8). */
int avlen_minibatch_gather(const void* src, void* dst, const int64_t* env, int T, int N, int n_mb, size_t D,
                           int elem_bytes, avlen_stream_t stream);
/* out[i] = a[i] (fp32 copy on stream; strided rows) -- storage insert helper. */
int avlen_copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, avlen_stream_t stream);
/* dst[r][0 .. cols) = src[index[r]][0 .. cols): rows of the external-memory ring read back as encoder features (PPO.update with
 * feature_reuse: policy.py:1035-1036 cuts the gradient in front of them, so the stored rows ARE what a recompute would give). */
int avlen_gather_rows(const float* src, int lds, const int* index, float* dst, int ldd, int rows, int cols, avlen_stream_t stream);
/* dst[i][0..nbytes[i]) = src[i][0..nbytes[i]) for i < n, in ONE launch per 32 pairs (host arrays of device pointers).
 * RolloutStorage.insert (rollout_storage.py:223-330 of the reference: ~25 `tensor[step].copy_()` calls per step). */
int avlen_multi_copy(const void* const* src, void* const* dst, const int64_t* nbytes, int n, avlen_stream_t stream);

/* Warm the L2 of every XCD with up to 8 byte ranges (16-byte aligned; whole 16-byte words are read) that a latency-bound kernel is
 * about to stream -- the 16-bit weight planes of the fused state-encoder chains of pi_q / pi_l (smt_state_encoder.py:152-166,
 * dialog_state_encoder.py:133-152 as one launch each): 62 us warm against ~100 us from HBM.  Reads only; speed only. */
int avlen_prefetch_l2(const void* const* ptrs, const int64_t* nbytes, int n, avlen_stream_t stream);

/* The path's one collective: the mean over the ranks of the trained parameters' flat gradient, once per optimiser step --
 * what DistributedDataParallel's reducer does for the reference (ss_baselines/savi/ddppo/algo/ddppo.py:75-96), as ONE in-place
 * ncclAllReduce(avg) over RCCL (xGMI inside a node) on the stream the backward ran on.  RCCL is bound at run time (dlopen).
 *   avlen_comm_unique_id: rank 0 draws the 128-byte id (HOST memory) and hands it to the others out of band (the host already has
 *                         torch.distributed / a store for that);
 *   avlen_comm_init_rank: every rank, collectively -> *comm;   avlen_comm_destroy at the end;
 *   avlen_grad_allreduce: bucket (device, `count` elements of dtype AVLEN_PREC_FP32 | AVLEN_PREC_BF16) <- mean over the ranks. */
int avlen_comm_unique_id(void* host_out, size_t bytes);
int avlen_comm_init_rank(void** comm, int nranks, const void* host_unique_id, int rank);
int avlen_comm_destroy(void* comm);
int avlen_grad_allreduce(void* bucket, size_t count, int dtype, void* comm, avlen_stream_t stream);

/* Step sequencer (csrc/sequencer.hip): a list of stream operations run by ONE call, in order.  The reference's rollout step has
 * two host round trips on its critical path (ppo_trainer.py:449-636: act_option -> host reads the option actions -> tokens ->
 * act_dialog -> envs.step); what the host enqueues after each of them is fixed once the argument buffers are known.
 *   AVLEN_CMD_GRAPH      launch the instantiated graph a (hipGraphExec_t) on stream b
 *   AVLEN_CMD_RECORD     record event a (hipEvent_t) on stream b
 *   AVLEN_CMD_WAIT       make stream a wait for event b
 *   AVLEN_CMD_MULTICOPY  avlen_multi_copy(a = src pointer array, b = dst pointer array, c = byte counts, n pairs) on stream d
 * All handles are HOST values (HIP runtime handles / host arrays of device pointers); nothing is allocated or synchronised. */
#define AVLEN_CMD_GRAPH 1
#define AVLEN_CMD_RECORD 2
#define AVLEN_CMD_WAIT 3
#define AVLEN_CMD_MULTICOPY 4
typedef struct { int op; int n; void* a; void* b; void* c; void* d; } avlen_cmd;
int avlen_cmds_run(const avlen_cmd* cmds, int n);

/* build / device info */
const char* avlen_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
