// Internal (non-ABI) entry points shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/avlen_hip.h"

// per-call context of the module-level helpers (modules.hip): stream, arithmetic mode, split-K scratch, optional ragged-batch
// descriptors, optional operand scratch of the large-M bf16 training products
struct avlen_ctx { hipStream_t st; int prec; void* gws; size_t gws_bytes; const int* live = nullptr; const int* seg = nullptr;
                   void* xs = nullptr; size_t xs_bytes = 0;
                   int x3 = 0; };      // fast paths: the 16-bit operands are compensated bf16 pairs (hi plane + lo plane)
// Operand options of the 16-bit MFMA GEMM family (igemm2.hip).  f16: the 16-bit operands / outputs are IEEE half instead of bf16.
// x3 ("bf16x3", compensated bf16): every operand is a pair of bf16 planes, hi = bf16(x) and lo = bf16(x - hi); the lo plane of A / W
// lies a_lo / b_lo BYTES behind the hi plane, the lo plane of the bf16 output c16_lo ELEMENTS behind C16 (0: hi only).
// rs_slabs / ln_slabs: deterministic row statistics (avlen_gemm_bf16_ln): the producer stores one partial slab [M][2] per 128-column
// tile (rs_slabs = ceil(N / 128), rowstats sized rs_slabs * M * 2, no zeroing), the consumer adds ln_slabs slabs in order.
struct avlen_g2_opts { int f16 = 0; int x3 = 0; long a_lo = 0, b_lo = 0, c16_lo = 0; int rs_slabs = 0, ln_slabs = 0; };
// Y = act(X W^T + b) + res;  dX = dY W (+ add);  G.w += dY^T X;  out[col] += sum_rows dY   (training products, modules.hip)
int avlen_i_linear(const avlen_ctx& c, const avlen_linear& L, const float* X, int ldx, float* Y, int ldy, int M, int act,
                   const float* res, int ldr);
int avlen_i_linear_dx(const avlen_ctx& c, const avlen_linear& L, const float* dY, int ldy, float* dX, int ldx, int M,
                      const float* add, int ldadd);
int avlen_i_linear_dw(const avlen_ctx& c, const avlen_linear& G, const float* dY, int ldy, const float* X, int ldx, int M);
// the AudioCNN's three convolutions as one launch, activations in LDS (audio3.hip); _ok: geometry / LDS budget covered
bool avlen_i_audio3_ok(const avlen_cnn3* n, int H, int W);
int avlen_i_audio3_fwd(const avlen_cnn3* const* nets, const float* x, const int* row_index, int groups, int B, int H, int W, void* const* outs,
                       hipStream_t st);
// single-query cross attention in memory space (cross1.hip; d = 256, 8 heads, S <= 320): per-head products against the rows of a
// [d][d] projection slice, the forward (scores + softmax + weighted memory sum) and the backward (dA, d memory rows)
bool avlen_i_cross1_ok(int d, int H, int S);
int avlen_i_cross1_expand(const float* X, int ldx, const float* W, int ldw, float* out, int B, hipStream_t st);
int avlen_i_cross1_reduce(const float* Z, const float* W, int ldw, const float* bias, float* Y, int ldy, int B, hipStream_t st);
int avlen_i_cross1_dw(const float* X, int ldx, const float* Z, float* dW, int ldw, int B, hipStream_t st);
int avlen_i_cross1_fwd(const float* A, const void* MEM16, long lo, const float* maskx, float* P, float* Mo, int B, int S, float scale,
                       hipStream_t st);
int avlen_i_cross1_bwd(const float* P, const float* DM, const float* A, const void* MEM16, long lo, float* dA, float* dMEM, int B, int S,
                       float scale, hipStream_t st);
// C [N1][N2] = beta C + A^T B over M rows, A / B row-major bf16 (gemm_tn.hip: the weight gradient without transposed operand copies)
size_t avlen_i_gemm_tn_workspace_bytes(long M, int N1, int N2);
int avlen_i_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, long M, int N1, int N2, float* C, int ldc, float beta,
                         void* ws, size_t ws_bytes, hipStream_t st);
int avlen_i_colsum_acc(const avlen_ctx& c, const float* dY, int ld, float* out, int rows, int N);
// conv weight gradient G.w[cout][KH*KW*C] += dY^T im2col(X) on the large-M bf16 route with the gather fused into the operand
// cast; AVLEN_NOT_BIG = route not applicable (caller: im2col + avlen_i_linear_dw)
#define AVLEN_NOT_BIG 100
int avlen_i_conv_dw16(const avlen_ctx& c, const avlen_linear& G, const float* dY, int ldy, const float* X, long B, int H, int W,
                      int C, int OH, int OW, int KH, int KW, int s, int pad);
// the same gradient WITHOUT any materialised operand (conv_bwd.hip): gw [cout][KH*KW*C] = dY^T im2col(X) (overwritten), gb[cout] +=
// column sums of dY (null: skipped); bf16 mode, valid convolutions, cout 32 / 64; partials in c.gws.  AVLEN_NOT_BIG: not applicable
int avlen_i_conv_dw_direct(const avlen_ctx& c, float* gw, float* gb, int cout, const float* dY, const float* X, long R, int H,
                           int W, int C, int OH, int OW, int KH, int KW, int s);
// 3-conv CNN forward on the 16-bit conv kernels that also keeps every conv's fp32 output (modules.hip; the training forward)
int avlen_i_cnn3_fwd16_keep(const avlen_cnn3* n, const float* x, int B, int H, int W, float* const* keep, float* out, int ld_out,
                            void* x16, void* const* a16, void* gws, size_t gws_bytes, hipStream_t st);
// conv data gradient without the im2col-shaped intermediate (conv_bwd.hip): dX = [act > 0] * conv_transpose(dY, W), W packed
// [cout][KH][KW][cin]; bf16 mode, cin 32 / 64, stride 1 / 2, valid convolutions.  AVLEN_NOT_BIG: not applicable
int avlen_i_conv_dx_direct(const avlen_ctx& c, const float* w, const float* dY, const float* act, float* dX, long R, int H, int W, int cin,
                           int OH, int OW, int cout, int KH, int KW, int s);

// few-row Linear against a tall weight matrix (train_gru.hip): out = x W^T + b, one wave per output column
bool avlen_i_skinny_linear_ok(int M, int K);
int avlen_i_skinny_linear(const float* x, int ldx, const float* W, const float* b, float* out, int ldo, int M, int N, int K,
                          hipStream_t st);
// fused masked GRU step (train_gru.hip): out = GRU(gi, hprev * mask), one wave per hidden unit
bool avlen_i_gru_step_ok(int N, int H);
int avlen_i_gru_step_fwd(const avlen_gru* p, const float* gi, const float* hprev, const float* mask, float* out, int N,
                         hipStream_t st);

int avlen_groupnorm_nhwc_ws(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                            int B, int HW, int C, int G, int relu, float eps, float* part, hipStream_t stream);
extern "C" size_t avlen_groupnorm_workspace_bytes(int B, int C);
int avlen_groupnorm_apply_bf16(const float* x, const float* stats, const float* gamma, const float* beta, const void* res16,
                               void* y16, int B, int HW, int C, int G, int relu, float eps, hipStream_t stream);
// avlen_layernorm_bwd that also leaves dx as bf16 rows [rows][d] (dx16) and adds its column sums to colsum[d] (both or neither)
int avlen_layernorm_bwd16(const float* dy, const float* xsum, const float* gamma, const float* mean, const float* rstd, float* dx,
                          float* dgamma, float* dbeta, int rows, int d, hipStream_t stream, void* dx16, float* colsum);
int avlen_layernorm_fwd16(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                          void* y16, float* mean, float* rstd, int rows, int d, float eps, hipStream_t stream);
int avlen_attention_fwd16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                          void* O16, int ldo16, const float* key_mask, float* lse, int B, int H, int Sq, int Sk, int D,
                          int causal, float scale, hipStream_t stream);
int avlen_zero_bytes(void* p, size_t bytes, hipStream_t stream);      // zero fill as kernel node(s) under capture
int avlen_preprocess_image_bf16(const void* x, int x_u8, void* y16, int B, int S, int C, float divisor, hipStream_t stream,
                                const int* row_index = nullptr);
// rows are grouped in items of `rows_per_item`; item i of the batch is item row_index[i] of src
int avlen_cast_bf16_indexed(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, const int* row_index,
                            int rows_per_item, hipStream_t stream, int fmt = 0);       // fmt: 0 bf16, 1 fp16
int avlen_groupnorm_apply_bf16_grouped(const void* const* x, int raw16, const float* const* stats,
                                       const float* const* gamma, const float* const* beta, const void* const* res16,
                                       void* const* y16, int groups, int B, int HW, int C, int G, int relu, float eps,
                                       hipStream_t stream, const float* const* rstats = nullptr,
                                       const float* const* rgamma = nullptr, const float* const* rbeta = nullptr, int rrelu = 0);
int avlen_conv2d_nhwc_bf16_grouped(const void* const* X, const void* const* Wp, float* const* Y32, void* const* Y16,
                                   float* const* gn_stats,
                                   int groups, int Bn, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                   void* ws, size_t ws_bytes, hipStream_t stream, const float* const* bias = nullptr,
                                   int act = 0, int stride_w = 0, const avlen_g2_opts* o = nullptr);
int avlen_gemm_bf16_grouped(const void* const* A, int lda, const void* const* B, int ldb, float* const* C32, int ldc32,
                            const float* const* bias, int groups, int M, int N, int K, int act, void* ws, size_t ws_bytes,
                            hipStream_t stream, const avlen_g2_opts* o = nullptr);
// y16_lo (elements): also write the low plane of the compensated bf16 pair at y16 + y16_lo
int avlen_layernorm_fwd16_dyn(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                              void* y16, float* mean, float* rstd, int rows, const int* rows_dev, int d, float eps,
                              hipStream_t stream, long y16_lo = 0);
int avlen_attention_fwd16_seg(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                              void* O16, int ldo16, const float* key_mask, float* lse, int B, int H, int Sq, int Sk, int D,
                              int causal, float scale, const int* seg_off, hipStream_t stream);
int avlen_gemm_bf16_ln(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16, int ldc16,
                       const float* bias, const float* residual, int ldr, int M, const int* M_dev, int N, int K, int act,
                       const float* ln_stats, const float* ln_s, float* rowstats, void* ws, size_t ws_bytes,
                       hipStream_t stream, const avlen_g2_opts* o = nullptr);
int avlen_attention_qkv16(const void* QKV16, int ld, void* O16, int ldo16, int B, int H, int S, int causal, float scale,
                          const int* seg_off, hipStream_t stream, int f16 = 0);
// lo planes (elements; 0 = plain bf16): qkv_lo behind QKV16, o_lo behind O16 -- compensated bf16 (three MFMAs per product, P split
// into hi + lo); only the <= 160-token instance exists in that mode
int avlen_attention_smt16(const void* QKV16, int ld, void* O16, int ldo16, int B, int H, int S, float scale,
                          const float* key_mask, const int* seg_off, hipStream_t stream, long qkv_lo = 0, long o_lo = 0,
                          float* O32 = nullptr, int ldo32 = 0, float* lse = nullptr);   // training forward: fp32 output + row log-sum-exp
// avlen_attention_bwd_bf16 with q | k | v taken from the packed bf16 projection the forward kept (QKV16 [R][ld16]; Q, K, V unused)
int avlen_attention_bwd_p16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                            const float* dO, int lddo, const float* key_mask, const float* lse, float* delta, float* dQ, int lddq,
                            float* dK, int lddk, float* dV, int lddv, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                            hipStream_t stream, const void* QKV16, int ld16, void* dQKV16 = nullptr, int ldd16 = 0,
                            float* colsum = nullptr);     // dQKV16: packed bf16 gradient rows + column sums instead of / beside fp32
int avlen_attention_q1(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* O16, int ldo16, int B,
                       int H, int Sk, float scale, const float* key_mask, const int* seg_off, hipStream_t stream, long o_lo = 0);
int avlen_gemm_bf16_dyn(const void* A, int lda, const void* B, int ldb, float* C32, int ldc32, void* C16, int ldc16,
                        const float* bias, const float* residual, int ldr, int M, const int* M_dev, int N, int K, int act,
                        void* ws, size_t ws_bytes, hipStream_t stream, const avlen_g2_opts* o = nullptr);
// in_stats / in_gamma / in_beta (optional, 16/16@64 and 32/32@32 only): X is a RAW conv output whose GroupNorm(16) + ReLU is
// applied while the halo is staged (saves the separate apply pass)
int avlen_dconv_bf16_grouped(const void* const* X, const void* const* Wp, void* const* Y16, float* const* gn_stats,
                             int groups, int B, int W, int Cin, int Cout, int K, hipStream_t stream,
                             const float* const* in_stats = nullptr, const float* const* in_gamma = nullptr,
                             const float* const* in_beta = nullptr);
bool avlen_dconv_supported(int W, int Cin, int Cout, int KH, int KW, int stride, int pad);

// ---- fused row-batch chain (chain.hip): a program of d=256 Linear / LayerNorm steps run by one kernel ----
#define AVLEN_CH_LOAD_X16 1   /* p0: bf16 [B][ld] rows -> LDS image `buf`, k columns (multiple of 8; zero-filled to a multiple of 64);
                                 compensated programs: p1 = the low plane of the same rows */
#define AVLEN_CH_LOAD_CUR 2   /* p0: fp32 [B/div][ld] (256 features), row r reads source row r/div -> registers, image -> `buf` */
#define AVLEN_CH_LINEAR 3     /* p0: bf16 W[256][ld], p1: fp32 bias[256] or null; input image `buf` (k columns, k % 8 == 0);
                                 act; res = 1 | 2 adds save slot 0 | 1; result -> registers and image `out_buf` */
#define AVLEN_CH_LAYERNORM 4  /* p0: gamma, p1: beta (eps 1e-5); result -> registers and image `out_buf` */
#define AVLEN_CH_SAVE 5       /* registers -> save slot `res` (0 | 1) */
#define AVLEN_CH_STORE 6      /* p0: fp32 [B/div][ld] or null, p1: bf16 [B/div][ld2] or null; div > 1: only the last row of
                                 every group of `div` rows is stored (to row r/div) */
#define AVLEN_CH_RECALL 7     /* save slot `res` -> registers and image `out_buf` */
#define AVLEN_CH_ATTN 8       /* 8-head attention inside groups of `seq` (1|2|4) consecutive rows: Q = save slot `res`,
                                 K = image `buf`, V = image `ld2`, p0: fp32 key mask [B/seq][seq] (1 = valid) or null,
                                 `scale`; result -> registers and image `out_buf` */
#define AVLEN_CH_ADD_PE 9     /* registers += p0[clamp(int(p1[r/div]), 0, k-1)][0..256): positional table p0 fp32 [k][256],
                                 p1 fp32 [B/div]; result -> registers and image `out_buf` */
#define AVLEN_CHAIN_MAX_OPS 40
typedef struct { int kind, k, ld, ld2, act, res, buf, out_buf, div, seq; float scale; int pad; const void* p0; const void* p1;
                 const void* p2; } avlen_chain_op;      /* p2: LINEAR: the low plane of the weights (compensated bf16 programs) */
typedef struct { int n; avlen_chain_op op[AVLEN_CHAIN_MAX_OPS]; } avlen_chain;
// x3: compensated bf16 (LINEAR needs p2; LOAD_X16 takes the low plane of its rows in p1); a block then takes 8 batch rows
int avlen_chain_run(const avlen_chain* prog, int B, hipStream_t stream, int x3 = 0);

// ---- preprocessing + stem + layers 1-2 of the ResNet towers (tower_head.hip): one workgroup per image, activations in LDS / registers ----
bool avlen_tower_head_supported(const avlen_resnet18* net, int S, int C);
int avlen_tower_head_bf16(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                          const float* divisors, const int* row_index, void* const* Y, int groups, int B, int S,
                          hipStream_t stream);

// ---- fused layers 3 + 4 of the ResNet towers (tower_tail.hip): one workgroup per image, activations in LDS ----
int avlen_tower_tail_bf16(const avlen_resnet18* const* nets, const void* const* X, void* const* Y, int groups, int B,
                          hipStream_t stream);

// ---- the ResNet tower in compensated bf16 (tower_x3.hip): stem + four band convs + layers 2-4 fused; fp32 layer-4 output ----
bool avlen_tower_x3_supported(const avlen_resnet18* net, int S, int C);
size_t avlen_tower_x3_workspace_bytes(int groups, int B);
int avlen_tower_x3_fwd(const avlen_resnet18* const* nets, const void* const* imgs, const int* img_u8, const int* channels,
                       const float* divisors, const int* row_index, void* const* Y, int groups, int B, int S, void* ws,
                       size_t ws_bytes, hipStream_t stream);
// clip_tower.hip: the 12 blocks of the CLIP text tower in one sequence-stationary launch -> E[b] = residual row at the EOT token
size_t avlen_clip_tower_stream_ws_bytes(int B);
// memo (device pointers, optional -- the memoised tower of avlen_clip_text_cached_fwd): the launch's work-list kernel compares
// tokens_new (B - 1 rows) with prev (B rows, the last one all zero; updated in place -- pass it as `tokens`) and only rows that differ
// go through the tower (E rows of the others stay untouched); hdr = [valid, rows through the tower, all-zero rows], zidx = the latter
struct avlen_clip_memo { const int64_t* tokens_new; int64_t* prev; int* hdr; int* zidx; };
int avlen_clip_tower_stream_fwd(const avlen_clip_text* p, const int64_t* tokens, float* E, int B, int f16, void* ws, size_t ws_bytes,
                                hipStream_t st, const avlen_clip_memo* memo = nullptr);
// fp32 rows -> compensated bf16 pair in one pass (hi plane at dst, lo plane `lo` elements behind it)
int avlen_cast_pair(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, long lo, hipStream_t stream);
int avlen_conv2d_nhwc_h16(const void* X, const void* Wp, const float* bias, const float* residual, float* Y32, void* Y16,
                          float* gn_stats, int Bn, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int act,
                          void* ws, size_t ws_bytes, hipStream_t stream, const avlen_g2_opts* o);
