/* vqa_hip.h — C ABI of libvqa_hip.so: the MI355X (gfx950) kernels behind VqaNet.forward/backward.
 *
 * The reference (OmerShubi/DL_VQA) has no FFI layer: its hot path is a chain of stock torch
 * operators inside models/model.py:53-67 and train.py:189-206.  Each entry point below replaces
 * one fused stage of that chain; the reference lines it stands in for are cited per function.
 *
 * Conventions:
 *   - every tensor argument is a raw DEVICE pointer into caller-owned memory (torch storage);
 *     the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - return value: 0 = success, otherwise a VQA_ERR_* code; vqa_last_error() gives the text;
 *   - no C++ exception crosses the ABI; functions are re-entrant (safe from any host thread, any device);
 *     the only state is the optional profiling hook, the VQA_* knobs (read once), a per-device record of
 *     kernels whose LDS limit has been raised and the cache of LSTM-sequence hipGraphs, all behind mutexes;
 *   - all floating point is IEEE fp32; contractions use v_mfma_f32_32x32x2_f32 (exact fp32);
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS; pointers and leading
 *     dimensions of GEMM operands must be multiples of 4 elements (16-byte vector loads);
 *   - images inside the library are NHWC; question activations are time-major [T][B][*].
 */
#ifndef VQA_HIP_H
#define VQA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported entry point changes its argument list or disappears (round 1: 1, round 2: 2, round 3: 3).
 * dl_vqa_amd/_lib.py parses this line and refuses a library that answers differently. */
#define VQA_ABI_VERSION 7

#define VQA_OK 0
#define VQA_ERR_INVALID 1 /* bad argument (shape, alignment, null pointer) */
#define VQA_ERR_HIP 2     /* a HIP runtime call or kernel launch failed */
#define VQA_ERR_WORKSPACE 3 /* workspace too small */

typedef void* vqa_stream_t; /* hipStream_t */

int vqa_abi_version(void);
const char* vqa_last_error(void);
/* 1 if a gfx950 device is visible to the HIP runtime, else 0 (never fails). */
int vqa_device_ok(void);

/* Re-read the VQA_* environment knobs (forced tile variants for parity tests, diagnostics).  They are read
 * once when the library is first used; tests that switch variants inside one process call this. */
int vqa_reload_knobs(void);

/* ---- profiling hook (bench.py: live HIP-event timing of one kernel family) ------------------ */
enum {
  VQA_K_GEMM = 0,
  VQA_K_CONV_FWD = 1,
  VQA_K_CONV_DGRAD = 2,
  VQA_K_CONV_WGRAD = 3,
  /* HBM-bound stages (bench.py reports GB/s against the 8 TB/s HBM3E peak for these) */
  VQA_K_L2NORM_FWD = 4,
  VQA_K_L2NORM_BWD = 5,
  VQA_K_ATT_SCORE_FWD = 6,
  VQA_K_ATT_SCORE_BWD = 7,
  VQA_K_ATT_APPLY_FWD = 8,
  VQA_K_ATT_APPLY_BWD = 9,
  VQA_K_ADAM = 10,
  VQA_K_SOFTCE = 11,
  VQA_K_DROPOUT = 12,
  VQA_K_LSTM_SEQ = 13, /* one whole vqa_lstm_seq_fwd / _bwd call (T launches or one graph replay) */
  VQA_K_COUNT = 14
};
/* Arm event bracketing for kernel family `kernel_id` whose launch tag equals `tag`
 * (tag < 0: any). kernel_id == VQA_K_COUNT arms every family; kernel_id < 0 disarms.
 * Resets the accumulated numbers. */
int vqa_prof_arm(int kernel_id, int tag);
/* The same for a SET of families: bit k of `mask` arms family k (mask 0 disarms). */
int vqa_prof_arm_mask(uint32_t mask, int tag);
/* Synchronise the recorded events; returns launches counted and their total device time. */
int vqa_prof_read(int* launches, float* total_ms);
/* Per (family, tag) totals of the recorded events: fills up to `cap` entries of ids / tags / launches /
 * total_ms and returns the number of distinct (family, tag) pairs seen. */
int vqa_prof_read_groups(int* ids, int* tags, int* launches, float* total_ms, int cap);

/* ---- generic GEMM ---------------------------------------------------------------------------
 * C[M][N] = act( A.B  (op) rowgroup + bias1 + bias2 ) (+ C if accumulate)
 *   transA = 0: A stored [M][K];  1: A stored [K][M]
 *   transB = 0: B stored [K][N];  1: B stored [N][K]   (torch nn.Linear weight => transB = 1)
 *   rowgroup (optional): value rowgroup[(m / rg_div) * rg_ld + n], rg_op 0 = add, 1 = multiply
 *     (the question projection tiled over image positions, models/model.py:187-193,224-231)
 *   relu: 0/1.   tag: free integer recorded by the profiling hook.
 * Replaces nn.Linear / 1x1 nn.Conv2d / LSTM projections (models/model.py:145,173-174,202,205)
 * and every dX / dW product of their backward passes.
 * Small outputs are split along K over workgroups; partial slabs live in `workspace`. */
int64_t vqa_gemm_workspace_bytes(int M, int N, int K);
int vqa_gemm(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
             float* C, int64_t ldc, int M, int N, int K, const float* bias1, const float* bias2,
             const float* rowgroup, int64_t rg_ld, int rg_div, int rg_op, int relu, int accumulate,
             float* aux /* optional [M][ldc]: the raw product A.B before the epilogue (v' for '*') */,
             float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream);

/* ---- image encoder: Conv2d(k=3, stride, pad=0) + ReLU + MaxPool2d(2,2) ----------------------
 * (models/model.py:72-84 ImageNet2). Activations NHWC, channel count padded to a multiple of 4
 * (CiP); weights re-packed per step from the torch layout [Co][Ci][3][3]. */
int vqa_nchw_to_nhwc4(const float* x_nchw, float* y_nhwc, int B, int C, int H, int W, vqa_stream_t stream);
/* wf[(ky*3+kx)*CiP + ci][co], wd[(ky*3+kx)*Co + co][ci]  (wd may be NULL) */
int vqa_conv_pack_weights(const float* w, float* wf, float* wd, int Co, int Ci, int CiP, vqa_stream_t stream);
/* pooled[B][Hp][Wp][Co], argmax[B][Hp][Wp][Co] in {0..3 = dy*2+dx of the winning pre-activation,
 * 4 = window dead (max <= 0, ReLU blocks the gradient)}; Hp = ((H-3)/stride+1)/2 (floor). */
int vqa_conv3x3_relu_pool_fwd(const float* x, const float* wf, const float* bias, float* pooled,
                              uint8_t* argmax, int B, int H, int W, int CiP, int Co, int stride,
                              int tag, vqa_stream_t stream);
/* dx[B][H][W][CiP] = gradient w.r.t. the conv input, routed through arg-max and ReLU. */
int vqa_conv3x3_dgrad(const float* dpooled, const uint8_t* argmax, const float* wd, float* dx, int B,
                      int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream);
/* dw[Co][Ci][3][3] and dbias[Co] (torch layouts; written, not accumulated). */
int64_t vqa_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride);
int vqa_conv3x3_wgrad(const float* x, const float* dpooled, const uint8_t* argmax, float* dw,
                      float* dbias, int B, int H, int W, int CiP, int Ci, int Co, int stride,
                      float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream);

/* Conv blocks with image.kernel_size != 3 (models/model.py:75-82: nn.Conv2d(kernel_size=k, stride) -> ReLU -> MaxPool2d(2,2);
 * utils/config_schema.py:59 admits any int, config.yaml:58 ships 3): the materialised form of the same product
 * (csrc/conv_generic.hip).  The caller writes the im2col matrix of a batch chunk, runs vqa_gemm on it (bias in the GEMM's
 * epilogue, relu = 0) and pools; backward routes the pooled gradient to the pre-pool rows, takes dW = dY^T.cols and
 * dcols = dY.wk with vqa_gemm and folds dcols back.  Ho = (H - ks)/stride + 1, rows r = (b*Ho + yo)*Wo + xo, K index
 * (ky*ks + kx)*CiP + ci; ks in 1..15, stride 1 or 2, CiP % 4 == 0, Co % 4 == 0; arg-max bytes as above (0..3, 4 = dead).
 *   wk [Co][ks*ks*CiP] from torch's [Co][Ci][ks][ks] (channels >= Ci zero); vqa_convk_unpack_wgrad is its inverse for dW. */
int vqa_convk_pack_weights(const float* w, float* wk, int Co, int Ci, int CiP, int ks, vqa_stream_t stream);
int vqa_convk_unpack_wgrad(const float* dwk, float* dw, int Co, int Ci, int CiP, int ks, vqa_stream_t stream);
int vqa_convk_im2col(const float* x, float* cols, int B, int H, int W, int CiP, int ks, int stride, vqa_stream_t stream);
/* y [B*Ho*Wo][Co] = the convolution output with its bias, before the ReLU */
int vqa_convk_relu_pool(const float* y, float* pooled, uint8_t* argmax, int B, int Ho, int Wo, int Co, vqa_stream_t stream);
int vqa_convk_route(const float* dpooled, const uint8_t* argmax, float* dy, int B, int Ho, int Wo, int Co, vqa_stream_t stream);
int vqa_convk_col2im(const float* dcols, float* dx, int B, int H, int W, int CiP, int ks, int stride, vqa_stream_t stream);

/* First conv block, dedicated path (Cin <= 3, stride 1, Co in {32, 64}, W % 4 == 0): reads the caller's
 * NCHW image directly (no layout conversion), weights/bias in torch layout, same pooled/argmax outputs
 * as vqa_conv3x3_relu_pool_fwd.  vqa_conv0_supported() tells whether a shape takes this path. */
int vqa_conv0_supported(int Ci, int H, int W, int Co, int stride);
int vqa_conv0_relu_pool_fwd(const void* x_nchw, int x_is_fp16 /* 1: __half NCHW as the dataset stores it, widened where
                            the patch is staged (preprocess_images.py:39-53); 0: float */, const float* w, const float* bias, void* pooled,
                            int pooled_is_bf16 /* 0: fp32 out; 1: fp32 MFMA, P_0 stored as bf16; 2 (bf16 path):
                            image and weights rounded to bf16, two 32x32x16 bf16 MFMA k-steps, bf16 out; 3 (fp32x3 path): fp32
                            MFMA, the output written x3-packed (vqa_x3_pack's form, 6 bytes per element); 4: as 2 with the
                            pooled map channel-blocked, [B][Co/16][Hp][Wp][16] (what vqa_pconv_* read; arg-max stays NHWC) */, uint8_t* argmax, int B, int Ci,
                            int H, int W, int Co, vqa_stream_t stream);
int64_t vqa_conv0_wgrad_workspace_bytes(int Co);
int vqa_conv0_wgrad(const void* x_nchw, int x_is_fp16, const float* dpooled, const uint8_t* argmax, float* dw, float* dbias,
                    int B, int Ci, int H, int W, int Co, float* workspace, int64_t workspace_bytes,
                    vqa_stream_t stream);

/* bf16 path: the first block's weight gradient on bf16 MFMA (image rounded to bf16 where it is staged, bf16 pooled
 * gradient as written by vqa_conv3x3_dgrad_bf16, fp32 accumulation, fp32 dw / dbias).  Same workspace query. */
int vqa_conv0_wgrad_bf16(const void* x_nchw, int x_is_fp16, const void* dpooled_bf16, const uint8_t* argmax, float* dw,
                         float* dbias, int B, int Ci, int H, int W, int Co, float* workspace,
                         int64_t workspace_bytes, vqa_stream_t stream);

/* ---- dropout (nn.Dropout, 7 sites: models/model.py:84,156,185,186,194,201,204) ---------------
 * y = x * keep(seed, i) / (1-p); keep() is a counter-based hash, so backward calls the same
 * function on the gradient. In-place allowed. */
int vqa_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, vqa_stream_t stream);
/* y += x * keep(seed, i) / (1-p): the backward of a dropout whose gradient joins another one (attention.drop on v:
 * d loss / d v = dropout-mask * d loss / d v_in + the weighted-sum branch, models/model.py:185,62) in ONE pass. */
int vqa_dropout_add(const float* x, float* y, int64_t n, float p, uint64_t seed, vqa_stream_t stream);

/* ---- L2 normalisation over channels (models/model.py:56), fused with image.drop (model.py:84) -
 * u = dropout(pooled); norm = ||u||_2 per row; vn = u / (norm + 1e-12).  rows = B*P, C channels.
 * vdrop (optional): a second output dropout_{p2, seed2}(vn), fp32 or bf16 -- attention.drop applied to v
 * (models/model.py:185), i.e. the v_conv operand, written in the same pass. */
int vqa_l2norm_fwd(const float* pooled, float* vn, float* norm, int64_t rows, int C, float p,
                   uint64_t seed, void* vdrop, int vdrop_is_bf16, float p2, uint64_t seed2, vqa_stream_t stream);
int vqa_l2norm_bwd(const float* dvn, const float* vn, const float* norm, void* dpooled,
                   int dpooled_mode /* 0 fp32 [rows][C]; 1 bf16 [rows][C] (bf16 path: no separate conversion pass); 2 bf16
                                       channel-blocked [rows / positions][C/16][positions][16] for vqa_pconv_dgrad / _wgrad */,
                   int64_t rows, int positions /* rows per image: mode 2 only */, int C, float p, uint64_t seed,
                   vqa_stream_t stream);
/* The same with the incoming gradient JOINED in the kernel instead of read from a [rows][C] tensor:
 *   d loss / d vn [b*P + p][:] = sum_g probs[b][g][p] * dout[b][g*C : (g+1)*C]            (the weighted-sum branch, model.py:62:
 *                                                                                          what vqa_att_apply_bwd writes as dvn)
 *                               + dropout_{p_v, seed_v}-mask * dv_in[b*P + p][:]           (attention.drop on v, model.py:185:
 *                                                                                          what vqa_dropout_add adds)
 * -- two passes over a [B*P][C] fp32 tensor less per step; vqa_att_apply_bwd is then called with dvn = NULL. */
int vqa_l2norm_bwd_joined(const float* dout, int64_t dout_ld, const float* probs, int G, const float* dv_in, float p_v,
                          uint64_t seed_v, const float* vn, const float* norm, void* dpooled, int dpooled_mode, int64_t rows,
                          int positions, int C, float p, uint64_t seed, vqa_stream_t stream);

/* ---- question encoder (models/model.py:134-166 questionNet) ----------------------------------
 * x[t][b][:] = tanh(dropout(emb[q[b][t]]))   (embedding -> drop -> tanh, model.py:155-157).
 * Token ids outside [0, V) -- nn.Embedding raises for them -- are counted into *bad_tokens (device int32,
 * may be NULL; the caller zeroes and reads it) and contribute a zero row, forward and backward alike. */
int vqa_embed_tanh_fwd(const int64_t* q, const float* emb, float* x, int B, int T, int E, int V,
                       float p, uint64_t seed, int32_t* bad_tokens, vqa_stream_t stream);
/* demb[v][:] = sum over the slots (b,t) with q[b][t] == v of dx * (1 - x^2) * dropmask, in slot order
 * (deterministic, no atomics); every row of demb [V][E] is WRITTEN (row 0 = padding_idx and unused rows: zeros). */
/* workspace (optional, vqa_embed_tanh_bwd_workspace_bytes): the slots are binned by token first and every row sums its own
 * (sorted) bin -- O(B*T + V) instead of the workspace-less form's scan of all B*T slots per vocabulary row; same bits. */
int64_t vqa_embed_tanh_bwd_workspace_bytes(int B, int T, int V);
int vqa_embed_tanh_bwd(const int64_t* q, const float* x, const float* dx, float* demb, int B, int T,
                       int E, int V, float p, uint64_t seed, void* workspace /* may be NULL */, int64_t workspace_bytes,
                       vqa_stream_t stream);
/* One LSTM time step for one direction (gate order i,f,g,o; nn.LSTM, model.py:145-149):
 *   pre = xg[b] + hg[b]   (xg = x W_ih^T + b_ih + b_hh for time t, hg = h_in W_hh^T, both [B][4H])
 *   rows with t >= q_len[b] keep (h,c) unchanged — the packed-sequence semantics of model.py:159-164.
 * Writes activated gates [B][4H], c_out, h_out; if c_final != NULL also c_final[b*cf_ld + j]. */
int vqa_lstm_cell_fwd(const float* xg, const float* hg, const float* c_in, const float* h_in,
                      const int64_t* q_len, int t, float* gates, float* c_out, float* h_out,
                      float* c_final, int64_t cf_ld, int B, int H, vqa_stream_t stream);
/* Backward of one step. In: dh [B][H] (grad w.r.t. h_out), dc [B][H] (grad w.r.t. c_out).
 * Out: dgates [B][4H] (pre-activation grads, 0 for inactive rows), dc <- grad w.r.t. c_in,
 * dh <- pass-through part of grad w.r.t. h_in (dh for inactive rows, 0 for active rows); the caller
 * then accumulates dgates.W_hh onto dh with vqa_gemm(accumulate=1). */
int vqa_lstm_cell_bwd(const float* gates, const float* c_in, const float* c_out, const int64_t* q_len,
                      int t, float* dh, float* dc, float* dgates, int B, int H, vqa_stream_t stream);

/* The whole recurrence of one question batch as ONE call (models/model.py:145-149, 159-164): T dependent launches,
 * each covering every direction; a forward step is h_{t-1} W_hh^T on the MFMA engine with the cell as its epilogue,
 * a backward step is dgates_t W_hh over the whole K = 4H inside one workgroup (split over its MFMA waves by gate,
 * combined through LDS) with the cell backward of the next time as its epilogue -- they replace vqa_gemm +
 * vqa_lstm_cell_fwd, resp. vqa_lstm_cell_bwd + split-K vqa_gemm + reduce, per step and direction.
 * Requires H % 32 == 0 (vqa_lstm_step_supported); other sizes use the unfused entry points above.
 *   use_graph != 0: the chain is replayed as an explicit hipGraph, built once per distinct argument set (shapes AND
 *   pointers) and cached inside the library (at most 32 graphs per process, least recently used dropped; a caller
 *   whose buffers move on every call is detected and served by plain launches).  vqa_lstm_graph_stats reports
 *   replays / graph builds / plain-launch fallbacks since load and returns the number of cached graphs.
 * Layouts: time-major.  The forward direction reads state slot t and writes slot t+1 of Hs / Cs [T+1][B][H] at time
 * t, the reverse direction reads slot t+1 and writes slot t; the caller zeroes the initial slot (0, resp. T). */
typedef struct {
  const float* w_hh; /* [4H][H], PyTorch gate order i,f,g,o */
  const float* xg;   /* fwd: [T][B][4H] = x_t W_ih^T + b_ih + b_hh (vqa_gemm) */
  float* gates;      /* [T][B][4H] gate activations, zero rows where t >= q_len[b] (fwd: out, bwd: in) */
  float* Hs;         /* [T+1][B][H] */
  float* Cs;         /* [T+1][B][H] */
  float* c_final;    /* fwd: optional [B][cf_ld], final cell state c_n (the question feature, model.py:164-166) */
  float* dgates;     /* bwd: [T][B][4H] out: gradient w.r.t. the gate pre-activations */
  float* dh;         /* bwd: [B][H] work; on entry d loss / d h_n (zeros for this model) */
  float* dc;         /* bwd: [B][H] work; on entry d loss / d c_n, on exit d loss / d c at the first processed time */
  int reverse;       /* 0: t = 0 .. T-1, 1: t = T-1 .. 0 */
} vqa_lstm_dir_t;
int vqa_lstm_step_supported(int H);
int vqa_lstm_seq_fwd(const vqa_lstm_dir_t* dirs, int ndir, const int64_t* q_len, int B, int T, int H,
                     int64_t cf_ld, int use_graph, vqa_stream_t stream);
int vqa_lstm_seq_bwd(const vqa_lstm_dir_t* dirs, int ndir, const int64_t* q_len, int B, int T, int H,
                     int use_graph, vqa_stream_t stream);
int vqa_lstm_graph_stats(int* replays, int* builds, int* plain);

/* ---- attention (models/model.py:169-195 Attention, 208-221 image_question_attention) ---------
 * x = relu(v' (+|*) q') comes from vqa_gemm(rowgroup = q') as xs[m][n]; for do_option '|' (model.py:192)
 * x = relu(cat[v', tile(q')]) has 2*mid channels: xs holds the v' half and qcat = q' [B][mid] the other.
 * score[b][g][p] = bx[g] + sum_n dropout(x[b*P+p][n]) * wx[g*wx_ld + n]            (x_conv, model.py:194) */
int vqa_att_score_fwd(const void* xs, int xs_is_bf16 /* the bf16 path stores x as bf16 */, const float* wx, int wx_ld,
                      const float* bx, float* score, int B, int P, int mid, int G, float p, uint64_t seed,
                      const float* qcat /* '|' only */, vqa_stream_t stream);
/* Backward of the score + combine stage, in place on xs.  mode: 0 '+', 1 '*', 2 '|'.
 *   xs      <- gradient w.r.t. v' (what the v_conv dW / dX GEMMs consume)
 *   dq_part[b*RS+rs][mid]      partial sums of the gradient w.r.t. q' (RS = vqa_att_row_splits(P))
 *   dwx_part[b*RS+rs][G][xld]  partial sums of the x_conv weight gradient, xld = mid (2*mid for '|')
 * '*' needs vprime = the raw v' (aux output of the forward GEMM) and qp = q'; '|' needs qp. */
int vqa_att_row_splits(int P);
int vqa_att_score_bwd(const float* dscore, const float* wx, int wx_ld, void* xs_inout, int xs_is_bf16, float* dwx_part,
                      float* dq_part, int B, int P, int mid, int G, float p, uint64_t seed, int mode,
                      const float* vprime, const float* qp, vqa_stream_t stream);
/* probs = softmax_p(score); out[b*out_ld + g*C + c] = sum_p probs[b][g][p] * vn[b][p][c] */
int vqa_att_apply_fwd(const float* score, const float* vn, float* probs, float* out, int64_t out_ld,
                      int B, int P, int C, int G, vqa_stream_t stream);
/* dscore[b][g][p] and dvn[b][p][c] (written; NULL: skipped, see vqa_l2norm_bwd_joined) from dout[b*dout_ld + g*C + c]; dscore_rowsum (optional)
 * [b][g] = sum_p dscore[b][g][p], the per-sample part of the x_conv bias gradient. */
int vqa_att_apply_bwd(const float* dout, int64_t dout_ld, const float* probs, const float* vn,
                      float* dscore, float* dvn, float* dscore_rowsum, int B, int P, int C, int G,
                      vqa_stream_t stream);

/* ---- loss head (train.py:190-207, utils/train_utils.py:12-25) -------------------------------
 * loss_rows[b] = sum_k -log_softmax(logits[b])[a_idx[b][k]-1] * a_val[b][k]/10 * inv_batch
 * score_rows[b] = min(1, 0.3 * a_val of the arg-max answer)
 * dlogits (optional) = d(sum_b loss_rows)/dlogits. a_idx is 1-based, 0 = padding. */
int vqa_softce_fwd_bwd(const float* logits, int64_t ld, const int64_t* a_idx, const int64_t* a_val,
                       int kmax, int B, int A, float inv_batch, float* loss_rows, float* score_rows,
                       float* dlogits, int64_t dld, vqa_stream_t stream);

/* ---- reductions / pointwise helpers -------------------------------------------------------- */
/* out[n] (+)= sum_m x[m*ld + n]; if mask != NULL rows of x where mask[m*cols+n] == 4 are skipped
 * (conv bias gradient over dead pool windows). workspace: vqa_colsum_workspace_bytes(rows, cols). */
int64_t vqa_colsum_workspace_bytes(int64_t rows, int cols);
int vqa_colsum(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols, float* out,
               int accumulate, float* workspace, int64_t workspace_bytes, vqa_stream_t stream);
/* out[g] = sum_{b,p} x[b][g][p] */
int vqa_sum_bgp(const float* x, float* out, int B, int G, int P, vqa_stream_t stream);
/* out[b][n] = sum_r part[(b*parts + r)*cols + n] */
int vqa_sum_parts(const float* part, float* out, int batch, int parts, int cols, vqa_stream_t stream);
/* dx = dy * dropmask * (y > 0): backward of dropout(relu(.)) given y = relu output (model.py:201-204) */
int vqa_relu_drop_bwd(const float* y, const float* dy, float* dx, int64_t n, float p, uint64_t seed,
                      vqa_stream_t stream);
/* y[r*ldy + c] = a[r*lda + c] + (b ? b[r*ldb + c] : 0) for r < rows, c < cols (strided add / copy;
 * in-place allowed): gradient joins such as d(combined) = d(cat[v, q]) (models/model.py:64). */
int vqa_add2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy,
              int64_t rows, int cols, vqa_stream_t stream);

/* x[i] *= *scalar (scalar is a DEVICE pointer): chains an upstream loss gradient without a host sync */
int vqa_scale_by(float* x, int64_t n, const float* scalar, vqa_stream_t stream);

/* ---- input pipeline ----------------------------------------------------- */
/* y[i] = (float)x[i] for n IEEE half values: the dataset stores image features as fp16 [N,3,S,S]
 * (preprocessing/preprocess_images.py:39-53) and the reference casts them to fp32 on the HOST per sample
 * (preprocessing/data_preprocessing.py:167-176); here the fp16 batch is uploaded as is (half the PCIe bytes)
 * and widened on the device.  x and y 16-byte aligned or n small; layout unchanged (NCHW). */
int vqa_half_to_float(const void* x_f16, float* y, int64_t n, vqa_stream_t stream);

/* ---- bf16 path (BASELINE configs[3]: bf16 MFMA conv / FC, fp32 accumulate, fp32 LSTM) -------------------
 * Opt-in second instantiation of the engine on v_mfma_f32_32x32x16_bf16.  Parameters, Adam state, the LSTM and
 * every reduction stay fp32; bf16 tensors are raw uint16 buffers (IEEE bfloat16, round to nearest even).
 * The fp32 entry points above remain the parity path. */
/* y = bf16(x) / y = float(x) element-wise; y[c][r] = bf16(x[r][c]) (a weight's bf16 copy in the other orientation) */
int vqa_f32_to_bf16(const float* x, void* y_bf16, int64_t n, vqa_stream_t stream);
int vqa_bf16_to_f32(const void* x_bf16, float* y, int64_t n, vqa_stream_t stream);
int vqa_f32_to_bf16_transpose(const float* x, void* y_bf16, int rows, int cols, vqa_stream_t stream);
/* y = bf16(x * keep(seed, i) / (1-p)): nn.Dropout fused with the bf16 copy (attention.drop on v, model.py:185) */
int vqa_dropout_to_bf16(const float* x, void* y_bf16, int64_t n, float p, uint64_t seed, vqa_stream_t stream);
/* vqa_gemm with bf16 A and B (same layouts / trans flags; leading dimensions in ELEMENTS, multiples of 8; K % 8 == 0;
 * a reduction-major operand needs its row length % 8 == 0), fp32 accumulation, the same fused epilogue; C is fp32,
 * or bf16 when c_is_bf16 (then no accumulate).  bias / rowgroup / aux stay fp32. */
int64_t vqa_gemm_bf16_workspace_bytes(int M, int N, int K);
int vqa_gemm_bf16(const void* A, int64_t lda, int transA, const void* B, int64_t ldb, int transB, void* C,
                  int64_t ldc, int c_is_bf16, int M, int N, int K, const float* bias1, const float* bias2,
                  const float* rowgroup, int64_t rg_ld, int rg_div, int rg_op, int relu, int accumulate,
                  float* aux, float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream);

/* Tall bf16 GEMM with a short reduction (csrc/gemm_tall_bf16.hip): C[M][N] = act(A[M][K] . W[N][K]^T (+|*) rowgroup), bf16
 * result -- the v_conv forward of the bf16 path (models/model.py:173,187-193: M = B * positions, N = mid, K = image
 * features).  Persistent 256 x 128 tiles, LDS-DMA staging across tile boundaries, 16-byte stores.  Operands as
 * vqa_gemm_bf16 with transA = 0, transB = 1; rowgroup (optional) [M / rg_div][N] fp32, rg_op 0 = add, 1 = multiply. */
int vqa_gemm_tall_bf16_supported(int M, int N, int K, int rg_div, int has_rowgroup);
int vqa_gemm_tall_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                       const float* rowgroup, int64_t rg_ld, int rg_div, int rg_op, int relu, int tag, vqa_stream_t stream);

/* Convolution blocks of the bf16 path (models/model.py:80-82 and their autograd): activations NHWC bf16, weights
 * re-packed per step to bf16 as wfT [Co][9*CiP] and wdT [CiP][9*Co] (K index = (tap, channel)), arg-max bytes and
 * semantics as in the fp32 entry points, fp32 accumulation, fp32 weight / bias gradients.
 *   forward: CiP % 64 == 0; pooled is bf16, or fp32 when pooled_is_bf16 == 0 (the last block feeds the fp32 L2 norm);
 *   dgrad:   Co % 64 == 0; dx is bf16, or fp32 when dx_is_bf16 == 0;
 *   wgrad:   CiP, Co multiples of 8; both operands reach the MFMAs through ds_read_b64_tr_b16 (reduction-major). */
int vqa_conv_pack_weights_bf16(const float* w, void* wfT, void* wdT /* may be NULL */, int Co, int Ci, int CiP,
                               vqa_stream_t stream);
int vqa_conv3x3_relu_pool_fwd_bf16(const void* x, const void* wfT, const float* bias, void* pooled,
                                   int pooled_is_bf16, uint8_t* argmax, int B, int H, int W, int CiP, int Co,
                                   int stride, int tag, vqa_stream_t stream);
int vqa_conv3x3_dgrad_bf16(const void* dpooled, const uint8_t* argmax, const void* wdT, void* dx, int dx_is_bf16,
                           int B, int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream);
int64_t vqa_conv3x3_wgrad_bf16_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride);
int vqa_conv3x3_wgrad_bf16(const void* x, const void* dpooled, const uint8_t* argmax, float* dw, float* dbias,
                           int B, int H, int W, int CiP, int Ci, int Co, int stride, float* workspace,
                           int64_t workspace_bytes, int tag, vqa_stream_t stream);

/* Patch convolutions of the bf16 path (csrc/conv_patch_bf16.hip; models/model.py:80-82 and their autograd), 3x3, stride 1:
 * a persistent workgroup keeps an input patch in LDS and takes the nine taps as shifted fragment reads of it, so every
 * input pixel is fetched once per workgroup instead of once per tap.  Same results contract as the vqa_conv3x3_*_bf16 entry
 * points (bf16 operands, fp32 accumulation, arg-max codes 0-3 / 4 = dead window); what differs:
 *   - every tensor a patch is cut from is channel-blocked "C16": [B][C/16][H][W][16] -- the activations between the blocks
 *     (bf16), the pooled gradients (bf16) AND the arg-max bytes;
 *   - the weights are packed per step into fragment-ordered images (vqa_pconv_pack_weights; wf_img for forward, wd_img =
 *     flipped + transposed for backward-data, vqa_pconv_weights_bytes each);
 *   - forward patches are copied by buffer_load ... lds; backward patches (the pre-pool gradient: dP where the arg-max byte
 *     names the pixel, zero elsewhere and on the border) are ROUTED in the kernel from dP + arg-max -- the pre-pool
 *     gradient, four times the pooled one, never exists in HBM.
 * Shapes: Ci % 16 == 0, Co % 64 == 0 (forward); additionally Ci % 64 == 0 for backward-data (vqa_pconv_supported). */
int vqa_pconv_supported(int H, int W, int Ci, int Co, int stride);
int64_t vqa_pconv_weights_bytes(int Ci, int Co);
int vqa_pconv_pack_weights(const float* w /* [Co][Ci][3][3] */, void* wf_img /* may be NULL */, void* wd_img /* may be NULL */,
                           int Co, int Ci, vqa_stream_t stream);
/* x C16 [B][Ci/16][H][W][16] -> pooled (bf16 C16 [B][Co/16][Hp][Wp][16] or fp32 NHWC [B][Hp][Wp][Co]) and arg-max bytes
 * (always C16 [B][Co/16][Hp][Wp][16]) */
int vqa_pconv_fwd(const void* x, const void* wf_img, const float* bias, void* pooled, int pooled_is_bf16, uint8_t* argmax,
                  int B, int H, int W, int Ci, int Co, int tag, vqa_stream_t stream);
/* dX of the block whose INPUT map is H x W x Ci, from the pooled gradient and arg-max bytes (both C16, [B][Co/16][Hp][Wp][16]).
 * dx_mode: 0 fp32 NHWC, 1 bf16 NHWC (feeds vqa_conv0_wgrad_bf16), 2 bf16 C16 (the pooled gradient of the block below) */
int vqa_pconv_dgrad(const void* dpooled, const uint8_t* argmax, const void* wd_img, void* dx, int dx_mode, int B, int H, int W,
                    int Ci, int Co, int tag, vqa_stream_t stream);
/* weight + bias gradient: dw [Co][Ci][3][3], dbias [Co] fp32.  A workgroup holds a whole [9 taps x 64 ci] x [128 co] block of
 * dW in its accumulators and streams 4 x 32-pixel tiles of x (C16; LDS patch, nine shifted transpose-reads) and of the routed
 * pre-pool gradient through LDS; one fp32 slab per workgroup, summed by a reduce kernel (deterministic).  Ci % 64 == 0,
 * Co % 128 == 0, and (Ci / 64) * (Co / 128) in {1, 2, 4, 8}; dbias is the masked sum of dpooled (arg-max != 4), summed by the
 * same kernel as the pieces of dpooled pass through it. */
int vqa_pconv_wgrad_supported(int H, int W, int Ci, int Co);
int64_t vqa_pconv_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co);
int vqa_pconv_wgrad(const void* x, const void* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B, int H, int W,
                    int Ci, int Co, float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream);

/* fp32 patch backward-data (csrc/conv_patch_f32.hip; the autograd of models/model.py:80-82 on the fp32 headline path): the same
 * dX as vqa_conv3x3_dgrad (stride 1) up to fp32 summation order, from the same operands (dpooled fp32 NHWC, argmax NHWC) --
 * a persistent workgroup builds the pre-pool gradient of an 8-channel slice in LDS once (routed from dpooled + argmax) and
 * takes the nine taps as shifted fragment reads over a FLATTENED segment of the padded map (no 2-D edge waste at 111- and
 * 54-pixel maps), exact fp32 MFMA.  wd_img = vqa_pconvf_pack_weights(w) (flipped, transposed, fragment order),
 * vqa_pconvf_weights_bytes.  Ci % 64 == 0, Co % 8 == 0, W <= 256 (vqa_pconvf_supported). */
int vqa_pconvf_supported(int H, int W, int Ci, int Co, int stride);
int64_t vqa_pconvf_weights_bytes(int Ci, int Co);
int vqa_pconvf_pack_weights(const float* w /* [Co][Ci][3][3] */, float* wd_img, int Co, int Ci, vqa_stream_t stream);
int vqa_pconvf_dgrad(const float* dpooled, const uint8_t* argmax, const float* wd_img, float* dx, int B, int H, int W, int Ci,
                     int Co, int tag, vqa_stream_t stream);

/* ---- fp32 on the bf16 matrix cores ("fp32x3": csrc/x3_core.hpp) -------------------------------
 * The same fp32 tensors, layouts and results contract as the fp32 entry points above; inside the K loop every
 * fp32 operand element is split exactly into three bf16 terms (8 + 8 + 8 significand bits) and a product is
 * accumulated in fp32 from its six partial products of weight >= 2^-16 (the dropped ones are below 2^-24 of the
 * product, less than one fp32 rounding): six v_mfma_f32_32x32x16_bf16 instead of eight v_mfma_f32_32x32x2_f32 per
 * 32x32x16 block, 2.67x the fp32 MFMA rate at fp32 accuracy.  Arguments as vqa_conv3x3_relu_pool_fwd /
 * vqa_conv3x3_dgrad / vqa_conv3x3_wgrad (models/model.py:72-84), except that the packed weights wf / wd are handed over
 * already split: three bf16 planes (hi, mid, lo), each in the layout of vqa_conv_pack_weights and one after the other
 * (vqa_x3_split of the fp32 packing, once per step); shapes: CiP, Co multiples of 32 and a conv output
 * row of at least 32 pixels (vqa_conv3x3_x3_supported), other layers keep the fp32 MFMA entry points. */
int vqa_conv3x3_x3_supported(int H, int W, int CiP, int Co, int stride);
/* the operand split of those kernels on its own (the packed weights, once per step; tests): x[n] -> three planes of n
 * bf16, x == hi + mid + lo exactly */
int vqa_x3_split(const float* x, void* hi, void* mid, void* lo, int64_t n /* multiple of 4 */, vqa_stream_t stream);
/* x3-packed activations: a tensor [pixels][C] (C % 4 == 0) split once per tensor instead of by every workgroup that
 * reads it: every four consecutive channels are 24 bytes hi[4] mid[4] lo[4] bf16 (6 bytes per element, same order).
 * The forward and wgrad entry points take their input x either as fp32 (x_packed = 0) or in this form (1). */
int vqa_x3_pack(const float* x, void* out /* 6 bytes per element */, int64_t n /* multiple of 4 */, vqa_stream_t stream);
int vqa_conv3x3_relu_pool_fwd_x3(const void* x, int x_packed, const void* wf_planes, const float* bias, void* pooled,
                                 int pooled_packed /* 1: the output is written x3-packed, for the next block */,
                                 uint8_t* argmax, int B, int H, int W, int CiP, int Co, int stride, int tag,
                                 vqa_stream_t stream);
/* dpooled: fp32 (dp_packed = 0) or x3-packed (1): the arg-max routing then is a mask on the packed halves */
int vqa_conv3x3_dgrad_x3(const void* dpooled, int dp_packed, const uint8_t* argmax, const void* wd_planes, float* dx, int B,
                         int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream);
int64_t vqa_conv3x3_wgrad_x3_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride);
/* One pass over a pooled gradient [windows][Co] for its two needs in the fp32x3 backward: the x3-packed copy (read by
 * dgrad and wgrad) and the bias gradient (sum over the windows whose ReLU was alive, arg-max byte != 4). */
int64_t vqa_x3_pack_pooled_grad_workspace_bytes(int Co);
int vqa_x3_pack_pooled_grad(const float* dpooled, const uint8_t* argmax, void* packed, float* dbias, int64_t windows, int Co,
                            float* workspace, int64_t workspace_bytes, vqa_stream_t stream);
/* dpooled (fp32): the bias gradient is summed from it unless dbias is NULL (the caller has it from
 * vqa_x3_pack_pooled_grad); dpooled_packed (optional): the same gradient x3-packed, read by the contraction instead */
int vqa_conv3x3_wgrad_x3(const void* x, int x_packed, const float* dpooled, const void* dpooled_packed,
                         const uint8_t* argmax, float* dw, float* dbias, int B, int H, int W, int CiP, int Ci, int Co,
                         int stride, float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream);

/* vqa_gemm on the split kernels: the same arguments and results contract (large contractions: the attention stage's
 * v_conv forward / dW / dX); 192 x 128 tiles, split-K slabs in `workspace` when the output has few tiles. */
int64_t vqa_gemm_x3_workspace_bytes(int M, int N, int K);
int vqa_gemm_x3(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc,
                int M, int N, int K, const float* bias1, const float* bias2, const float* rowgroup, int64_t rg_ld,
                int rg_div, int rg_op, int relu, int accumulate, float* aux, float* workspace, int64_t workspace_bytes,
                int tag, vqa_stream_t stream);

/* ---- optimiser: torch.optim.Adam defaults over one flat buffer (train.py:55,80) ------------- */
int vqa_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
             float beta1, float beta2, float eps, int step, float grad_scale, vqa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQA_HIP_H */
