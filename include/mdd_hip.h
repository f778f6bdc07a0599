/* mdd_hip.h -- C ABI of libmdd_hip.so: the MI355X (gfx950) engine for the hot path of
 * kushal-bhargav/multimodal_dataset_distillation: distill.py's inner `syn_steps` unrolled
 * training + bi-trajectory matching loop.
 *
 * Plain pointers and sizes only; all `*_dev` pointers are DEVICE pointers (hipMalloc'ed /
 * torch.Tensor.data_ptr()), `stream` is a hipStream_t cast to void* (NULL = default stream).
 * Every function returns 0 on success; on failure it returns non-zero and `mdd_last_error()`
 * describes it.  HIP out-of-memory is reported with the substring "out of memory" so the
 * reference's OOM handling (distill.py:525-545, 568-575 matches on that substring) keeps
 * working when the Python host re-raises it as RuntimeError.
 *
 * Reference interface each entry point replaces (file:line in /root/reference):
 *   mdd_engine_create / _param_*     student construction CLIPModel_full + ReparamModule x2
 *                                    distill.py:440-442, reparam_module.py:18-75 (flatten order)
 *   mdd_img_forward                  img_student_net(x, flat_param=theta)   distill.py:524
 *                                    -> ReparamModule.forward reparam_module.py:148-159
 *                                    -> ImageEncoder.forward networks.py:678-682 (timm nfnet_l0)
 *   mdd_txt_forward                  txt_student_net(y, flat_param=theta)   distill.py:537
 *                                    -> ProjectionHead.forward networks.py:639-646
 *   mdd_contrastive                  x/||x||, y/||y||, syn_lr_img * x @ y.T, symmetric CE
 *                                    distill.py:533,546-551
 *   mdd_img_backward / mdd_txt_backward
 *                                    torch.autograd.grad(loss, theta, create_graph=True)
 *                                    distill.py:562-567
 *   mdd_*_tangent_forward/backward   the double-backward that grand_loss.backward() performs
 *                                    through those gradients, distill.py:606
 *   mdd_flat_*                       theta' = theta - lr*g (:582-583), sum-MSE (:588-595),
 *                                    SGD momentum step (:233-241, :611-613)
 *   mdd_unrolled_match               the whole iteration distill.py:509-606 in one call
 */
#ifndef MDD_HIP_H
#define MDD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDD_DTYPE_F32 0  /* fp32 storage + exact-fp32 MFMA (parity mode)                    */
#define MDD_DTYPE_BF16 1 /* bf16 activations/weights, bf16 MFMA, fp32 accumulate, fp32 theta */
#define MDD_DTYPE_BF16X2 2 /* fp32 storage; every contraction operand split into hi+lo bf16 (16
                              significand bits), bf16 MFMA with fp32 accumulate: the fast mode that
                              meets the 1e-3 parity bar                        */
#define MDD_DTYPE_F32_BF16OPS 3 /* experiment: fp32 storage, operands rounded to ONE bf16 (hi only):
                                   separates stash rounding from MFMA operand rounding            */

typedef struct mdd_engine mdd_engine;

typedef struct mdd_config {
  const char* variant; /* "nfnet_l0" | "nfnet_l1" | "nfnet_tiny" */
  int32_t batch;       /* pairs per inner step (mini_batch_size, distill.py:511)          */
  int32_t num_queries; /* rows of image_syn / text_syn (distill.py:630)                    */
  int32_t image_size;  /* square input, multiple of 32                                     */
  int32_t d_txt;       /* text embedding dim (768 for BERT, networks.py:824)               */
  int32_t syn_steps;   /* number of activation stashes kept (unroll depth)                 */
  int32_t dtype;       /* MDD_DTYPE_*                                                      */
  int32_t keep_steps;  /* activation stash policy of mdd_unrolled_match: the first keep_steps inner
                          steps keep their forward/backward activations in HBM for the reverse sweep;
                          the others share ONE stash slot and are recomputed (forward + inner
                          gradient at the kept theta_k) when the sweep reaches them.  < 0 or
                          >= syn_steps: keep every step (no recompute).  Workspace shrinks from
                          syn_steps+1 to keep_steps+2 activation sets.              */
} mdd_config;

const char* mdd_last_error(void);
/* ABI version of this header: 2 = round 2 (mdd_config.keep_steps, MDD_DTYPE_BF16X2 / _F32_BF16OPS,
 * profile kind 4); 3 = + mdd_comm_* / mdd_allreduce_syn_grads; 4 = + the ViT building-block ops; 5 = + mdd_engine_set_pass_precision; 6 = + mdd_op_conv2d_wgrad2, mdd_set_pipe_kernels.  A binding built against another version
 * must refuse to load. */
#define MDD_ABI_VERSION 6
int mdd_version(void);
/* The wide bf16 pointwise contractions (>= 768 channels in and out, >= 8192 rows: the ViT linears) run on 256 x 256-tile
 * pipelined kernels; 0 routes them to the general kernels instead (process-wide; returns the previous setting).  For the
 * parity tests, which compare an iteration both ways, and for A/B timing. */
int mdd_set_pipe_kernels(int enable);

/* ---- engine lifetime and memory (the caller owns device memory: PyTorch caching allocator) */
int mdd_engine_create(const mdd_config* cfg, mdd_engine** out);
void mdd_engine_destroy(mdd_engine* e);
int64_t mdd_engine_workspace_bytes(const mdd_engine* e);
int mdd_engine_bind_workspace(mdd_engine* e, void* ws_dev, int64_t bytes, void* stream);

/* ---- parameter table in reference flatten order (reparam_module.py:28-39).  which: 0 = image
 * encoder, 1 = text projection.  shape4 is right-padded with 1s; ndim in *ndim. */
int64_t mdd_engine_param_numel(const mdd_engine* e, int which);
int mdd_engine_param_count(const mdd_engine* e, int which);
int mdd_engine_param_info(const mdd_engine* e, int which, int index, char* name, int name_cap,
                          int64_t* shape4, int* ndim, int64_t* offset);
int mdd_engine_feature_dim(const mdd_engine* e);
/* named stash buffer lookup for tests: byte offset into the bound workspace */
int mdd_engine_find_buffer(const mdd_engine* e, const char* name, int slot, int64_t* byte_offset,
                           int64_t* elems, int* is_float32);

/* ---- image encoder passes (slot = inner step index whose activations are kept) */
int mdd_img_forward(mdd_engine* e, int slot, const float* theta_dev, const float* image_syn_dev,
                    const int64_t* idx_dev, float* feat_out_dev, void* stream);
/* gtheta = J_theta^T feat_bar.  If dimage_accum_dev != NULL also
 * dimage_accum[idx] += coef*mul * J_image^T feat_bar.  stash != 0 keeps the backward signals of
 * this slot for the tangent pass (the inner-gradient call); stash == 0 leaves them untouched (a
 * second backward through the same forward, as the outer backward of distill.py:606 performs). */
int mdd_img_backward(mdd_engine* e, int slot, const float* theta_dev, const float* feat_bar_dev,
                     float* gtheta_out_dev, float* dimage_accum_dev, const int64_t* idx_dev,
                     const float* coef_dev, float mul, int stash, void* stream);
int mdd_img_tangent_forward(mdd_engine* e, int slot, const float* theta_dev,
                            const float* theta_dot_dev, float* feat_dot_out_dev, void* stream);
/* dimage_accum[idx] += coef*mul * d/d(image) ; coef_dev may be NULL (=1) */
int mdd_img_tangent_backward(mdd_engine* e, int slot, const float* theta_dev,
                             const float* theta_dot_dev, const float* feat_bar_dot_dev,
                             float* htheta_out_dev, float* dimage_accum_dev,
                             const int64_t* idx_dev, const float* coef_dev, float mul,
                             void* stream);

/* ---- text projection passes; drop_mask_dev: [batch, feat] already scaled by 1/(1-p), or NULL */
int mdd_txt_forward(mdd_engine* e, int slot, const float* theta_dev, const float* text_syn_dev,
                    const int64_t* idx_dev, const float* drop_mask_dev, float* feat_out_dev,
                    void* stream);
int mdd_txt_backward(mdd_engine* e, int slot, const float* theta_dev, const float* feat_bar_dev,
                     float* gtheta_out_dev, float* dtext_accum_dev, const int64_t* idx_dev,
                     const float* coef_dev, float mul, int stash, void* stream);
int mdd_txt_tangent_forward(mdd_engine* e, int slot, const float* theta_dev,
                            const float* theta_dot_dev, float* feat_dot_out_dev, void* stream);
int mdd_txt_tangent_backward(mdd_engine* e, int slot, const float* theta_dev,
                             const float* theta_dot_dev, const float* feat_bar_dot_dev,
                             float* htheta_out_dev, float* dtext_accum_dev, const int64_t* idx_dev,
                             const float* coef_dev, float mul, void* stream);

/* ---- contrastive head: loss + d/dx, d/dy, d/dscale; tangent variant when x_dot/y_dot given.
 * scale_dev NULL => scale_const is used (upstream distill_original.py:430 behaviour). */
int mdd_contrastive(mdd_engine* e, const float* x_dev, const float* y_dev, const float* scale_dev,
                    float scale_const, float* loss_dev, float* xbar_dev, float* ybar_dev,
                    float* sbar_dev, void* stream);
int mdd_contrastive_tangent(mdd_engine* e, const float* x_dev, const float* y_dev,
                            const float* x_dot_dev, const float* y_dot_dev, const float* scale_dev,
                            float scale_const, float* xbar_dot_dev, float* ybar_dot_dev,
                            float* sbar_dot_dev, void* stream);

/* The same head without an engine, for any row count: sample-sharded runs (SURVEY 8e mode B, what
 * `--distributed` / nn.DataParallel does at distill.py:443-445, 515-517) evaluate it on the all-gathered
 * [n, d] features of every rank.  x_dot/y_dot NULL: loss, xbar, ybar, sbar; non-NULL: their directional
 * derivatives into the same output pointers (loss unused).  work: mdd_op_contrastive_workspace_floats(n, d). */
int64_t mdd_op_contrastive_workspace_floats(int n, int d);
int mdd_op_contrastive(int n, int d, const float* x_dev, const float* y_dev, const float* x_dot_dev,
                       const float* y_dot_dev, const float* scale_dev, float scale_const, float* work_dev,
                       float* loss_dev, float* xbar_dev, float* ybar_dev, float* sbar_dev, void* stream);

/* ---- flat fp32 parameter streams */
int mdd_flat_axpy(float* out_dev, const float* x_dev, const float* g_dev, const float* lr_dev,
                  float sign, int64_t n, void* stream);       /* out = x + sign*lr*g */
int mdd_flat_sqdist(const float* a_dev, const float* b_dev, double* out_dev, int64_t n,
                    void* stream);                             /* out[0] += sum (a-b)^2 */
int mdd_flat_sgd_momentum(float* p_dev, const float* g_dev, float* buf_dev, float lr, float mom,
                          int first, int64_t n, void* stream);
/* the same step, a no-op when skip_flag_dev[0] != 0: the reference leaves its loop on a NaN loss BEFORE the
 * optimiser steps (distill.py:599-613); with the decision taken on the device the host can enqueue the next
 * iteration without first reading this one's loss back */
int mdd_flat_sgd_momentum_guarded(float* p_dev, const float* g_dev, float* buf_dev, float lr,
                                  float mom, int first, int64_t n, const float* skip_flag_dev,
                                  void* stream);

/* ---- one whole outer iteration (reference distill.py:509-606) */
typedef struct mdd_iter_args {
  const float* image_syn;    /* [num_queries,3,S,S] fp32 NCHW                                  */
  const float* text_syn;     /* [num_queries,d_txt]                                            */
  const float* lr_img;       /* device scalar syn_lr_img (also the logit scale, distill.py:548) */
  const float* lr_txt;       /* device scalar syn_lr_txt                                       */
  const float* theta0_img;   /* expert start  (distill.py:473-476)                             */
  const float* theta0_txt;
  const float* target_img;   /* expert target (distill.py:469-472)                             */
  const float* target_txt;
  const int64_t* perms;      /* [syn_steps, batch] minibatch indices (distill.py:510-511)      */
  const float* drop_masks;   /* [syn_steps, batch, feat] or NULL                               */
  int32_t syn_steps;
  int32_t use_lr_as_scale;   /* 1: fork behaviour (:548); 0: use logit_scale_const             */
  float logit_scale_const;
  float* grad_image_syn;     /* out, same shape as image_syn (overwritten)                     */
  float* grad_text_syn;      /* out                                                            */
  float* grad_lr;            /* out [2]: d/d lr_img, d/d lr_txt                                */
  float* losses;             /* out [3 + syn_steps]: grand, img, txt, contrastive per step     */
} mdd_iter_args;
int mdd_unrolled_match(mdd_engine* e, const mdd_iter_args* a, void* stream);

/* ---- per-kernel HIP-event timing of the contraction launches (bench.py roofline accounting).
 * kind: 0 = conv_gemm 128x32 tile, 1 = conv_gemm 256x64, 2 = conv_gemm 128x128, 3 = conv_wgrad (the
 * contraction kernel alone), 4 = the split-M reduce kernel that follows each conv_wgrad.
 * out4 = {launches, total milliseconds, algorithmic FLOPs, algorithmic bytes} since enable. */
int mdd_engine_profile(mdd_engine* e, int enable);
/* fp32-storage engines (MDD_DTYPE_F32 / _BF16X2 / _F32_BF16OPS): operand arithmetic of the image encoder's
 * contractions per pass -- F (reference distill.py:524), B (:562-567), T-F and T-B (what :606 differentiates):
 * 0 = the engine's own mode, 1 = split-bf16 (hi + lo), 2 = one bf16 per operand, 3 = exact fp32 MFMA, 4 = one fp16 per
 * operand (experiment: what an fp16-storage engine's matrix cores would see).  Used for the
 * per-pass error attribution and for mixed parity-grade modes. */
int mdd_engine_set_pass_precision(mdd_engine* e, int fwd, int bwd, int tan_fwd, int tan_bwd);
int mdd_engine_profile_read(mdd_engine* e, int kind, double* out4);
/* one CSV row per contraction launch since enable (geometry, ms, TFLOP/s, GB/s) */
int mdd_engine_profile_dump(mdd_engine* e, const char* path);

/* ---- single-op entry points (parity tests; NHWC activations, dtype = MDD_DTYPE_*) */
int mdd_op_conv2d(int dtype, int transposed, int nimg, int hin, int win, int cin, int cout, int k,
                  int stride, int pad, int groups, const void* a_dev, const void* w_packed_dev,
                  const float* bias_dev, void* out_dev, void* stream);
int mdd_op_conv2d_wgrad(int dtype, int nimg, int hin, int win, int cin, int cout, int k, int stride,
                        int pad, int groups, const void* dy_dev, const void* x_dev,
                        float* dw_packed_dev, float* dbias_dev, void* stream);
/* the engine's form of the weight gradient: an optional second pair (tangent pass: dW_t = dy1^T x1 + dy2^T x2) and
 * a split-M workspace of ws_floats floats (partial tiles are summed by a second kernel into dw, which is then
 * OVERWRITTEN; without a workspace the partials are added into dw with fp32 atomics).  The bias gradient is the
 * column sum of dy1 (added into dbias). */
int mdd_op_conv2d_wgrad2(int dtype, int nimg, int hin, int win, int cin, int cout, int k, int stride,
                         int pad, int groups, const void* dy1_dev, const void* x1_dev, const void* dy2_dev,
                         const void* x2_dev, float* dw_packed_dev, float* dbias_dev, float* ws_dev,
                         long long ws_floats, void* stream);

/* ---- synthetic-set evaluation: retrieval ranks.
 * Replaces the per-row host loops of reference epoch.py:140-156 (image->text: argsort of one similarity
 * row, position of the best ground-truth caption), :163-192 (text->image) and itm_eval epoch.py:219-237.
 * img_feat [n_img, dim], txt_feat [n_txt, dim]: un-normalised fp32 embeddings (normalised here as
 * epoch.py:126/:148 do); scale = exp(log(1/0.07)) in the reference (epoch.py:106-107).
 * img2txt: CSR lists of ground-truth caption ids per image (dataset.img2txt); txt2img[n_txt].
 * scores_ws: n_img*n_txt floats, left holding the scaled similarity matrix; norm_ws: n_img+n_txt floats.
 * rank_*: 0-based ranks = number of strictly better-scoring candidates.  All pointers device memory. */
int mdd_retrieval_ranks(const float* img_feat, const float* txt_feat, const int* img2txt_off,
                        const int* img2txt_idx, const int* txt2img, int n_img, int n_txt, int dim,
                        float scale, float* scores_ws, float* norm_ws, int* rank_i2t, int* rank_t2i,
                        void* stream);

/* ---- caption decode of the synthetic text embeddings: for every query row the index
 * of the most cosine-similar bank row.  Replaces reference distill.py:89-95 (`nearest_neighbor`: one
 * sklearn cosine_similarity + np.argmax per query against all ~145k train caption embeddings), called at
 * distill.py:244 and :374.  Ties resolve to the lowest index (np.argmax).
 * query [n_query, dim], bank [n_bank, dim] fp32; scores_ws n_query*n_bank floats (left holding the cosine
 * matrix); norm_ws n_query+n_bank floats; idx_out [n_query].  All device memory. */
int mdd_nearest_neighbor(const float* query, const float* bank, int n_query, int n_bank, int dim,
                         float* scores_ws, float* norm_ws, int* idx_out, void* stream);

/* ---- RCCL all-reduce of the synthetic-set gradient (SURVEY 8b `allreduce_syn_grads`; mode A's one exchange
 * per outer iteration -- north_star: "RCCL all-reduce of the matching-loss gradient over xGMI").  The reference
 * has no counterpart (its only multi-GPU code is nn.DataParallel, distill.py:443-445).  RCCL is bound at run
 * time from the librccl.so.1 the process already holds (torch's), else /opt/rocm's; every entry fails with
 * "mdd: RCCL unavailable" when there is none.  One communicator per rank = per GPU:
 *   rank 0: mdd_comm_unique_id(id)  ->  the host moves the MDD_COMM_ID_BYTES to every rank (any channel)
 *   all   : mdd_comm_create(id, rank, world, device_id, &c)          (collective: returns when all joined)
 *   all   : mdd_allreduce_syn_grads(c, buf, n, average, stream)      (in place, fp32 sum; average != 0 then
 *           divides by world in the same stream order: bit-identical to all_reduce(SUM) + div_(world)) */
#define MDD_COMM_ID_BYTES 128
typedef struct mdd_comm mdd_comm;
int mdd_comm_unique_id(void* id_out);
int mdd_comm_create(const void* id, int rank, int world, int device_id, mdd_comm** out);
void mdd_comm_destroy(mdd_comm* c);
int mdd_comm_world(const mdd_comm* c);
int mdd_allreduce_syn_grads(mdd_comm* c, float* buf_dev, int64_t n, int average, void* stream);

/* ---- ViT building blocks (BASELINE configs[4]: ViT-B/16 image encoder; csrc/vit.hip).
 * The operators of timm 0.6.7's VisionTransformer block (what the reference reaches through timm / clip,
 * networks.py:661,668) that the NFNet path does not have, each as forward, backward and -- when the `*_t` tangent
 * pointers are non-NULL -- the tangent of either (the double backward of distill.py:606).  A tangent call reads the
 * primal operands and writes ONLY the `*_t` outputs.  dtype: MDD_DTYPE_F32 or MDD_DTYPE_BF16 = the storage type of
 * the activation tensors (`void*`); parameters, scores and probabilities are fp32.  Row-major, 16-byte aligned.
 *   layernorm      y = (x - mean) * rstd * gamma + beta over rows of `dim`            (eps 1e-6 in ViT)
 *   layernorm_bwd  dx (+ `res` when non-NULL: the gradient arriving over the residual connection), and dgamma / dbeta
 *                  ACCUMULATED (+=, fp32 atomics) into the caller's zeroed/partial sums
 *   gelu           a = GELU(c) (exact, erf);  gelu_bwd  cbar = abar * GELU'(c)
 *   softmax        p = softmax(scale * s) over `cols` of each row (row stride ld); softmax_bwd  ds = scale*p*(dp - <p,dp>);
 *                  dtype = the storage type of the score tensors (the engine's bf16 mode keeps them in bf16: the matrix-core
 *                  contractions round them to bf16 anyway)
 *   bgemm          C[b] = alpha * A[b] B[b] for b = (o < outer, q < inner), every operand addressed by element strides
 *                  (transposes and the head slices of a fused qkv tensor need no copy).  bf16 dtype: a_is_f32 /
 *                  c_is_f32 say which of A, C are fp32 tensors (scores, probabilities); B is always activation-typed.
 *                  dtype MDD_DTYPE_BF16X2: fp32 tensors throughout, every product as hi*hi + hi*lo + lo*hi of split-bf16
 *                  operands on the matrix cores (the engine's bf16x2 mode); MDD_DTYPE_F32: exact fp32 FMA. */
typedef struct mdd_bgemm_desc {
  int32_t m, n, k, outer, inner;
  int64_t a_row, a_col, b_row, b_col, c_row, c_col;               /* A(i,kk) = A[i*a_row + kk*a_col], B(kk,j) = B[kk*b_row + j*b_col] */
  int64_t a_outer, a_inner, b_outer, b_inner, c_outer, c_inner;   /* batch strides in elements */
  float alpha;
} mdd_bgemm_desc;
/* Fused multi-head attention of the bf16 ViT path (csrc/attn.hip): head dimension 64, tokens <= 224; qkv is the fused
 * [n, tokens, 3, heads, 64] tensor, dout / out the [n, tokens, heads, 64] attention output (modes 0, 1) or, for the
 * gradient modes, out = d qkv in qkv's layout.  Nothing of size tokens x tokens leaves the CU: every mode recomputes
 * the score tiles it needs and exchanges per-row statistics [n, heads, tokens] (m, l: softmax max / sum, written by
 * mode 0; r = rowsum(P S_t), written by mode 1; D = rowsum(P dP) = rowsum(O dO), by mode 2 from o and dout; D_t, by
 * mode 3 from o, o_t, dout, dout_t).
 *   0  O = softmax(scale Q K^T) V          1  O_t (tangent along qkv_t)          2  dQ = dS K      (r_tan: dS K_t, accum)
 *   3  dQ_t part: dS_t K                   4  dV = P^T dO, dK = dS^T Q           (r_tan: dS^T Q_t; skip0: dK part only; accum)
 *   5  dV_t = P_t^T dO + P^T dO_t          6  dK_t part: dS_t^T Q
 * A full tangent of the backward is 3, 2(r_tan, accum), 5, 6, 4(r_tan, skip0, accum) on d qkv_t. */
int mdd_op_attention(int mode, int n, int tokens, int heads, float scale, const void* qkv, const void* qkv_t,
                     const void* dout, const void* dout_t, const void* o, const void* o_t, void* out, float* m, float* l,
                     float* r, float* D, float* D_t, int r_tan, int accum, int skip0, void* stream);
int mdd_op_layernorm(int dtype, int rows, int dim, float eps, const void* x, const void* x_t, const float* gamma,
                     const float* gamma_t, const float* beta, const float* beta_t, void* y, void* y_t, void* stream);
/* LayerNorm of a residual sum in one pass: s = a + b (stored in the storage type, normalised as stored), y = LN(s).
 * Tangent call (gamma_t given): s is the stashed primal sum (read), s_t = a_t + b_t (stored), y_t the tangent. */
int mdd_op_add_layernorm(int dtype, int rows, int dim, float eps, const void* a, const void* a_t, const void* b,
                         const void* b_t, void* s, void* s_t, const float* gamma, const float* gamma_t, const float* beta,
                         const float* beta_t, void* y, void* y_t, void* stream);
int mdd_op_layernorm_bwd(int dtype, int rows, int dim, float eps, const void* x, const void* x_t, const void* dy,
                         const void* dy_t, const float* gamma, const float* gamma_t, const void* res,
                         const void* res_t, void* dx, void* dx_t, float* dgamma, float* dgamma_t, float* dbeta,
                         float* dbeta_t, void* stream);
int mdd_op_gelu(int dtype, int64_t n, const void* c, const void* c_t, void* a, void* a_t, void* stream);
int mdd_op_gelu_bwd(int dtype, int64_t n, const void* c, const void* c_t, const void* abar, const void* abar_t,
                    void* cbar, void* cbar_t, void* stream);
int mdd_op_softmax(int dtype, int64_t rows, int cols, int ld, float scale, const void* s, const void* s_t, void* p,
                   void* p_t, void* stream);
int mdd_op_softmax_bwd(int dtype, int64_t rows, int cols, int ld, float scale, const void* p, const void* p_t,
                       const void* dp, const void* dp_t, void* ds, void* ds_t, void* stream);
int mdd_op_bgemm(int dtype, int a_is_f32, int c_is_f32, const mdd_bgemm_desc* d, const void* A, const void* A_t,
                 const void* B, const void* B_t, void* C, void* C_t, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDD_HIP_H */
