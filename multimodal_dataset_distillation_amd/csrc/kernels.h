// Launcher declarations shared by the engine (engine.cpp) and the kernel translation units.
#pragma once
#include "common.h"

int mdd_set_error(hipError_t e, const char* what);
int mdd_set_error_msg(int code, const char* msg);

// ---------------------------------------------------------------- flat_ops.hip (fp32 theta)
void launch_axpy_out(float* y, const float* x, const float* g, const float* lr, float sign,
                     int64_t n, hipStream_t st);
void launch_scale_out(float* y, const float* x, const float* coef, float mul, int64_t n,
                      hipStream_t st);
void launch_sqdist(const float* a, const float* b, double* out, int64_t n, hipStream_t st);
void launch_dot(const float* a, const float* b, double* out, double sign, int64_t n,
                hipStream_t st);
void launch_lambda_init(float* lam, const float* thK, const float* tgt, const double* dist0,
                        int64_t n, hipStream_t st);
void launch_sub_inplace(float* y, const float* x, int64_t n, hipStream_t st);
void launch_sgd_momentum(float* p, const float* g, float* buf, float lr, float mom, int first,
                         int64_t n, const float* skip, hipStream_t st);
void launch_match_finalize(const double* s, float* out, hipStream_t st);
void launch_d2f(float* out, const double* in, float mul, int accum, int n, hipStream_t st);
void launch_accum_f2d(double* out, const float* in, double sign, hipStream_t st);

// ---------------------------------------------------------------- ws.hip
// One entry per weight-standardised conv (timm ScaledStdConv2d).  Packed weights:
//   wf : [cout][k*k][cin_pad_g]            (B operand of the forward implicit GEMM)
//   wt : [groups][cin_pad_g][k*k][cout_g]  (B operand of the dgrad implicit GEMM)
struct WsDesc {
  int64_t off_w, off_b, off_g;  // float offsets into flat theta (weight OIHW, bias, gain)
  int64_t off_wf, off_wt;       // element offsets into the packed buffers
  int cout, cin_g, ksq, groups, cin_pad_g, cout_g;
  float scale, eps;
  int row_start;                // prefix sum of cout (global row id of this conv's channel 0)
  int tile_start;               // prefix sum of groups * ceil(cout_g / ws_tile_rows()) (ws_forward blocks)
};
int ws_tile_rows();
// theta_t == nullptr: primal (writes wf, wt).  theta_t != nullptr: tangent pass (writes wf_t,
// wt_t, and re-derives the primal wf, wt from theta when those pointers are non-null).
template <class AT>
void launch_ws_forward(const WsDesc* descs_dev, int nconv, int total_rows, int total_tiles, const float* theta,
                       const float* theta_t, AT* wf, AT* wt, AT* wf_t, AT* wt_t, hipStream_t st);
// dwf (+dwf_t): gradient w.r.t. the standardised weight, fp32, wf layout.
// out: gradient w.r.t. raw weight (OIHW) and gain, written into gtheta at off_w / off_g
// (tangent pass: writes the tangent into gtheta, i.e. the Hessian-vector product rows).
void launch_ws_backward(const WsDesc* descs_dev, int nconv, int total_rows, const float* theta,
                        const float* theta_t, const float* dwf, const float* dwf_t, float* gtheta,
                        hipStream_t st);

// ---------------------------------------------------------------- conv_gemm.hip
// Implicit-GEMM convolution on NHWC activations, forward and transposed (dgrad) in one kernel,
// optional second (A2,B2) source accumulated into the same MFMA accumulators (tangent pass).
struct ConvEpi {
  int mode;               // EPI_*
  const float* bias;      // [cout_tot] fp32 or null          (forward modes)
  const float* bias_t;    // tangent bias or null
  void* out_raw;          // AT*: raw result (c / a-bar) or null
  void* out_act;          // AT*: activated / chained result or null
  const void* c;          // AT*: stashed pre-activation (derivative point)
  const void* c_t;        // AT*: its tangent
  const void* abar;       // AT*: stashed raw a-bar (tangent backward)
  const void* add1;       // AT*: added to the accumulator before anything else (or null)
  const void* add2;       // AT*: added to out_act result (or null)
  float beta;
  // per-image bias of the pointwise data-gradient modes (EPI_BWD: ib, EPI_BWD_T: ib_t; ib non-null enables it):
  // row m of the accumulator gets ib[m / ib_hw][channel] * ib_mul added before anything else
  const float* ib;        // [nimg, co_tot] fp32 or null
  const float* ib_t;
  float ib_mul;
  int ib_hw;
  int act;                // activation of the fused chain rule: 0 SiLU, 1 exact GELU (conv_gemm_supports_gelu)
};
enum {
  EPI_FWD = 0,        // c = acc + bias ; out_raw = c ; out_act = beta*silu(c)
  EPI_FWD_T = 1,      // c = Dual(c, acc + bias_t + add1); out_raw = c.t ; out_act = (beta*silu(c)).t
  EPI_BWD = 2,        // a = acc + add1 ; out_raw = a ; out_act = beta*dsilu(c)*a + add2
  EPI_BWD_T = 3,      // a = Dual(abar, acc + add1); out_act = (beta*dsilu(Dual(c,c_t))*a).t + add2
  EPI_BWD_LIN = 4,    // a = acc + add1 ; out_raw = a     (no activation in between)
};
struct ConvGeom {
  int nimg;
  int ha, wa, ca_tot;     // spatial dims / pixel stride (elems) of the gathered operand A
  int ho, wo, co_tot;     // spatial dims / pixel stride of the output
  int kc;                 // reduction channels per group per tap (padded, multiple of chunk)
  int nc;                 // output channels per group
  int groups;
  int k, stride, pad;     // conv kernel size / stride / padding (of the FORWARD conv)
  int transposed;         // 0: forward gather, 1: dgrad gather
  int prec = 0;           // fp32 storage only: 0 exact-fp32 MFMA, 1 split-bf16 (hi+lo) MFMA, 2 hi only
};
template <class AT>
void launch_conv_gemm(const ConvGeom& g, const AT* A1, const AT* B1, const AT* A2, const AT* B2,
                      const ConvEpi& ep, hipStream_t st);
bool conv_gemm_supports_gelu(const ConvGeom& g);
// the 256 x 256 pipelined kernels (k_gemm_pipe, k_wgrad_pipe) may be switched off at run time: the parity tests compare
// an iteration with and without them (mdd_set_pipe_kernels)
bool pipe_kernels_enabled();
void set_pipe_kernels(bool on);   // ConvEpi::act == 1 may be requested for this geometry

// ---------------------------------------------------------------- conv_wgrad.hip
// dW[g][co][tap][kc] = sum_m dy[m][g*nc+co] * x[src(m,tap)][g*kc+kc], split over M.
//   slab != null (slab_floats >= the layer's dW size): two-phase combine -- partial tiles by plain
//     stores into slab[split][...], then a reduce kernel OVERWRITES dW (no zero-fill needed);
//   slab == null: fp32 atomics into dW, which the caller must have zeroed.
// optional second pair (dy2, x2) accumulated as well; optional bias grad db[co] += sum_m dy[m][co]
// (always atomics into a zeroed db: cout floats).  ev_mid (optional, profiling): recorded between the
// contraction kernel and the reduce kernel.
template <class AT>
void launch_conv_wgrad(const ConvGeom& g, const AT* dy1, const AT* x1, const AT* dy2, const AT* x2,
                       float* dW, float* dbias, float* slab, int64_t slab_floats, hipEvent_t ev_mid,
                       hipStream_t st);
// one launch for up to four wide pointwise bf16 layers that share the pixel count and the pairing (a ViT layer's four
// linears); false = not taken, launch them one by one
struct WgradItem { ConvGeom g; const void *dy1, *x1, *dy2, *x2; float* dW; float* dbias; };
bool launch_conv_wgrad_group(const WgradItem* items, int n, float* slab, int64_t slab_floats, hipStream_t st);
bool conv_wgrad_group_takes(const ConvGeom& g);   // such a layer (bf16 storage)

// ---------------------------------------------------------------- elementwise.hip
template <class AT>
void launch_img_gather_nhwc(AT* x0, const float* image, const int64_t* idx, int n, int c, int h,
                            int w, int cpad, hipStream_t st);
template <class AT>
void launch_img_scatter_grad(float* dimage, const AT* x0bar, const int64_t* idx, const float* coef,
                             float mul, int n, int c, int h, int w, int cpad, hipStream_t st);
// Data gradient of the FIRST stem conv (3x3, stride 2, pad 1, 3 real input channels) accumulated straight
// into the NCHW fp32 image gradient: dimage[idx[n]] += coef*mul * (dy1 (*) wt1 + dy2 (*) wt2).
// wt*: packed dgrad weights [cin_pad][9][cout].  Returns false when cout is not a supported width (the
// caller then runs the implicit-GEMM data gradient + launch_img_scatter_grad).
template <class AT>
bool launch_stem_dgrad_image(float* dimage, const AT* dy1, const AT* wt1, const AT* dy2, const AT* wt2,
                             const int64_t* idx, const float* coef, float mul, int n, int s, int cout,
                             int cpad, hipStream_t st);
template <class AT>
void launch_avgpool2(AT* out, const AT* in, int n, int h, int w, int c, int stride, hipStream_t st);
template <class AT>
void launch_avgpool2_bwd(AT* din, const AT* dout, int n, int h, int w, int c, int stride,
                         hipStream_t st);
// p[n,c] = mean_hw x[n,hw,c]
template <class AT>
void launch_pool_mean(float* p, const AT* x, int n, int hw, int c, hipStream_t st);
// X' = C3*gate*ga + SC ; A' = beta*silu(X')   (tangent: *_t pointers non-null)
template <class AT>
void launch_se_apply(const AT* c3, const AT* c3_t, const float* gate, const float* gate_t,
                     const AT* sc, const AT* sc_t, AT* xo, AT* xo_t, AT* ao, AT* ao_t, float ga,
                     float beta, int n, int hw, int c, hipStream_t st);
// zbar[n,c] = dgate*gate*(1-gate) with dgate[n,c] = ga * sum_hw xbar*c3  (sigmoid backward fused)
template <class AT>
void launch_se_gate_grad(float* zbar, float* zbar_t, const AT* xbar, const AT* xbar_t,
                         const AT* c3, const AT* c3_t, const float* gate, const float* gate_t,
                         float ga, int n, int hw, int c, hipStream_t st);
// the same reduction, and c3bar = xbar*gate*ga written in the same pass (the pooled path's gradient reaches the
// residual branch through the mid-channel pooled vector: Eng::img_backward)
template <class AT>
void launch_se_gate_grad_c3b(float* zbar, float* zbar_t, AT* c3bar, AT* c3bar_t, const AT* xbar, const AT* xbar_t,
                             const AT* c3, const AT* c3_t, const float* gate, const float* gate_t, float ga, int n,
                             int hw, int c, hipStream_t st);
// c3bar = xbar*gate*ga + pbar/hw
template <class AT>
void launch_se_apply_bwd(AT* c3bar, AT* c3bar_t, const AT* xbar, const AT* xbar_t,
                         const float* gate, const float* gate_t, const float* pbar,
                         const float* pbar_t, float ga, int n, int hw, int c, hipStream_t st);
// y[n,f] = mean_hw silu(cf)
template <class AT>
void launch_final_pool(float* y, float* y_t, const AT* cf, const AT* cf_t, int n, int hw, int c,
                       hipStream_t st);
// cfbar = ybar/hw * dsilu(cf)
template <class AT>
void launch_final_pool_bwd(AT* cfbar, AT* cfbar_t, const float* ybar, const float* ybar_t,
                           const AT* cf, const AT* cf_t, int n, int hw, int c, hipStream_t st);

// ---------------------------------------------------------------- linear.hip (fp32, small M)
// Split-K scratch of the small-M SGEMM: per-tile accumulators, bias-gradient row sums and arrival
// counters (zeroed once at bind; every launch leaves them zero).  One per stream that runs linears.
constexpr int64_t LIN_PART_FLOATS = 256 * 4096, LIN_PART_RS_FLOATS = 256 * 64;
constexpr int LIN_CTR_N = 256;
struct LinScratch {
  float* part = nullptr; float* part_rs = nullptr; unsigned* ctr = nullptr;
  int64_t part_floats = 0, part_rs_floats = 0; int ctr_n = 0;
  bool mfma = false;     // small-batch layer: route to the one-workgroup-per-tile matrix-core kernel (no scratch used)
};
int64_t lin_scratch_bytes();
LinScratch lin_scratch_carve(void* base);
// act: 0 none, 1 relu, 2 sigmoid.  Tangent inputs may be null individually (treated as 0).
// Tangent forward reads the stashed post-activation primal from y and writes y_t only.
void launch_linear_fwd(float* y, float* y_t, const float* x, const float* x_t, const float* W,
                       const float* W_t, const float* b, const float* b_t, int n, int k, int j,
                       int act, const LinScratch& ws, hipStream_t st);
// dx[n,k] = sum_j dy[n,j] W[j,k], times [relu_of[n,k] > 0] when relu_of is given
void launch_linear_dgrad(float* dx, float* dx_t, const float* dy, const float* dy_t,
                         const float* W, const float* W_t, const float* relu_of, int n, int k,
                         int j, const LinScratch& ws, hipStream_t st);
// dW[j,k] = sum_n dy[n,j] x[n,k] ; db[j] = sum_n dy[n,j]   (tangent pass writes tangents only)
void launch_linear_wgrad(float* dW, float* db, const float* dy, const float* dy_t, const float* x,
                         const float* x_t, int n, int k, int j, const LinScratch& ws,
                         hipStream_t st);

// y = alpha * (x W^T) + b with W [j,k] in the activation storage type (packed standardised conv weights)
template <class TW>
void launch_linear_fwd_w(float* y, float* y_t, const float* x, const float* x_t, const TW* W, const TW* W_t,
                         const float* b, const float* b_t, int n, int k, int j, float alpha, hipStream_t st);
// dW[j,k] += sum_n dy[n,j] x[n,k] ; db[j] += sum_n dy[n,j]  (accumulating form; db may be null)
void launch_linear_wgrad_accum(float* dW, float* db, const float* dy, const float* dy_t, const float* x,
                               const float* x_t, int n, int k, int j, hipStream_t st);

// out[j] = sum_c W[j,c] v[c] + b[j] (primal: out_t == null); tangent call writes out_t only
void launch_matvec_bias(float* out, float* out_t, const float* W, const float* W_t, const float* v, const float* v_t,
                        const float* b, const float* b_t, int rows, int cols, hipStream_t st);

// ---------------------------------------------------------------- retrieval.hip (fp32)
// scores[b,n] = scale * <img_hat[i], txt_hat[j]> ; rank_i2t[i], rank_t2i[j] (see retrieval.hip).
// img2txt as CSR (off[b+1], idx[]); rn_ws: b + n floats of scratch.
void launch_retrieval_ranks(int* rank_i2t, int* rank_t2i, float* scores, float* rn_ws,
                            const float* img_feat, const float* txt_feat, const int* img2txt_off,
                            const int* img2txt_idx, const int* txt2img, int b, int n, int d,
                            float scale, hipStream_t st);

// idx_out[i] = argmax_j cos(query[i], bank[j]) (first maximum); scores [q,n] and rn_ws [q+n] scratch
void launch_nearest_neighbor(int* idx_out, float* scores, float* rn_ws, const float* query,
                             const float* bank, int q, int n, int d, hipStream_t st);

// ---------------------------------------------------------------- head.hip (fp32)
void launch_gather_rows(float* out, const float* in, const int64_t* idx, int n, int d,
                        hipStream_t st);
void launch_scatter_rows_axpy(float* out, const float* in, const int64_t* idx, const float* coef,
                              float mul, int n, int d, hipStream_t st);
void launch_gelu(float* g, float* g_t, const float* p, const float* p_t, int64_t n, hipStream_t st);
// r = f*mask + p ; y = LN(r)*gamma + beta
void launch_ln_fwd(float* y, float* y_t, float* r, float* r_t, const float* f, const float* f_t,
                   const float* mask, const float* p, const float* p_t, const float* gamma,
                   const float* gamma_t, const float* beta, const float* beta_t, int n, int d,
                   float eps, hipStream_t st);
// rbar (LN backward wrt r), fbar = rbar*mask; dgamma/dbeta
void launch_ln_bwd(float* rbar, float* rbar_t, float* fbar, float* fbar_t, float* dgamma,
                   float* dbeta, float* stats /*[n*4] scratch*/, const float* ybar, const float* ybar_t, const float* r,
                   const float* r_t, const float* mask, const float* gamma, const float* gamma_t,
                   int n, int d, float eps, hipStream_t st);
// pbar = rbar + gbar*dgelu(p)
void launch_gelu_bwd(float* pbar, float* pbar_t, const float* rbar, const float* rbar_t,
                     const float* gbar, const float* gbar_t, const float* p, const float* p_t,
                     int64_t n, hipStream_t st);

// contrastive head (reference distill.py:533,546-551) forward+backward in one go.
// primal: writes loss[0], xbar, ybar, sbar[0].  tangent (x_t,y_t given): writes xbar_t, ybar_t,
// sbar_t[0] -- the directional derivative of (xbar, ybar, sbar).
struct LossWork {  // device scratch, sized by loss_work_floats(n, d)
  float* xh; float* yh; float* xh_t; float* yh_t; float* rnx; float* rny; float* rnx_t;
  float* rny_t; float* G; float* G_t; float* Gb; float* Gb_t;
};
int64_t loss_work_floats(int n, int d);
LossWork loss_work_carve(float* base, int n, int d);
void launch_contrastive(const LossWork& w, float* loss, float* xbar, float* ybar, float* sbar,
                        float* xbar_t, float* ybar_t, float* sbar_t, const float* x,
                        const float* y, const float* x_t, const float* y_t, const float* scale_ptr,
                        float scale_const, int n, int d, hipStream_t st);

// ---------------------------------------------------------------- vit.hip (ViT topology, BASELINE configs[4])
template <class AT> void launch_patchify(AT* cols, const float* image, const int64_t* idx, int n, int s, int patch, hipStream_t st);
template <class AT> void launch_unpatchify_accum(float* dimage, const AT* colsbar, const int64_t* idx, const float* coef,
                                                 float mul, int n, int s, int patch, hipStream_t st);
template <class AT> void launch_vit_embed(AT* x0, AT* x0_t, const AT* pe, const AT* pe_t, const float* cls, const float* cls_t,
                                          const float* pos, const float* pos_t, int n, int tok, int dim, hipStream_t st);
template <class AT> void launch_vit_embed_bwd(const AT* xb, const AT* xb_t, AT* peb, AT* peb_t, float* dcls, float* dpos,
                                              int n, int tok, int dim, hipStream_t st);
template <class AT> void launch_add2(AT* o, AT* o_t, const AT* a, const AT* a_t, const AT* b, const AT* b_t, int64_t elems,
                                     hipStream_t st);
template <class AT> void launch_cls_gather(AT* out, const AT* x, int n, int tok, int dim, hipStream_t st);
template <class AT> void launch_cls_scatter(AT* xb, const AT* in, int n, int tok, int dim, hipStream_t st);
template <class AT> void launch_act_to_f32(float* o, const AT* a, int64_t n, hipStream_t st);
template <class AT> void launch_f32_to_act(AT* o, const float* a, int64_t n, hipStream_t st);
template <class AT> void launch_lin_pack(AT* wf, AT* wt, const float* w, int out, int in, hipStream_t st);
struct LinPackDesc { int64_t off_w, off_p; int out, in, tile_start, pad_; };
template <class AT> void launch_lin_pack_all(const LinPackDesc* descs, int nd, int total_tiles, const float* th, const float* th_t,
                                             AT* wf, AT* wt, AT* wf_t, AT* wt_t, hipStream_t st);

// ---------------------------------------------------------------- attn.hip (bf16; fused attention of the ViT path)
// mode: 0 O = P V (writes m, l; optionally P) | 1 O_t (reads m, l; writes r) | 2 dQ = dS K (writes D; r_tan: K_t,
// accum) | 3 dS_t K (reads r, D; writes D_t) | 4 dV = P^T dO, dK = dS^T Q (r_tan: Q_t; skip0; accum) | 5 dV_t | 6 dS_t^T Q
int launch_attention(int mode, const void* qkv, const void* qkv_t, const void* dout, const void* dout_t, void* out,
                     const void* o, const void* o_t, float* m, float* l, float* r, float* D, float* D_t, int n, int tokens,
                     int heads, float scale, int r_tan, int accum, int skip0, hipStream_t st);
bool attention_fused_supported(int tokens, int head_dim);
