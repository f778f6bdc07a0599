// Host-side engine of libmdd_hip.so: network spec + parameter table in the reference's flatten
// order, HBM workspace plan, and the walkers that launch the gfx950 kernels for
//   F  : student forward                                   (reference distill.py:524, 537)
//   B  : inner gradient  d contrastive / d theta           (reference distill.py:562-567)
//   T  : tangent (forward-over-reverse) of F and B         (what distill.py:606 differentiates)
// plus mdd_unrolled_match: the whole outer iteration (distill.py:509-606) as an explicit tape:
//   theta_{k+1} = theta_k - lr*g_k ;  lambda_K = 2(theta_K - theta*)/||theta_0 - theta*||^2 ;
//   for k = K-1..0:  v = lr*lambda ; (H v, d/dX, d/ds) = R_v{grad L}(theta_k) ;
//                    lambda -= H v ; dX -= d/dX ; dlr -= <g_k, lambda> ...
// Activations of every step are kept in HBM (288 GB): forward stash {c, silu(c)} and backward
// stash {a-bar, c-bar} per conv site, so the tangent pass never recomputes a primal contraction.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <algorithm>
#include <string>
#include <functional>
#include <vector>

#include "kernels.h"
#include "mdd_hip.h"

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
int mdd_set_error(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s%s", (int)e, hipGetErrorString(e), what,
           e == hipErrorOutOfMemory ? " : out of memory" : "");
  g_err = buf;
  return 1000 + (int)e;
}
int mdd_set_error_msg(int code, const char* msg) {
  g_err = msg;
  return code;
}
#ifndef MDD_SIDE_PRIORITY
#define MDD_SIDE_PRIORITY 0   // bit 0: weight-gradient stream at least priority; bit 1: text stream (experiment builds)
#endif
#ifndef MDD_SE_SIDE
#define MDD_SE_SIDE 2            // forward passes: squeeze-excite gate chain beside conv3 -- 0: no (main stream),
#endif                           // 1: on the side stream, 2: on a stream of its own with high queue priority
#ifndef MDD_SE_W13
#define MDD_SE_W13 0             // experiment (r03_experiments.md: +1 % time, rejected): squeeze-excite chain through
                                 // W13 = W1 W3_hat (se_prep) -- one product fewer on the dependent chain, 24-48 more
                                 // small products per pass off it; 0: p, h, gate / pb, qb in sequence
#endif
#ifndef MDD_VIT_FUSED_ATTENTION
#define MDD_VIT_FUSED_ATTENTION 1   // bf16 ViT: attention through csrc/attn.hip (0: bgemm + softmax kernels, scores in HBM)
#endif
#ifndef MDD_VIT_FUSE_GELU
#define MDD_VIT_FUSE_GELU 1      // ViT MLP: exact GELU (and its chain rule) in the epilogues of fc1 / fc2's data gradient
#endif
#ifndef MDD_TF_SPLIT
#define MDD_TF_SPLIT 0           // experiment (profiles/r03_experiments.md: +2.7 % time, rejected): tangent-forward pass,
#endif                           // the second source of every contraction, conv(a, w_t), depends on the primal stash only --
                                 // it runs ahead on the side stream (idle in forward passes) into the tangent buffer and
                                 // the main chain's conv(a_t, w) adds it in its epilogue
#ifndef MDD_GRAPH
#define MDD_GRAPH 0      // 1: mdd_unrolled_match replays a captured hipGraph (experiment build)
#endif
#define CHECK_ARG(cond, msg) \
  do { if (!(cond)) return mdd_set_error_msg(2, "mdd: invalid argument: " msg); } while (0)
#define POST_LAUNCH(what) HIP_CHECK_RET(hipGetLastError())
// inside Eng<AT> walkers: launch errors + latched stream-ordering / event errors
#define POST_WALK(what)                                  \
  do {                                                   \
    HIP_CHECK_RET(hipGetLastError());                    \
    if (int rc_ = async_status()) return rc_;            \
  } while (0)

// ------------------------------------------------------------------------------------------ spec
namespace {

struct NfCfg {
  int depths[4];
  int channels[4];
  int stem_chs, group_size;
  float bottle_ratio, feat_mult, se_rd_ratio;
  float alpha, attn_gain, eps, gamma;
};
int make_divisible(double v, int divisor, double round_limit = 0.9) {
  int nv = std::max(divisor, (int)(v + divisor / 2.0) / divisor * divisor);
  if (nv < round_limit * v) nv += divisor;
  return nv;
}
bool get_cfg(const char* variant, NfCfg& c) {
  // timm 0.6.7 nfnet.py: _nfnet_cfg(depths, feat_mult, group_size=64, bottle_ratio=0.25,
  //   attn_kwargs=dict(rd_ratio=0.25, rd_divisor=8), act_layer='silu'); alpha 0.2, attn_gain 2.0,
  //   std_conv_eps 1e-5, gamma = _nonlin_gamma['silu']
  NfCfg base = {{1, 2, 6, 3}, {256, 512, 1536, 1536}, 128, 64, 0.25f, 1.5f, 0.25f,
                0.2f, 2.0f, 1e-5f, 1.7881293296813965f};
  if (!strcmp(variant, "nfnet_l0")) { c = base; return true; }
  if (!strcmp(variant, "nfnet_l1")) {
    c = base; int d[4] = {2, 4, 12, 6}; memcpy(c.depths, d, sizeof d); c.feat_mult = 2.0f; return true;
  }
  if (!strcmp(variant, "nfnet_tiny")) {  // build-defined miniature (tests)
    c = base; int d[4] = {1, 2, 2, 1}; int ch[4] = {64, 128, 192, 192};
    memcpy(c.depths, d, sizeof d); memcpy(c.channels, ch, sizeof ch);
    c.stem_chs = 64; c.group_size = 16; return true;
  }
  return false;
}

// ViT topology (BASELINE configs[4]; timm 0.6.7 VisionTransformer as restated in oracle/vit_ref.py)
struct VitCfg { int patch, dim, depth, heads; float eps; int head = 0; };   // head > 0: classifier Linear(dim, head)
bool get_vit_cfg(const char* variant, VitCfg& c) {
  if (!strcmp(variant, "vit_b16")) { c = {16, 768, 12, 12, 1e-6f}; return true; }      // vit_base_patch16_224
  if (!strcmp(variant, "vit_tiny16")) { c = {16, 192, 12, 3, 1e-6f}; return true; }    // vit_tiny_patch16_224 (networks.py:668)
  if (!strcmp(variant, "vit_micro")) { c = {8, 64, 2, 2, 1e-6f}; return true; }        // build-defined miniature (tests)
  // the reference's 'vit' as it stands (networks.py:668: no num_classes=0): timm's 1000-way head stays on the model
  if (!strcmp(variant, "vit_tiny16_cls")) { c = {16, 192, 12, 3, 1e-6f, 1000}; return true; }
  if (!strcmp(variant, "vit_micro_cls")) { c = {8, 64, 2, 2, 1e-6f, 24}; return true; }
  return false;
}

struct ParamInfo { std::string name; int64_t shape[4]; int ndim; int64_t offset, numel; };

struct ConvL {
  int cin, cout, k, stride, pad, groups;
  int hin, hout;          // square spatial dims
  int cin_pad;            // total input channels as stored (conv1: 8)
  int64_t off_w, off_b, off_g;
  int64_t off_p;          // offset into packed buffers (same for wf / wt / dwf)
  int tokens = 0;         // > 0: a linear over batch*tokens rows (ViT), no spatial extent
  int64_t packed() const { return (int64_t)cout * k * k * (cin_pad / groups); }
};
struct VitBlkL { int qkv, proj, fc1, fc2; int64_t ln1_w, ln1_b, ln2_w, ln2_b; };
struct SeL { int c, rd; int64_t off_w1, off_b1, off_w2, off_b2; };
struct Blk {
  int ds, c1, c2, c2b, c3;  // conv indices (ds = -1: identity shortcut)
  SeL se;
  int stride, cin, cout, mid, hin, hout;
  float beta;               // pre-activation scale of THIS block's input
};

struct Arena {
  int64_t cur = 0;
  int64_t take(int64_t bytes) {
    int64_t o = cur;
    cur += (bytes + 255) / 256 * 256;
    return o;
  }
};
struct NamedBuf { std::string name; int slot; int64_t off, elems; bool f32; };

}  // namespace

// ------------------------------------------------------------------------------------------ engine
struct mdd_engine {
  mdd_config cfg;
  std::string variant;
  virtual ~mdd_engine() {}
  virtual int64_t workspace_bytes() const = 0;
  virtual int bind(void* ws, int64_t bytes, hipStream_t st) = 0;
  virtual int img_forward(bool T, int slot, const float* th, const float* th_t, const float* image,
                          const int64_t* idx, float* feat_out, hipStream_t st) = 0;
  virtual int img_backward(bool T, int slot, const float* th, const float* th_t, const float* ybar,
                           const float* ybar_t, float* gout, float* dimage, const int64_t* idx,
                           const float* coef, float mul, bool repack, bool stash, hipStream_t st) = 0;
  virtual int txt_forward(bool T, int slot, const float* th, const float* th_t, const float* text,
                          const int64_t* idx, const float* mask, float* feat_out,
                          hipStream_t st) = 0;
  virtual int txt_backward(bool T, int slot, const float* th, const float* th_t, const float* ybar,
                           const float* ybar_t, float* gout, float* dtext, const int64_t* idx,
                           const float* coef, float mul, bool stash, hipStream_t st) = 0;
  virtual int contrastive(bool T, const float* x, const float* y, const float* x_t,
                          const float* y_t, const float* scale_dev, float scale_const, float* loss,
                          float* xbar, float* ybar, float* sbar, hipStream_t st) = 0;
  virtual int unrolled_match(const mdd_iter_args* a, hipStream_t st) = 0;
  virtual int set_pass_prec(const int* q4) = 0;
  virtual void profile_enable(bool on) = 0;
  virtual int profile_read(int kind, double* out4) = 0;
  virtual int profile_dump(const char* path) = 0;
  std::vector<ParamInfo> pimg, ptxt;
  int64_t P_img = 0, P_txt = 0;
  int feat = 0;
  std::vector<NamedBuf> names;
};

namespace {

template <class AT>
struct Eng : mdd_engine {
  NfCfg nf;
  bool is_vit = false;     // the image encoder is a Vision Transformer (vit_forward / vit_backward below)
  bool fused_attn = false; // bf16 ViT with head dimension 64 and <= 224 tokens: attention by attn.hip, no score-sized tensors
  VitCfg vit;
  int Tk = 0, sld = 0;     // tokens per image (1 + patches), row stride of the score / probability matrices
  std::vector<VitBlkL> vblk;
  int pe_conv = -1;
  int64_t off_cls = 0, off_pos = 0, off_nw = 0, off_nb = 0, off_hw = 0, off_hb = 0;
  int prec = 0;   // ConvGeom::prec of every contraction (fp32 storage: 0 exact, 1 split-bf16, 2 hi only)
  // fp32-storage engines: operand arithmetic of the image encoder's contractions per pass {F, B, T-F, T-B}
  // (0 = the engine's mode; mdd_engine_set_pass_precision -- the attribution table of DESIGN.md section 5)
  int pass_prec[4] = {0, 0, 0, 0};
  int cur_prec = 0;
  void select_prec(bool T, bool bwd) {
    const int q = pass_prec[(T ? 2 : 0) + (bwd ? 1 : 0)];
    cur_prec = (sizeof(AT) == 4 && q != 0) ? (q == 3 ? 0 : (q == 4 ? 3 : q)) : prec;
  }
  int set_pass_prec(const int* q) override {
    for (int i = 0; i < 4; ++i) {
      CHECK_ARG(q[i] >= 0 && q[i] <= 4, "pass precision must be 0 (engine mode), 1 (split bf16), 2 (one bf16), 3 (exact fp32), 4 (one fp16)");
      pass_prec[i] = q[i];
    }
    CHECK_ARG(sizeof(AT) == 4 || (q[0] | q[1] | q[2] | q[3]) == 0, "per-pass precision needs an fp32-storage engine");
    return 0;
  }
  int N, S, Dt, K;
  // Activation stash policy: steps k < keep own stash slot k; steps k >= keep share slot
  // `keep` and are recomputed by the reverse sweep.  nslots = activation sets besides the tangent set.
  int keep = 0, nslots = 0;
  int slot_of(int k) const { return k < keep ? k : keep; }
  std::vector<ConvL> convs;
  std::vector<Blk> blks;
  int stem[4], fin;
  std::vector<int> xh, xc;  // stream spatial dim / channels: X[b], b = 0..nb
  int64_t packed_total = 0;
  int total_rows = 0, total_tiles = 0;
  std::vector<WsDesc> descs;
  // text head offsets
  int64_t t_pw, t_pb, t_fw, t_fb, t_lw, t_lb;

  // ---------------- workspace plan (byte offsets), resolved to pointers at bind()
  struct BlockActs {
    AT *P, *SC, *C1, *A1, *C2, *A2, *C2b, *A2b, *C3;
    float *q, *p, *h, *gate;   // SE: mid-channel pooled vector, its image under conv3 (= pooled conv3 output), MLP
    AT *C3B, *A2bB, *C2bB, *A2B, *C2B, *A1B, *C1B, *AinB;
    float *zB, *hB, *pB, *qB;
  };
  struct VitActs {
    AT *N1, *QKV, *O, *X2, *N2, *C, *A;       // LayerNorm 1, fused qkv, attention output, stream after attention, LayerNorm 2, fc1, GELU
    AT* P;                                    // attention probabilities [batch*heads*tokens, sld], in the storage type
    AT *N1B, *QKVB, *OB, *X2B, *N2B, *CB, *AB;
    AT *PB, *SB;                              // gradients of the probabilities / of the scores
    float *sm, *sl, *sr, *sD, *sDt;           // fused attention (attn.hip): per-row statistics [batch, heads, tokens]
  };
  struct ActSet {
    AT *VCOL = nullptr, *VPE = nullptr, *VCOLB = nullptr, *VPEB = nullptr;   // patch columns, patch embedding (+ grads)
    AT *CLS = nullptr, *CLSN = nullptr, *CLSNB = nullptr, *CLSB = nullptr, *TMP = nullptr;
    float *hin = nullptr, *hinB = nullptr;     // fp32 copies of the normalised class token / its gradient (classifier head)
    std::vector<VitActs> vb;
    AT* X0;
    AT *Cs[3], *As[3];
    std::vector<AT*> X, A, XB;  // stream, pre-activated stream, stream grads
    std::vector<BlockActs> blk;
    AT *CF, *CFB;
    AT *AsB[3], *CsB[3];
    AT* X0B;
    float *y, *yB;              // image features / their grads
    float* dwf;                 // grad wrt standardised weights (fp32, packed layout)
    // text head
    float *tx, *tp, *tg, *tf, *tr, *ty;
    float *tyB, *trB, *tfB, *tgB, *tpB, *txB;
  };
  std::vector<ActSet> sets;  // per slot
  ActSet tn;                 // tangent set
  int64_t ws_bytes = 0;
  char* base = nullptr;
  Arena arena;
  // persistent / scratch
  int64_t o_descs; WsDesc* d_descs = nullptr;
  AT *wf = nullptr, *wt = nullptr, *wf_t = nullptr, *wt_t = nullptr;
  AT *dsS = nullptr, *dsF = nullptr;       // downsample dgrad scratch (pooled / full res)
  float* ln_stats = nullptr;
  float* wslab[2] = {nullptr, nullptr};             // split-M partial-sum slabs of conv_wgrad: main / side stream
  int64_t wslab_floats = 0;
  char* lin_mem[3] = {nullptr, nullptr, nullptr};   // split-K scratch: main / side / text stream
  LinScratch lin_main, lin_side, lin_txt;
  float* lossw = nullptr;
  // unrolled-match state
  std::vector<float*> thI, thT, gI, gT;    // theta_k (k=1..K), g_k
  float *lamI, *lamT, *nuI, *nuT, *hI, *hT;
  float *fx_t, *fy_t, *xbar, *ybar, *xbar_t, *ybar_t, *sbar, *sbar_t;
  double* dsc;                              // [0..3] sqdist, [4],[5] dlr
  std::vector<std::pair<void**, int64_t>> fix;  // pointer fix-ups (address of pointer, offset)

  template <class T> void plan(T** slot_ptr, int64_t elems, const char* name, int slot) {
    int64_t off = arena.take(elems * (int64_t)sizeof(T));
    fix.push_back({(void**)slot_ptr, off});
    if (name) names.push_back({name, slot, off, elems, sizeof(T) == 4 && std::is_same<T, float>::value});
  }

  // ------------------------------------------------------------------ construction
  int build(const mdd_config& c) {
    cfg = c; variant = c.variant; cfg.variant = variant.c_str();
    N = c.batch; S = c.image_size; Dt = c.d_txt; K = c.syn_steps;
    is_vit = get_vit_cfg(c.variant, vit);
    CHECK_ARG(is_vit || get_cfg(c.variant, nf), "unknown variant");
    CHECK_ARG(S % 32 == 0 && S >= 32, "image_size must be a multiple of 32");
    CHECK_ARG(N >= 2 && N <= 1024, "batch must be in [2,1024]");
    CHECK_ARG(K >= 1 && K <= 64, "syn_steps must be in [1,64]");
    CHECK_ARG(Dt >= 1, "d_txt");
    keep = (c.keep_steps < 0 || c.keep_steps >= K) ? K : c.keep_steps;
    nslots = keep >= K ? K : keep + 1;
    int64_t off = 0;
    auto add_param = [&](std::vector<ParamInfo>& tab, const std::string& name,
                         std::initializer_list<int64_t> shp) {
      ParamInfo p; p.name = name; p.ndim = (int)shp.size(); p.numel = 1;
      int i = 0; for (auto s : shp) { p.shape[i++] = s; p.numel *= s; }
      for (; i < 4; ++i) p.shape[i] = 1;
      p.offset = off; off += p.numel; tab.push_back(p);
      return p.offset;
    };
    auto add_conv = [&](const std::string& name, int cin, int cout, int k, int stride, int groups,
                        int hin) {
      ConvL L; L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.groups = groups;
      L.pad = ((stride - 1) + (k - 1)) / 2;
      L.hin = hin; L.hout = (hin + 2 * L.pad - k) / stride + 1;
      L.cin_pad = cin < 8 ? 8 : cin;
      L.off_w = add_param(pimg, name + ".weight", {cout, cin / groups, k, k});
      L.off_b = add_param(pimg, name + ".bias", {cout});
      L.off_g = add_param(pimg, name + ".gain", {cout, 1, 1, 1});
      L.off_p = packed_total; packed_total += L.packed();
      convs.push_back(L);
      return (int)convs.size() - 1;
    };
    if (is_vit) {
      // timm registration order: cls_token, pos_embed, patch_embed.proj, blocks.i.{norm1, attn.qkv, attn.proj,
      // norm2, mlp.fc1, mlp.fc2}, norm (oracle/vit_ref.py; reparam_module.py:28-39 flattens in this order)
      CHECK_ARG(S % vit.patch == 0, "image_size must be a multiple of the patch size");
      const int D = vit.dim, gp = S / vit.patch;
      Tk = 1 + gp * gp; sld = (Tk + 7) & ~7;
      CHECK_ARG(D % vit.heads == 0 && Tk <= 512, "heads must divide dim; at most 512 tokens");
      fused_attn = MDD_VIT_FUSED_ATTENTION && sizeof(AT) == 2 && attention_fused_supported(Tk, D / vit.heads);
      auto add_lin = [&](const std::string& name, int cin, int cout, int tokens, bool conv_shape) {
        ConvL L; L.cin = cin; L.cout = cout; L.k = 1; L.stride = 1; L.groups = 1; L.pad = 0;
        L.hin = 1; L.hout = 1; L.cin_pad = cin; L.tokens = tokens;
        if (conv_shape) L.off_w = add_param(pimg, name + ".weight", {cout, 3, vit.patch, vit.patch});
        else L.off_w = add_param(pimg, name + ".weight", {cout, cin});
        L.off_b = add_param(pimg, name + ".bias", {cout});
        L.off_g = -1;
        L.off_p = packed_total; packed_total += L.packed();
        convs.push_back(L);
        return (int)convs.size() - 1;
      };
      off_cls = add_param(pimg, "model.cls_token", {1, 1, D});
      off_pos = add_param(pimg, "model.pos_embed", {1, Tk, D});
      pe_conv = add_lin("model.patch_embed.proj", 3 * vit.patch * vit.patch, D, Tk - 1, true);
      for (int l = 0; l < vit.depth; ++l) {
        std::string pre = "model.blocks." + std::to_string(l);
        VitBlkL B;
        B.ln1_w = add_param(pimg, pre + ".norm1.weight", {D}); B.ln1_b = add_param(pimg, pre + ".norm1.bias", {D});
        B.qkv = add_lin(pre + ".attn.qkv", D, 3 * D, Tk, false);
        B.proj = add_lin(pre + ".attn.proj", D, D, Tk, false);
        B.ln2_w = add_param(pimg, pre + ".norm2.weight", {D}); B.ln2_b = add_param(pimg, pre + ".norm2.bias", {D});
        B.fc1 = add_lin(pre + ".mlp.fc1", D, 4 * D, Tk, false);
        B.fc2 = add_lin(pre + ".mlp.fc2", 4 * D, D, Tk, false);
        vblk.push_back(B);
      }
      off_nw = add_param(pimg, "model.norm.weight", {D}); off_nb = add_param(pimg, "model.norm.bias", {D});
      if (vit.head > 0) {
        off_hw = add_param(pimg, "model.head.weight", {vit.head, D}); off_hb = add_param(pimg, "model.head.bias", {vit.head});
      }
      int ce = 16 / (int)sizeof(AT);
      CHECK_ARG(D % ce == 0 && (3 * vit.patch * vit.patch) % ce == 0 && (D / vit.heads) % 4 == 0,
                "dims must be multiples of the 16-byte chunk");
      P_img = off; feat = vit.head > 0 ? vit.head : D;
      stem[0] = stem[1] = stem[2] = stem[3] = fin = -1;
      for (auto& L : convs) {
        LinPackDesc d; d.off_w = L.off_w; d.off_p = L.off_p; d.out = L.cout; d.in = L.cin; d.tile_start = lpd_tiles; d.pad_ = 0;
        lpd_tiles += ((L.cin + 63) / 64) * ((L.cout + 63) / 64);      // k_lin_pack_all: 64 x 64 tiles
        lpd.push_back(d);
      }
    } else {
    // stem 'deep_quad': 3x3 convs, channels (sc/8, sc/4, sc/2, sc), strides (2,1,1,2)
    int sc = nf.stem_chs, chs[4] = {sc / 8, sc / 4, sc / 2, sc}, strides[4] = {2, 1, 1, 2};
    int prev = 3, h = S;
    for (int i = 0; i < 4; ++i) {
      stem[i] = add_conv("model.stem.conv" + std::to_string(i + 1), prev, chs[i], 3, strides[i], 1, h);
      h = convs[stem[i]].hout; prev = chs[i];
    }
    xh.push_back(h); xc.push_back(prev);
    double expected_var = 1.0;
    for (int si = 0; si < 4; ++si) {
      for (int bi = 0; bi < nf.depths[si]; ++bi) {
        Blk B; std::string pre = "model.stages." + std::to_string(si) + "." + std::to_string(bi);
        int stride = (bi == 0 && si > 0) ? 2 : 1;
        int out = make_divisible(nf.channels[si], 8);
        int mid = make_divisible(out * nf.bottle_ratio, 8);
        int groups = nf.group_size ? mid / nf.group_size : 1;
        if (nf.group_size && nf.group_size % 8 == 0) mid = nf.group_size * groups;
        B.stride = stride; B.cin = prev; B.cout = out; B.mid = mid; B.hin = h;
        B.beta = (float)(1.0 / std::sqrt(expected_var));
        int hout = stride == 2 ? (h + 1) / 2 : h;
        B.ds = -1;
        if (prev != out || stride != 1) B.ds = add_conv(pre + ".downsample.conv", prev, out, 1, 1, 1, hout);
        B.c1 = add_conv(pre + ".conv1", prev, mid, 1, 1, 1, h);
        B.c2 = add_conv(pre + ".conv2", mid, mid, 3, stride, groups, h);
        B.c2b = add_conv(pre + ".conv2b", mid, mid, 3, 1, groups, hout);
        B.c3 = add_conv(pre + ".conv3", mid, out, 1, 1, 1, hout);
        CHECK_ARG(convs[B.c2].hout == hout, "internal: spatial plan");
        B.se.c = out; B.se.rd = make_divisible(out * nf.se_rd_ratio, 8, 0.0);
        B.se.off_w1 = add_param(pimg, pre + ".attn_last.fc1.weight", {B.se.rd, out, 1, 1});
        B.se.off_b1 = add_param(pimg, pre + ".attn_last.fc1.bias", {B.se.rd});
        B.se.off_w2 = add_param(pimg, pre + ".attn_last.fc2.weight", {out, B.se.rd, 1, 1});
        B.se.off_b2 = add_param(pimg, pre + ".attn_last.fc2.bias", {out});
        B.hout = hout;
        blks.push_back(B);
        if (bi == 0) expected_var = 1.0;
        expected_var += (double)nf.alpha * nf.alpha;
        prev = out; h = hout;
        xh.push_back(h); xc.push_back(prev);
      }
    }
    feat = (int)(nf.channels[3] * nf.feat_mult);
    fin = add_conv("model.final_conv", prev, feat, 1, 1, 1, h);
    P_img = off;
    }   // NFNet topology
    CHECK_ARG(feat <= 4096, "feature dim > 4096 unsupported by the contrastive head kernel");
    // text head (reference networks.py:625-646)
    off = 0;
    t_pw = add_param(ptxt, "projection.weight", {feat, Dt});
    t_pb = add_param(ptxt, "projection.bias", {feat});
    t_fw = add_param(ptxt, "fc.weight", {feat, feat});
    t_fb = add_param(ptxt, "fc.bias", {feat});
    t_lw = add_param(ptxt, "layer_norm.weight", {feat});
    t_lb = add_param(ptxt, "layer_norm.bias", {feat});
    P_txt = off;
    // weight-standardisation descriptors
    int row = 0, tile = 0;
    const int tr = ws_tile_rows();
    for (auto& L : convs) {
      if (is_vit) break;      // plain linears: cast / transpose pack (vit_pack), no weight standardisation
      WsDesc d; d.off_w = L.off_w; d.off_b = L.off_b; d.off_g = L.off_g;
      d.off_wf = L.off_p; d.off_wt = L.off_p;
      d.cout = L.cout; d.cin_g = L.cin / L.groups; d.ksq = L.k * L.k; d.groups = L.groups;
      d.cin_pad_g = L.cin_pad / L.groups; d.cout_g = L.cout / L.groups;
      d.scale = nf.gamma / std::sqrt((float)(d.cin_g * d.ksq)); d.eps = nf.eps;
      d.row_start = row; row += L.cout;
      d.tile_start = tile; tile += d.groups * ((d.cout_g + tr - 1) / tr);
      descs.push_back(d);
      int ce = 16 / (int)sizeof(AT);
      CHECK_ARG(d.cin_pad_g % ce == 0 && d.cout_g % ce == 0, "channel counts must be multiples of the 16-byte chunk");
    }
    total_rows = row; total_tiles = tile;
    plan_workspace();
    return 0;
  }

  void plan_set_vit(ActSet& s, int slot) {
    const int64_t n = N, D = vit.dim, M = n * Tk, kk = 3 * vit.patch * vit.patch, np = Tk - 1;
    const int64_t pr = n * vit.heads * Tk * sld;
    plan(&s.VCOL, n * np * kk, "vit.COL", slot); plan(&s.VPE, n * np * D, "vit.PE", slot);
    plan(&s.VCOLB, n * np * kk, "vit.COLB", slot); plan(&s.VPEB, n * np * D, "vit.PEB", slot);
    plan(&s.CLS, n * D, "vit.CLS", slot); plan(&s.CLSN, n * D, "vit.CLSN", slot);
    plan(&s.CLSNB, n * D, "vit.CLSNB", slot); plan(&s.CLSB, n * D, "vit.CLSB", slot);
    plan(&s.TMP, M * D, "vit.TMP", slot);
    plan(&s.hin, n * D, "vit.hin", slot); plan(&s.hinB, n * D, "vit.hinB", slot);
    s.X.resize(vit.depth + 1); s.XB.resize(vit.depth + 1); s.vb.resize(vit.depth);
    for (int l = 0; l <= vit.depth; ++l) {
      plan(&s.X[l], M * D, ("X" + std::to_string(l)).c_str(), slot);
      plan(&s.XB[l], M * D, ("XB" + std::to_string(l)).c_str(), slot);
    }
    for (int l = 0; l < vit.depth; ++l) {
      VitActs& a = s.vb[l];
      std::string p = "v" + std::to_string(l) + ".";
      plan(&a.N1, M * D, (p + "N1").c_str(), slot);   plan(&a.QKV, M * 3 * D, (p + "QKV").c_str(), slot);
      if (!fused_attn) plan(&a.P, pr, (p + "P").c_str(), slot); else a.P = nullptr;
      plan(&a.O, M * D, (p + "O").c_str(), slot);
      plan(&a.X2, M * D, (p + "X2").c_str(), slot);   plan(&a.N2, M * D, (p + "N2").c_str(), slot);
      plan(&a.C, M * 4 * D, (p + "C").c_str(), slot); plan(&a.A, M * 4 * D, (p + "A").c_str(), slot);
      plan(&a.N1B, M * D, (p + "N1B").c_str(), slot); plan(&a.QKVB, M * 3 * D, (p + "QKVB").c_str(), slot);
      if (!fused_attn) { plan(&a.PB, pr, (p + "PB").c_str(), slot); plan(&a.SB, pr, (p + "SB").c_str(), slot); }
      else {
        a.PB = a.SB = nullptr;
        const int64_t ns = n * vit.heads * Tk;
        plan(&a.sm, ns, nullptr, slot); plan(&a.sl, ns, nullptr, slot); plan(&a.sr, ns, nullptr, slot);
        plan(&a.sD, ns, nullptr, slot); plan(&a.sDt, ns, nullptr, slot);
      }
      plan(&a.OB, M * D, (p + "OB").c_str(), slot);   plan(&a.X2B, M * D, (p + "X2B").c_str(), slot);
      plan(&a.N2B, M * D, (p + "N2B").c_str(), slot); plan(&a.CB, M * 4 * D, (p + "CB").c_str(), slot);
      plan(&a.AB, M * 4 * D, (p + "AB").c_str(), slot);
    }
    plan(&s.y, n * feat, "y", slot); plan(&s.yB, n * feat, "yB", slot);
    plan(&s.tx, n * Dt, "tx", slot); plan(&s.tp, n * feat, "tp", slot); plan(&s.tg, n * feat, "tg", slot);
    plan(&s.tf, n * feat, "tf", slot); plan(&s.tr, n * feat, "tr", slot); plan(&s.ty, n * feat, "ty", slot);
    plan(&s.tyB, n * feat, "tyB", slot); plan(&s.trB, n * feat, "trB", slot); plan(&s.tfB, n * feat, "tfB", slot);
    plan(&s.tgB, n * feat, "tgB", slot); plan(&s.tpB, n * feat, "tpB", slot); plan(&s.txB, n * Dt, "txB", slot);
  }
  void plan_set(ActSet& s, int slot) {
    if (is_vit) { plan_set_vit(s, slot); return; }
    int nb = (int)blks.size();
    auto nm = [&](const std::string& n) { return n; };
    int64_t n = N;
    plan(&s.X0, n * S * S * 8, "IN", slot);
    for (int i = 0; i < 3; ++i) {
      const ConvL& L = convs[stem[i]];
      int64_t e = n * L.hout * L.hout * L.cout;
      plan(&s.Cs[i], e, ("stem.C" + std::to_string(i)).c_str(), slot);
      plan(&s.As[i], e, ("stem.A" + std::to_string(i)).c_str(), slot);
      plan(&s.AsB[i], e, ("stem.AB" + std::to_string(i)).c_str(), slot);
      plan(&s.CsB[i], e, ("stem.CB" + std::to_string(i)).c_str(), slot);
    }
    plan(&s.X0B, n * S * S * 8, "INB", slot);
    s.X.resize(nb + 1); s.A.resize(nb + 1); s.XB.resize(nb + 1); s.blk.resize(nb);
    for (int b = 0; b <= nb; ++b) {
      int64_t e = n * xh[b] * xh[b] * xc[b];
      plan(&s.X[b], e, ("X" + std::to_string(b)).c_str(), slot);
      plan(&s.XB[b], e, ("XB" + std::to_string(b)).c_str(), slot);
      if (b < nb) plan(&s.A[b], e, ("A" + std::to_string(b)).c_str(), slot); else s.A[b] = nullptr;
    }
    for (int b = 0; b < nb; ++b) {
      const Blk& B = blks[b]; BlockActs& a = s.blk[b];
      std::string p = "b" + std::to_string(b) + ".";
      int64_t ein = n * B.hin * B.hin, eout = n * B.hout * B.hout;
      a.P = nullptr; a.SC = nullptr;
      if (B.ds >= 0 && B.stride == 2) plan(&a.P, eout * B.cin, (p + "P").c_str(), slot);
      if (B.ds >= 0) plan(&a.SC, eout * B.cout, (p + "SC").c_str(), slot);
      plan(&a.C1, ein * B.mid, (p + "C1").c_str(), slot);   plan(&a.A1, ein * B.mid, (p + "A1").c_str(), slot);
      plan(&a.C2, eout * B.mid, (p + "C2").c_str(), slot);  plan(&a.A2, eout * B.mid, (p + "A2").c_str(), slot);
      plan(&a.C2b, eout * B.mid, (p + "C2b").c_str(), slot); plan(&a.A2b, eout * B.mid, (p + "A2b").c_str(), slot);
      plan(&a.C3, eout * B.cout, (p + "C3").c_str(), slot);
      plan(&a.q, n * B.mid, (p + "q").c_str(), slot); plan(&a.qB, n * B.mid, (p + "qB").c_str(), slot);
      plan(&a.p, n * B.se.c, (p + "p").c_str(), slot); plan(&a.h, n * B.se.rd, (p + "h").c_str(), slot);
      plan(&a.gate, n * B.se.c, (p + "gate").c_str(), slot);
      plan(&a.C3B, eout * B.cout, (p + "C3B").c_str(), slot);
      plan(&a.A2bB, eout * B.mid, (p + "A2bB").c_str(), slot); plan(&a.C2bB, eout * B.mid, (p + "C2bB").c_str(), slot);
      plan(&a.A2B, eout * B.mid, (p + "A2B").c_str(), slot);   plan(&a.C2B, eout * B.mid, (p + "C2B").c_str(), slot);
      plan(&a.A1B, ein * B.mid, (p + "A1B").c_str(), slot);    plan(&a.C1B, ein * B.mid, (p + "C1B").c_str(), slot);
      plan(&a.AinB, ein * B.cin, (p + "AinB").c_str(), slot);
      plan(&a.zB, n * B.se.c, (p + "zB").c_str(), slot);
      plan(&a.hB, n * B.se.rd, (p + "hB").c_str(), slot);      plan(&a.pB, n * B.se.c, (p + "pB").c_str(), slot);
    }
    int64_t ef = n * xh[nb] * xh[nb] * feat;
    plan(&s.CF, ef, "CF", slot); plan(&s.CFB, ef, "CFB", slot);
    plan(&s.y, n * feat, "y", slot); plan(&s.yB, n * feat, "yB", slot);
    plan(&s.dwf, packed_total, "dwf", slot);
    plan(&s.tx, n * Dt, "tx", slot); plan(&s.tp, n * feat, "tp", slot); plan(&s.tg, n * feat, "tg", slot);
    plan(&s.tf, n * feat, "tf", slot); plan(&s.tr, n * feat, "tr", slot); plan(&s.ty, n * feat, "ty", slot);
    plan(&s.tyB, n * feat, "tyB", slot); plan(&s.trB, n * feat, "trB", slot); plan(&s.tfB, n * feat, "tfB", slot);
    plan(&s.tgB, n * feat, "tgB", slot); plan(&s.tpB, n * feat, "tpB", slot); plan(&s.txB, n * Dt, "txB", slot);
    (void)nm;
  }

  void plan_workspace() {
    arena = Arena(); fix.clear(); names.clear();
    o_descs = arena.take((int64_t)descs.size() * sizeof(WsDesc));
    o_lpd = arena.take((int64_t)lpd.size() * sizeof(LinPackDesc));
    plan(&wf, packed_total, "wf", -2); plan(&wt, packed_total, "wt", -2);
    plan(&wf_t, packed_total, "wf_t", -2); plan(&wt_t, packed_total, "wt_t", -2);
    int64_t dsmax = 0, dfmax = 0;
    for (auto& B : blks) {
      if (B.ds >= 0) {
        dsmax = std::max(dsmax, (int64_t)N * B.hout * B.hout * B.cin);
        dfmax = std::max(dfmax, (int64_t)N * B.hin * B.hin * B.cin);
      }
    }
    plan(&dsS, dsmax, "dsS", -2); plan(&dsF, dfmax, "dsF", -2);
    for (int i = 0; i < 3; ++i) plan(&lin_mem[i], lin_scratch_bytes(), nullptr, -2);
    // conv_wgrad split-M slabs: room for ~16 partial copies of the largest dW, at least 64 MiB, per stream
    {
      int64_t big = 0;
      for (auto& L : convs) big = std::max(big, L.packed());
      wslab_floats = std::max<int64_t>(16 * big, (int64_t)16 << 20);
      plan(&wslab[0], wslab_floats, nullptr, -2); plan(&wslab[1], wslab_floats, nullptr, -2);
    }
    plan(&ln_stats, (int64_t)N * 4, nullptr, -2);
    // squeeze-excite shortcut operands, per block (se_prep): W13 = W1 W3_hat [rd, mid], b13 = W1 b3 + b1, tangents
    se13.resize(blks.size());
    for (size_t b = 0; b < blks.size(); ++b) {
      const int64_t rm = (int64_t)blks[b].se.rd * blks[b].mid;
      plan(&se13[b].W, rm, nullptr, -2); plan(&se13[b].W_t, rm, nullptr, -2);
      plan(&se13[b].b, blks[b].se.rd, nullptr, -2); plan(&se13[b].b_t, blks[b].se.rd, nullptr, -2);
    }
    plan(&lossw, loss_work_floats(N, feat), nullptr, -2);
    sets.resize(nslots);
    for (int k = 0; k < nslots; ++k) plan_set(sets[k], k);
    plan_set(tn, -1);
    thI.assign(K + 1, nullptr); thT.assign(K + 1, nullptr); gI.assign(K, nullptr); gT.assign(K, nullptr);
    for (int k = 1; k <= K; ++k) { plan(&thI[k], P_img, nullptr, -2); plan(&thT[k], P_txt, nullptr, -2); }
    for (int k = 0; k < K; ++k) { plan(&gI[k], P_img, nullptr, -2); plan(&gT[k], P_txt, nullptr, -2); }
    plan(&lamI, P_img, nullptr, -2); plan(&lamT, P_txt, nullptr, -2);
    plan(&nuI, P_img, nullptr, -2); plan(&nuT, P_txt, nullptr, -2);
    plan(&hI, P_img, nullptr, -2); plan(&hT, P_txt, nullptr, -2);
    int64_t nf_ = (int64_t)N * feat;
    plan(&fx_t, nf_, nullptr, -2); plan(&fy_t, nf_, nullptr, -2);
    plan(&xbar, nf_, nullptr, -2); plan(&ybar, nf_, nullptr, -2);
    plan(&xbar_t, nf_, nullptr, -2); plan(&ybar_t, nf_, nullptr, -2);
    plan(&sbar, 64, nullptr, -2); plan(&sbar_t, 64, nullptr, -2);
    plan(&dsc, 16, nullptr, -2);
    ws_bytes = arena.cur;
  }
  int64_t workspace_bytes() const override { return ws_bytes; }

  // The engine belongs to the device its workspace lives on: every entry point makes that device current for
  // the duration of the call (a caller that drives several GPUs from one process, as the reference's
  // nn.DataParallel path does, may arrive with another device current) and restores the caller's afterwards.
  int device_id = -1;
  struct DevGuard {
    int prev = -1, want;
    explicit DevGuard(int d) : want(d) {
      if (d >= 0 && hipGetDevice(&prev) == hipSuccess && prev != d) (void)hipSetDevice(d); else prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  };
  int bind(void* ws, int64_t bytes, hipStream_t st) override {
    CHECK_ARG(ws != nullptr, "workspace is null");
    CHECK_ARG(bytes >= ws_bytes, "workspace too small");
    CHECK_ARG(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
    {
      hipPointerAttribute_t attr;
      HIP_CHECK_RET(hipPointerGetAttributes(&attr, ws));
      device_id = attr.device;
    }
    DevGuard guard(device_id);
    base = (char*)ws;
    for (auto& f : fix) *f.first = (void*)(base + f.second);
    d_descs = (WsDesc*)(base + o_descs);
    if (!descs.empty())
      HIP_CHECK_RET(hipMemcpyAsync(d_descs, descs.data(), descs.size() * sizeof(WsDesc),
                                   hipMemcpyHostToDevice, st));
    d_lpd = (LinPackDesc*)(base + o_lpd);
    if (!lpd.empty())
      HIP_CHECK_RET(hipMemcpyAsync(d_lpd, lpd.data(), lpd.size() * sizeof(LinPackDesc), hipMemcpyHostToDevice, st));
    // packed weight buffers carry zero padding (conv1: 3 -> 8 input channels) that is never rewritten
    HIP_CHECK_RET(hipMemsetAsync(wf, 0, packed_total * sizeof(AT), st));
    HIP_CHECK_RET(hipMemsetAsync(wt, 0, packed_total * sizeof(AT), st));
    HIP_CHECK_RET(hipMemsetAsync(wf_t, 0, packed_total * sizeof(AT), st));
    HIP_CHECK_RET(hipMemsetAsync(wt_t, 0, packed_total * sizeof(AT), st));
    for (int i = 0; i < 3; ++i) HIP_CHECK_RET(hipMemsetAsync(lin_mem[i], 0, lin_scratch_bytes(), st));
    lin_main = lin_scratch_carve(lin_mem[0]); lin_side = lin_scratch_carve(lin_mem[1]);
    lin_txt = lin_scratch_carve(lin_mem[2]);
    HIP_CHECK_RET(hipStreamSynchronize(st));  // descs.data() is host memory
    if (!side) {
#ifdef MDD_DEBUG_SWITCHES
      const char* env = getenv("MDD_SIDE_STREAM");
      use_side = !(env && env[0] == '0');
#endif
#if MDD_SIDE_PRIORITY
      // experiment: the weight-gradient (1) / text (2) streams below the caller's stream in queue priority
      int prio_lo = 0, prio_hi = 0;
      HIP_CHECK_RET(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));   // lo = least priority (largest number)
      if (use_side) HIP_CHECK_RET(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, (MDD_SIDE_PRIORITY & 1) ? prio_lo : 0));
      if (use_side) HIP_CHECK_RET(hipStreamCreateWithPriority(&tside, hipStreamNonBlocking, (MDD_SIDE_PRIORITY & 2) ? prio_lo : 0));
#else
      if (use_side) HIP_CHECK_RET(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      if (use_side) HIP_CHECK_RET(hipStreamCreateWithFlags(&tside, hipStreamNonBlocking));
#endif
      if (use_side && MDD_SE_SIDE == 2) {
        int prio_lo_ = 0, prio_hi_ = 0;
        HIP_CHECK_RET(hipDeviceGetStreamPriorityRange(&prio_lo_, &prio_hi_));   // hi = greatest priority (smallest number)
        HIP_CHECK_RET(hipStreamCreateWithPriority(&gstream, hipStreamNonBlocking, prio_hi_));
      }
    }
    return 0;
  }

  // ------------------------------------------------------------------ geometry helpers
  ConvGeom gfwd(const ConvL& L) const {
    ConvGeom g; g.nimg = L.tokens ? N * L.tokens : N; g.ha = L.hin; g.wa = L.hin; g.ca_tot = L.cin_pad;
    g.ho = L.hout; g.wo = L.hout; g.co_tot = L.cout; g.kc = L.cin_pad / L.groups;
    g.nc = L.cout / L.groups; g.groups = L.groups; g.k = L.k; g.stride = L.stride; g.pad = L.pad;
    g.transposed = 0; g.prec = cur_prec; return g;
  }
  ConvGeom gdgrad(const ConvL& L) const {
    ConvGeom g; g.nimg = L.tokens ? N * L.tokens : N; g.ha = L.hout; g.wa = L.hout; g.ca_tot = L.cout;
    g.ho = L.hin; g.wo = L.hin; g.co_tot = L.cin_pad; g.kc = L.cout / L.groups;
    g.nc = L.cin_pad / L.groups; g.groups = L.groups; g.k = L.k; g.stride = L.stride; g.pad = L.pad;
    g.transposed = 1; g.prec = cur_prec; return g;
  }

  // ---- side stream: weight-gradient contractions only depend on (dy, x) of their own layer, so
  // they run beside the data-gradient chain instead of in it (both are latency-bound on their own).
  hipStream_t side = nullptr;
  bool use_side = true;
  ~Eng() override {   // streams / events created at bind(); the workspace belongs to the caller
#if MDD_GRAPH
    if (gexec) (void)hipGraphExecDestroy(gexec);
#endif
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (tside) { (void)hipStreamSynchronize(tside); (void)hipStreamDestroy(tside); }
    if (gstream) { (void)hipStreamSynchronize(gstream); (void)hipStreamDestroy(gstream); }
    for (auto e : evs) (void)hipEventDestroy(e);
    for (auto e : tf_ev) if (e) (void)hipEventDestroy(e);
    if (se13_ev) (void)hipEventDestroy(se13_ev);
    for (auto& p : prof) { if (!p.shared_a) (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  }
  std::vector<hipEvent_t> evs;
  size_t evi = 0;
  // The stream-ordering helpers below are called from inside the walkers (void lambdas, queued
  // closures); the first failing HIP call is latched here and turned into the pass's return code by
  // ASYNC_CHECK at the end of every walker -- a failed event record / stream wait would otherwise
  // silently drop a dependency between the main and the side stream.
  hipError_t aerr = hipSuccess;
  const char* aerr_what = "";
  void ck(hipError_t e, const char* what) {
    if (e != hipSuccess && aerr == hipSuccess) { aerr = e; aerr_what = what; }
  }
  int async_status() {
    if (aerr == hipSuccess) return 0;
    hipError_t e = aerr; aerr = hipSuccess;
    return mdd_set_error(e, aerr_what);
  }
  hipEvent_t next_event() {
    if (evs.size() < 512) {
      hipEvent_t e = nullptr;
      ck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
      if (e) { evs.push_back(e); return e; }
    }
    return evs.empty() ? nullptr : evs[evi++ % evs.size()];
  }
  void order(hipStream_t waiter, hipStream_t signaller) {   // waiter waits for everything enqueued on signaller
    hipEvent_t e = next_event();
    if (!e) { ck(hipErrorOutOfMemory, "no event for stream ordering"); return; }
    ck(hipEventRecord(e, signaller), "hipEventRecord");
    ck(hipStreamWaitEvent(waiter, e, 0), "hipStreamWaitEvent");
  }
  void fork(hipStream_t st) {  // side waits for everything enqueued on st so far
    if (!use_side) return;
    order(side, st);
  }
  void join(hipStream_t st) {  // st waits for everything enqueued on side so far
    if (!use_side) return;
    order(st, side);
  }
  hipStream_t wstream(hipStream_t st) const { return use_side ? side : st; }
  const LinScratch& wlin() const { return use_side ? lin_side : lin_main; }
  hipStream_t tside = nullptr;   // text-projection stream
  hipStream_t gstream = nullptr; // squeeze-excite gate chain of the forward passes (high queue priority: its small
                                 // kernels take the next CU slots a conv3 block frees instead of queueing behind the grid)
  void fork_to(hipStream_t to, hipStream_t from) {
    if (to == from) return;
    order(to, from);
  }
  void join_from(hipStream_t from, hipStream_t to) { fork_to(to, from); }

  // ---- optional HIP-event timing of every contraction launch (bench.py roofline accounting)
  struct Prof { int kind; double flops, bytes; hipEvent_t a, b; ConvGeom g; int ns; bool shared_a = false; };
  bool prof_on = false;
  std::vector<Prof> prof;
  void profile_enable(bool on) override {
    for (auto& p : prof) { if (!p.shared_a) (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    prof.clear(); prof_on = on;
  }
  int profile_read(int kind, double* out) override {  // {launches, total ms, flops, bytes}
    out[0] = out[1] = out[2] = out[3] = 0;
    for (auto& p : prof) if (p.kind == kind) {
      float ms = 0; HIP_CHECK_RET(hipEventSynchronize(p.b)); HIP_CHECK_RET(hipEventElapsedTime(&ms, p.a, p.b));
      out[0] += 1; out[1] += ms; out[2] += p.flops; out[3] += p.bytes;
    }
    return 0;
  }
  int profile_dump(const char* path) override {
    FILE* f = fopen(path, "w");
    if (!f) return mdd_set_error_msg(4, "mdd: cannot open profile dump file");
    fprintf(f, "kind,transposed,nsrc,M,nc,groups,kc,k,stride,ha,ho,ms,gflops,gbytes,tflops_s,gb_s\n");
    for (auto& p : prof) {
      float ms = 0;
      if (hipEventSynchronize(p.b) != hipSuccess || hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) {
        fclose(f);
        return mdd_set_error_msg(4, "mdd: profile events could not be read");
      }
      fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.3f,%.4f,%.1f,%.1f\n", p.kind, p.g.transposed,
              p.ns, p.g.nimg * p.g.ho * p.g.wo, p.g.nc, p.g.groups, p.g.kc, p.g.k, p.g.stride, p.g.ha,
              p.g.ho, ms, p.flops / 1e9, p.bytes / 1e9, p.flops / ms / 1e9, p.bytes / ms / 1e6);
    }
    fclose(f);
    return 0;
  }
  double conv_macs(const ConvL& L) const {
    return (double)N * (L.tokens ? L.tokens : 1) * L.hout * L.hout * L.cout * (L.cin / L.groups) * L.k * L.k;
  }
  void gemm(const ConvL& L, const ConvGeom& g, const AT* A1, const AT* B1, const AT* A2, const AT* B2,
            const ConvEpi& e, hipStream_t st) {
    if (!prof_on) { launch_conv_gemm<AT>(g, A1, B1, A2, B2, e, st); return; }
    Prof p; int ns = A2 ? 2 : 1;
    p.kind = g.nc <= 32 ? 0 : (g.nc <= 64 ? 1 : 2);
    p.flops = 2.0 * conv_macs(L) * ns;
    double ain = (double)g.nimg * g.ha * g.wa * g.ca_tot, aout = (double)g.nimg * g.ho * g.wo * g.co_tot;
    int nio = (e.out_raw ? 1 : 0) + (e.out_act ? 1 : 0) + (e.c ? 1 : 0) + (e.c_t ? 1 : 0) +
              (e.abar ? 1 : 0) + (e.add1 ? 1 : 0) + (e.add2 ? 1 : 0);
    p.bytes = (ain * ns + (double)L.packed() * ns + aout * nio) * sizeof(AT);
    ck(hipEventCreate(&p.a), "hipEventCreate"); ck(hipEventCreate(&p.b), "hipEventCreate");
    ck(hipEventRecord(p.a, st), "hipEventRecord");
    launch_conv_gemm<AT>(g, A1, B1, A2, B2, e, st);
    ck(hipEventRecord(p.b, st), "hipEventRecord");
    p.g = g; p.ns = ns; prof.push_back(p);
  }
  void wgrad(const ConvL& L, const ConvGeom& g, const AT* dy1, const AT* x1, const AT* dy2, const AT* x2,
             float* dW, float* db, hipStream_t st) {
    float* slab = (use_side && st == side) ? wslab[1] : wslab[0];
    if (!prof_on) { launch_conv_wgrad<AT>(g, dy1, x1, dy2, x2, dW, db, slab, wslab_floats, nullptr, st); return; }
    Prof p; int ns = dy2 ? 2 : 1;
    p.kind = 3;
    p.flops = 2.0 * conv_macs(L) * ns;
    double ain = (double)g.nimg * g.ha * g.wa * g.ca_tot, aout = (double)g.nimg * g.ho * g.wo * g.co_tot;
    p.bytes = (ain + aout) * ns * sizeof(AT) + (double)L.packed() * 4;
    ck(hipEventCreate(&p.a), "hipEventCreate"); ck(hipEventCreate(&p.b), "hipEventCreate");
    ck(hipEventRecord(p.a, st), "hipEventRecord");
    // [a, b] brackets the contraction kernel alone (b is recorded by the launcher before the split-M
    // reduce kernel), [b, c] the reduce kernel: kind 4
    Prof r; r.kind = 4; r.flops = 0; r.bytes = 0; r.g = g; r.ns = ns; r.a = p.b;
    ck(hipEventCreate(&r.b), "hipEventCreate");
    launch_conv_wgrad<AT>(g, dy1, x1, dy2, x2, dW, db, slab, wslab_floats, p.b, st);
    ck(hipEventRecord(r.b, st), "hipEventRecord");
    p.g = g; p.ns = ns; prof.push_back(p);
    r.shared_a = true; prof.push_back(r);
  }

  // c = conv(in) + bias ; C <- c ; A <- beta*silu(c).   T: tangent of the same (primal from stash)
  void conv_fwd(bool T, const ConvL& L, const AT* in, const AT* in_t, AT* C, AT* C_t, AT* A,
                AT* A_t, float beta, const float* th, const float* th_t, hipStream_t st, int act = 0) {
    ConvGeom g = gfwd(L);
    ConvEpi e; memset(&e, 0, sizeof e);
    e.beta = beta; e.act = act;
    if (!T) {
      e.mode = EPI_FWD; e.bias = th + L.off_b; e.out_raw = C; e.out_act = A;
      gemm(L, g, in, wf + L.off_p, nullptr, nullptr, e, st);
    } else {
      e.mode = EPI_FWD_T; e.bias_t = th_t + L.off_b; e.out_raw = C_t; e.out_act = A_t; e.c = C;
      if (in_t && tf_split_active) {   // conv(a, w_t) is already in C_t (tf_side_prepass): add it in the epilogue
        e.add1 = C_t;
        gemm(L, g, in_t, wf + L.off_p, nullptr, nullptr, e, st);
      }
      else if (in_t) gemm(L, g, in_t, wf + L.off_p, in, wf_t + L.off_p, e, st);
      else gemm(L, g, in, wf_t + L.off_p, nullptr, nullptr, e, st);
    }
  }
  // Tangent-forward pass, second sources: raw conv(a, w_t) of every layer whose input carries a tangent, written into the
  // layer's tangent buffer on the side stream.  They depend on the primal stash and the packed tangent weights only, so
  // the whole sequence is enqueued at the start of the pass and runs beside the main chain; one event per block.
  bool tf_split_active = false;
  std::vector<hipEvent_t> tf_ev;     // [0] stem, [1 + b] block b, [nb + 1] final conv
  void tf_raw(const ConvL& L, const AT* in, AT* dst, hipStream_t s2) {
    ConvGeom g = gfwd(L);
    ConvEpi e; memset(&e, 0, sizeof e);
    e.mode = EPI_FWD; e.out_raw = dst; e.beta = 1.f;
    gemm(L, g, in, wf_t + L.off_p, nullptr, nullptr, e, s2);
  }
  void tf_side_prepass(ActSet& P, ActSet& Q, hipStream_t st) {
    const int nb = (int)blks.size();
    if (tf_ev.size() < (size_t)nb + 2) {
      tf_ev.resize(nb + 2, nullptr);
      for (auto& e : tf_ev)
        if (!e) ck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    fork(st);                         // after ws_forward: the packed tangent weights exist
    hipStream_t s2 = side;
    for (int i = 1; i < 4; ++i) tf_raw(convs[stem[i]], P.As[i - 1], i == 3 ? Q.X[0] : Q.Cs[i], s2);
    ck(hipEventRecord(tf_ev[0], s2), "hipEventRecord");
    for (int b = 0; b < nb; ++b) {
      const Blk& B = blks[b]; BlockActs& pa = P.blk[b]; BlockActs& qa = Q.blk[b];
      if (B.ds >= 0) tf_raw(convs[B.ds], B.stride == 2 ? pa.P : P.A[b], qa.SC, s2);
      tf_raw(convs[B.c1], P.A[b], qa.C1, s2);
      tf_raw(convs[B.c2], pa.A1, qa.C2, s2);
      tf_raw(convs[B.c2b], pa.A2, qa.C2b, s2);
      tf_raw(convs[B.c3], pa.A2b, qa.C3, s2);
      ck(hipEventRecord(tf_ev[1 + b], s2), "hipEventRecord");
    }
    tf_raw(convs[fin], P.X[nb], Q.CF, s2);
    ck(hipEventRecord(tf_ev[nb + 1], s2), "hipEventRecord");
  }
  void tf_wait(int i, hipStream_t st) {
    if (tf_split_active) ck(hipStreamWaitEvent(st, tf_ev[i], 0), "hipStreamWaitEvent");
  }

  // Weight-gradient work is queued while a block's data-gradient chain is enqueued on the main
  // stream and released to the side stream with ONE event per block (flush_w): an event record is a
  // barrier packet on the main queue (~8-12 us bubble each), and nothing on the main stream waits for
  // these results before the end of the pass.
  static constexpr bool MAIN_TAIL = true;
#ifndef MDD_VIT_FUSE_ADD_LN
#define MDD_VIT_FUSE_ADD_LN 1    // ViT: a residual sum is formed by the LayerNorm kernel that reads it (mdd_op_add_layernorm)
#endif
#ifndef MDD_WGRAD_GROUP
#define MDD_WGRAD_GROUP 1        // the wide pointwise layers queued between two flushes share one weight-gradient launch
#endif
#ifndef MDD_FLUSH_EVERY
#define MDD_FLUSH_EVERY 1        // release queued weight gradients to the side stream every this many blocks
#endif
#ifndef MDD_TAIL_MAIN
#define MDD_TAIL_MAIN 2          // how many of the stem's weight gradients run on the main stream at the pass's tail
#endif
#ifndef MDD_FWD_SHORTCUT_SIDE
#define MDD_FWD_SHORTCUT_SIDE 1  // forward passes: shortcut branch (avg-pool + 1x1 conv) on the side stream
#endif
  std::vector<std::function<void(hipStream_t)>> wq;
  // wide pointwise layers (the ViT linears) queued since the last flush: launched together, up to four per launch
  // (csrc/conv_wgrad.hip launch_conv_wgrad_group: a layer's 108 tiles need two pixel chunks instead of seven)
  struct WgQ { const ConvL* L; WgradItem it; };
  std::vector<WgQ> wgroup;
  void flush_group(hipStream_t ws_) {
    size_t i = 0;
    while (i < wgroup.size()) {
      size_t j = i + 1;
      while (j < wgroup.size() && j - i < 4 && (wgroup[j].it.dy2 != nullptr) == (wgroup[i].it.dy2 != nullptr)) ++j;
      WgradItem items[4];
      for (size_t k = i; k < j; ++k) items[k - i] = wgroup[k].it;
      float* slab = (use_side && ws_ == side) ? wslab[1] : wslab[0];
      if (j - i < 2 || !launch_conv_wgrad_group(items, (int)(j - i), slab, wslab_floats, ws_)) {
        for (size_t k = i; k < j; ++k) {
          const WgradItem& w = wgroup[k].it;
          wgrad(*wgroup[k].L, w.g, (const AT*)w.dy1, (const AT*)w.x1, (const AT*)w.dy2, (const AT*)w.x2, w.dW, w.dbias, ws_);
        }
      }
      i = j;
    }
    wgroup.clear();
  }
  void flush_w(hipStream_t st, bool on_main = false) {
    if (wq.empty() && wgroup.empty()) return;
    if (on_main) {   // tail of the pass: the side stream is the one lagging, the main stream helps out
      for (auto& f : wq) f(st);
      wq.clear();
      flush_group(st);
      return;
    }
    fork(st);
    hipStream_t ws_ = wstream(st);
    for (auto& f : wq) f(ws_);
    wq.clear();
    flush_group(ws_);
  }
  // weight + bias gradient of conv L (dy = grad wrt its raw output, x = its input)
  void conv_bwd_w(bool T, const ConvL& L, const AT* dy, const AT* dy_t, const AT* x, const AT* x_t,
                  float* dwf_, float* dwf_t_, float* gout, hipStream_t) {
    const ConvL* Lp = &L;
    if (sizeof(AT) == 2 && MDD_WGRAD_GROUP && !prof_on && conv_wgrad_group_takes(gfwd(L))) {
      WgQ q; q.L = Lp; q.it.g = gfwd(L);
      if (!T) { q.it.dy1 = dy; q.it.x1 = x; q.it.dy2 = nullptr; q.it.x2 = nullptr; q.it.dW = dwf_ + L.off_p; }
      else { q.it.dy1 = dy_t; q.it.x1 = x; q.it.dy2 = x_t ? dy : nullptr; q.it.x2 = x_t; q.it.dW = dwf_t_ + L.off_p; }
      q.it.dbias = gout + L.off_b;
      wgroup.push_back(q);
      return;
    }
    wq.push_back([=](hipStream_t ws_) {
      ConvGeom g = gfwd(*Lp);
      if (!T) wgrad(*Lp, g, dy, x, nullptr, nullptr, dwf_ + Lp->off_p, gout + Lp->off_b, ws_);
      else wgrad(*Lp, g, dy_t, x, x_t ? dy : nullptr, x_t, dwf_t_ + Lp->off_p, gout + Lp->off_b, ws_);
    });
  }
  // data gradient of conv L with fused epilogue
  void conv_bwd_d(bool T, const ConvL& L, const AT* dy, const AT* dy_t, ConvEpi e, hipStream_t st) {
    ConvGeom g = gdgrad(L);
    if (!T) gemm(L, g, dy, wt + L.off_p, nullptr, nullptr, e, st);
    else gemm(L, g, dy_t, wt + L.off_p, dy, wt_t + L.off_p, e, st);
  }
  ConvEpi epi_act(bool T, AT* raw, AT* act, AT* act_t, const AT* c, const AT* c_t, float beta,
                  const AT* add1, const AT* add2) {
    ConvEpi e; memset(&e, 0, sizeof e);
    e.beta = beta; e.add1 = add1; e.add2 = add2; e.c = c;
    if (!T) { e.mode = EPI_BWD; e.out_raw = raw; e.out_act = act; }
    else { e.mode = EPI_BWD_T; e.out_raw = nullptr; e.out_act = act_t; e.c_t = c_t; e.abar = raw; }
    return e;
  }
  ConvEpi epi_lin(AT* raw, const AT* add1) {
    ConvEpi e; memset(&e, 0, sizeof e);
    e.mode = EPI_BWD_LIN; e.out_raw = raw; e.add1 = add1; e.beta = 1.f; return e;
  }

  // ================================================================== ViT image encoder (BASELINE configs[4])
  // timm 0.6.7 VisionTransformer, class token as the feature (oracle/vit_ref.py).  The linears run on k_conv_gemm
  // MODE 0 over batch*tokens rows (ConvL::tokens) with their weight gradients written by k_conv_wgrad straight
  // into the flat gradient (a linear's packed layout is its parameter layout); LayerNorm / GELU / softmax / the
  // attention contractions are the S-generic kernels of vit.hip (mdd_op_*), whose tangent calls write only `_t`.
  static constexpr int VDT = sizeof(AT) == 4 ? MDD_DTYPE_F32 : MDD_DTYPE_BF16;
  // the attention contractions follow the engine's precision mode: exact FMA (f32), split-bf16 MFMA (bf16x2), bf16 MFMA
  int bdt() const { return sizeof(AT) == 4 ? (prec == 1 ? MDD_DTYPE_BF16X2 : MDD_DTYPE_F32) : MDD_DTYPE_BF16; }
  std::vector<LinPackDesc> lpd; LinPackDesc* d_lpd = nullptr; int64_t o_lpd = 0; int lpd_tiles = 0;
  void vit_pack(const float* th, const float* th_t, hipStream_t st) {
    launch_lin_pack_all<AT>(d_lpd, (int)lpd.size(), lpd_tiles, th, th_t, wf, wt, wf_t, wt_t, st);
  }
  // the weight gradient of a linear goes to gout + off_w: conv_bwd_w adds the layer's PACKED offset to its base
  void lin_bwd_w(bool T, const ConvL& L, const AT* dy, const AT* dy_t, const AT* x, const AT* x_t, float* gout,
                 hipStream_t st) {
    float* basep = gout + (L.off_w - L.off_p);
    conv_bwd_w(T, L, dy, dy_t, x, x_t, basep, basep, gout, st);
  }
  struct AttnDesc { mdd_bgemm_desc s, o, dp, dv, dq, dk; };
  AttnDesc attn_desc() const {
    const int64_t D = vit.dim, H = vit.heads, hd = D / H, R = 3 * D, T_ = Tk, ld = sld;
    auto mk_ = [&](int m, int n, int k, int64_t ar, int64_t ac, int64_t ao, int64_t aq, int64_t br, int64_t bc,
                   int64_t bo, int64_t bq, int64_t cr, int64_t cc, int64_t co, int64_t cq) {
      mdd_bgemm_desc d; memset(&d, 0, sizeof d);
      d.m = m; d.n = n; d.k = k; d.outer = N; d.inner = (int)H;
      d.a_row = ar; d.a_col = ac; d.a_outer = ao; d.a_inner = aq;
      d.b_row = br; d.b_col = bc; d.b_outer = bo; d.b_inner = bq;
      d.c_row = cr; d.c_col = cc; d.c_outer = co; d.c_inner = cq; d.alpha = 1.f;
      return d;
    };
    const int64_t po = H * T_ * ld, pq = T_ * ld;      // batch strides of the score matrices
    AttnDesc a;
    a.s = mk_(Tk, Tk, (int)hd, R, 1, T_ * R, hd, 1, R, T_ * R, hd, ld, 1, po, pq);        // q k^T
    a.o = mk_(Tk, (int)hd, Tk, ld, 1, po, pq, R, 1, T_ * R, hd, D, 1, T_ * D, hd);         // p v
    a.dp = mk_(Tk, Tk, (int)hd, D, 1, T_ * D, hd, 1, R, T_ * R, hd, ld, 1, po, pq);        // do v^T
    a.dv = mk_(Tk, (int)hd, Tk, 1, ld, po, pq, D, 1, T_ * D, hd, R, 1, T_ * R, hd);        // p^T do
    a.dq = mk_(Tk, (int)hd, Tk, ld, 1, po, pq, R, 1, T_ * R, hd, R, 1, T_ * R, hd);        // ds k
    a.dk = mk_(Tk, (int)hd, Tk, 1, ld, po, pq, R, 1, T_ * R, hd, R, 1, T_ * R, hd);        // ds^T q
    return a;
  }
#define VIT_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
  int vit_forward(bool T, int slot, const float* th, const float* th_t, const float* image, const int64_t* idx,
                  float* feat_out, hipStream_t st) {
    ActSet& P = sets[slot]; ActSet& Q = tn;
    const int D = vit.dim, M = N * Tk;
    const int64_t rows = (int64_t)N * vit.heads * Tk;
    const float scale = 1.f / std::sqrt((float)(D / vit.heads));
    const AttnDesc ad = attn_desc();
    auto tp = [&](const float* q) { return T ? q : (const float*)nullptr; };
    vit_pack(th, T ? th_t : nullptr, st);
    if (!T) launch_patchify<AT>(P.VCOL, image, idx, N, S, vit.patch, st);
    conv_fwd(T, convs[pe_conv], P.VCOL, nullptr, P.VPE, Q.VPE, nullptr, nullptr, 1.f, th, th_t, st);
    launch_vit_embed<AT>(P.X[0], T ? Q.X[0] : nullptr, P.VPE, T ? Q.VPE : nullptr, th + off_cls,
                         tp(th_t + off_cls), th + off_pos, tp(th_t + off_pos), N, Tk, D, st);
    for (int l = 0; l < vit.depth; ++l) {
      const VitBlkL& B = vblk[l]; VitActs& pa = P.vb[l]; VitActs& qa = Q.vb[l];
      const AT *x = P.X[l], *x_t = T ? Q.X[l] : nullptr;
      if (MDD_VIT_FUSE_ADD_LN && l > 0) {
        // x_l = x2_{l-1} + fc2 output (still in TMP) is formed by the kernel that normalises it
        VitActs& pp = P.vb[l - 1]; VitActs& qp = Q.vb[l - 1];
        VIT_RC(mdd_op_add_layernorm(VDT, M, D, vit.eps, pp.X2, T ? qp.X2 : nullptr, P.TMP, T ? Q.TMP : nullptr, P.X[l],
                                    T ? Q.X[l] : nullptr, th + B.ln1_w, tp(th_t + B.ln1_w), th + B.ln1_b, tp(th_t + B.ln1_b),
                                    pa.N1, T ? qa.N1 : nullptr, st));
      } else {
        VIT_RC(mdd_op_layernorm(VDT, M, D, vit.eps, x, x_t, th + B.ln1_w, tp(th_t + B.ln1_w), th + B.ln1_b,
                                tp(th_t + B.ln1_b), pa.N1, T ? qa.N1 : nullptr, st));
      }
      conv_fwd(T, convs[B.qkv], pa.N1, qa.N1, pa.QKV, qa.QKV, nullptr, nullptr, 1.f, th, th_t, st);
      // attention: scores -> probabilities (in place) -> weighted values.  Tangent of the softmax from the stashed
      // probabilities: p_t = scale * p * (s_t - <p, s_t>) -- the softmax Jacobian is symmetric, so it is the
      // backward kernel applied to the score tangent.
      const AT *q = pa.QKV, *k = pa.QKV + D, *v = pa.QKV + 2 * D;
      const AT *q_t = T ? qa.QKV : nullptr, *k_t = T ? qa.QKV + D : nullptr, *v_t = T ? qa.QKV + 2 * D : nullptr;
      if (fused_attn) {
        // one launch: the score tiles are recomputed on the matrix cores, only O (O_t) and the row statistics leave the CU
        if (!T) VIT_RC(launch_attention(0, pa.QKV, nullptr, nullptr, nullptr, pa.O, nullptr, nullptr, pa.sm, pa.sl, nullptr, nullptr,
                                        nullptr, N, Tk, vit.heads, scale, 0, 0, 0, st));
        else VIT_RC(launch_attention(1, pa.QKV, qa.QKV, nullptr, nullptr, qa.O, nullptr, nullptr, pa.sm, pa.sl, qa.sr, nullptr, nullptr,
                                     N, Tk, vit.heads, scale, 0, 0, 0, st));
      } else if (!T) {
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.s, q, nullptr, k, nullptr, pa.P, nullptr, st));
        VIT_RC(mdd_op_softmax(VDT, rows, Tk, sld, scale, pa.P, nullptr, pa.P, nullptr, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.o, pa.P, nullptr, v, nullptr, pa.O, nullptr, st));
      } else {
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.s, q, q_t, k, k_t, nullptr, qa.P, st));
        VIT_RC(mdd_op_softmax_bwd(VDT, rows, Tk, sld, scale, pa.P, nullptr, qa.P, nullptr, qa.P, nullptr, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.o, pa.P, qa.P, v, v_t, nullptr, qa.O, st));
      }
      conv_fwd(T, convs[B.proj], pa.O, qa.O, P.TMP, Q.TMP, nullptr, nullptr, 1.f, th, th_t, st);
      if (MDD_VIT_FUSE_ADD_LN) {
        VIT_RC(mdd_op_add_layernorm(VDT, M, D, vit.eps, x, x_t, P.TMP, T ? Q.TMP : nullptr, pa.X2, T ? qa.X2 : nullptr,
                                    th + B.ln2_w, tp(th_t + B.ln2_w), th + B.ln2_b, tp(th_t + B.ln2_b), pa.N2,
                                    T ? qa.N2 : nullptr, st));
      } else {
        launch_add2<AT>(pa.X2, T ? qa.X2 : nullptr, x, x_t, P.TMP, T ? Q.TMP : nullptr, (int64_t)M * D, st);
        VIT_RC(mdd_op_layernorm(VDT, M, D, vit.eps, pa.X2, T ? qa.X2 : nullptr, th + B.ln2_w, tp(th_t + B.ln2_w),
                                th + B.ln2_b, tp(th_t + B.ln2_b), pa.N2, T ? qa.N2 : nullptr, st));
      }
      // fc1 with the exact GELU in its epilogue (C and A = gelu(C) written by the contraction; tangent: C_t and
      // A_t = gelu'(C) C_t) wherever the GELU instance of the kernel exists for this width
      if (MDD_VIT_FUSE_GELU && conv_gemm_supports_gelu(gfwd(convs[B.fc1]))) {
        conv_fwd(T, convs[B.fc1], pa.N2, qa.N2, pa.C, qa.C, pa.A, qa.A, 1.f, th, th_t, st, 1);
      } else {
        conv_fwd(T, convs[B.fc1], pa.N2, qa.N2, pa.C, qa.C, nullptr, nullptr, 1.f, th, th_t, st);
        VIT_RC(mdd_op_gelu(VDT, (int64_t)M * 4 * D, pa.C, T ? qa.C : nullptr, pa.A, T ? qa.A : nullptr, st));
      }
      conv_fwd(T, convs[B.fc2], pa.A, qa.A, P.TMP, Q.TMP, nullptr, nullptr, 1.f, th, th_t, st);
      if (!(MDD_VIT_FUSE_ADD_LN && l + 1 < vit.depth))    // otherwise formed by the next layer's first LayerNorm
        launch_add2<AT>(P.X[l + 1], T ? Q.X[l + 1] : nullptr, pa.X2, T ? qa.X2 : nullptr, P.TMP, T ? Q.TMP : nullptr,
                        (int64_t)M * D, st);
    }
    // feature = LayerNorm(x_L)[:, 0]: normalise the class rows only
    launch_cls_gather<AT>(T ? Q.CLS : P.CLS, T ? Q.X[vit.depth] : P.X[vit.depth], N, Tk, D, st);
    VIT_RC(mdd_op_layernorm(VDT, N, D, vit.eps, P.CLS, T ? Q.CLS : nullptr, th + off_nw, tp(th_t + off_nw),
                            th + off_nb, tp(th_t + off_nb), P.CLSN, T ? Q.CLSN : nullptr, st));
    if (vit.head > 0) {
      // classifier head on the normalised class token (timm default; reference networks.py:668): small-batch fp32 linear
      launch_act_to_f32<AT>(T ? Q.hin : P.hin, T ? Q.CLSN : P.CLSN, (int64_t)N * D, st);
      launch_linear_fwd(P.y, T ? feat_out : nullptr, P.hin, T ? Q.hin : nullptr, th + off_hw, tp(th_t + off_hw),
                        th + off_hb, tp(th_t + off_hb), N, D, vit.head, 0, lin_main, st);
      if (!T && feat_out && feat_out != P.y)
        HIP_CHECK_RET(hipMemcpyAsync(feat_out, P.y, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    } else if (!T) {
      launch_act_to_f32<AT>(P.y, P.CLSN, (int64_t)N * D, st);
      if (feat_out && feat_out != P.y)
        HIP_CHECK_RET(hipMemcpyAsync(feat_out, P.y, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    } else {
      launch_act_to_f32<AT>(feat_out, Q.CLSN, (int64_t)N * D, st);
    }
    POST_WALK("vit_forward");
    return 0;
  }
  int vit_backward(bool T, int slot, const float* th, const float* th_t, const float* ybar_in,
                   const float* ybar_t_in, float* gout, float* dimage, const int64_t* idx, const float* coef,
                   float mul, bool repack, bool stash, hipStream_t st) {
    ActSet& P = sets[slot]; ActSet& Q = tn;
    ActSet& O = (!T && !stash) ? tn : P;       // where the primal backward signals are written
    const int D = vit.dim, M = N * Tk, L_ = vit.depth;
    const int64_t rows = (int64_t)N * vit.heads * Tk;
    const float scale = 1.f / std::sqrt((float)(D / vit.heads));
    const AttnDesc ad = attn_desc();
    auto tp = [&](const float* q) { return T ? q : (const float*)nullptr; };
    // gradient slices: a primal call fills gout with d/d theta, a tangent call with its tangent
    auto gp = [&](int64_t off) { return T ? (float*)nullptr : gout + off; };
    auto gt = [&](int64_t off) { return T ? gout + off : (float*)nullptr; };
    if (repack) vit_pack(th, T ? th_t : nullptr, st);
    HIP_CHECK_RET(hipMemsetAsync(gout, 0, P_img * 4, st));
    if (!T && ybar_in != O.yB)
      HIP_CHECK_RET(hipMemcpyAsync(O.yB, ybar_in, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    if (vit.head > 0) {
      launch_linear_wgrad(gout + off_hw, gout + off_hb, O.yB, T ? ybar_t_in : nullptr, P.hin, T ? Q.hin : nullptr, N, D,
                          vit.head, lin_main, st);
      launch_linear_dgrad(O.hinB, T ? Q.hinB : nullptr, O.yB, T ? ybar_t_in : nullptr, th + off_hw, tp(th_t + off_hw),
                          nullptr, N, D, vit.head, lin_main, st);
      launch_f32_to_act<AT>(T ? Q.CLSNB : O.CLSNB, T ? Q.hinB : O.hinB, (int64_t)N * D, st);
    } else
    launch_f32_to_act<AT>(T ? Q.CLSNB : O.CLSNB, T ? ybar_t_in : O.yB, (int64_t)N * D, st);
    VIT_RC(mdd_op_layernorm_bwd(VDT, N, D, vit.eps, P.CLS, T ? Q.CLS : nullptr, O.CLSNB, T ? Q.CLSNB : nullptr,
                                th + off_nw, tp(th_t + off_nw), nullptr, nullptr, T ? nullptr : O.CLSB,
                                T ? Q.CLSB : nullptr, gp(off_nw), gt(off_nw), gp(off_nb), gt(off_nb), st));
    launch_cls_scatter<AT>(T ? Q.XB[L_] : O.XB[L_], T ? Q.CLSB : O.CLSB, N, Tk, D, st);
    for (int l = L_ - 1; l >= 0; --l) {
      const VitBlkL& B = vblk[l]; VitActs& pa = P.vb[l]; VitActs& qa = Q.vb[l]; VitActs& oa = O.vb[l];
      const AT *xb = O.XB[l + 1], *xb_t = Q.XB[l + 1];
      // MLP branch
      lin_bwd_w(T, convs[B.fc2], xb, xb_t, pa.A, qa.A, gout, st);
      if (MDD_VIT_FUSE_GELU && conv_gemm_supports_gelu(gdgrad(convs[B.fc2]))) {
        // fc2's data gradient with the GELU chain rule in its epilogue: raw AB (stashed: the tangent pass needs it)
        // and CB = gelu'(C) AB; tangent: CB_t = (gelu'(Dual(C, C_t)) Dual(AB, AB_t)).t from the stashed C, C_t, AB
        ConvEpi eg = epi_act(T, oa.AB, oa.CB, qa.CB, pa.C, qa.C, 1.f, nullptr, nullptr);
        eg.act = 1;
        conv_bwd_d(T, convs[B.fc2], xb, xb_t, eg, st);
      } else {
        conv_bwd_d(T, convs[B.fc2], xb, xb_t, epi_lin(T ? qa.AB : oa.AB, nullptr), st);
        VIT_RC(mdd_op_gelu_bwd(VDT, (int64_t)M * 4 * D, pa.C, T ? qa.C : nullptr, oa.AB, T ? qa.AB : nullptr,
                               T ? nullptr : oa.CB, T ? qa.CB : nullptr, st));
      }
      lin_bwd_w(T, convs[B.fc1], oa.CB, qa.CB, pa.N2, qa.N2, gout, st);
      conv_bwd_d(T, convs[B.fc1], oa.CB, qa.CB, epi_lin(T ? qa.N2B : oa.N2B, nullptr), st);
      VIT_RC(mdd_op_layernorm_bwd(VDT, M, D, vit.eps, pa.X2, T ? qa.X2 : nullptr, oa.N2B, T ? qa.N2B : nullptr,
                                  th + B.ln2_w, tp(th_t + B.ln2_w), xb, T ? xb_t : nullptr, T ? nullptr : oa.X2B,
                                  T ? qa.X2B : nullptr, gp(B.ln2_w), gt(B.ln2_w), gp(B.ln2_b), gt(B.ln2_b), st));
      // attention branch
      lin_bwd_w(T, convs[B.proj], oa.X2B, qa.X2B, pa.O, qa.O, gout, st);
      conv_bwd_d(T, convs[B.proj], oa.X2B, qa.X2B, epi_lin(T ? qa.OB : oa.OB, nullptr), st);
      const AT *q = pa.QKV, *k = pa.QKV + D, *v = pa.QKV + 2 * D;
      if (fused_attn) {
        const int H_ = vit.heads;
        if (!T) {
          // d qkv = (dS K | dS^T Q | P^T dO): query-owner launch (writes D = rowsum(P dP)), then the key-owner one
          VIT_RC(launch_attention(2, pa.QKV, nullptr, oa.OB, nullptr, oa.QKVB, pa.O, nullptr, pa.sm, pa.sl, nullptr, oa.sD, nullptr,
                                  N, Tk, H_, scale, 0, 0, 0, st));
          VIT_RC(launch_attention(4, pa.QKV, nullptr, oa.OB, nullptr, oa.QKVB, nullptr, nullptr, pa.sm, pa.sl, nullptr, oa.sD, nullptr,
                                  N, Tk, H_, scale, 0, 0, 0, st));
        } else {
          // tangent: dQ_t = dS_t K + dS K_t ; dV_t = P_t^T dO + P^T dO_t ; dK_t = dS_t^T Q + dS^T Q_t
          VIT_RC(launch_attention(3, pa.QKV, qa.QKV, oa.OB, qa.OB, qa.QKVB, pa.O, qa.O, pa.sm, pa.sl, qa.sr, oa.sD, qa.sDt, N, Tk,
                                  H_, scale, 0, 0, 0, st));
          VIT_RC(launch_attention(2, pa.QKV, qa.QKV, oa.OB, nullptr, qa.QKVB, pa.O, nullptr, pa.sm, pa.sl, nullptr, qa.sD, nullptr, N,
                                  Tk, H_, scale, 1, 1, 0, st));
          VIT_RC(launch_attention(5, pa.QKV, qa.QKV, oa.OB, qa.OB, qa.QKVB, nullptr, nullptr, pa.sm, pa.sl, qa.sr, nullptr, nullptr, N, Tk,
                                  H_, scale, 0, 0, 0, st));
          VIT_RC(launch_attention(6, pa.QKV, qa.QKV, oa.OB, qa.OB, qa.QKVB, nullptr, nullptr, pa.sm, pa.sl, qa.sr, oa.sD, qa.sDt, N, Tk,
                                  H_, scale, 0, 0, 0, st));
          VIT_RC(launch_attention(4, pa.QKV, qa.QKV, oa.OB, nullptr, qa.QKVB, nullptr, nullptr, pa.sm, pa.sl, nullptr, oa.sD, nullptr, N,
                                  Tk, H_, scale, 1, 1, 1, st));
        }
      } else if (!T) {
        AT* z = oa.QKVB;
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dp, oa.OB, nullptr, v, nullptr, oa.PB, nullptr, st));
        VIT_RC(mdd_op_softmax_bwd(VDT, rows, Tk, sld, scale, pa.P, nullptr, oa.PB, nullptr, oa.SB, nullptr, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dv, pa.P, nullptr, oa.OB, nullptr, z + 2 * D, nullptr, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dq, oa.SB, nullptr, k, nullptr, z, nullptr, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dk, oa.SB, nullptr, q, nullptr, z + D, nullptr, st));
      } else {
        const AT *q_t = qa.QKV, *k_t = qa.QKV + D, *v_t = qa.QKV + 2 * D;
        AT* z = qa.QKVB;
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dp, oa.OB, qa.OB, v, v_t, nullptr, qa.PB, st));
        VIT_RC(mdd_op_softmax_bwd(VDT, rows, Tk, sld, scale, pa.P, qa.P, oa.PB, qa.PB, nullptr, qa.SB, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dv, pa.P, qa.P, oa.OB, qa.OB, nullptr, z + 2 * D, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dq, oa.SB, qa.SB, k, k_t, nullptr, z, st));
        VIT_RC(mdd_op_bgemm(bdt(), 0, 0, &ad.dk, oa.SB, qa.SB, q, q_t, nullptr, z + D, st));
      }
      lin_bwd_w(T, convs[B.qkv], oa.QKVB, qa.QKVB, pa.N1, qa.N1, gout, st);
      conv_bwd_d(T, convs[B.qkv], oa.QKVB, qa.QKVB, epi_lin(T ? qa.N1B : oa.N1B, nullptr), st);
      VIT_RC(mdd_op_layernorm_bwd(VDT, M, D, vit.eps, P.X[l], T ? Q.X[l] : nullptr, oa.N1B, T ? qa.N1B : nullptr,
                                  th + B.ln1_w, tp(th_t + B.ln1_w), oa.X2B, T ? qa.X2B : nullptr,
                                  T ? nullptr : O.XB[l], T ? Q.XB[l] : nullptr, gp(B.ln1_w), gt(B.ln1_w),
                                  gp(B.ln1_b), gt(B.ln1_b), st));
      flush_w(st);
    }
    launch_vit_embed_bwd<AT>(O.XB[0], T ? Q.XB[0] : nullptr, T ? nullptr : O.VPEB, T ? Q.VPEB : nullptr,
                             gout + off_cls, gout + off_pos, N, Tk, D, st);
    lin_bwd_w(T, convs[pe_conv], O.VPEB, Q.VPEB, P.VCOL, nullptr, gout, st);
    if (dimage) {
      AT* cb = T ? Q.VCOLB : O.VCOLB;
      conv_bwd_d(T, convs[pe_conv], O.VPEB, Q.VPEB, epi_lin(cb, nullptr), st);
      launch_unpatchify_accum<AT>(dimage, cb, idx, coef, mul, N, S, vit.patch, st);
    }
    flush_w(st, MAIN_TAIL);
    join(st);
    POST_WALK("vit_backward");
    return 0;
  }
#undef VIT_RC

  // ------------------------------------------------------------------ image encoder: F / T-fwd
  int img_forward(bool T, int slot, const float* th, const float* th_t, const float* image,
                  const int64_t* idx, float* feat_out, hipStream_t st) override {
    CHECK_ARG(base, "workspace not bound");
    DevGuard guard(device_id);
    CHECK_ARG(slot >= 0 && slot < nslots, "slot out of range");
    select_prec(T, false);
    if (is_vit) return vit_forward(T, slot, th, th_t, image, idx, feat_out, st);
    ActSet& P = sets[slot]; ActSet& Q = tn;
    int nb = (int)blks.size();
    launch_ws_forward<AT>(d_descs, (int)descs.size(), total_rows, total_tiles, th, T ? th_t : nullptr, wf, wt,
                          wf_t, wt_t, st);
    if (!T) launch_img_gather_nhwc<AT>(P.X0, image, idx, N, 3, S, S, 8, st);
    se_prep(T, th, th_t, st);
    tf_split_active = T && use_side && MDD_TF_SPLIT;
    if (tf_split_active) { tf_side_prepass(P, Q, st); tf_wait(0, st); }
    const AT* in = P.X0; const AT* in_t = nullptr;
    for (int i = 0; i < 4; ++i) {
      bool last = i == 3;
      AT* C = last ? P.X[0] : P.Cs[i];  AT* C_t = last ? Q.X[0] : Q.Cs[i];
      AT* A = last ? P.A[0] : P.As[i];  AT* A_t = last ? Q.A[0] : Q.As[i];
      conv_fwd(T, convs[stem[i]], in, in_t, C, C_t, A, A_t, last ? blks[0].beta : 1.f, th, th_t, st);
      in = A; in_t = A_t;
    }
    const float ga = nf.attn_gain * nf.alpha;
    for (int b = 0; b < nb; ++b) {
      const Blk& B = blks[b]; BlockActs& pa = P.blk[b]; BlockActs& qa = Q.blk[b];
      const AT *x = P.X[b], *x_t = Q.X[b], *a = P.A[b], *a_t = Q.A[b];
      const AT *sc = x, *sc_t = x_t;
      bool sc_forked = false;
      tf_wait(1 + b, st);
      if (B.ds >= 0) {
        // the shortcut (avg-pool + 1x1 conv) only meets the residual branch at the SE apply: it runs on
        // the side stream, which is idle during forward passes (not in a split tangent pass: the side stream is busy
        // with the second sources then)
        hipStream_t ss = st;
        if (use_side && MDD_FWD_SHORTCUT_SIDE && !tf_split_active) { fork(st); ss = side; sc_forked = true; }
        const AT *din = a, *din_t = a_t;
        if (B.stride == 2) {
          launch_avgpool2<AT>(T ? qa.P : pa.P, T ? a_t : a, N, B.hin, B.hin, B.cin, 2, ss);
          din = pa.P; din_t = qa.P;
        }
        conv_fwd(T, convs[B.ds], din, din_t, pa.SC, qa.SC, nullptr, nullptr, 1.f, th, th_t, ss);
        sc = pa.SC; sc_t = qa.SC;
      }
      conv_fwd(T, convs[B.c1], a, a_t, pa.C1, qa.C1, pa.A1, qa.A1, 1.f, th, th_t, st);
      conv_fwd(T, convs[B.c2], pa.A1, qa.A1, pa.C2, qa.C2, pa.A2, qa.A2, 1.f, th, th_t, st);
      conv_fwd(T, convs[B.c2b], pa.A2, qa.A2, pa.C2b, qa.C2b, pa.A2b, qa.A2b, 1.f, th, th_t, st);
      int hw = B.hout * B.hout, c = B.se.c, rd = B.se.rd;
      // Squeeze-excite gate.  conv3 is a pointwise (linear) conv, so the pooled conv3 output is conv3 applied to
      // the pooled MID-channel activation: p = W3_hat q + b3 with q = mean_hw(A2b) -- a quarter of the bytes of
      // pooling C3, and the whole gate chain (pool, three small-batch linears) no longer waits for conv3: it runs
      // on the side stream beside it and meets the main stream at the SE apply.
      bool g_forked = false;
      {
        hipStream_t gs = st;
        if (use_side && MDD_SE_SIDE == 1) { fork(st); gs = side; sc_forked = true; }
        if (use_side && MDD_SE_SIDE == 2 && gstream) { fork_to(gstream, st); gs = gstream; g_forked = true; }
        se_prep_wait(gs);
        launch_pool_mean<AT>(T ? qa.q : pa.q, T ? qa.A2b : pa.A2b, N, hw, B.mid, gs);
        if (MDD_SE_W13) {
          // h = relu(q W13^T + b13)  (= relu(W1 (W3_hat q + b3) + b1): one K = mid product instead of two)
          launch_linear_fwd(pa.h, T ? qa.h : nullptr, pa.q, T ? qa.q : nullptr, se13[b].W, T ? se13[b].W_t : nullptr,
                            se13[b].b, T ? se13[b].b_t : nullptr, N, B.mid, rd, 1, lin_main, gs);
        } else {
          const ConvL& L3 = convs[B.c3];
          launch_linear_fwd_w<AT>(pa.p, T ? qa.p : nullptr, pa.q, T ? qa.q : nullptr, wf + L3.off_p,
                                  T ? wf_t + L3.off_p : nullptr, th + L3.off_b, T ? th_t + L3.off_b : nullptr, N,
                                  B.mid, c, 1.f, gs);
          launch_linear_fwd(pa.h, T ? qa.h : nullptr, pa.p, T ? qa.p : nullptr, th + B.se.off_w1,
                            T ? th_t + B.se.off_w1 : nullptr, th + B.se.off_b1,
                            T ? th_t + B.se.off_b1 : nullptr, N, c, rd, 1, lin_main, gs);
        }
        launch_linear_fwd(pa.gate, T ? qa.gate : nullptr, pa.h, T ? qa.h : nullptr, th + B.se.off_w2,
                          T ? th_t + B.se.off_w2 : nullptr, th + B.se.off_b2,
                          T ? th_t + B.se.off_b2 : nullptr, N, rd, c, 2, lin_main, gs);
      }
      conv_fwd(T, convs[B.c3], pa.A2b, qa.A2b, pa.C3, qa.C3, nullptr, nullptr, 1.f, th, th_t, st);
      bool lastb = b == nb - 1;
      if (sc_forked) join(st);
      if (g_forked) join_from(gstream, st);
      if (MDD_SE_W13) {
        // p = q W3_hat^T + b3 (the pooled conv3 output itself) is only needed by the weight gradients of the backward
        // passes: formed behind the gate (after the event the main stream waits for), off the dependent chain
        hipStream_t ps = g_forked ? gstream : st;
        const ConvL& L3 = convs[B.c3];
        launch_linear_fwd_w<AT>(pa.p, T ? qa.p : nullptr, pa.q, T ? qa.q : nullptr, wf + L3.off_p,
                                T ? wf_t + L3.off_p : nullptr, th + L3.off_b, T ? th_t + L3.off_b : nullptr, N,
                                B.mid, c, 1.f, ps);
      }
      launch_se_apply<AT>(pa.C3, T ? qa.C3 : nullptr, pa.gate, T ? qa.gate : nullptr, sc,
                          T ? sc_t : nullptr, P.X[b + 1], T ? Q.X[b + 1] : nullptr,
                          lastb ? nullptr : P.A[b + 1], (T && !lastb) ? Q.A[b + 1] : nullptr, ga,
                          lastb ? 1.f : blks[b + 1].beta, N, hw, c, st);
    }
    tf_wait(nb + 1, st);
    conv_fwd(T, convs[fin], P.X[nb], Q.X[nb], P.CF, Q.CF, nullptr, nullptr, 1.f, th, th_t, st);
    if (tf_split_active) { join(st); tf_split_active = false; }
    if (use_side && gstream) join_from(gstream, st);     // the trailing p products of the gate stream
    int hwf = xh[nb] * xh[nb];
    launch_final_pool<AT>(P.y, T ? feat_out : nullptr, P.CF, T ? Q.CF : nullptr, N, hwf, feat, st);
    if (!T && feat_out && feat_out != P.y)
      HIP_CHECK_RET(hipMemcpyAsync(feat_out, P.y, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    POST_WALK("img_forward");
    return 0;
  }

  // ------------------------------------------------------------------ image encoder: B / T-bwd
  int img_backward(bool T, int slot, const float* th, const float* th_t, const float* ybar_in,
                   const float* ybar_t_in, float* gout, float* dimage, const int64_t* idx,
                   const float* coef, float mul, bool repack, bool stash, hipStream_t st) override {
    CHECK_ARG(base, "workspace not bound");
    DevGuard guard(device_id);
    CHECK_ARG(slot >= 0 && slot < nslots, "slot out of range");
    select_prec(T, true);
    if (is_vit) return vit_backward(T, slot, th, th_t, ybar_in, ybar_t_in, gout, dimage, idx, coef, mul, repack, stash, st);
    ActSet& P = sets[slot]; ActSet& Q = tn;
    ActSet& O = (!T && !stash) ? tn : P;  // where primal backward signals are written
    int nb = (int)blks.size();
    if (repack) {
      launch_ws_forward<AT>(d_descs, (int)descs.size(), total_rows, total_tiles, th, T ? th_t : nullptr, wf, wt,
                            wf_t, wt_t, st);
      se_prep(T, th, th_t, st);
    }
    se_prep_wait(st);
    float* dw = O.dwf; float* dw_t = Q.dwf;
    // (no zero-fill of dwf: every conv's two-phase weight gradient overwrites its whole packed slice)
    HIP_CHECK_RET(hipMemsetAsync(gout, 0, P_img * 4, st));
    if (!T && ybar_in != O.yB)
      HIP_CHECK_RET(hipMemcpyAsync(O.yB, ybar_in, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    const float ga = nf.attn_gain * nf.alpha;
    int hwf = xh[nb] * xh[nb];
    launch_final_pool_bwd<AT>(O.CFB, T ? Q.CFB : nullptr, O.yB, T ? ybar_t_in : nullptr, P.CF,
                              T ? Q.CF : nullptr, N, hwf, feat, st);
    conv_bwd_w(T, convs[fin], O.CFB, Q.CFB, P.X[nb], Q.X[nb], dw, dw_t, gout, st);
    conv_bwd_d(T, convs[fin], O.CFB, Q.CFB, epi_lin(T ? Q.XB[nb] : O.XB[nb], nullptr), st);
    for (int b = nb - 1; b >= 0; --b) {
      const Blk& B = blks[b]; BlockActs& pa = P.blk[b]; BlockActs& qa = Q.blk[b];
      BlockActs& oa = O.blk[b];
      const AT *xb = O.XB[b + 1], *xb_t = Q.XB[b + 1];
      int hw = B.hout * B.hout, c = B.se.c, rd = B.se.rd;
      // squeeze-excite backward.  One pass over (xb, C3) gives the gate gradient AND conv3's output gradient
      // C3B = ga * gate * xb: with the pooled vector taken from the mid channels (img_forward) the pooled path's
      // gradient no longer goes through C3 -- it reaches conv3's weights as the small outer product pb^T q, its
      // bias as sum_n pb, and the residual branch as the per-image vector qb / hw added in conv3's data-gradient
      // epilogue.
      launch_se_gate_grad_c3b<AT>(oa.zB, T ? qa.zB : nullptr, oa.C3B, T ? qa.C3B : nullptr, xb, T ? xb_t : nullptr,
                                  pa.C3, T ? qa.C3 : nullptr, pa.gate, T ? qa.gate : nullptr, ga, N, hw, c, st);
      // conv3's weight gradient only needs C3B: release it (with what the previous block queued) now, so the
      // side stream works while the main stream walks the small-batch chain below
      conv_bwd_w(T, convs[B.c3], oa.C3B, qa.C3B, pa.A2b, qa.A2b, dw, dw_t, gout, st);
      flush_w(st);
      const ConvL& L3 = convs[B.c3];
      {
        const float *zB = oa.zB, *zB_t = T ? qa.zB : nullptr, *hh = pa.h, *hh_t = T ? qa.h : nullptr;
        const float *hB = oa.hB, *hB_t = T ? qa.hB : nullptr, *pp = pa.p, *pp_t = T ? qa.p : nullptr;
        const float *pB = oa.pB, *pB_t = T ? qa.pB : nullptr, *qq = pa.q, *qq_t = T ? qa.q : nullptr;
        float *gw2 = gout + B.se.off_w2, *gb2 = gout + B.se.off_b2, *gw1 = gout + B.se.off_w1,
              *gb1 = gout + B.se.off_b1, *gw3 = (T ? dw_t : dw) + L3.off_p, *gb3 = gout + L3.off_b;
        const int n_ = N, mid_ = B.mid;
        float *pBw = oa.pB, *pBw_t = T ? qa.pB : nullptr;
        const float *W1 = th + B.se.off_w1, *W1_t = T ? th_t + B.se.off_w1 : nullptr;
        wq.push_back([=](hipStream_t ws_) {
          // pb = hb W1 (gradient w.r.t. the pooled conv3 output): only the weight gradients need it
          if (MDD_SE_W13) launch_linear_dgrad(pBw, pBw_t, hB, hB_t, W1, W1_t, nullptr, n_, c, rd, wlin(), ws_);
          launch_linear_wgrad(gw2, gb2, zB, zB_t, hh, hh_t, n_, rd, c, wlin(), ws_);
          launch_linear_wgrad(gw1, gb1, hB, hB_t, pp, pp_t, n_, c, rd, wlin(), ws_);
          // after conv3's own weight gradient (queued above: same stream, in order) has overwritten its slice
          launch_linear_wgrad_accum(gw3, gb3, pB, pB_t, qq, qq_t, n_, mid_, c, ws_);
        });
      }
      launch_linear_dgrad(oa.hB, T ? qa.hB : nullptr, oa.zB, T ? qa.zB : nullptr,
                          th + B.se.off_w2, T ? th_t + B.se.off_w2 : nullptr, pa.h, N, rd, c,
                          lin_main, st);
      if (MDD_SE_W13) {
        // qb = hb W13 (= (hb W1) W3_hat): the gradient w.r.t. the mid-channel pooled vector in ONE K = rd product
        launch_linear_dgrad(oa.qB, T ? qa.qB : nullptr, oa.hB, T ? qa.hB : nullptr, se13[b].W, T ? se13[b].W_t : nullptr,
                            nullptr, N, B.mid, rd, lin_main, st);
      } else {
        launch_linear_dgrad(oa.pB, T ? qa.pB : nullptr, oa.hB, T ? qa.hB : nullptr, th + B.se.off_w1,
                            T ? th_t + B.se.off_w1 : nullptr, nullptr, N, c, rd, lin_main, st);
        launch_linear_fwd_w<AT>(oa.qB, T ? qa.qB : nullptr, oa.pB, T ? qa.pB : nullptr, wt + L3.off_p,
                                T ? wt_t + L3.off_p : nullptr, nullptr, nullptr, N, c, B.mid, 1.f, st);
      }
      // residual branch
      {
        ConvEpi e3 = epi_act(T, oa.A2bB, oa.C2bB, qa.C2bB, pa.C2b, qa.C2b, 1.f, nullptr, nullptr);
        e3.ib = oa.qB; e3.ib_t = qa.qB; e3.ib_mul = 1.f / (float)hw; e3.ib_hw = hw;
        conv_bwd_d(T, convs[B.c3], oa.C3B, qa.C3B, e3, st);
      }
      conv_bwd_w(T, convs[B.c2b], oa.C2bB, qa.C2bB, pa.A2, qa.A2, dw, dw_t, gout, st);
      conv_bwd_d(T, convs[B.c2b], oa.C2bB, qa.C2bB,
                 epi_act(T, oa.A2B, oa.C2B, qa.C2B, pa.C2, qa.C2, 1.f, nullptr, nullptr), st);
      conv_bwd_w(T, convs[B.c2], oa.C2B, qa.C2B, pa.A1, qa.A1, dw, dw_t, gout, st);
      conv_bwd_d(T, convs[B.c2], oa.C2B, qa.C2B,
                 epi_act(T, oa.A1B, oa.C1B, qa.C1B, pa.C1, qa.C1, 1.f, nullptr, nullptr), st);
      conv_bwd_w(T, convs[B.c1], oa.C1B, qa.C1B, P.A[b], Q.A[b], dw, dw_t, gout, st);
      // shortcut branch
      const AT* add1 = nullptr;
      if (B.ds >= 0) {
        const AT* dsin = B.stride == 2 ? pa.P : P.A[b];
        const AT* dsin_t = B.stride == 2 ? qa.P : Q.A[b];
        conv_bwd_w(T, convs[B.ds], xb, xb_t, dsin, dsin_t, dw, dw_t, gout, st);
        conv_bwd_d(T, convs[B.ds], xb, xb_t, epi_lin(dsS, nullptr), st);
        if (B.stride == 2) {
          launch_avgpool2_bwd<AT>(dsF, dsS, N, B.hin, B.hin, B.cin, 2, st);
          add1 = dsF;
        } else {
          add1 = dsS;
        }
      }
      // conv1 dgrad + pre-activation chain rule + identity shortcut:  XB[b] = beta*silu'(X[b])*Abar (+ XB[b+1])
      conv_bwd_d(T, convs[B.c1], oa.C1B, qa.C1B,
                 epi_act(T, oa.AinB, O.XB[b], Q.XB[b], P.X[b], Q.X[b], B.beta, add1,
                         B.ds >= 0 ? nullptr : (T ? xb_t : xb)), st);
      if (b == 0) flush_w(st);
    }
    // stem (conv4 output is the raw stream X[0]; its grad is XB[0])
    conv_bwd_w(T, convs[stem[3]], O.XB[0], Q.XB[0], P.As[2], Q.As[2], dw, dw_t, gout, st);
    flush_w(st);   // stem: release each weight gradient at once (they are the tail of the pass)
    conv_bwd_d(T, convs[stem[3]], O.XB[0], Q.XB[0],
               epi_act(T, O.AsB[2], O.CsB[2], Q.CsB[2], P.Cs[2], Q.Cs[2], 1.f, nullptr, nullptr), st);
    if (MDD_TAIL_MAIN < 3) {
      conv_bwd_w(T, convs[stem[2]], O.CsB[2], Q.CsB[2], P.As[1], Q.As[1], dw, dw_t, gout, st);
      flush_w(st);   // stem: release each weight gradient at once (they are the tail of the pass)
    }
    conv_bwd_d(T, convs[stem[2]], O.CsB[2], Q.CsB[2],
               epi_act(T, O.AsB[1], O.CsB[1], Q.CsB[1], P.Cs[1], Q.Cs[1], 1.f, nullptr, nullptr), st);
    conv_bwd_d(T, convs[stem[1]], O.CsB[1], Q.CsB[1],
               epi_act(T, O.AsB[0], O.CsB[0], Q.CsB[0], P.Cs[0], Q.Cs[0], 1.f, nullptr, nullptr), st);
    if (dimage) {
      const ConvL& L0 = convs[stem[0]];
      // 3 real input channels: the image gradient is accumulated directly by a pixel-parallel kernel
      // (tangent pass: dy_t (*) w + dy (*) w_t); other stem widths fall back to the implicit GEMM + scatter
      const bool fused = L0.k == 3 && L0.stride == 2 && L0.pad == 1 && L0.cin == 3 && L0.groups == 1 &&
                         launch_stem_dgrad_image<AT>(dimage, T ? Q.CsB[0] : O.CsB[0], wt + L0.off_p,
                                                     T ? O.CsB[0] : (const AT*)nullptr,
                                                     T ? wt_t + L0.off_p : (const AT*)nullptr, idx, coef, mul, N,
                                                     S, L0.cout, L0.cin_pad, st);
      if (!fused) {
        AT* x0b = T ? Q.X0B : O.X0B;
        conv_bwd_d(T, L0, O.CsB[0], Q.CsB[0], epi_lin(x0b, nullptr), st);
        launch_img_scatter_grad<AT>(dimage, x0b, idx, coef, mul, N, 3, S, S, 8, st);
      }
    }
    // the last two weight gradients run on the main stream: at this point the side stream still has
    // the high-resolution layers' weight gradients queued and would otherwise be waited for
    if (MDD_TAIL_MAIN >= 3) conv_bwd_w(T, convs[stem[2]], O.CsB[2], Q.CsB[2], P.As[1], Q.As[1], dw, dw_t, gout, st);
    conv_bwd_w(T, convs[stem[1]], O.CsB[1], Q.CsB[1], P.As[0], Q.As[0], dw, dw_t, gout, st);
    conv_bwd_w(T, convs[stem[0]], O.CsB[0], Q.CsB[0], P.X0, nullptr, dw, dw_t, gout, st);
    flush_w(st, MAIN_TAIL);
    join(st);
    launch_ws_backward(d_descs, (int)descs.size(), total_rows, th, T ? th_t : nullptr, dw,
                       T ? dw_t : nullptr, gout, st);
    POST_WALK("img_backward");
    return 0;
  }

  // ------------------------------------------------------------------ text projection head
  int txt_forward(bool T, int slot, const float* th, const float* th_t, const float* text,
                  const int64_t* idx, const float* mask, float* feat_out, hipStream_t st) override {
    CHECK_ARG(base, "workspace not bound");
    DevGuard guard(device_id);
    CHECK_ARG(slot >= 0 && slot < nslots, "slot out of range");
    ActSet& P = sets[slot]; ActSet& Q = tn;
    if (!T) launch_gather_rows(P.tx, text, idx, N, Dt, st);
    auto tt = [&](int64_t o) { return T ? th_t + o : nullptr; };
    launch_linear_fwd(P.tp, T ? Q.tp : nullptr, P.tx, nullptr, th + t_pw, tt(t_pw), th + t_pb,
                      tt(t_pb), N, Dt, feat, 0, lin_txt, st);
    launch_gelu(P.tg, T ? Q.tg : nullptr, P.tp, T ? Q.tp : nullptr, (int64_t)N * feat, st);
    launch_linear_fwd(P.tf, T ? Q.tf : nullptr, P.tg, T ? Q.tg : nullptr, th + t_fw, tt(t_fw),
                      th + t_fb, tt(t_fb), N, feat, feat, 0, lin_txt, st);
    // the dropout mask of this slot is kept for the backward / tangent passes
    if (!T) slot_mask_[slot] = mask;
    launch_ln_fwd(P.ty, T ? feat_out : nullptr, P.tr, T ? Q.tr : nullptr, P.tf, T ? Q.tf : nullptr,
                  slot_mask_[slot], P.tp, T ? Q.tp : nullptr, th + t_lw, tt(t_lw), th + t_lb,
                  tt(t_lb), N, feat, 1e-5f, st);
    if (!T && feat_out && feat_out != P.ty)
      HIP_CHECK_RET(hipMemcpyAsync(feat_out, P.ty, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    POST_WALK("txt_forward");
    return 0;
  }
  std::vector<const float*> slot_mask_ = std::vector<const float*>(65, nullptr);

  // ---- squeeze-excite shortcut (round 3).  h_pre = W1 (W3_hat q + b3) + b1 = W13 q + b13 with W13 = W1 W3_hat
  // [rd, mid]: the dependent chain of a block is pool -> (K = mid) -> (K = rd) instead of pool -> (K = mid) -> (K = C)
  // -> (K = rd), and in the backward passes qb = hb W13 replaces pb = hb W1, qb = pb W3_hat on the critical path (pb
  // and p are still formed, off that path, for the weight gradients).  W13 / b13 depend on theta only: they are built
  // on the gate stream at the start of every pass that (re)packs the weights, beside the stem.
  struct Se13 { float *W = nullptr, *W_t = nullptr, *b = nullptr, *b_t = nullptr; };
  std::vector<Se13> se13;
  hipEvent_t se13_ev = nullptr;
  void se_prep(bool T, const float* th, const float* th_t, hipStream_t st) {
    if (!MDD_SE_W13) return;
    hipStream_t s2 = (use_side && gstream) ? gstream : st;
    fork_to(s2, st);                   // after ws_forward: the packed (transposed) conv3 weights exist
    for (size_t b = 0; b < blks.size(); ++b) {
      const Blk& B = blks[b]; const ConvL& L3 = convs[B.c3];
      const int C = B.se.c, rd = B.se.rd, mid = B.mid;
      const float *W1 = th + B.se.off_w1, *b1 = th + B.se.off_b1, *b3 = th + L3.off_b;
      // primal always (a tangent pass follows a repack at ITS theta_k); rows of W1 play the role of the batch
      launch_linear_fwd_w<AT>(se13[b].W, nullptr, W1, nullptr, wt + L3.off_p, nullptr, nullptr, nullptr, rd, C, mid, 1.f, s2);
      launch_matvec_bias(se13[b].b, nullptr, W1, nullptr, b3, nullptr, b1, nullptr, rd, C, s2);
      if (T) {
        const float *W1_t = th_t + B.se.off_w1, *b1_t = th_t + B.se.off_b1, *b3_t = th_t + L3.off_b;
        launch_linear_fwd_w<AT>(se13[b].W, se13[b].W_t, W1, W1_t, wt + L3.off_p, wt_t + L3.off_p, nullptr, nullptr, rd, C,
                                mid, 1.f, s2);
        launch_matvec_bias(nullptr, se13[b].b_t, W1, W1_t, b3, b3_t, nullptr, b1_t, rd, C, s2);
      }
    }
    if (s2 != st) {
      if (!se13_ev) ck(hipEventCreateWithFlags(&se13_ev, hipEventDisableTiming), "hipEventCreateWithFlags");
      ck(hipEventRecord(se13_ev, s2), "hipEventRecord");
    }
  }
  void se_prep_wait(hipStream_t s) {   // a stream about to read W13 / b13
    if (se13_ev && use_side && gstream && s != gstream) ck(hipStreamWaitEvent(s, se13_ev, 0), "hipStreamWaitEvent");
  }

  int txt_backward(bool T, int slot, const float* th, const float* th_t, const float* ybar_in,
                   const float* ybar_t_in, float* gout, float* dtext, const int64_t* idx,
                   const float* coef, float mul, bool stash, hipStream_t st) override {
    CHECK_ARG(base, "workspace not bound");
    DevGuard guard(device_id);
    CHECK_ARG(slot >= 0 && slot < nslots, "slot out of range");
    ActSet& P = sets[slot]; ActSet& Q = tn;
    ActSet& O = (!T && !stash) ? tn : P;
    auto tt = [&](int64_t o) { return T ? th_t + o : nullptr; };
    if (!T && ybar_in != O.tyB)
      HIP_CHECK_RET(hipMemcpyAsync(O.tyB, ybar_in, (size_t)N * feat * 4, hipMemcpyDeviceToDevice, st));
    launch_ln_bwd(O.trB, T ? Q.trB : nullptr, O.tfB, T ? Q.tfB : nullptr, gout + t_lw, gout + t_lb,
                  ln_stats, O.tyB, T ? ybar_t_in : nullptr, P.tr, T ? Q.tr : nullptr,
                  slot_mask_[slot], th + t_lw, tt(t_lw), N, feat, 1e-5f, st);
    launch_linear_wgrad(gout + t_fw, gout + t_fb, O.tfB, T ? Q.tfB : nullptr, P.tg,
                        T ? Q.tg : nullptr, N, feat, feat, lin_txt, st);
    launch_linear_dgrad(O.tgB, T ? Q.tgB : nullptr, O.tfB, T ? Q.tfB : nullptr, th + t_fw, tt(t_fw),
                        nullptr, N, feat, feat, lin_txt, st);
    launch_gelu_bwd(O.tpB, T ? Q.tpB : nullptr, O.trB, T ? Q.trB : nullptr, O.tgB,
                    T ? Q.tgB : nullptr, P.tp, T ? Q.tp : nullptr, (int64_t)N * feat, st);
    launch_linear_wgrad(gout + t_pw, gout + t_pb, O.tpB, T ? Q.tpB : nullptr, P.tx, nullptr, N, Dt,
                        feat, lin_txt, st);
    if (dtext) {
      float* xb = T ? Q.txB : O.txB;
      launch_linear_dgrad(O.txB, T ? Q.txB : nullptr, O.tpB, T ? Q.tpB : nullptr, th + t_pw,
                          tt(t_pw), nullptr, N, Dt, feat, lin_txt, st);
      launch_scatter_rows_axpy(dtext, xb, idx, coef, mul, N, Dt, st);
    }
    POST_WALK("txt_backward");
    return 0;
  }

  int contrastive(bool T, const float* x, const float* y, const float* x_t, const float* y_t,
                  const float* scale_dev, float scale_const, float* loss, float* xb, float* yb,
                  float* sb, hipStream_t st) override {
    CHECK_ARG(base, "workspace not bound");
    DevGuard guard(device_id);
    LossWork w = loss_work_carve(lossw, N, feat);
    if (!T) launch_contrastive(w, loss, xb, yb, sb, nullptr, nullptr, nullptr, x, y, nullptr,
                               nullptr, scale_dev, scale_const, N, feat, st);
    else launch_contrastive(w, nullptr, nullptr, nullptr, nullptr, xb, yb, sb, x, y, x_t, y_t,
                            scale_dev, scale_const, N, feat, st);
    POST_WALK("contrastive");
    return 0;
  }

  // ------------------------------------------------------------------ whole outer iteration
#if MDD_GRAPH
  // Experiment (profiles/r02_experiments.md): the whole outer iteration captured into ONE hipGraph (three
  // streams, their event edges included) and replayed while the call's arguments stay the same.
  hipGraphExec_t gexec = nullptr;
  std::vector<uint64_t> gkey;
  int unrolled_match(const mdd_iter_args* a, hipStream_t st) override {
    if (!st || prof_on) return unrolled_match_enqueue(a, st);     // the legacy default stream cannot be captured
    DevGuard guard(device_id);
    std::vector<uint64_t> key;
    const void* ptrs[] = {a->image_syn, a->text_syn, a->lr_img, a->lr_txt, a->theta0_img, a->theta0_txt,
                          a->target_img, a->target_txt, a->perms, a->drop_masks, a->grad_image_syn,
                          a->grad_text_syn, a->grad_lr, a->losses, (const void*)st};
    for (const void* q : ptrs) key.push_back((uint64_t)(uintptr_t)q);
    key.push_back((uint64_t)a->syn_steps); key.push_back((uint64_t)a->use_lr_as_scale);
    uint32_t lsb; memcpy(&lsb, &a->logit_scale_const, 4); key.push_back(lsb);
    if (!gexec || key != gkey) {
      if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
      HIP_CHECK_RET(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      int rc = unrolled_match_enqueue(a, st);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(st, &g);
      if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      HIP_CHECK_RET(e);
      e = hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      HIP_CHECK_RET(e);
      gkey = key;
    }
    HIP_CHECK_RET(hipGraphLaunch(gexec, st));
    return 0;
  }
  int unrolled_match_enqueue(const mdd_iter_args* a, hipStream_t st) {
#else
  int unrolled_match(const mdd_iter_args* a, hipStream_t st) override {
#endif
    CHECK_ARG(base, "workspace not bound");
    DevGuard guard(device_id);
    CHECK_ARG(a->syn_steps >= 1 && a->syn_steps <= K, "syn_steps exceeds the engine's slots");
    const int Ks = a->syn_steps;
    const float* scale_dev = a->use_lr_as_scale ? a->lr_img : nullptr;
    const int64_t nfeat = (int64_t)N * feat;
    hipStream_t ts = (use_side && tside) ? tside : st;
    int rc;
    std::vector<const float*> tI(Ks + 1), tT(Ks + 1);
    tI[0] = a->theta0_img; tT[0] = a->theta0_txt;
    for (int k = 1; k <= Ks; ++k) { tI[k] = thI[k]; tT[k] = thT[k]; }
    HIP_CHECK_RET(hipMemsetAsync(dsc, 0, 16 * sizeof(double), st));
    // ---- unrolled student training (distill.py:509-583)
    // primal forward + contrastive head + inner gradient of step k at theta_k, activations into `slot`
    auto primal_step = [&](int k, int slot) -> int {
      const int64_t* idx = a->perms ? a->perms + (int64_t)k * N : nullptr;
      const float* mask = a->drop_masks ? a->drop_masks + (int64_t)k * nfeat : nullptr;
      ActSet& Pk = sets[slot];
      // the text projection runs on its own stream beside the image encoder; they meet at the
      // contrastive head
      fork_to(ts, st);
      if ((rc = txt_forward(false, slot, tT[k], nullptr, a->text_syn, idx, mask, nullptr, ts))) return rc;
      if ((rc = img_forward(false, slot, tI[k], nullptr, a->image_syn, idx, nullptr, st))) return rc;
      join_from(ts, st);
      if ((rc = contrastive(false, Pk.y, Pk.ty, nullptr, nullptr, scale_dev,
                            a->logit_scale_const, a->losses + 3 + k, Pk.yB, Pk.tyB, sbar, st)))
        return rc;
      fork_to(ts, st);
      if ((rc = txt_backward(false, slot, tT[k], nullptr, Pk.tyB, nullptr, gT[k], nullptr, nullptr,
                             nullptr, 0.f, true, ts))) return rc;
      if ((rc = img_backward(false, slot, tI[k], nullptr, Pk.yB, nullptr, gI[k], nullptr, nullptr,
                             nullptr, 0.f, false, true, st))) return rc;
      return 0;
    };
    int shared_holds = -1;   // which step's activations the shared (recompute) slot currently holds
    for (int k = 0; k < Ks; ++k) {
      const int slot = slot_of(k);
      if ((rc = primal_step(k, slot))) return rc;
      if (slot == keep) shared_holds = k;
      launch_axpy_out(thT[k + 1], tT[k], gT[k], a->lr_txt, -1.f, P_txt, ts);
      launch_axpy_out(thI[k + 1], tI[k], gI[k], a->lr_img, -1.f, P_img, st);
    }
    join_from(ts, st);
    // ---- trajectory-matching loss (distill.py:584-598)
    launch_sqdist(tI[Ks], a->target_img, dsc + 0, P_img, st);
    launch_sqdist(a->theta0_img, a->target_img, dsc + 1, P_img, st);
    launch_sqdist(tT[Ks], a->target_txt, dsc + 2, P_txt, st);
    launch_sqdist(a->theta0_txt, a->target_txt, dsc + 3, P_txt, st);
    launch_match_finalize(dsc, a->losses, st);
    // ---- outer backward (distill.py:606) as an explicit reverse sweep
    launch_lambda_init(lamI, tI[Ks], a->target_img, dsc + 1, P_img, st);
    launch_lambda_init(lamT, tT[Ks], a->target_txt, dsc + 3, P_txt, st);
    HIP_CHECK_RET(hipMemsetAsync(a->grad_image_syn, 0, (size_t)cfg.num_queries * 3 * S * S * 4, st));
    HIP_CHECK_RET(hipMemsetAsync(a->grad_text_syn, 0, (size_t)cfg.num_queries * Dt * 4, st));
    for (int k = Ks - 1; k >= 0; --k) {
      const int64_t* idx = a->perms ? a->perms + (int64_t)k * N : nullptr;
      const int slot = slot_of(k);
      if (slot == keep && keep < K && shared_holds != k) {
        // stash policy: this step's activations were not kept -- recompute its forward pass and inner
        // gradient at the kept theta_k into the shared slot (3 of the 12 contractions this step then costs)
        join_from(ts, st);
        if ((rc = primal_step(k, slot))) return rc;
        shared_holds = k;
      }
      ActSet& Pk = sets[slot];
      fork_to(ts, st);
      launch_dot(gT[k], lamT, dsc + 5, -1.0, P_txt, ts);
      launch_scale_out(nuT, lamT, a->lr_txt, 1.f, P_txt, ts);
      if ((rc = txt_forward(true, slot, tT[k], nuT, nullptr, nullptr, nullptr, fy_t, ts))) return rc;
      launch_dot(gI[k], lamI, dsc + 4, -1.0, P_img, st);   // d/d lr_img  -= <g_k, lambda>
      launch_scale_out(nuI, lamI, a->lr_img, 1.f, P_img, st);  // direction v = lr * lambda
      if ((rc = img_forward(true, slot, tI[k], nuI, nullptr, nullptr, fx_t, st))) return rc;
      join_from(ts, st);
      if ((rc = contrastive(true, Pk.y, Pk.ty, fx_t, fy_t, scale_dev, a->logit_scale_const,
                            nullptr, xbar_t, ybar_t, sbar_t, st))) return rc;
      fork_to(ts, st);
      if ((rc = txt_backward(true, slot, tT[k], nuT, nullptr, ybar_t, hT, a->grad_text_syn, idx, nullptr,
                             -1.f, true, ts))) return rc;
      launch_sub_inplace(lamT, hT, P_txt, ts);
      if ((rc = img_backward(true, slot, tI[k], nuI, nullptr, xbar_t, hI, a->grad_image_syn, idx, nullptr,
                             -1.f, false, true, st))) return rc;
      if (a->use_lr_as_scale) launch_accum_f2d(dsc + 4, sbar_t, -1.0, st);
      launch_sub_inplace(lamI, hI, P_img, st);
    }
    join_from(ts, st);
    launch_d2f(a->grad_lr, dsc + 4, 1.f, 0, 2, st);
    POST_WALK("unrolled_match");
    return 0;
  }
};

}  // namespace

// ========================================================================================== C ABI
extern "C" {

const char* mdd_last_error(void) { return g_err.c_str(); }
int mdd_version(void) { return MDD_ABI_VERSION; }
int mdd_set_pipe_kernels(int enable) {
  const int prev = pipe_kernels_enabled() ? 1 : 0;
  set_pipe_kernels(enable != 0);
  return prev;
}

int mdd_engine_create(const mdd_config* cfg, mdd_engine** out) {
  CHECK_ARG(cfg && out && cfg->variant, "null config");
  *out = nullptr;
  if (cfg->dtype == MDD_DTYPE_F32) {
    auto* e = new Eng<float>();
    int rc = e->build(*cfg);
    if (rc) { delete e; return rc; }
    *out = e;
  } else if (cfg->dtype == MDD_DTYPE_BF16) {
    auto* e = new Eng<bf16>();
    int rc = e->build(*cfg);
    if (rc) { delete e; return rc; }
    *out = e;
  } else if (cfg->dtype == MDD_DTYPE_BF16X2 || cfg->dtype == MDD_DTYPE_F32_BF16OPS) {
    auto* e = new Eng<float>();
    e->prec = cfg->dtype == MDD_DTYPE_BF16X2 ? 1 : 2;
    int rc = e->build(*cfg);
    if (rc) { delete e; return rc; }
    *out = e;
  } else {
    return mdd_set_error_msg(2, "mdd: invalid argument: dtype");
  }
  return 0;
}
void mdd_engine_destroy(mdd_engine* e) { delete e; }
int64_t mdd_engine_workspace_bytes(const mdd_engine* e) { return e ? e->workspace_bytes() : -1; }
int mdd_engine_bind_workspace(mdd_engine* e, void* ws, int64_t bytes, void* stream) {
  CHECK_ARG(e, "null engine");
  return e->bind(ws, bytes, (hipStream_t)stream);
}
int64_t mdd_engine_param_numel(const mdd_engine* e, int which) {
  return !e ? -1 : (which == 0 ? e->P_img : e->P_txt);
}
int mdd_engine_param_count(const mdd_engine* e, int which) {
  return !e ? -1 : (int)(which == 0 ? e->pimg.size() : e->ptxt.size());
}
int mdd_engine_param_info(const mdd_engine* e, int which, int index, char* name, int name_cap,
                          int64_t* shape4, int* ndim, int64_t* offset) {
  CHECK_ARG(e, "null engine");
  const auto& tab = which == 0 ? e->pimg : e->ptxt;
  CHECK_ARG(index >= 0 && index < (int)tab.size(), "param index");
  const ParamInfo& p = tab[index];
  if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (shape4) memcpy(shape4, p.shape, sizeof p.shape);
  if (ndim) *ndim = p.ndim;
  if (offset) *offset = p.offset;
  return 0;
}
int mdd_engine_feature_dim(const mdd_engine* e) { return e ? e->feat : -1; }
int mdd_engine_find_buffer(const mdd_engine* e, const char* name, int slot, int64_t* byte_offset,
                           int64_t* elems, int* is_float32) {
  CHECK_ARG(e && name, "null");
  for (const auto& b : e->names)
    if (b.slot == slot && b.name == name) {
      if (byte_offset) *byte_offset = b.off;
      if (elems) *elems = b.elems;
      if (is_float32) *is_float32 = b.f32 ? 1 : 0;
      return 0;
    }
  return mdd_set_error_msg(3, "mdd: buffer not found");
}

int mdd_img_forward(mdd_engine* e, int slot, const float* th, const float* image,
                    const int64_t* idx, float* feat_out, void* stream) {
  CHECK_ARG(e && th && image, "null pointer");
  return e->img_forward(false, slot, th, nullptr, image, idx, feat_out, (hipStream_t)stream);
}
int mdd_img_backward(mdd_engine* e, int slot, const float* th, const float* fb, float* gout,
                     float* dimage, const int64_t* idx, const float* coef, float mul, int stash,
                     void* stream) {
  CHECK_ARG(e && th && fb && gout, "null pointer");
  return e->img_backward(false, slot, th, nullptr, fb, nullptr, gout, dimage, idx, coef, mul, true,
                         stash != 0, (hipStream_t)stream);
}
int mdd_img_tangent_forward(mdd_engine* e, int slot, const float* th, const float* thd,
                            float* fdot, void* stream) {
  CHECK_ARG(e && th && thd && fdot, "null pointer");
  return e->img_forward(true, slot, th, thd, nullptr, nullptr, fdot, (hipStream_t)stream);
}
int mdd_img_tangent_backward(mdd_engine* e, int slot, const float* th, const float* thd,
                             const float* fbd, float* hout, float* dimage, const int64_t* idx,
                             const float* coef, float mul, void* stream) {
  CHECK_ARG(e && th && thd && fbd && hout, "null pointer");
  return e->img_backward(true, slot, th, thd, nullptr, fbd, hout, dimage, idx, coef, mul, true,
                         true, (hipStream_t)stream);
}
int mdd_txt_forward(mdd_engine* e, int slot, const float* th, const float* text,
                    const int64_t* idx, const float* mask, float* feat_out, void* stream) {
  CHECK_ARG(e && th && text, "null pointer");
  return e->txt_forward(false, slot, th, nullptr, text, idx, mask, feat_out, (hipStream_t)stream);
}
int mdd_txt_backward(mdd_engine* e, int slot, const float* th, const float* fb, float* gout,
                     float* dtext, const int64_t* idx, const float* coef, float mul, int stash,
                     void* stream) {
  CHECK_ARG(e && th && fb && gout, "null pointer");
  return e->txt_backward(false, slot, th, nullptr, fb, nullptr, gout, dtext, idx, coef, mul,
                         stash != 0, (hipStream_t)stream);
}
int mdd_txt_tangent_forward(mdd_engine* e, int slot, const float* th, const float* thd,
                            float* fdot, void* stream) {
  CHECK_ARG(e && th && thd && fdot, "null pointer");
  return e->txt_forward(true, slot, th, thd, nullptr, nullptr, nullptr, fdot, (hipStream_t)stream);
}
int mdd_txt_tangent_backward(mdd_engine* e, int slot, const float* th, const float* thd,
                             const float* fbd, float* hout, float* dtext, const int64_t* idx,
                             const float* coef, float mul, void* stream) {
  CHECK_ARG(e && th && thd && fbd && hout, "null pointer");
  return e->txt_backward(true, slot, th, thd, nullptr, fbd, hout, dtext, idx, coef, mul, true,
                         (hipStream_t)stream);
}
int mdd_contrastive(mdd_engine* e, const float* x, const float* y, const float* scale_dev,
                    float scale_const, float* loss, float* xbar, float* ybar, float* sbar,
                    void* stream) {
  CHECK_ARG(e && x && y && loss && xbar && ybar && sbar, "null pointer");
  return e->contrastive(false, x, y, nullptr, nullptr, scale_dev, scale_const, loss, xbar, ybar,
                        sbar, (hipStream_t)stream);
}
int mdd_contrastive_tangent(mdd_engine* e, const float* x, const float* y, const float* xd,
                            const float* yd, const float* scale_dev, float scale_const,
                            float* xbd, float* ybd, float* sbd, void* stream) {
  CHECK_ARG(e && x && y && xd && yd && xbd && ybd && sbd, "null pointer");
  return e->contrastive(true, x, y, xd, yd, scale_dev, scale_const, nullptr, xbd, ybd, sbd,
                        (hipStream_t)stream);
}
int64_t mdd_op_contrastive_workspace_floats(int n, int d) { return loss_work_floats(n, d); }
int mdd_op_contrastive(int n, int d, const float* x, const float* y, const float* xd, const float* yd,
                       const float* scale_dev, float scale_const, float* work, float* loss, float* xbar,
                       float* ybar, float* sbar, void* stream) {
  CHECK_ARG(n > 0 && d > 0 && x && y && work && xbar && ybar && sbar, "null pointer / empty problem");
  CHECK_ARG((xd == nullptr) == (yd == nullptr), "x_dot and y_dot come together");
  CHECK_ARG(xd || loss, "loss output is null");
  LossWork w = loss_work_carve(work, n, d);
  if (!xd) launch_contrastive(w, loss, xbar, ybar, sbar, nullptr, nullptr, nullptr, x, y, nullptr, nullptr,
                              scale_dev, scale_const, n, d, (hipStream_t)stream);
  else launch_contrastive(w, nullptr, nullptr, nullptr, nullptr, xbar, ybar, sbar, x, y, xd, yd, scale_dev,
                          scale_const, n, d, (hipStream_t)stream);
  POST_LAUNCH("op_contrastive");
  return 0;
}
int mdd_flat_axpy(float* out, const float* x, const float* g, const float* lr, float sign,
                  int64_t n, void* stream) {
  CHECK_ARG(out && x && g && lr && n >= 0, "null pointer");
  launch_axpy_out(out, x, g, lr, sign, n, (hipStream_t)stream);
  POST_LAUNCH("axpy");
  return 0;
}
int mdd_flat_sqdist(const float* a, const float* b, double* out, int64_t n, void* stream) {
  CHECK_ARG(a && b && out && n >= 0, "null pointer");
  launch_sqdist(a, b, out, n, (hipStream_t)stream);
  POST_LAUNCH("sqdist");
  return 0;
}
int mdd_flat_sgd_momentum(float* p, const float* g, float* buf, float lr, float mom, int first,
                          int64_t n, void* stream) {
  CHECK_ARG(p && g && buf && n >= 0, "null pointer");
  launch_sgd_momentum(p, g, buf, lr, mom, first, n, nullptr, (hipStream_t)stream);
  POST_LAUNCH("sgd");
  return 0;
}
int mdd_flat_sgd_momentum_guarded(float* p, const float* g, float* buf, float lr, float mom, int first,
                                  int64_t n, const float* skip_flag, void* stream) {
  CHECK_ARG(p && g && buf && skip_flag && n >= 0, "null pointer");
  launch_sgd_momentum(p, g, buf, lr, mom, first, n, skip_flag, (hipStream_t)stream);
  POST_LAUNCH("sgd_guarded");
  return 0;
}
int mdd_engine_set_pass_precision(mdd_engine* e, int fwd, int bwd, int tan_fwd, int tan_bwd) {
  CHECK_ARG(e, "null engine");
  const int q[4] = {fwd, bwd, tan_fwd, tan_bwd};
  return e->set_pass_prec(q);
}
int mdd_engine_profile(mdd_engine* e, int enable) {
  CHECK_ARG(e, "null engine");
  e->profile_enable(enable != 0);
  return 0;
}
int mdd_engine_profile_dump(mdd_engine* e, const char* path) {
  CHECK_ARG(e && path, "profile_dump");
  return e->profile_dump(path);
}
int mdd_engine_profile_read(mdd_engine* e, int kind, double* out4) {
  CHECK_ARG(e && out4 && kind >= 0 && kind < 5, "profile_read");
  return e->profile_read(kind, out4);
}
int mdd_unrolled_match(mdd_engine* e, const mdd_iter_args* a, void* stream) {
  CHECK_ARG(e && a, "null pointer");
  CHECK_ARG(a->image_syn && a->text_syn && a->lr_img && a->lr_txt && a->theta0_img &&
                a->theta0_txt && a->target_img && a->target_txt && a->grad_image_syn &&
                a->grad_text_syn && a->grad_lr && a->losses, "null pointer in mdd_iter_args");
  return e->unrolled_match(a, (hipStream_t)stream);
}

static ConvGeom op_geom(int transposed, int nimg, int hin, int win, int cin, int cout, int k,
                        int stride, int pad, int groups) {
  int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  ConvGeom g;
  g.nimg = nimg; g.groups = groups; g.k = k; g.stride = stride; g.pad = pad; g.transposed = transposed;
  if (!transposed) {
    g.ha = hin; g.wa = win; g.ca_tot = cin; g.ho = hout; g.wo = wout; g.co_tot = cout;
    g.kc = cin / groups; g.nc = cout / groups;
  } else {
    g.ha = hout; g.wa = wout; g.ca_tot = cout; g.ho = hin; g.wo = win; g.co_tot = cin;
    g.kc = cout / groups; g.nc = cin / groups;
  }
  return g;
}
int mdd_op_conv2d(int dtype, int transposed, int nimg, int hin, int win, int cin, int cout, int k,
                  int stride, int pad, int groups, const void* a, const void* w, const float* bias,
                  void* out, void* stream) {
  CHECK_ARG(a && w && out, "null pointer");
  CHECK_ARG(dtype >= 0 && dtype <= 3, "dtype");
  CHECK_ARG(stride == 1 || stride == 2, "stride must be 1 or 2");
  ConvGeom g = op_geom(transposed, nimg, hin, win, cin, cout, k, stride, pad, groups);
  ConvEpi e; memset(&e, 0, sizeof e);
  e.mode = transposed ? EPI_BWD_LIN : EPI_FWD; e.bias = bias; e.out_raw = out; e.beta = 1.f;
  if (dtype == MDD_DTYPE_BF16X2) g.prec = 1;
  if (dtype == MDD_DTYPE_F32_BF16OPS) g.prec = 2;
  if (dtype != MDD_DTYPE_BF16)
    launch_conv_gemm<float>(g, (const float*)a, (const float*)w, nullptr, nullptr, e, (hipStream_t)stream);
  else
    launch_conv_gemm<bf16>(g, (const bf16*)a, (const bf16*)w, nullptr, nullptr, e, (hipStream_t)stream);
  POST_LAUNCH("op_conv2d");
  return 0;
}
int mdd_op_conv2d_wgrad(int dtype, int nimg, int hin, int win, int cin, int cout, int k, int stride,
                        int pad, int groups, const void* dy, const void* x, float* dw, float* db,
                        void* stream) {
  CHECK_ARG(dy && x && dw, "null pointer");
  CHECK_ARG(dtype >= 0 && dtype <= 3, "dtype");
  ConvGeom g = op_geom(0, nimg, hin, win, cin, cout, k, stride, pad, groups);
  if (dtype == MDD_DTYPE_BF16X2) g.prec = 1;
  if (dtype == MDD_DTYPE_F32_BF16OPS) g.prec = 2;
  if (dtype != MDD_DTYPE_BF16)
    launch_conv_wgrad<float>(g, (const float*)dy, (const float*)x, nullptr, nullptr, dw, db, nullptr, 0, nullptr, (hipStream_t)stream);
  else
    launch_conv_wgrad<bf16>(g, (const bf16*)dy, (const bf16*)x, nullptr, nullptr, dw, db, nullptr, 0, nullptr, (hipStream_t)stream);
  POST_LAUNCH("op_conv2d_wgrad");
  return 0;
}

int mdd_op_conv2d_wgrad2(int dtype, int nimg, int hin, int win, int cin, int cout, int k, int stride,
                         int pad, int groups, const void* dy1, const void* x1, const void* dy2, const void* x2,
                         float* dw, float* db, float* ws, long long ws_floats, void* stream) {
  CHECK_ARG(dy1 && x1 && dw, "null pointer");
  CHECK_ARG((dy2 == nullptr) == (x2 == nullptr), "the second pair is given whole or not at all");
  CHECK_ARG(dtype >= 0 && dtype <= 3, "dtype");
  CHECK_ARG(ws_floats >= 0 && (ws != nullptr || ws_floats == 0), "workspace");
  ConvGeom g = op_geom(0, nimg, hin, win, cin, cout, k, stride, pad, groups);
  if (dtype == MDD_DTYPE_BF16X2) g.prec = 1;
  if (dtype == MDD_DTYPE_F32_BF16OPS) g.prec = 2;
  if (dtype != MDD_DTYPE_BF16)
    launch_conv_wgrad<float>(g, (const float*)dy1, (const float*)x1, (const float*)dy2, (const float*)x2, dw, db, ws,
                             (int64_t)ws_floats, nullptr, (hipStream_t)stream);
  else
    launch_conv_wgrad<bf16>(g, (const bf16*)dy1, (const bf16*)x1, (const bf16*)dy2, (const bf16*)x2, dw, db, ws,
                            (int64_t)ws_floats, nullptr, (hipStream_t)stream);
  POST_LAUNCH("op_conv2d_wgrad2");
  return 0;
}

int mdd_retrieval_ranks(const float* img_feat, const float* txt_feat, const int* img2txt_off,
                        const int* img2txt_idx, const int* txt2img, int n_img, int n_txt, int dim,
                        float scale, float* scores_ws, float* norm_ws, int* rank_i2t, int* rank_t2i,
                        void* stream) {
  CHECK_ARG(img_feat && txt_feat && img2txt_off && img2txt_idx && txt2img, "null input pointer");
  CHECK_ARG(scores_ws && norm_ws && rank_i2t && rank_t2i, "null output / workspace pointer");
  CHECK_ARG(n_img > 0 && n_txt > 0 && dim > 0, "empty problem");
  launch_retrieval_ranks(rank_i2t, rank_t2i, scores_ws, norm_ws, img_feat, txt_feat, img2txt_off,
                         img2txt_idx, txt2img, n_img, n_txt, dim, scale, (hipStream_t)stream);
  POST_LAUNCH("retrieval_ranks");
  return 0;
}

int mdd_nearest_neighbor(const float* query, const float* bank, int n_query, int n_bank, int dim,
                         float* scores_ws, float* norm_ws, int* idx_out, void* stream) {
  CHECK_ARG(query && bank && scores_ws && norm_ws && idx_out, "null pointer");
  CHECK_ARG(n_query > 0 && n_bank > 0 && dim > 0, "empty problem");
  launch_nearest_neighbor(idx_out, scores_ws, norm_ws, query, bank, n_query, n_bank, dim, (hipStream_t)stream);
  POST_LAUNCH("nearest_neighbor");
  return 0;
}

}  // extern "C"
