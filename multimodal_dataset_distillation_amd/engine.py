"""UnrollEngine: Python host of the MI355X bi-trajectory matching engine (libmdd_hip.so).

Mirrors, for the hot path only, what reference distill.py:439-613 does with
CLIPModel_full + ReparamModule + torch.autograd: student forward through flat theta, inner
gradient, unrolled update, trajectory-matching loss and the outer gradient w.r.t.
(image_syn, text_syn, syn_lr_img, syn_lr_txt).  Device memory comes from the PyTorch caching
allocator (one workspace tensor); all arithmetic runs in hand-written HIP kernels.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DTYPE_BF16, DTYPE_BF16X2, DTYPE_F32, DTYPE_F32_BF16OPS, MddConfig, MddIterArgs, check

# f32: exact-fp32 MFMA; bf16: bf16 storage + MFMA; bf16x2: fp32 storage, split hi+lo bf16 operands (the
# fast mode within the 1e-3 parity bar); f32_bf16ops: experiment (fp32 storage, single-bf16 operands)
_DT = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32, "bf16": DTYPE_BF16,
       "bfloat16": DTYPE_BF16, "bf16x2": DTYPE_BF16X2, "f32_bf16ops": DTYPE_F32_BF16OPS}
FP32_STORAGE = ("f32", "fp32", "float32", "bf16x2", "f32_bf16ops")


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "engine tensors must be contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    assert t.dtype == torch.float32, "expected float32"
    return t


class UnrollEngine:
    def __init__(self, variant="nfnet_l0", batch=100, num_queries=None, image_size=224, d_txt=768,
                 syn_steps=8, dtype="bf16", device="cuda", bind=True, keep_steps=None):
        """keep_steps: how many inner steps keep their activations in HBM for the outer backward
        (None = all: nothing is recomputed); the rest are recomputed in one shared slot."""
        self.lib = _lib.load()
        self.variant, self.batch = variant, int(batch)
        self.num_queries = int(num_queries if num_queries is not None else batch)
        self.image_size, self.d_txt, self.syn_steps = int(image_size), int(d_txt), int(syn_steps)
        self.dtype = dtype
        self._variant_b = variant.encode()
        self.keep_steps = self.syn_steps if keep_steps is None else max(0, min(int(keep_steps), self.syn_steps))
        cfg = MddConfig(self._variant_b, self.batch, self.num_queries, self.image_size, self.d_txt,
                        self.syn_steps, _DT[dtype], -1 if keep_steps is None else int(keep_steps))
        self.num_slots = self.syn_steps if self.keep_steps >= self.syn_steps else self.keep_steps + 1
        h = C.c_void_p()
        check(self.lib.mdd_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.feature_dim = self.lib.mdd_engine_feature_dim(h)
        self.P_img = self.lib.mdd_engine_param_numel(h, 0)
        self.P_txt = self.lib.mdd_engine_param_numel(h, 1)
        self.workspace_bytes = self.lib.mdd_engine_workspace_bytes(h)
        self.workspace = None
        self.device = torch.device(device)
        if bind:
            self.bind()

    # ------------------------------------------------------------------ memory
    def bind(self):
        if not torch.cuda.is_available():
            raise RuntimeError("UnrollEngine needs an MI355X (HIP device); there is no CPU path")
        try:
            self.workspace = torch.empty(self.workspace_bytes + 256, dtype=torch.uint8,
                                         device=self.device)
        except RuntimeError as e:  # keep the reference's "out of memory" convention
            raise RuntimeError("HIP out of memory allocating %.1f GiB engine workspace: %s"
                               % (self.workspace_bytes / 2**30, e))
        off = (-self.workspace.data_ptr()) % 256
        self._ws_view = self.workspace[off:off + self.workspace_bytes]
        check(self.lib.mdd_engine_bind_workspace(self.h, C.c_void_p(self._ws_view.data_ptr()),
                                                 self.workspace_bytes, _stream()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mdd_engine_destroy(self.h)
            self.h = None
        self.workspace = None
        self._ws_view = None      # the view keeps the allocation alive otherwise

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameter table
    def param_table(self, which):
        """[(name, shape, offset)] in reference flatten order (reparam_module.py:28-39)."""
        w = {"img": 0, "txt": 1}[which] if isinstance(which, str) else which
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        ndim, off = C.c_int(), C.c_int64()
        for i in range(self.lib.mdd_engine_param_count(self.h, w)):
            check(self.lib.mdd_engine_param_info(self.h, w, i, name, 256, shape, C.byref(ndim),
                                                 C.byref(off)))
            out.append((name.value.decode(), tuple(shape[:ndim.value]), off.value))
        return out

    def buffer(self, name, slot):
        """Debug view of a named stash buffer (tests). slot=-1: tangent set, -2: shared."""
        off, elems, isf = C.c_int64(), C.c_int64(), C.c_int()
        check(self.lib.mdd_engine_find_buffer(self.h, name.encode(), slot, C.byref(off),
                                              C.byref(elems), C.byref(isf)))
        esz = 4 if (isf.value or self.dtype in FP32_STORAGE) else 2
        raw = self._ws_view[off.value: off.value + elems.value * esz]
        if isf.value or esz == 4:
            return raw.view(torch.float32)
        return raw.view(torch.bfloat16)

    # ------------------------------------------------------------------ passes
    def _new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def img_forward(self, slot, theta, image_syn, idx=None):
        out = self._new(self.batch, self.feature_dim)
        check(self.lib.mdd_img_forward(self.h, slot, _ptr(_f32(theta)), _ptr(_f32(image_syn)),
                                       _ptr(idx), _ptr(out), _stream()))
        return out

    def img_backward(self, slot, theta, feat_bar, dimage=None, idx=None, coef=None, mul=1.0,
                     stash=True):
        g = self._new(self.P_img)
        check(self.lib.mdd_img_backward(self.h, slot, _ptr(theta), _ptr(_f32(feat_bar)), _ptr(g),
                                        _ptr(dimage), _ptr(idx), _ptr(coef), float(mul),
                                        1 if stash else 0, _stream()))
        return g

    def img_tangent_forward(self, slot, theta, theta_dot):
        out = self._new(self.batch, self.feature_dim)
        check(self.lib.mdd_img_tangent_forward(self.h, slot, _ptr(theta), _ptr(_f32(theta_dot)),
                                               _ptr(out), _stream()))
        return out

    def img_tangent_backward(self, slot, theta, theta_dot, feat_bar_dot, dimage=None, idx=None,
                             coef=None, mul=1.0):
        h = self._new(self.P_img)
        check(self.lib.mdd_img_tangent_backward(self.h, slot, _ptr(theta), _ptr(theta_dot),
                                                _ptr(_f32(feat_bar_dot)), _ptr(h), _ptr(dimage),
                                                _ptr(idx), _ptr(coef), float(mul), _stream()))
        return h

    def txt_forward(self, slot, theta, text_syn, idx=None, drop_mask=None):
        out = self._new(self.batch, self.feature_dim)
        self._keep = getattr(self, "_keep", {})
        self._keep[("mask", slot)] = drop_mask  # the engine reads it again in backward/tangent
        check(self.lib.mdd_txt_forward(self.h, slot, _ptr(_f32(theta)), _ptr(_f32(text_syn)),
                                       _ptr(idx), _ptr(drop_mask), _ptr(out), _stream()))
        return out

    def txt_backward(self, slot, theta, feat_bar, dtext=None, idx=None, coef=None, mul=1.0,
                     stash=True):
        g = self._new(self.P_txt)
        check(self.lib.mdd_txt_backward(self.h, slot, _ptr(theta), _ptr(_f32(feat_bar)), _ptr(g),
                                        _ptr(dtext), _ptr(idx), _ptr(coef), float(mul),
                                        1 if stash else 0, _stream()))
        return g

    def txt_tangent_forward(self, slot, theta, theta_dot):
        out = self._new(self.batch, self.feature_dim)
        check(self.lib.mdd_txt_tangent_forward(self.h, slot, _ptr(theta), _ptr(_f32(theta_dot)),
                                               _ptr(out), _stream()))
        return out

    def txt_tangent_backward(self, slot, theta, theta_dot, feat_bar_dot, dtext=None, idx=None,
                             coef=None, mul=1.0):
        h = self._new(self.P_txt)
        check(self.lib.mdd_txt_tangent_backward(self.h, slot, _ptr(theta), _ptr(theta_dot),
                                                _ptr(_f32(feat_bar_dot)), _ptr(h), _ptr(dtext),
                                                _ptr(idx), _ptr(coef), float(mul), _stream()))
        return h

    def contrastive(self, x, y, scale):
        """scale: device scalar tensor (the fork's syn_lr_img, distill.py:548) or python float."""
        loss, sbar = self._new(1), self._new(1)
        xbar, ybar = torch.empty_like(x), torch.empty_like(y)
        sdev, sconst = (scale, 0.0) if torch.is_tensor(scale) else (None, float(scale))
        check(self.lib.mdd_contrastive(self.h, _ptr(x), _ptr(y), _ptr(sdev), sconst, _ptr(loss),
                                       _ptr(xbar), _ptr(ybar), _ptr(sbar), _stream()))
        return loss, xbar, ybar, sbar

    def contrastive_tangent(self, x, y, x_dot, y_dot, scale):
        sbd = self._new(1)
        xbd, ybd = torch.empty_like(x), torch.empty_like(y)
        sdev, sconst = (scale, 0.0) if torch.is_tensor(scale) else (None, float(scale))
        check(self.lib.mdd_contrastive_tangent(self.h, _ptr(x), _ptr(y), _ptr(x_dot), _ptr(y_dot),
                                               _ptr(sdev), sconst, _ptr(xbd), _ptr(ybd), _ptr(sbd),
                                               _stream()))
        return xbd, ybd, sbd

    def set_pass_precision(self, fwd=0, bwd=0, tan_fwd=0, tan_bwd=0):
        """fp32-storage engines: operand arithmetic of the image encoder's contractions per pass
        (0 engine mode, 1 split bf16, 2 one bf16, 3 exact fp32) -- include/mdd_hip.h."""
        check(self.lib.mdd_engine_set_pass_precision(self.h, int(fwd), int(bwd), int(tan_fwd), int(tan_bwd)))

    # ------------------------------------------------------------------ whole iteration
    def unrolled_match(self, image_syn, text_syn, lr_img, lr_txt, theta0_img, theta0_txt,
                       target_img, target_txt, perms=None, drop_masks=None, syn_steps=None,
                       logit_scale=None, out=None):
        """One outer iteration (reference distill.py:509-606).  Returns dict with `grand_loss`,
        `img_loss`, `txt_loss`, `contrastive` [K] and grads `image_syn`, `text_syn`, `lr` [2]
        (all device tensors; nothing is synchronised)."""
        K = self.syn_steps if syn_steps is None else int(syn_steps)
        if out is None:
            out = dict(image_syn=torch.empty_like(image_syn), text_syn=torch.empty_like(text_syn),
                       lr=self._new(2), losses=self._new(3 + K))
        a = MddIterArgs()
        a.image_syn, a.text_syn = image_syn.data_ptr(), text_syn.data_ptr()
        a.lr_img, a.lr_txt = lr_img.data_ptr(), lr_txt.data_ptr()
        a.theta0_img, a.theta0_txt = theta0_img.data_ptr(), theta0_txt.data_ptr()
        a.target_img, a.target_txt = target_img.data_ptr(), target_txt.data_ptr()
        a.perms = perms.data_ptr() if perms is not None else None
        a.drop_masks = drop_masks.data_ptr() if drop_masks is not None else None
        a.syn_steps = K
        a.use_lr_as_scale = 1 if logit_scale is None else 0
        a.logit_scale_const = 0.0 if logit_scale is None else float(logit_scale)
        a.grad_image_syn, a.grad_text_syn = out["image_syn"].data_ptr(), out["text_syn"].data_ptr()
        a.grad_lr, a.losses = out["lr"].data_ptr(), out["losses"].data_ptr()
        for t in (image_syn, text_syn, lr_img, lr_txt, theta0_img, theta0_txt, target_img, target_txt):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        if perms is not None:
            assert perms.dtype == torch.int64 and perms.is_contiguous() and perms.shape == (K, self.batch)
        check(self.lib.mdd_unrolled_match(self.h, C.byref(a), _stream()))
        L = out["losses"]
        out.update(grand_loss=L[0], img_loss=L[1], txt_loss=L[2], contrastive=L[3:3 + K])
        return out
