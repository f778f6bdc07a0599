"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement (PyTorch autograd) of the reference's hot
path: distill.py's inner `syn_steps` unrolled-training + bi-trajectory-matching loop.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product package (multimodal_dataset_distillation_amd) never does.

Follows, line by line:
  reference distill.py:466-476   flatten start / target expert params
  reference distill.py:509-583   inner loop: minibatch pick, student forward via flat theta,
                                 L2-normalise, logits = syn_lr_img * x @ y.T (fork quirk: the inner
                                 LR doubles as logit scale, :548), symmetric CE, two
                                 autograd.grad(create_graph=True), theta' = theta - lr * g
  reference distill.py:584-598   normalised sum-MSE trajectory-matching loss
  reference distill.py:603-613   backward + three SGD(momentum=0.5) steps (distill.py:233-241)
  reference reparam_module.py:110-115,148-159   flat theta -> split/view -> module forward
  reference networks.py:625-646  ProjectionHead

Pinning: tests/golden/*.npz are produced by oracle/gen_golden.py which runs THIS restatement
side by side with the reference's importable pieces (the real `ReparamModule` and the
AST-extracted `ProjectionHead`) and asserts equality before writing.  The image encoder
(timm NFNet, third-party, absent) is PARITY UNPINNED -- see oracle/nfnet_ref.py.
"""
import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.func import functional_call


class ProjectionHead(nn.Module):
    """reference networks.py:625-646.  `drop_mask` (already scaled by 1/(1-p)) replaces
    nn.Dropout so parity runs can inject the mask; None = dropout off."""

    def __init__(self, embedding_dim, projection_dim=768, dropout=0.1):
        super().__init__()
        self.projection = nn.Linear(embedding_dim, projection_dim)
        self.fc = nn.Linear(projection_dim, projection_dim)
        self.layer_norm = nn.LayerNorm(projection_dim)
        self.p = dropout

    def forward(self, x, drop_mask=None):
        projected = self.projection(x)
        x = F.gelu(projected)
        x = self.fc(x)
        if drop_mask is not None:
            x = x * drop_mask
        x = x + projected
        return self.layer_norm(x)


class FlatModule:
    """reference reparam_module.py:18-75 (flatten order) + :110-115 (split/view) + :148-159
    (forward with a caller-supplied flat_param), restated with torch.func.functional_call."""

    def __init__(self, module: nn.Module):
        self.module = module
        self.names, self.shapes, self.numels = [], [], []
        for mn, m in module.named_modules():
            for n, p in m.named_parameters(recurse=False):
                self.names.append(f"{mn}.{n}" if mn else n)
                self.shapes.append(tuple(p.shape))
                self.numels.append(p.numel())
        self.param_numel = sum(self.numels)

    def flat_param(self):
        sd = dict(self.module.named_parameters())
        return torch.cat([sd[n].detach().reshape(-1) for n in self.names])

    def __call__(self, *inputs, flat_param, **kw):
        flat_param = torch.squeeze(flat_param)
        views = {n: t.view(s) for n, t, s in
                 zip(self.names, flat_param.split(self.numels), self.shapes)}
        return functional_call(self.module, views, inputs, kw)


def contrastive_loss(x, y, scale):
    """reference distill.py:533,546-551."""
    x = x / x.norm(dim=1, keepdim=True)
    y = y / y.norm(dim=1, keepdim=True)
    logits = scale * x.float() @ y.float().t()
    gt = torch.arange(len(logits))
    return (F.cross_entropy(logits, gt) + F.cross_entropy(logits.t(), gt)) / 2


def unrolled_match(img_net: FlatModule, txt_net: FlatModule, image_syn, text_syn,
                   syn_lr_img, syn_lr_txt, th0_img, th0_txt, tgt_img, tgt_txt,
                   perms: Sequence[torch.Tensor], drop_masks: Optional[List] = None,
                   logit_scale=None):
    """One outer iteration up to grand_loss (reference distill.py:466-598).

    perms[k] is the index vector `these_indices` of step k (distill.py:510-511).
    logit_scale=None reproduces the fork (scale = syn_lr_img, :548); a float reproduces
    upstream distill_original.py:430.
    Returns grand_loss, dict(contrastive=[...], img_loss, txt_loss, theta_img, theta_txt).
    """
    img_params = [th0_img.detach().clone().requires_grad_(True)]
    txt_params = [th0_txt.detach().clone().requires_grad_(True)]
    ce = []
    for k, idx in enumerate(perms):
        x = image_syn[idx]
        y = text_syn[idx]
        x = img_net(x, flat_param=img_params[-1])
        dm = None if drop_masks is None else drop_masks[k]
        y = txt_net(y, flat_param=txt_params[-1], drop_mask=dm)
        scale = syn_lr_img if logit_scale is None else logit_scale
        loss = contrastive_loss(x, y, scale)
        g_img = torch.autograd.grad(loss, img_params[-1], create_graph=True)[0]
        g_txt = torch.autograd.grad(loss, txt_params[-1], create_graph=True)[0]
        ce.append(loss.detach())
        img_params.append(img_params[-1] - syn_lr_img * g_img)
        txt_params.append(txt_params[-1] - syn_lr_txt * g_txt)
    img_loss = F.mse_loss(img_params[-1], tgt_img, reduction="sum")
    img_dist = F.mse_loss(th0_img, tgt_img, reduction="sum")
    txt_loss = F.mse_loss(txt_params[-1], tgt_txt, reduction="sum")
    txt_dist = F.mse_loss(th0_txt, tgt_txt, reduction="sum")
    img_loss = img_loss / img_dist
    txt_loss = txt_loss / txt_dist
    grand = img_loss + txt_loss
    return grand, dict(contrastive=ce, img_loss=img_loss.detach(), txt_loss=txt_loss.detach(),
                       theta_img=img_params[-1].detach(), theta_txt=txt_params[-1].detach())


def outer_grads(grand, image_syn, text_syn, syn_lr_img, syn_lr_txt):
    """reference distill.py:603-606 (zero_grad + backward), returned instead of accumulated."""
    return torch.autograd.grad(grand, [image_syn, text_syn, syn_lr_img, syn_lr_txt])


class SGDMomentum:
    """torch.optim.SGD(momentum=m, dampening=0, nesterov=False) restated (distill.py:233-241):
    buf = g on first step, else buf = m*buf + g; p -= lr*buf."""

    def __init__(self, lr, momentum=0.5):
        self.lr, self.m, self.buf = lr, momentum, None

    def step(self, p, g):
        self.buf = g.clone() if self.buf is None else self.m * self.buf + g
        return p - self.lr * self.buf


def synthetic_inputs(n, image_size, d_txt, seed=0):
    """BASELINE.md synthetic inputs (upstream noise-init constants, distill_original.py:139-147)."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor([-0.0626, -0.0221, 0.0680]).view(1, 3, 1, 1)
    std = torch.tensor([1.0451, 1.0752, 1.0539]).view(1, 3, 1, 1)
    image_syn = torch.randn(n, 3, image_size, image_size, generator=g) * std + mean
    text_syn = torch.randn(n, d_txt, generator=g) * 0.5253 - 0.0094
    return image_syn, text_syn


def is_nan_break(img_loss):
    """reference distill.py:599."""
    return math.isnan(float(img_loss))
