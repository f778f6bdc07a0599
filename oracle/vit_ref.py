"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the ViT image encoder of BASELINE configs[4].

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the
product package never does.

What it restates
----------------
BASELINE.json configs[4] names "ViT-B/16 + CLIP-text encoders swapped in via networks.py".  The reference's
``ImageEncoder`` (networks.py:648-682) reaches a Vision Transformer two ways -- ``'clip'`` -> ViT-B/32 through the
``clip`` package (:661), ``'vit'`` -> ``timm.create_model('vit_tiny_patch16_224')`` (:668) -- and has no ViT-B/16
path: the config is a build-defined extension.  What is restated here is timm 0.6.7's published
``VisionTransformer`` (``timm/models/vision_transformer.py``; timm is pinned in requirements.yaml:282, not vendored,
not installed) with ``num_classes=0`` and the class token as the pooled feature:

    x = patch_embed(img)                       Conv2d(3, D, kernel=P, stride=P) -> [N, (S/P)^2, D]
    x = cat(cls_token, x) + pos_embed          [N, T = 1 + (S/P)^2, D]
    for each block:
        x = x + proj(attention(qkv(LayerNorm(x))))       heads H, head dim D/H, scale (D/H)^-0.5, qkv_bias=True
        x = x + fc2(GELU(fc1(LayerNorm(x))))             hidden 4 D, exact (erf) GELU
    y = LayerNorm(x)[:, 0]                     eps = 1e-6 in every LayerNorm

PARITY UNPINNED: neither timm nor a fixture of it exists here; structural anchors checked in tests/test_oracle.py:
ViT-B/16 has 85,798,656 parameters without a head (86,567,656 with timm's 1000-way head = its published 86.6 M),
vit_tiny_patch16_224 5,524,416 (5,717,416 with head = published 5.7 M), feature dims 768 / 192.

Parameter order (theta is ONE flat vector in ``named_modules() x named_parameters(recurse=False)`` order, reference
reparam_module.py:28-39): the root module's own parameters first (cls_token, pos_embed), then patch_embed.proj,
blocks.i.{norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2}, norm -- timm's registration order.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VitCfg:
    img_size: int = 224
    patch: int = 16
    dim: int = 768
    depth: int = 12
    heads: int = 12
    mlp_ratio: float = 4.0
    eps: float = 1e-6
    num_classes: int = 0      # > 0: timm's default classifier head Linear(D, num_classes) on the normalised class token


VARIANTS = {
    "vit_b16": VitCfg(),                                                    # vit_base_patch16_224
    "vit_tiny16": VitCfg(dim=192, depth=12, heads=3),                       # vit_tiny_patch16_224 (networks.py:668)
    "vit_micro": VitCfg(img_size=32, patch=8, dim=64, depth=2, heads=2),    # plumbing size for goldens / GPU tests
    # the reference's 'vit' as it stands: timm.create_model('vit_tiny_patch16_224', pretrained=True) WITHOUT
    # num_classes=0 (networks.py:668) keeps the 1000-way head -- 5,717,416 parameters, 1000-d features into the
    # projection (networks.py:819: image_embedding = 1000)
    "vit_tiny16_cls": VitCfg(dim=192, depth=12, heads=3, num_classes=1000),
    "vit_micro_cls": VitCfg(img_size=32, patch=8, dim=64, depth=2, heads=2, num_classes=24),
}


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, C = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]                      # [B, H, T, hd]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, T, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))                  # nn.GELU() default = exact erf


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)        # [N, (S/P)^2, D], row-major over the patch grid


class VisionTransformer(nn.Module):
    def __init__(self, cfg: VitCfg, img_size=None):
        super().__init__()
        self.cfg = cfg
        size = img_size or cfg.img_size
        assert size % cfg.patch == 0
        self.num_tokens = 1 + (size // cfg.patch) ** 2
        self.num_features = cfg.num_classes if cfg.num_classes > 0 else cfg.dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.num_tokens, cfg.dim) * 0.02)
        self.patch_embed = PatchEmbed(cfg.patch, cfg.dim)
        self.blocks = nn.Sequential(*[Block(cfg.dim, cfg.heads, cfg.mlp_ratio, cfg.eps) for _ in range(cfg.depth)])
        self.norm = nn.LayerNorm(cfg.dim, eps=cfg.eps)
        if cfg.num_classes > 0:
            self.head = nn.Linear(cfg.dim, cfg.num_classes)      # registered after `norm`, as in timm
        nn.init.normal_(self.cls_token, std=1e-6)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        x = self.blocks(x)
        x = self.norm(x)[:, 0]
        return self.head(x) if self.cfg.num_classes > 0 else x


class ImageEncoder(nn.Module):
    """Same wrapper shape as oracle/nfnet_ref.ImageEncoder (reference networks.py:648-682: `self.model`)."""

    def __init__(self, variant="vit_b16", img_size=None):
        super().__init__()
        self.model = VisionTransformer(VARIANTS[variant], img_size)

    def forward(self, x):
        return self.model(x)


def randomize_like_trained(module: nn.Module, seed: int):
    """Break the symmetric default init (unit LayerNorm gains, zero biases, 1e-6 class token) so that every
    parameter carries a gradient of ordinary size in the parity tests."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.add_(0.02 * torch.randn(p.shape, generator=g))
            elif name.endswith("cls_token"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
