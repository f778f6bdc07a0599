"""HIP-backed student networks with the reference's module surface (reference networks.py):

  ImageEncoder(args)   networks.py:648-682   `.model` = NFNet-l0 (timm `nfnet_l0`, num_classes=0,
                                             global_pool="avg"), forward(x) -> [N, 2304]
  ProjectionHead(...)  networks.py:625-646   Linear -> GELU -> Linear -> Dropout -> +res -> LayerNorm
  CLIPModel_full(args) networks.py:805-843   container with .image_encoder / .text_projection

The modules own `nn.Parameter`s with exactly the reference's names, shapes and registration order
(the table comes from the C-ABI library, which is the single source of truth for the flat layout),
so `ReparamModule(module)` flattens them identically and expert buffers written by the
reference's buffer.py (`[p for p in net.parameters()]`) load unchanged.  All arithmetic runs in
libmdd_hip.so: `forward_flat(theta, x)` is one twice-differentiable autograd op per network
(functional.py).  There is no torch/ATen implementation of these forwards in the product.
"""
import math

import torch
import torch.nn as nn

from .engine import UnrollEngine
from . import functional as Fn

# reference networks.py:660-676 names -> engine topologies.  'vit' there is timm.create_model('vit_tiny_patch16_224')
# WITHOUT num_classes=0 (:668): the model keeps timm's 1000-way head (5,717,416 parameters, 1000-d features into the
# projection, networks.py:819) = 'vit_tiny16_cls' here, so reference-format 'vit' expert buffers flatten to P_img.
# 'vit_tiny16' (headless, 192-d class token) and 'vit_b16' (BASELINE configs[4]'s ViT-B/16; the reference's 'clip' is
# ViT-B/32 through the clip package) are build-defined.
VARIANTS = {"nfnet": "nfnet_l0", "nfnet_l0": "nfnet_l0", "nfnet_l1": "nfnet_l1",
            "nfnet_tiny": "nfnet_tiny", "vit": "vit_tiny16_cls", "vit_tiny16_cls": "vit_tiny16_cls",
            "vit_tiny16": "vit_tiny16", "vit_b16": "vit_b16", "vit_micro": "vit_micro", "vit_micro_cls": "vit_micro_cls"}

_ENGINES = {}


def get_engine(variant, batch, image_size, d_txt, syn_steps, dtype, device="cuda", num_queries=None):
    """Engines are shared between the image encoder and the text projection of one student."""
    key = (variant, int(batch), int(image_size), int(d_txt), int(syn_steps), dtype, str(device),
           int(num_queries or batch))
    if key not in _ENGINES:
        _ENGINES[key] = UnrollEngine(variant, batch=batch, num_queries=num_queries or batch,
                                     image_size=image_size, d_txt=d_txt, syn_steps=syn_steps,
                                     dtype=dtype, device=device)
    return _ENGINES[key]


def release_engines():
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()


def _param_table(variant, which, d_txt=768):
    eng = UnrollEngine(variant, batch=2, image_size=32, d_txt=d_txt, syn_steps=1, dtype="f32",
                       bind=False)
    tab = eng.param_table(which)
    feat = eng.feature_dim
    eng.close()
    return tab, feat


def _attach(root, dotted, param):
    *path, leaf = dotted.split(".")
    m = root
    for part in path:
        if not hasattr(m, part):
            m.add_module(part, nn.Module())
        m = getattr(m, part)
    m.register_parameter(leaf, param)


def init_image_params(table, generator=None, trained_like=False):
    """timm NormFreeNet init: kaiming_normal_(fan_in, linear) conv weights, zero bias, gain 1 (conv3:
    0).  trained_like=True perturbs gains/biases the way oracle/nfnet_ref.randomize_like_trained
    does (synthetic expert snapshots; default init zeroes most of the theta-gradient)."""
    out = []
    if any(name == "model.cls_token" for name, _, _ in table):
        # timm VisionTransformer init: trunc_normal(0.02) linear weights and position embedding, zero biases, unit
        # LayerNorm gains, class token N(0, 1e-6); trained_like perturbs gains / biases / class token
        for name, shape, _ in table:
            if name.endswith("cls_token"):
                t = (0.02 if trained_like else 1e-6) * torch.randn(shape, generator=generator)
            elif name.endswith("pos_embed"):
                t = 0.02 * torch.randn(shape, generator=generator)
            elif "norm" in name and name.endswith(".weight"):
                t = 1.0 + (0.1 * torch.randn(shape, generator=generator) if trained_like else 0.0) * torch.ones(shape)
            elif name.endswith(".weight"):
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= int(d)
                t = torch.randn(shape, generator=generator) * (1.0 / math.sqrt(fan_in) if "patch_embed" in name else 0.02)
            else:
                t = 0.02 * torch.randn(shape, generator=generator) if trained_like else torch.zeros(shape)
            out.append(t)
        return out
    for name, shape, _ in table:
        if name.endswith(".weight"):
            fan_in = int(shape[1] * shape[2] * shape[3])
            t = torch.randn(shape, generator=generator) / math.sqrt(fan_in)
        elif name.endswith(".gain"):
            t = torch.zeros(shape) if (".conv3." in name and not trained_like) else torch.ones(shape)
            if trained_like:
                t = 1.0 + 0.1 * torch.randn(shape, generator=generator)
        else:
            t = 0.02 * torch.randn(shape, generator=generator) if trained_like else torch.zeros(shape)
        out.append(t)
    return out


class ImageEncoder(nn.Module):
    """reference networks.py:648-682 for args.image_encoder == 'nfnet' (the hot path's encoder)."""

    def __init__(self, args=None, eval_stage=False, variant=None, dtype=None):
        super().__init__()
        name = variant or getattr(args, "image_encoder", "nfnet")
        if name not in VARIANTS:
            raise NotImplementedError(
                "MI355X engine implements the image encoders %s; got %r" % (sorted(VARIANTS), name))
        self.model_name = name
        self.variant = VARIANTS[name]
        self.compute_dtype = dtype or getattr(args, "compute_dtype", "bf16")
        self.syn_steps = int(getattr(args, "syn_steps", 8) or 8)
        self.d_txt = 768
        table, self.num_features = _param_table(self.variant, "img")
        self.model = nn.Module()
        for (pname, shape, _), t in zip(table, init_image_params(table)):
            assert pname.startswith("model.")
            _attach(self, pname, nn.Parameter(t))
        self._table = table
        self._slot = 0

    def _engine(self, x):
        return get_engine(self.variant, x.shape[0], x.shape[-1], self.d_txt, self.syn_steps,
                          self.compute_dtype, x.device)

    def forward_flat(self, flat_param, x):
        eng = self._engine(x)
        slot = self._slot % eng.syn_steps
        self._slot += 1
        return Fn.image_encoder(eng, slot, flat_param, x)

    def forward(self, x):
        flat = torch.cat([p.reshape(-1) for p in self.parameters()])
        return self.forward_flat(flat, x)


class ProjectionHead(nn.Module):
    """reference networks.py:625-646.  Dropout is active in train mode (distill.py:446-447): the mask
    is drawn with torch's device RNG and replayed by the engine in backward / double-backward."""

    def __init__(self, embedding_dim, projection_dim=768, dropout=0.1, image_size=224,
                 variant="nfnet_l0", syn_steps=8, dtype="bf16"):
        super().__init__()
        self.projection = nn.Linear(embedding_dim, projection_dim)
        self.gelu = nn.GELU()
        self.fc = nn.Linear(projection_dim, projection_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(projection_dim)
        self.embedding_dim, self.projection_dim, self.p = embedding_dim, projection_dim, dropout
        self.image_size, self.variant, self.syn_steps, self.compute_dtype = image_size, variant, syn_steps, dtype
        self._slot = 0

    def forward_flat(self, flat_param, x):
        eng = get_engine(self.variant, x.shape[0], self.image_size, self.embedding_dim,
                         self.syn_steps, self.compute_dtype, x.device)
        if eng.feature_dim != self.projection_dim:
            raise RuntimeError("projection_dim %d does not match the %s feature dim %d"
                               % (self.projection_dim, self.variant, eng.feature_dim))
        slot = self._slot % eng.syn_steps
        self._slot += 1
        mask = None
        if self.training and self.p > 0:
            mask = (torch.rand(x.shape[0], self.projection_dim, device=x.device) >= self.p).float()
            mask = mask / (1.0 - self.p)
        return Fn.text_projection(eng, slot, flat_param, x, mask)

    def forward(self, x):
        flat = torch.cat([p.reshape(-1) for p in self.parameters()])
        return self.forward_flat(flat, x)


class CLIPModel_full(nn.Module):
    """reference networks.py:805-843, restricted to what the distillation hot path constructs:
    `.image_encoder` and `.text_projection` (the frozen BERT/CLIP text encoder runs only at
    synthetic-set initialisation and is outside the hot path -- SURVEY 8f)."""

    def __init__(self, args, temperature=1.0, eval_stage=False):
        super().__init__()
        self.image_encoder = ImageEncoder(args, eval_stage=eval_stage)
        self.image_embedding = self.image_encoder.num_features
        text_encoder = getattr(args, "text_encoder", "bert")
        if text_encoder == "bert":
            self.text_embedding = 768
        elif text_encoder == "clip":
            self.text_embedding = 512
        else:
            raise NotImplementedError("Unsupported text encoder:", text_encoder)
        self.image_encoder.d_txt = self.text_embedding
        self.text_projection = ProjectionHead(
            embedding_dim=self.text_embedding, projection_dim=self.image_embedding,
            image_size=getattr(args, "image_size", 224), variant=self.image_encoder.variant,
            syn_steps=self.image_encoder.syn_steps, dtype=self.image_encoder.compute_dtype)
        self.temperature = temperature
        self.args = args
        self.distill = getattr(args, "distill", False)


# ------------------------------------------------------------------------------------------------
def synthetic_expert_params(engine, seed, device="cuda"):
    """Flat (theta_img, theta_txt) of a synthetic 'trained-like' expert snapshot, built on the host
    with torch's CPU RNG (deterministic) in the engine's flatten order."""
    g = torch.Generator().manual_seed(seed)
    ti = torch.cat([t.reshape(-1) for t in
                    init_image_params(engine.param_table("img"), g, trained_like=True)])
    parts = []
    for name, shape, _ in engine.param_table("txt"):
        if name.endswith("weight") and len(shape) == 2:
            bound = 1.0 / math.sqrt(shape[1])
            parts.append((torch.rand(shape, generator=g) * 2 - 1) * bound)
        elif name == "layer_norm.weight":
            parts.append(1.0 + 0.05 * torch.randn(shape, generator=g))
        else:
            parts.append(0.02 * torch.randn(shape, generator=g))
    tt = torch.cat([t.reshape(-1) for t in parts])
    return ti.to(device), tt.to(device)


def student_move_normalised_targets(engine, th0_img, th0_txt, image_syn, text_syn, lr, syn_steps, generator,
                                    frac=0.1):
    """Synthetic expert targets theta* = theta0 + sigma * N(0, I) with sigma chosen so that
    |theta* - theta0| equals the norm of the student's own syn_steps-step move (frac * K * |g0|, g0 = the
    inner gradient at theta0 on the full synthetic set) -- the construction oracle/gen_golden.py uses.
    With it the normalised matching loss (reference distill.py:596-597) and the outer gradients depend
    O(1) on the inner path; a fixed tiny displacement makes grand_loss = 2 regardless of the kernels.
    Runs the engine's first-order passes in slot 0.  Returns (target_img, target_txt, sigma_img, sigma_txt)."""
    x0 = engine.img_forward(0, th0_img, image_syn)
    y0 = engine.txt_forward(0, th0_txt, text_syn)
    _, xb0, yb0, _ = engine.contrastive(x0, y0, lr[0:1])
    gi0 = engine.img_backward(0, th0_img, xb0)
    gt0 = engine.txt_backward(0, th0_txt, yb0)
    sig_i = float(frac * syn_steps * gi0.norm() / th0_img.numel() ** 0.5)
    sig_t = float(frac * syn_steps * gt0.norm() / th0_txt.numel() ** 0.5)
    dev = th0_img.device
    tgi = th0_img + sig_i * torch.randn(th0_img.shape, device=dev, generator=generator)
    tgt = th0_txt + sig_t * torch.randn(th0_txt.shape, device=dev, generator=generator)
    return tgi, tgt, sig_i, sig_t
