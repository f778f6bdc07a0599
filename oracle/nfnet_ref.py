"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the image encoder on the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product package never does.

What it restates
----------------
``ImageEncoder('nfnet')`` (reference networks.py:648-682) is
``timm.create_model('nfnet_l0', num_classes=0, global_pool="avg")`` (networks.py:666).  timm is a
third-party dependency (pinned ``timm=0.6.7``, reference requirements.yaml:282) that is NOT
vendored in /root/reference and NOT installed in this image, so the arithmetic below is a
restatement of timm 0.6.7's published ``NormFreeNet`` (``timm/models/nfnet.py``) and
``ScaledStdConv2d`` (``timm/models/layers/std_conv.py``) algorithm.

PARITY UNPINNED for the image encoder: the reference holds no test, golden vector or known-answer
value at this boundary.  Structural anchors that ARE checked (tests/test_oracle.py, tests/test_abi.py):
feature dim 2304 (networks.py:811), 32,769,488 parameters without a head (35,074,488 with the
1000-way head = timm's published 35.07 M), parameter order (weight, bias, gain) per conv.

Parameter order (matters: theta is one flat vector in ``named_modules() x named_parameters``
order, reference reparam_module.py:28-39) follows timm's module registration order:
stem.conv1..4 -> stages.S.B.{downsample.conv, conv1, conv2, conv2b, conv3, attn_last.fc1,
attn_last.fc2} -> final_conv.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

GAMMA_SILU = 1.7881293296813965  # timm _nonlin_gamma['silu']


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


@dataclass
class NfCfg:
    depths: Tuple[int, ...] = (1, 2, 6, 3)
    channels: Tuple[int, ...] = (256, 512, 1536, 1536)
    stem_chs: int = 128
    group_size: int = 64
    bottle_ratio: float = 0.25
    feat_mult: float = 1.5
    se_rd_ratio: float = 0.25
    se_rd_divisor: int = 8
    alpha: float = 0.2
    attn_gain: float = 2.0
    std_conv_eps: float = 1e-5
    gamma: float = GAMMA_SILU
    ch_div: int = 8

    @property
    def num_features(self):
        return int(self.channels[-1] * self.feat_mult)


CONFIGS = {
    # timm: nfnet_l0=_nfnet_cfg(depths=(1,2,6,3), feat_mult=1.5, group_size=64, bottle_ratio=0.25,
    #        attn_kwargs=dict(rd_ratio=0.25, rd_divisor=8), act_layer='silu')
    "nfnet_l0": NfCfg(),
    "nfnet_l1": NfCfg(depths=(2, 4, 12, 6), feat_mult=2.0),
    # build-defined miniature of the same topology for fast full-tensor parity tests
    "nfnet_tiny": NfCfg(depths=(1, 2, 2, 1), channels=(64, 128, 192, 192), stem_chs=64,
                        group_size=16, feat_mult=1.5),
}


class ScaledStdConv2d(nn.Conv2d):
    """timm ScaledStdConv2d: weight standardised per out-channel on EVERY forward (so it is inside
    the theta-graph), scaled by gain * gamma * fan_in**-0.5.  Param order: weight, bias, gain."""

    def __init__(self, in_chs, out_chs, kernel_size, stride=1, groups=1, gamma=1.0, eps=1e-5,
                 gain_init=1.0):
        padding = ((stride - 1) + (kernel_size - 1)) // 2
        super().__init__(in_chs, out_chs, kernel_size, stride=stride, padding=padding,
                         groups=groups, bias=True)
        self.gain = nn.Parameter(torch.full((out_chs, 1, 1, 1), gain_init))
        self.scale = gamma * self.weight[0].numel() ** -0.5
        self.eps = eps

    def standardized_weight(self):
        return F.batch_norm(
            self.weight.reshape(1, self.out_channels, -1), None, None,
            weight=(self.gain * self.scale).view(-1),
            training=True, momentum=0., eps=self.eps).reshape_as(self.weight)

    def forward(self, x):
        return F.conv2d(x, self.standardized_weight(), self.bias, self.stride, self.padding,
                        self.dilation, self.groups)


class SEModule(nn.Module):
    def __init__(self, channels, rd_ratio, rd_divisor):
        super().__init__()
        rd = make_divisible(channels * rd_ratio, rd_divisor, round_limit=0.)
        self.fc1 = nn.Conv2d(channels, rd, 1, bias=True)
        self.fc2 = nn.Conv2d(rd, channels, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = F.relu(self.fc1(s))
        s = self.fc2(s)
        return x * s.sigmoid()


class DownsampleAvg(nn.Module):
    def __init__(self, in_chs, out_chs, stride, conv):
        super().__init__()
        self.pool = (nn.AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False)
                     if stride > 1 else nn.Identity())
        self.conv = conv(in_chs, out_chs, 1, stride=1)

    def forward(self, x):
        return self.conv(self.pool(x))


class NormFreeBlock(nn.Module):
    def __init__(self, in_chs, out_chs, stride, alpha, beta, cfg: NfCfg, conv):
        super().__init__()
        mid = make_divisible(out_chs * cfg.bottle_ratio, cfg.ch_div)
        groups = 1 if not cfg.group_size else mid // cfg.group_size
        if cfg.group_size and cfg.group_size % cfg.ch_div == 0:
            mid = cfg.group_size * groups
        self.alpha, self.beta, self.attn_gain = alpha, beta, cfg.attn_gain
        self.downsample = (DownsampleAvg(in_chs, out_chs, stride, conv)
                           if in_chs != out_chs or stride != 1 else None)
        self.conv1 = conv(in_chs, mid, 1)
        self.conv2 = conv(mid, mid, 3, stride=stride, groups=groups)
        self.conv2b = conv(mid, mid, 3, stride=1, groups=groups)
        self.conv3 = conv(mid, out_chs, 1, gain_init=0.)
        self.attn_last = SEModule(out_chs, cfg.se_rd_ratio, cfg.se_rd_divisor)

    def forward(self, x):
        out = F.silu(x) * self.beta
        shortcut = x
        if self.downsample is not None:
            shortcut = self.downsample(out)
        out = self.conv1(out)
        out = self.conv2(F.silu(out))
        out = self.conv2b(F.silu(out))
        out = self.conv3(F.silu(out))
        out = self.attn_gain * self.attn_last(out)
        return out * self.alpha + shortcut


class NormFreeNet(nn.Module):
    """timm NormFreeNet with stem_type='deep_quad', num_classes=0, global_pool='avg'."""

    def __init__(self, cfg: NfCfg, in_chans=3):
        super().__init__()
        self.cfg = cfg

        def conv(i, o, k, stride=1, groups=1, gain_init=1.0):
            return ScaledStdConv2d(i, o, k, stride=stride, groups=groups, gamma=cfg.gamma,
                                   eps=cfg.std_conv_eps, gain_init=gain_init)

        sc = cfg.stem_chs
        stem_chs = (sc // 8, sc // 4, sc // 2, sc)
        strides = (2, 1, 1, 2)
        stem = OrderedDict()
        prev = in_chans
        for i, (c, s) in enumerate(zip(stem_chs, strides)):
            stem[f"conv{i + 1}"] = conv(prev, c, 3, stride=s)
            if i != 3:
                stem[f"act{i + 2}"] = nn.SiLU()
            prev = c
        self.stem = nn.Sequential(stem)

        stages = []
        expected_var = 1.0
        for si, depth in enumerate(cfg.depths):
            stride = 1 if si == 0 else 2  # deep_quad stem stride is 4
            blocks = []
            for bi in range(depth):
                out_chs = make_divisible(cfg.channels[si], cfg.ch_div)
                blocks.append(NormFreeBlock(prev, out_chs, stride if bi == 0 else 1, cfg.alpha,
                                            1. / expected_var ** 0.5, cfg, conv))
                if bi == 0:
                    expected_var = 1.
                expected_var += cfg.alpha ** 2
                prev = out_chs
            stages.append(nn.Sequential(*blocks))
        self.stages = nn.Sequential(*stages)
        self.final_conv = conv(prev, cfg.num_features, 1)
        self.num_features = cfg.num_features

        # timm init: kaiming_normal_(fan_in, linear) on conv weights, zero bias
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="linear")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.stem(x)
        x = self.stages(x)
        x = F.silu(self.final_conv(x))
        return x.mean((2, 3))


class ImageEncoder(nn.Module):
    """reference networks.py:648-682 -- `.model` attribute holds the backbone (param names gain a
    `model.` prefix; irrelevant to the flat layout)."""

    def __init__(self, variant="nfnet_l0"):
        super().__init__()
        self.model = NormFreeNet(CONFIGS[variant])

    def forward(self, x):
        return self.model(x)


def randomize_like_trained(module: nn.Module, seed: int, gain_std=0.1, bias_std=0.02):
    """Default timm init leaves every conv3.gain at 0 and all biases at 0, which zeroes most of
    the theta-gradient; expert snapshots are TRAINED params.  Perturb gains/biases so synthetic
    theta exercises every term.  Deterministic under torch CPU RNG."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("gain"):
                p.copy_(1.0 + gain_std * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(bias_std * torch.randn(p.shape, generator=g))
    return module
