"""torch.optim.SGD semantics on flat fp32 device vectors through the C-ABI flat kernels.

  g += weight_decay * theta ; buf = g (first step) | momentum * buf + g ; theta -= lr * buf

Used by the stage-1 expert training step (reference buffer.py:59-60: SGD(lr, momentum=args.mom,
weight_decay=args.l2)), by evaluate_synset (reference epoch.py:361-362: momentum 0.9, wd 5e-4) and
by the stage-2 synthetic-set optimisers (reference distill.py:233-241: momentum 0.5).
"""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


_ONE = {}


def _one(device):
    key = str(device)
    if key not in _ONE:
        _ONE[key] = torch.ones(1, device=device)
    return _ONE[key]


def sgd_step(theta, grad, buf, lr, momentum=0.0, weight_decay=0.0, first=False, stream=None):
    """In place on theta / buf (and on grad when weight_decay != 0, like torch.optim.SGD's d_p)."""
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if stream is None else stream
    for t in (theta, grad, buf):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
    n = theta.numel()
    assert grad.numel() == n and buf.numel() == n
    if weight_decay:
        # g <- g + wd * theta
        _lib.check(lib.mdd_flat_axpy(_p(grad), _p(grad), _p(theta), _p(_one(theta.device)),
                                     float(weight_decay), n, st))
    _lib.check(lib.mdd_flat_sgd_momentum(_p(theta), _p(grad), _p(buf), float(lr), float(momentum),
                                         1 if first else 0, n, st))
