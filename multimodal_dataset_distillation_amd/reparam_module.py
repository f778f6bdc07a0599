"""ReparamModule -- drop-in for the reference's functional re-parametrisation wrapper
(reference reparam_module.py:9-159): `ReparamModule(module)(x, flat_param=theta)` runs `module`
with ALL of its parameters taken from the caller's flat vector, differentiably (twice).

Same public surface as the reference: ctor `ReparamModule(module)`, attributes `flat_param`,
`param_numel`, `_param_infos`, `_shared_param_infos`, `_param_numels`, `_param_shapes`,
`_buffer_infos`, methods `forward(*inputs, flat_param=None, buffers=None, **kw)`,
`clear_views()`, `trace()`.  Flatten order is `named_modules() x named_parameters(recurse=False)`
(reference :28-39), so expert buffers written by the reference's buffer.py load unchanged.

MI355X-native fast path: a wrapped module that implements `forward_flat(flat_param, *inputs)`
(our HIP-backed ImageEncoder / ProjectionHead in networks.py) receives the flat vector as ONE
tensor -- no per-parameter split/view/setattr (~225 python setattrs per call in the reference,
:110-115) and the HIP engine reads theta in place.  Any other nn.Module goes through the generic
split/view path with identical semantics to the reference.

Deliberate difference: `flat_param=None` falls back to `self.flat_param` (the reference crashes on
`torch.squeeze(None)`, :149 -- SURVEY Appendix B).
"""
from contextlib import contextmanager

import torch
import torch.nn as nn


def _resolve(root, path):
    m = root
    if path:
        for part in path.split("."):
            m = getattr(m, part)
    return m


class ReparamModule(nn.Module):
    def __init__(self, module):
        super().__init__()
        self.module = module
        infos, shared, tensors = [], [], []
        first_owner = {}
        for mod_path, mod in self.named_modules():
            for pname, p in mod.named_parameters(recurse=False):
                if p is None:
                    continue
                if p in first_owner:
                    shared.append((mod_path, pname) + first_owner[p])
                else:
                    first_owner[p] = (mod_path, pname)
                    infos.append((mod_path, pname))
                    tensors.append(p.detach())
        if len({t.dtype for t in tensors}) > 1:
            raise AssertionError("expects all parameters in module to have same dtype")
        self._param_infos = tuple(infos)
        self._shared_param_infos = tuple(shared)
        self._param_numels = tuple(t.numel() for t in tensors)
        self._param_shapes = tuple(t.size() for t in tensors)
        self._native = hasattr(module, "forward_flat")

        flat = nn.Parameter(torch.cat([t.reshape(-1) for t in tensors], 0))
        self.register_parameter("flat_param", flat)
        self.param_numel = flat.numel()

        # the wrapped module's own Parameters go away; views of a flat vector take their place
        for mod_path, pname in self._param_infos:
            delattr(_resolve(self, mod_path), pname)
        for mod_path, pname, _, _ in self._shared_param_infos:
            delattr(_resolve(self, mod_path), pname)
        self._bind_views(self.flat_param)

        self._buffer_infos = tuple((mp, bn, b) for mp, m in self.named_modules()
                                   for bn, b in m.named_buffers(recurse=False) if b is not None)
        self._traced_self = None

    # ------------------------------------------------------------------ views
    def _bind_views(self, flat_param):
        pieces = flat_param.split(self._param_numels)
        for (mod_path, pname), piece, shape in zip(self._param_infos, pieces, self._param_shapes):
            setattr(_resolve(self, mod_path), pname, piece.view(shape))
        for mod_path, pname, src_path, src_name in self._shared_param_infos:
            setattr(_resolve(self, mod_path), pname, getattr(_resolve(self, src_path), src_name))

    # reference name kept for callers that reach into it
    _unflatten_param = _bind_views

    def clear_views(self):
        for mod_path, pname in self._param_infos:
            setattr(_resolve(self, mod_path), pname, None)

    @contextmanager
    def unflattened_param(self, flat_param):
        saved = [getattr(_resolve(self, mp), pn) for mp, pn in self._param_infos]
        self._bind_views(flat_param)
        try:
            yield
        finally:
            for (mp, pn), old in zip(self._param_infos, saved):
                setattr(_resolve(self, mp), pn, old)
            for mp, pn, sp, sn in self._shared_param_infos:
                setattr(_resolve(self, mp), pn, getattr(_resolve(self, sp), sn))

    @contextmanager
    def replaced_buffers(self, buffers):
        for (mp, bn, _), nb in zip(self._buffer_infos, buffers):
            setattr(_resolve(self, mp), bn, nb)
        try:
            yield
        finally:
            for mp, bn, ob in self._buffer_infos:
                setattr(_resolve(self, mp), bn, ob)

    # ------------------------------------------------------------------ forward
    def _forward_with_param(self, flat_param, *inputs, **kwinputs):
        if self._native:
            return self.module.forward_flat(flat_param, *inputs, **kwinputs)
        with self.unflattened_param(flat_param):
            return self.module(*inputs, **kwinputs)

    def _forward_with_param_and_buffers(self, flat_param, buffers, *inputs, **kwinputs):
        with self.replaced_buffers(buffers):
            return self._forward_with_param(flat_param, *inputs, **kwinputs)

    def forward(self, *inputs, flat_param=None, buffers=None, **kwinputs):
        if flat_param is None:
            flat_param = self.flat_param
        # nn.DataParallel hands each replica a [1, P] row of the expanded theta (distill.py:515-517)
        flat_param = torch.squeeze(flat_param)
        if buffers is None:
            return self._forward_with_param(flat_param, *inputs, **kwinputs)
        return self._forward_with_param_and_buffers(flat_param, tuple(buffers), *inputs, **kwinputs)

    def trace(self, example_input, **trace_kwargs):
        """reference :77-98 (unused by distill.py).  The HIP path is already a single fused call;
        tracing is only meaningful for the generic path."""
        if self._native:
            return self
        assert self._traced_self is None, "This ReparamModule is already traced"
        if isinstance(example_input, torch.Tensor):
            example_input = (example_input,)
        example_param = (self.flat_param.detach().clone(),)
        self._traced_self = torch.jit.trace_module(
            self, inputs=dict(_forward_with_param=example_param + tuple(example_input)),
            **trace_kwargs)
        self._forward_with_param = self._traced_self._forward_with_param
        return self

    def _apply(self, *args, **kwargs):
        if self._traced_self is not None:
            self._traced_self._apply(*args, **kwargs)
            return self
        return super()._apply(*args, **kwargs)
