"""Twice-differentiable autograd ops over the HIP engine passes, so the reference's own loop
(distill.py:509-606: forward through `ReparamModule(...)(x, flat_param=theta)`,
`torch.autograd.grad(loss, theta, create_graph=True)`, `grand_loss.backward()`) runs unchanged
with every network FLOP in libmdd_hip.so.

Each network is ONE op with three levels:
  F  (theta, x)        -> y                         engine.*_forward        stash activations
  B  (theta, x, ybar)  -> (gtheta, gx)              engine.*_backward       stash backward signals
  BB cotangent (u) of gtheta -> cotangents of (theta, x, ybar):
       ybar-cot = J u                               engine.*_tangent_forward
       (theta-cot, x-cot) = d/de [J(theta+e u)^T ybar]   engine.*_tangent_backward (ybar_dot = 0)
The remaining second-order term (through y -> loss head -> ybar) is an ordinary B with a different
ybar: autograd calls F.backward again, which runs B with stash=False.
Cotangents arriving for gx (second order in x alone) are not produced by the reference's loop
(`autograd.grad` there is w.r.t. theta only) and are rejected loudly.  The B ops turn gradient
materialisation off (`ctx.set_materialize_grads(False)`), so "no cotangent" arrives as None and the
rejection needs no look at device values: there is NO host synchronisation in any backward.
"""
import torch


def _chk(t):
    return t.contiguous() if not t.is_contiguous() else t


class _ImgF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, x, eng, slot):
        theta_c, x_c = _chk(theta.detach()), _chk(x.detach())
        y = eng.img_forward(slot, theta_c, x_c)
        ctx.eng, ctx.slot = eng, slot
        ctx.save_for_backward(theta, x)
        ctx.first = True
        return y

    @staticmethod
    def backward(ctx, ybar):
        theta, x = ctx.saved_tensors
        # first call = the inner gradient (its signals are stashed for the tangent pass); later calls
        # are the outer backward's pass through y and must leave the stash alone.
        stash = ctx.first
        ctx.first = False
        gtheta, gx = _ImgB.apply(theta, x, ybar, ctx.eng, ctx.slot, stash)
        return gtheta, gx, None, None


class _ImgB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, x, ybar, eng, slot, stash):
        gx = torch.zeros_like(x)
        g = eng.img_backward(slot, _chk(theta.detach()), _chk(ybar.detach()), dimage=gx, stash=stash)
        ctx.eng, ctx.slot = eng, slot
        ctx.save_for_backward(theta, x)
        ctx.set_materialize_grads(False)
        return g, gx

    @staticmethod
    def backward(ctx, u, ux):
        theta, x = ctx.saved_tensors
        eng, slot = ctx.eng, ctx.slot
        if ux is not None:
            raise RuntimeError("second-order terms in the image alone are not part of this path")
        if u is None:
            return None, None, None, None, None, None
        th = _chk(theta.detach())
        u = _chk(u)
        ydot = eng.img_tangent_forward(slot, th, u)
        dx = torch.zeros_like(x)
        zeros = torch.zeros_like(ydot)
        h = eng.img_tangent_backward(slot, th, u, zeros, dimage=dx)
        return h, dx, ydot, None, None, None


class _TxtF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, x, eng, slot, mask):
        y = eng.txt_forward(slot, _chk(theta.detach()), _chk(x.detach()), None, mask)
        ctx.eng, ctx.slot = eng, slot
        ctx.save_for_backward(theta, x)
        ctx.first = True
        return y

    @staticmethod
    def backward(ctx, ybar):
        theta, x = ctx.saved_tensors
        stash = ctx.first
        ctx.first = False
        gtheta, gx = _TxtB.apply(theta, x, ybar, ctx.eng, ctx.slot, stash)
        return gtheta, gx, None, None, None


class _TxtB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, x, ybar, eng, slot, stash):
        gx = torch.zeros_like(x)
        g = eng.txt_backward(slot, _chk(theta.detach()), _chk(ybar.detach()), dtext=gx, stash=stash)
        ctx.eng, ctx.slot = eng, slot
        ctx.save_for_backward(theta, x)
        ctx.set_materialize_grads(False)
        return g, gx

    @staticmethod
    def backward(ctx, u, ux):
        theta, x = ctx.saved_tensors
        eng, slot = ctx.eng, ctx.slot
        if ux is not None:
            raise RuntimeError("second-order terms in the text embedding alone are not part of this path")
        if u is None:
            return None, None, None, None, None, None
        th = _chk(theta.detach())
        u = _chk(u)
        ydot = eng.txt_tangent_forward(slot, th, u)
        dx = torch.zeros_like(x)
        h = eng.txt_tangent_backward(slot, th, u, torch.zeros_like(ydot), dtext=dx)
        return h, dx, ydot, None, None, None


def image_encoder(eng, slot, theta, x):
    """y = NFNet(x; theta): [N,3,S,S] fp32 NCHW -> [N, feat] fp32."""
    return _ImgF.apply(theta, x, eng, slot)


def text_projection(eng, slot, theta, x, drop_mask=None):
    """y = ProjectionHead(x; theta) with an optional pre-scaled dropout mask."""
    return _TxtF.apply(theta, x, eng, slot, drop_mask)


class _Contrastive(torch.autograd.Function):
    """loss(x, y, s) of reference distill.py:533,546-551 on the HIP contrastive head, with
    first and second derivatives (the second through engine.contrastive_tangent)."""

    @staticmethod
    def forward(ctx, x, y, s, eng):
        L, xb, yb, sb = eng.contrastive(_chk(x.detach()), _chk(y.detach()), _chk(s.detach()).view(1))
        ctx.eng = eng
        ctx.save_for_backward(x, y, s)
        return L.view(())

    @staticmethod
    def backward(ctx, lbar):
        x, y, s = ctx.saved_tensors
        gx, gy, gs = _ContrastiveB.apply(x, y, s, ctx.eng)
        return lbar * gx, lbar * gy, lbar * gs, None


class _ContrastiveB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, s, eng):
        _, xb, yb, sb = eng.contrastive(_chk(x.detach()), _chk(y.detach()), _chk(s.detach()).view(1))
        ctx.eng = eng
        ctx.save_for_backward(x, y, s)
        ctx.set_materialize_grads(False)
        return xb, yb, sb.view(s.shape)

    @staticmethod
    def backward(ctx, ux, uy, us):
        # the Hessian of the loss in (x, y, s) is symmetric: cotangent (ux, uy, us) maps to
        # H [ux, uy, us]; the engine's tangent gives the (x, y) columns, the s column is closed-form
        x, y, s = ctx.saved_tensors
        eng = ctx.eng
        if us is not None:
            raise RuntimeError("second order in the logit scale alone is not part of this path")
        if ux is None and uy is None:
            return None, None, None, None
        xd = _chk(ux if ux is not None else torch.zeros_like(x))
        yd = _chk(uy if uy is not None else torch.zeros_like(y))
        xbd, ybd, sbd = eng.contrastive_tangent(_chk(x.detach()), _chk(y.detach()), xd, yd,
                                                _chk(s.detach()).view(1))
        return xbd, ybd, sbd.view(s.shape), None


def contrastive_loss(eng, x, y, scale):
    return _Contrastive.apply(x, y, scale, eng)
