"""GPU parity tests of the ViT building-block operators (csrc/vit.hip, include/mdd_hip.h `mdd_op_layernorm*`,
`mdd_op_gelu*`, `mdd_op_softmax*`, `mdd_op_bgemm`; BASELINE configs[4]) against torch in fp64 on the host:
forward, backward, and the tangents of both (what the double backward of reference distill.py:606 needs), the
latter against `torch.func.jvp` of the op and of its vjp.  f32 storage to 2e-6, bf16 storage to the rounding of
its outputs (5e-3).  The last test composes the ops into one attention
layer exactly as the engine will address a fused qkv tensor (head slices by strides, no copies)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F
from torch.func import jvp

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from multimodal_dataset_distillation_amd import _lib
    return _lib, _lib.load()


def P(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def act(t, dtype):      # host fp64 -> device activation tensor of the storage type (bf16x2 = fp32 storage)
    return t.to(DEV, torch.bfloat16 if dtype == "bf16" else torch.float32).contiguous()


def back(t):            # device tensor -> host fp64
    return t.detach().double().cpu()


DT = {"f32": 0, "bf16": 1, "bf16x2": 2}
TOL = {"f32": 2e-6, "bf16": 5e-3}     # measured: f32 <= 1.3e-7, bf16 1.7e-3 (the rounding of its bf16 outputs)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("rows,dim", [(37, 64), (50, 192), (197 * 3, 768)])
def test_layernorm_forward_backward_and_tangents(dtype, rows, dim, report):
    mod, lib = _lib()
    g = torch.Generator().manual_seed(rows + dim)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x, xt, dy, dyt = rnd(rows, dim), rnd(rows, dim), rnd(rows, dim), rnd(rows, dim)
    gam, gamt, bet, bett = 1 + 0.2 * rnd(dim), 0.3 * rnd(dim), 0.1 * rnd(dim), 0.2 * rnd(dim)
    eps = 1e-6
    # storage rounding is part of the op's contract: the reference values start from what the device holds
    xd, xtd, dyd, dytd = act(x, dtype), act(xt, dtype), act(dy, dtype), act(dyt, dtype)
    x, xt, dy, dyt = back(xd), back(xtd), back(dyd), back(dytd)
    f = lambda a, b, c: F.layer_norm(a, (dim,), b, c, eps)
    y_ref, yt_ref = jvp(f, (x, gam, bet), (xt, gamt, bett))

    # backward and its tangent via functional autograd
    def grads(a, b, d):
        from torch.func import vjp
        out, pull = vjp(lambda aa, bb, cc: F.layer_norm(aa, (dim,), bb, cc, eps), a, b, torch.zeros(dim, dtype=torch.float64))
        return pull(d)
    (dx_ref, dg_ref, db_ref), (dxt_ref, dgt_ref, dbt_ref) = jvp(grads, (x, gam, dy), (xt, gamt, dyt))

    gd, gtd, bd, btd = (t.float().to(DEV) for t in (gam, gamt, bet, bett))
    y, yt = torch.empty_like(xd), torch.empty_like(xd)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mod.check(lib.mdd_op_layernorm(DT[dtype], rows, dim, eps, P(xd), None, P(gd), None, P(bd), None, P(y), None, st))
    mod.check(lib.mdd_op_layernorm(DT[dtype], rows, dim, eps, P(xd), P(xtd), P(gd), P(gtd), P(bd), P(btd), P(y), P(yt), st))
    dx, dxt = torch.empty_like(xd), torch.empty_like(xd)
    dg, dgt, db, dbt = (torch.zeros(dim, device=DEV) for _ in range(4))
    mod.check(lib.mdd_op_layernorm_bwd(DT[dtype], rows, dim, eps, P(xd), None, P(dyd), None, P(gd), None, None, None,
                                       P(dx), None, P(dg), None, P(db), None, st))
    mod.check(lib.mdd_op_layernorm_bwd(DT[dtype], rows, dim, eps, P(xd), P(xtd), P(dyd), P(dytd), P(gd), P(gtd), None, None,
                                       P(dx), P(dxt), P(dg), P(dgt), P(db), P(dbt), st))
    # with the gradient of the residual connection folded in (as the engine calls it): dx + res
    dxr, dxrt = torch.empty_like(xd), torch.empty_like(xd)
    scratch = [torch.zeros(dim, device=DEV) for _ in range(4)]
    mod.check(lib.mdd_op_layernorm_bwd(DT[dtype], rows, dim, eps, P(xd), None, P(dyd), None, P(gd), None, P(dyd), None,
                                       P(dxr), None, P(scratch[0]), None, P(scratch[1]), None, st))
    mod.check(lib.mdd_op_layernorm_bwd(DT[dtype], rows, dim, eps, P(xd), P(xtd), P(dyd), P(dytd), P(gd), P(gtd), P(dyd), P(dytd),
                                       P(dxr), P(dxrt), P(scratch[0]), P(scratch[2]), P(scratch[1]), P(scratch[3]), st))
    torch.cuda.synchronize()
    assert rel_err(back(dxr), dx_ref + dy) < TOL[dtype] and rel_err(back(dxrt), dxt_ref + dyt) < TOL[dtype]
    e = dict(y=rel_err(back(y), y_ref), y_t=rel_err(back(yt), yt_ref), dx=rel_err(back(dx), dx_ref),
             dgamma=rel_err(back(dg), dg_ref), dbeta=rel_err(back(db), db_ref), dx_t=rel_err(back(dxt), dxt_ref),
             dgamma_t=rel_err(back(dgt), dgt_ref), dbeta_t=rel_err(back(dbt), dbt_ref))
    report(f"layernorm {rows}x{dim} {dtype}: " + " ".join(f"{k} {float(v):.1e}" for k, v in e.items()))
    assert all(float(v) < TOL[dtype] for v in e.values()), e


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_add_layernorm_equals_add_then_layernorm(dtype, report):
    """mdd_op_add_layernorm (the residual sum formed by the LayerNorm kernel that reads it) against the two separate steps:
    bit-identical sum and output, primal and tangent call -- the sum is normalised as stored."""
    mod, lib = _lib()
    rows, dim, eps = 197 * 2, 768, 1e-6
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s_: torch.randn(*s_, generator=g, dtype=torch.float64)
    a, b, at, bt = (act(rnd(rows, dim), dtype) for _ in range(4))
    gam, gamt, bet, bett = ((t.float().to(DEV)) for t in (1 + 0.2 * rnd(dim), 0.3 * rnd(dim), 0.1 * rnd(dim), 0.2 * rnd(dim)))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # separate: s = a + b in the storage type, then LayerNorm
    s_ref = (a.float() + b.float()).to(a.dtype)
    st_ref = (at.float() + bt.float()).to(a.dtype)
    y_ref, yt_ref = torch.empty_like(a), torch.empty_like(a)
    mod.check(lib.mdd_op_layernorm(DT[dtype], rows, dim, eps, P(s_ref), None, P(gam), None, P(bet), None, P(y_ref), None, st))
    mod.check(lib.mdd_op_layernorm(DT[dtype], rows, dim, eps, P(s_ref), P(st_ref), P(gam), P(gamt), P(bet), P(bett), P(y_ref),
                                   P(yt_ref), st))
    s_, st_, y, yt = (torch.empty_like(a) for _ in range(4))
    mod.check(lib.mdd_op_add_layernorm(DT[dtype], rows, dim, eps, P(a), None, P(b), None, P(s_), None, P(gam), None, P(bet), None,
                                       P(y), None, st))
    mod.check(lib.mdd_op_add_layernorm(DT[dtype], rows, dim, eps, None, P(at), None, P(bt), P(s_), P(st_), P(gam), P(gamt), P(bet),
                                       P(bett), None, P(yt), st))
    torch.cuda.synchronize()
    for got, want, name in ((s_, s_ref, "s"), (st_, st_ref, "s_t"), (y, y_ref, "y"), (yt, yt_ref, "y_t")):
        assert torch.equal(got, want), name
    report(f"add + layernorm {rows}x{dim} {dtype}: bit-identical to add then layernorm (s, s_t, y, y_t)")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gelu_forward_backward_and_tangents(dtype, report):
    mod, lib = _lib()
    g = torch.Generator().manual_seed(3)
    n = 40 * 3072
    c, ct, ab, abt = (torch.randn(n, generator=g, dtype=torch.float64) * s for s in (1.5, 1.0, 1.0, 1.0))
    cd, ctd, abd, abtd = act(c, dtype), act(ct, dtype), act(ab, dtype), act(abt, dtype)
    c, ct, ab, abt = back(cd), back(ctd), back(abd), back(abtd)
    a_ref, at_ref = jvp(F.gelu, (c,), (ct,))

    def bwd(cc, gg):
        from torch.func import vjp
        _, pull = vjp(F.gelu, cc)
        return pull(gg)[0]
    cb_ref, cbt_ref = jvp(bwd, (c, ab), (ct, abt))
    a, at, cb, cbt = (torch.empty_like(cd) for _ in range(4))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mod.check(lib.mdd_op_gelu(DT[dtype], n, P(cd), None, P(a), None, st))
    mod.check(lib.mdd_op_gelu(DT[dtype], n, P(cd), P(ctd), P(a), P(at), st))
    mod.check(lib.mdd_op_gelu_bwd(DT[dtype], n, P(cd), None, P(abd), None, P(cb), None, st))
    mod.check(lib.mdd_op_gelu_bwd(DT[dtype], n, P(cd), P(ctd), P(abd), P(abtd), P(cb), P(cbt), st))
    torch.cuda.synchronize()
    e = dict(a=rel_err(back(a), a_ref), a_t=rel_err(back(at), at_ref), cbar=rel_err(back(cb), cb_ref),
             cbar_t=rel_err(back(cbt), cbt_ref))
    report(f"gelu {dtype}: " + " ".join(f"{k} {float(v):.1e}" for k, v in e.items()))
    assert all(float(v) < TOL[dtype] for v in e.values()), e


@pytest.mark.parametrize("rows,cols,ld", [(2 * 3 * 17, 17, 17), (12 * 197, 197, 200), (5, 512, 512)])
def test_softmax_forward_backward_and_tangents(rows, cols, ld, report):
    mod, lib = _lib()
    g = torch.Generator().manual_seed(cols)
    rnd = lambda: torch.randn(rows, cols, generator=g, dtype=torch.float64)
    s, st_, dp, dpt = rnd() * 3, rnd(), rnd(), rnd()
    scale = 0.125
    f = lambda a: torch.softmax(a * scale, -1)
    p_ref, pt_ref = jvp(f, (s,), (st_,))

    def bwd(a, gg):
        from torch.func import vjp
        _, pull = vjp(f, a)
        return pull(gg)[0]
    ds_ref, dst_ref = jvp(bwd, (s, dp), (st_, dpt))

    def pad(t):
        o = torch.zeros(rows, ld, dtype=torch.float32)
        o[:, :cols] = t.float()
        return o.to(DEV)
    sd, std, dpd, dptd = pad(s), pad(st_), pad(dp), pad(dpt)
    p, pt, ds, dst = (torch.zeros(rows, ld, device=DEV) for _ in range(4))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mod.check(lib.mdd_op_softmax(0, rows, cols, ld, scale, P(sd), None, P(p), None, st))
    mod.check(lib.mdd_op_softmax(0, rows, cols, ld, scale, P(sd), P(std), P(p), P(pt), st))
    # the backward consumes the probabilities and their tangents (as the engine will: both are stashed)
    mod.check(lib.mdd_op_softmax_bwd(0, rows, cols, ld, scale, P(p), None, P(dpd), None, P(ds), None, st))
    mod.check(lib.mdd_op_softmax_bwd(0, rows, cols, ld, scale, P(p), P(pt), P(dpd), P(dptd), P(ds), P(dst), st))
    torch.cuda.synchronize()
    cut = lambda t: back(t)[:, :cols]
    e = dict(p=rel_err(cut(p), p_ref), p_t=rel_err(cut(pt), pt_ref), ds=rel_err(cut(ds), ds_ref),
             ds_t=rel_err(cut(dst), dst_ref))
    report(f"softmax {rows}x{cols} (ld {ld}): " + " ".join(f"{k} {float(v):.1e}" for k, v in e.items()))
    assert all(float(v) < 2e-6 for v in e.values()), e
    assert float(p[:, cols:].abs().max() if ld > cols else 0.0) == 0.0      # the padding columns are never written


def test_softmax_on_bf16_scores(report):
    """the engine's bf16 mode keeps scores / probabilities / their gradients in bf16 (dtype 1)"""
    mod, lib = _lib()
    rows, cols, ld = 12 * 197, 197, 200
    g = torch.Generator().manual_seed(1)
    rnd = lambda: torch.randn(rows, cols, generator=g, dtype=torch.float64)
    pad = lambda t: torch.nn.functional.pad(t, (0, ld - cols)).to(DEV, torch.bfloat16).contiguous()
    sd, std, dpd, dptd = pad(rnd() * 3), pad(rnd()), pad(rnd()), pad(rnd())
    s, st_, dp, dpt = (back(t)[:, :cols] for t in (sd, std, dpd, dptd))
    scale = 0.125
    f = lambda a: torch.softmax(a * scale, -1)
    p_ref, pt_ref = jvp(f, (s,), (st_,))
    p, pt, ds, dst = (torch.zeros(rows, ld, device=DEV, dtype=torch.bfloat16) for _ in range(4))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mod.check(lib.mdd_op_softmax(1, rows, cols, ld, scale, P(sd), None, P(p), None, st))
    mod.check(lib.mdd_op_softmax(1, rows, cols, ld, scale, P(sd), P(std), P(p), P(pt), st))
    # backward from the ROUNDED probabilities the device holds (that is what the engine stashes)
    pr, ptr = back(p)[:, :cols], back(pt)[:, :cols]
    ds_ref = scale * pr * (dp - (pr * dp).sum(-1, keepdim=True))
    dst_ref = scale * (ptr * (dp - (pr * dp).sum(-1, keepdim=True)) + pr * (dpt - (ptr * dp + pr * dpt).sum(-1, keepdim=True)))
    mod.check(lib.mdd_op_softmax_bwd(1, rows, cols, ld, scale, P(p), None, P(dpd), None, P(ds), None, st))
    mod.check(lib.mdd_op_softmax_bwd(1, rows, cols, ld, scale, P(p), P(pt), P(dpd), P(dptd), P(ds), P(dst), st))
    torch.cuda.synchronize()
    e = dict(p=rel_err(pr, p_ref), p_t=rel_err(ptr, pt_ref), ds=rel_err(back(ds)[:, :cols], ds_ref),
             ds_t=rel_err(back(dst)[:, :cols], dst_ref))
    report("softmax on bf16 scores: " + " ".join(f"{k} {float(v):.1e}" for k, v in e.items()))
    assert all(float(v) < 5e-3 for v in e.values()), e


def _desc(mod, **kw):
    d = mod.MddBgemmDesc()
    for k, v in kw.items():
        setattr(d, k, v)
    return d


@pytest.mark.parametrize("dtype", ["f32", "bf16x2", "bf16"])
def test_attention_layer_composed_from_the_ops(dtype, report):
    """softmax(q k^T / sqrt(hd)) v per (image, head) on a FUSED qkv tensor [N*T, 3*D] (timm layout: q | k | v, heads
    contiguous inside each), forward, backward and both tangents, against torch."""
    mod, lib = _lib()
    N, T, H, hd = 3, 17, 2, 32
    D = H * hd
    scale = hd ** -0.5
    g = torch.Generator().manual_seed(9)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    qkv, qkvt, do, dot_ = rnd(N * T, 3 * D), rnd(N * T, 3 * D), rnd(N * T, D), rnd(N * T, D)
    qkvd, qkvtd, dod, dotd = act(qkv, dtype), act(qkvt, dtype), act(do, dtype), act(dot_, dtype)
    qkv, qkvt, do, dot_ = back(qkvd), back(qkvtd), back(dod), back(dotd)

    def attn(z):
        z5 = z.reshape(N, T, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = z5[0], z5[1], z5[2]
        p = torch.softmax((q @ k.transpose(-2, -1)) * scale, -1)
        return (p @ v).transpose(1, 2).reshape(N * T, D)
    o_ref, ot_ref = jvp(attn, (qkv,), (qkvt,))

    def bwd(z, gg):
        from torch.func import vjp
        _, pull = vjp(attn, z)
        return pull(gg)[0]
    dz_ref, dzt_ref = jvp(bwd, (qkv, do), (qkvt, dot_))

    es = 2 if dtype == "bf16" else 4
    adt = torch.bfloat16 if dtype == "bf16" else torch.float32
    ld = 20                                              # score rows padded to a multiple of four floats
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    off = lambda t, elems: C.c_void_p(t.data_ptr() + elems * es)
    S = torch.zeros(N * H * T, ld, device=DEV); St = torch.zeros_like(S)
    Pm = torch.zeros_like(S); Pt = torch.zeros_like(S)
    dP = torch.zeros_like(S); dPt = torch.zeros_like(S); dS = torch.zeros_like(S); dSt = torch.zeros_like(S)
    O = torch.zeros(N * T, D, device=DEV, dtype=adt); Ot = torch.zeros_like(O)
    dZ = torch.zeros(N * T, 3 * D, device=DEV, dtype=adt); dZt = torch.zeros_like(dZ)
    R = 3 * D            # row stride of the fused tensor
    qk = dict(outer=N, inner=H, a_outer=T * R, a_inner=hd, b_outer=T * R, b_inner=hd)
    # scores[n,h] (T x T, fp32) = q k^T : A = q (rows i, cols d), B(kk = d, j) = k[j, d]
    d_s = _desc(mod, m=T, n=T, k=hd, a_row=R, a_col=1, b_row=1, b_col=R, c_row=ld, c_col=1,
                c_outer=H * T * ld, c_inner=T * ld, alpha=1.0, **qk)
    # out[n,h] (T x hd) = p v : A = p fp32, B(kk = j, d) = v[j, d]
    d_o = _desc(mod, m=T, n=hd, k=T, outer=N, inner=H, a_row=ld, a_col=1, a_outer=H * T * ld, a_inner=T * ld,
                b_row=R, b_col=1, b_outer=T * R, b_inner=hd, c_row=D, c_col=1, c_outer=T * D, c_inner=hd, alpha=1.0)
    # dp = do v^T : A = do (T x hd), B(kk = d, j) = v[j, d]
    d_dp = _desc(mod, m=T, n=T, k=hd, outer=N, inner=H, a_row=D, a_col=1, a_outer=T * D, a_inner=hd,
                 b_row=1, b_col=R, b_outer=T * R, b_inner=hd, c_row=ld, c_col=1, c_outer=H * T * ld, c_inner=T * ld, alpha=1.0)
    # dv = p^T do : A(i = j, kk = i') = p[i', j], B = do
    d_dv = _desc(mod, m=T, n=hd, k=T, outer=N, inner=H, a_row=1, a_col=ld, a_outer=H * T * ld, a_inner=T * ld,
                 b_row=D, b_col=1, b_outer=T * D, b_inner=hd, c_row=R, c_col=1, c_outer=T * R, c_inner=hd, alpha=1.0)
    # dq = ds k ; dk = ds^T q
    d_dq = _desc(mod, m=T, n=hd, k=T, outer=N, inner=H, a_row=ld, a_col=1, a_outer=H * T * ld, a_inner=T * ld,
                 b_row=R, b_col=1, b_outer=T * R, b_inner=hd, c_row=R, c_col=1, c_outer=T * R, c_inner=hd, alpha=1.0)
    d_dk = _desc(mod, m=T, n=hd, k=T, outer=N, inner=H, a_row=1, a_col=ld, a_outer=H * T * ld, a_inner=T * ld,
                 b_row=R, b_col=1, b_outer=T * R, b_inner=hd, c_row=R, c_col=1, c_outer=T * R, c_inner=hd, alpha=1.0)
    q0, k0, v0 = 0, D, 2 * D
    bg = lambda a32, c32, d, A, At, B, Bt, Cc, Ct: mod.check(lib.mdd_op_bgemm(DT[dtype], a32, c32, C.byref(d), A, At, B, Bt, Cc, Ct, st))
    rows = N * H * T
    for tangent in (False, True):
        z, zt = qkvd, (qkvtd if tangent else None)
        tq = (lambda t, o: None if t is None else off(t, o))
        # forward
        bg(0, 1, d_s, off(z, q0), tq(zt, q0), off(z, k0), tq(zt, k0), P(S), P(St) if tangent else None)
        mod.check(lib.mdd_op_softmax(0, rows, T, ld, scale, P(S), P(St) if tangent else None, P(Pm), P(Pt) if tangent else None, st))
        bg(1, 0, d_o, P(Pm), P(Pt) if tangent else None, off(z, v0), tq(zt, v0), P(O), P(Ot) if tangent else None)
        # backward
        g_, gt_ = dod, (dotd if tangent else None)
        bg(0, 1, d_dp, P(g_), P(gt_), off(z, v0), tq(zt, v0), P(dP), P(dPt) if tangent else None)
        mod.check(lib.mdd_op_softmax_bwd(0, rows, T, ld, scale, P(Pm), P(Pt) if tangent else None, P(dP),
                                         P(dPt) if tangent else None, P(dS), P(dSt) if tangent else None, st))
        out, outt = dZ, (dZt if tangent else None)
        bg(1, 0, d_dv, P(Pm), P(Pt) if tangent else None, P(g_), P(gt_), off(out, v0), tq(outt, v0))
        bg(1, 0, d_dq, P(dS), P(dSt) if tangent else None, off(z, k0), tq(zt, k0), off(out, q0), tq(outt, q0))
        bg(1, 0, d_dk, P(dS), P(dSt) if tangent else None, off(z, q0), tq(zt, q0), off(out, k0), tq(outt, k0))
    torch.cuda.synchronize()
    e = dict(o=rel_err(back(O), o_ref), o_t=rel_err(back(Ot), ot_ref), dqkv=rel_err(back(dZ), dz_ref),
             dqkv_t=rel_err(back(dZt), dzt_ref))
    report(f"attention layer from the ops, N={N} T={T} H={H} hd={hd} {dtype}: " + " ".join(f"{k} {float(v):.1e}" for k, v in e.items()))
    # the score / probability tensors are fp32 in both modes; bf16 storage rounds q, k, v, do on the way in and o, dqkv out
    # bf16x2: fp32 storage, split-bf16 products on the matrix cores (measured ~1e-5)
    assert all(float(v) < {"f32": 2e-6, "bf16x2": 1e-4, "bf16": 5e-3}[dtype] for v in e.values()), e
