"""GPU parity of the engine passes (through the C ABI) against the CPU oracle (torch autograd on
oracle/nfnet_ref.py + oracle/distill_ref.py), on the miniature `nfnet_tiny` topology, plus the
committed golden fixtures.  Tolerances: f32 mode 1e-3 relative (north_star's bar, norm-wise on
tensors); bf16 mode is reported and bounded loosely (operands carry 8 mantissa bits)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

# f32 / bf16x2: the north_star bar.  bf16: ~2x the errors measured on MI355X for these miniature cases
# (single-bf16 operands through two derivative orders; gpurun_out/test_report.txt)
TOL = {"f32": 1e-3, "bf16x2": 1e-3, "bf16": 8e-2}


def make_oracle(variant, d_txt, seed):
    from oracle import distill_ref as dr, nfnet_ref as nr
    torch.manual_seed(seed)
    enc = nr.ImageEncoder(variant)
    nr.randomize_like_trained(enc, seed + 1)
    head = dr.ProjectionHead(d_txt, enc.model.num_features)
    with torch.no_grad():
        head.layer_norm.weight.add_(0.1 * torch.randn_like(head.layer_norm.weight))
        head.layer_norm.bias.add_(0.1 * torch.randn_like(head.layer_norm.bias))
    return dr.FlatModule(enc), dr.FlatModule(head)


@pytest.fixture(scope="module", params=["f32", "bf16x2", "bf16"])
def setup(request):
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr
    dtype = request.param
    n, size, d_txt = 4, 64, 32
    fi, ft = make_oracle("nfnet_tiny", d_txt, 7)
    eng = UnrollEngine("nfnet_tiny", batch=n, num_queries=n, image_size=size, d_txt=d_txt,
                       syn_steps=2, dtype=dtype)
    img, txt = dr.synthetic_inputs(n, size, d_txt, seed=5)
    return dict(eng=eng, fi=fi, ft=ft, img=img, txt=txt, n=n, dtype=dtype)


BF16_TINY_IT1 = 0.3   # iteration 1 of the tiny golden in bf16 (pixels ~5e2 after one lr=1000 step): ~2x measured


def test_param_table_matches_oracle(setup):
    eng, fi, ft = setup["eng"], setup["fi"], setup["ft"]
    for tab, fm in ((eng.param_table("img"), fi), (eng.param_table("txt"), ft)):
        assert [t[0] for t in tab] == fm.names
        assert [t[1] for t in tab] == fm.shapes
        offs = np.cumsum([0] + fm.numels[:-1]).tolist()
        assert [t[2] for t in tab] == offs


def _cmp_buffers(eng, fi, th, img, report, dtype):
    """Layer-by-layer forward comparison (localises a failing kernel)."""
    acts = {}
    m = fi.module.model
    hooks = []

    def hook(name):
        def f(mod, inp, out):
            acts[name] = out.detach()
        return f
    for i in range(4):
        hooks.append(getattr(m.stem, f"conv{i + 1}").register_forward_hook(hook(f"stem{i}")))
    b = 0
    for st in m.stages:
        for blk in st:
            hooks.append(blk.conv1.register_forward_hook(hook(f"b{b}.C1")))
            hooks.append(blk.conv2.register_forward_hook(hook(f"b{b}.C2")))
            hooks.append(blk.conv2b.register_forward_hook(hook(f"b{b}.C2b")))
            hooks.append(blk.conv3.register_forward_hook(hook(f"b{b}.C3")))
            hooks.append(blk.register_forward_hook(hook(f"X{b + 1}")))
            b += 1
    hooks.append(m.final_conv.register_forward_hook(hook("CF")))
    with torch.no_grad():
        fi(img, flat_param=th)
    for h in hooks:
        h.remove()
    names = {"stem0": "stem.C0", "stem1": "stem.C1", "stem2": "stem.C2", "stem3": "X0"}
    worst = 0.0
    for k, ref in acts.items():
        bn = names.get(k, k)
        got = eng.buffer(bn, 0).float().cpu().view(ref.shape[0], ref.shape[2], ref.shape[3],
                                                   ref.shape[1]).permute(0, 3, 1, 2)
        e = rel_err(got, ref)
        worst = max(worst, e)
        report(f"  fwd buffer {dtype} {bn}: {e:.2e}")
    return worst


def test_image_passes(setup, report):
    eng, fi, img, n, dtype = setup["eng"], setup["fi"], setup["img"], setup["n"], setup["dtype"]
    tol = TOL[dtype]
    th = fi.flat_param()
    dev = "cuda"
    thd, imgd = th.to(dev), img.to(dev)
    idx = torch.tensor([2, 0, 3, 1])
    # ---- forward
    y = eng.img_forward(0, thd, imgd, idx.to(dev))
    thr = th.clone().requires_grad_(True)
    imr = img.clone().requires_grad_(True)
    y_ref = fi(imr[idx], flat_param=thr)
    worst = _cmp_buffers(eng, fi, th, img[idx], report, dtype)
    e_y = rel_err(y, y_ref)
    report(f"img forward {dtype}: feat {e_y:.2e} worst-buffer {worst:.2e}")
    # ---- backward
    torch.manual_seed(3)
    yb = torch.randn_like(y_ref)
    g = eng.img_backward(0, thd, yb.to(dev))
    g_ref, = torch.autograd.grad(y_ref, thr, yb, create_graph=True)
    e_g = rel_err(g, g_ref)
    # per-parameter breakdown
    off = 0
    bad = []
    for name, numel in zip(fi.names, fi.numels):
        e = rel_err(g[off:off + numel], g_ref[off:off + numel])
        if e > tol:
            bad.append((name, e))
        off += numel
    report(f"img backward {dtype}: gtheta {e_g:.2e}; params over tol: {bad[:12]}")
    # ---- tangent forward: J v
    v = torch.randn_like(th) * th.abs().mean()
    ydot = eng.img_tangent_forward(0, thd, v.to(dev))
    _, ydot_ref = torch.func.jvp(lambda t: fi(img[idx], flat_param=t), (th,), (v,))
    e_jv = rel_err(ydot, ydot_ref)
    report(f"img tangent-forward {dtype}: {e_jv:.2e}")
    # ---- tangent backward: d/de [ J(theta+e v)^T (yb + e ybd) ] and the same for d/d image
    ybd = torch.randn_like(yb)
    dimage = torch.zeros_like(imgd)
    h = eng.img_tangent_backward(0, thd, v.to(dev), ybd.to(dev), dimage=dimage, idx=idx.to(dev),
                                 mul=1.0)

    def G(t, ybar):
        xi = img[idx].clone()
        out, vjp = torch.func.vjp(lambda tt, xx: fi(xx, flat_param=tt), t, xi)
        return vjp(ybar)
    (_, _), (h_ref, dx_ref) = torch.func.jvp(G, (th, yb), (v, ybd))
    dimage_ref = torch.zeros_like(img)
    dimage_ref[idx] = dx_ref
    e_h, e_dx = rel_err(h, h_ref), rel_err(dimage, dimage_ref)
    report(f"img tangent-backward {dtype}: Hv {e_h:.2e} dimage {e_dx:.2e}")
    assert e_y < tol and e_g < tol and e_jv < tol and e_h < tol and e_dx < tol


def test_text_and_loss_passes(setup, report):
    eng, ft, txt, n, dtype = setup["eng"], setup["ft"], setup["txt"], setup["n"], setup["dtype"]
    from oracle import distill_ref as dr
    tol = 1e-3  # text head and contrastive head are fp32 in both modes
    dev = "cuda"
    th = ft.flat_param()
    idx = torch.tensor([1, 3, 0, 2])
    torch.manual_seed(9)
    mask = (torch.rand(n, eng.feature_dim) > 0.1).float() / 0.9
    y = eng.txt_forward(1, th.to(dev), txt.to(dev), idx.to(dev), mask.to(dev))
    thr = th.clone().requires_grad_(True)
    tx = txt.clone().requires_grad_(True)
    y_ref = ft(tx[idx], flat_param=thr, drop_mask=mask)
    e_y = rel_err(y, y_ref)
    yb = torch.randn_like(y_ref)
    g = eng.txt_backward(1, th.to(dev), yb.to(dev))
    g_ref, = torch.autograd.grad(y_ref, thr, yb)
    e_g = rel_err(g, g_ref)
    v = torch.randn_like(th) * 0.05
    ydot = eng.txt_tangent_forward(1, th.to(dev), v.to(dev))
    _, ydot_ref = torch.func.jvp(lambda t: ft(txt[idx], flat_param=t, drop_mask=mask), (th,), (v,))
    e_jv = rel_err(ydot, ydot_ref)
    ybd = torch.randn_like(yb)
    dtext = torch.zeros(n, txt.shape[1], device=dev)
    h = eng.txt_tangent_backward(1, th.to(dev), v.to(dev), ybd.to(dev), dtext=dtext,
                                 idx=idx.to(dev), mul=1.0)

    def G(t, ybar):
        xi = txt[idx].clone()
        out, vjp = torch.func.vjp(lambda tt, xx: ft(xx, flat_param=tt, drop_mask=mask), t, xi)
        return vjp(ybar)
    (_, _), (h_ref, dx_ref) = torch.func.jvp(G, (th, yb), (v, ybd))
    dt_ref = torch.zeros_like(txt)
    dt_ref[idx] = dx_ref
    e_h, e_dx = rel_err(h, h_ref), rel_err(dtext, dt_ref)
    report(f"txt passes {dtype}: fwd {e_y:.2e} bwd {e_g:.2e} jvp {e_jv:.2e} Hv {e_h:.2e} dtext {e_dx:.2e}")
    assert max(e_y, e_g, e_jv, e_h, e_dx) < tol

    # ---- contrastive head
    x = torch.randn(n, eng.feature_dim)
    yy = torch.randn(n, eng.feature_dim)
    s = torch.tensor(0.7)
    xr, yr, sr = x.clone().requires_grad_(True), yy.clone().requires_grad_(True), s.clone().requires_grad_(True)
    L_ref = dr.contrastive_loss(xr, yr, sr)
    gx, gy, gs = torch.autograd.grad(L_ref, [xr, yr, sr], create_graph=True)
    sd = s.view(1).to(dev)
    L, xb, yb2, sb = eng.contrastive(x.to(dev), yy.to(dev), sd)
    e0 = abs(L.item() - L_ref.item()) / abs(L_ref.item())
    e1, e2, e3 = rel_err(xb, gx), rel_err(yb2, gy), abs(sb.item() - gs.item()) / abs(gs.item())
    xd, yd = torch.randn_like(x), torch.randn_like(yy)
    xbd, ybd2, sbd = eng.contrastive_tangent(x.to(dev), yy.to(dev), xd.to(dev), yd.to(dev), sd)
    # directional derivative of (gx, gy, gs) along (xd, yd) by double backward
    def gradfun(a, b):
        a = a.clone(); b = b.clone()
        return torch.func.grad(lambda aa, bb, ss: dr.contrastive_loss(aa, bb, ss), argnums=(0, 1, 2))(a, b, s)
    _, (gxd, gyd, gsd) = torch.func.jvp(gradfun, (x, yy), (xd, yd))
    e4, e5 = rel_err(xbd, gxd), rel_err(ybd2, gyd)
    e6 = abs(sbd.item() - gsd.item()) / (abs(gsd.item()) + 1e-12)
    report(f"contrastive {dtype}: L {e0:.2e} xbar {e1:.2e} ybar {e2:.2e} sbar {e3:.2e} "
           f"T: {e4:.2e} {e5:.2e} {e6:.2e}")
    assert max(e0, e1, e2, e3, e4, e5, e6) < tol


@pytest.mark.parametrize("dtype", ["f32", "bf16x2", "bf16"])
def test_unrolled_match_golden_tiny(dtype, report):
    """tests/golden/unroll_tiny.npz: two consecutive outer iterations incl. SGD momentum."""
    from multimodal_dataset_distillation_amd import _lib
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, "unroll_tiny.npz"))
    n, size, d_txt, K = int(g["n"]), int(g["size"]), int(g["d_txt"]), int(g["K"])
    dev = "cuda"
    eng = UnrollEngine(str(g["variant"]), batch=n, num_queries=n, image_size=size, d_txt=d_txt,
                       syn_steps=K, dtype=dtype)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    image_syn, text_syn = T("image_syn0").clone(), T("text_syn0").clone()
    lr = torch.tensor([0.1, 0.1], device=dev)
    th0i, th0t, tgi, tgt = T("theta0_img"), T("theta0_txt"), T("target_img"), T("target_txt")
    bufs = [torch.zeros_like(image_syn), torch.zeros_like(text_syn), torch.zeros(2, device=dev)]
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    tol = TOL[dtype]
    for it in range(2):
        perms = torch.from_numpy(g["perms"][it]).to(dev)
        out = eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms)
        torch.cuda.synchronize()
        e = dict(
            grand=abs(out["grand_loss"].item() - g[f"it{it}_grand"]) / abs(g[f"it{it}_grand"]),
            ces=rel_err(out["contrastive"], torch.from_numpy(g[f"it{it}_ces"])),
            g_img=rel_err(out["image_syn"], torch.from_numpy(g[f"it{it}_g_image_syn"])),
            g_txt=rel_err(out["text_syn"], torch.from_numpy(g[f"it{it}_g_text_syn"])),
            g_lri=abs(out["lr"][0].item() - g[f"it{it}_g_lr_img"]) / abs(g[f"it{it}_g_lr_img"]),
            g_lrt=abs(out["lr"][1].item() - g[f"it{it}_g_lr_txt"]) / abs(g[f"it{it}_g_lr_txt"]),
        )
        report(f"unrolled_match golden tiny {dtype} it{it}: " + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
        # bf16: iteration 1 starts from pixels of magnitude ~5e2 (one lr=1000 step on a toy net);
        # operand rounding then dominates d/d(image) -- reported, bounded loosely.
        tol_it = tol if (dtype != "bf16" or it == 0) else BF16_TINY_IT1
        assert all(float(v) < tol_it for v in e.values()), e
        # the three SGD(momentum=0.5) steps (distill.py:233-241, 611-613) through the C ABI
        for p, gr, b, lrv in ((image_syn, out["image_syn"], bufs[0], 1000.0),
                              (text_syn, out["text_syn"], bufs[1], 1000.0), (lr, out["lr"], bufs[2], 1e-3)):
            _lib.check(lib.mdd_flat_sgd_momentum(P(p), P(gr), P(b), lrv, 0.5, 1 if it == 0 else 0,
                                                 p.numel(), st))
        e_after = max(rel_err(image_syn, torch.from_numpy(g[f"it{it}_image_syn_after"])),
                      rel_err(text_syn, torch.from_numpy(g[f"it{it}_text_syn_after"])),
                      rel_err(lr, torch.from_numpy(g[f"it{it}_lr_after"])))
        report(f"  after SGD step: {e_after:.2e}")
        assert e_after < tol
        # continue from the golden state so iteration 1 checks ONE iteration, not compounded drift
        image_syn.copy_(T(f"it{it}_image_syn_after"))
        text_syn.copy_(T(f"it{it}_text_syn_after"))
        lr.copy_(T(f"it{it}_lr_after"))


def test_unrolled_match_minibatch_subset_dropout_and_fixed_scale(report):
    """Edge cases of the loop the goldens do not exercise: mini_batch_size < num_queries (index subsets
    that overlap between steps, so image/text gradients of one row accumulate over steps -- reference
    distill.py:510-513), student dropout masks (distill.py:446-447 puts the text projection in train
    mode) and the upstream constant logit scale (distill_original.py:430).  f32 mode vs the oracle."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr
    nq, batch, size, d_txt, K = 7, 4, 64, 32, 3
    fi, ft = make_oracle("nfnet_tiny", d_txt, 21)
    feat = ft.module.fc.out_features
    img, txt = dr.synthetic_inputs(nq, size, d_txt, seed=9)
    g = torch.Generator().manual_seed(5)
    perms = [torch.randperm(nq, generator=g)[:batch] for _ in range(K)]
    masks = [(torch.rand(batch, feat, generator=g) >= 0.1).float() / 0.9 for _ in range(K)]
    th0i, th0t = fi.flat_param(), ft.flat_param()
    tgi = th0i + 2e-3 * torch.randn(th0i.shape, generator=g)
    tgt = th0t + 2e-3 * torch.randn(th0t.shape, generator=g)
    dev = "cuda"
    for scale in (None, 1.0 / 0.07):
        im, tx = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
        lri = torch.tensor(0.1, requires_grad=True); lrt = torch.tensor(0.05, requires_grad=True)
        grand, info = dr.unrolled_match(fi, ft, im, tx, lri, lrt, th0i, th0t, tgi, tgt, perms,
                                        drop_masks=masks, logit_scale=scale)
        if scale is None:
            gi, gt_, gli, glt = dr.outer_grads(grand, im, tx, lri, lrt)
        else:   # a constant scale takes syn_lr_img out of the logits: its gradient comes from the updates only
            gi, gt_, gli, glt = torch.autograd.grad(grand, [im, tx, lri, lrt])
        eng = UnrollEngine("nfnet_tiny", batch=batch, num_queries=nq, image_size=size, d_txt=d_txt,
                           syn_steps=K, dtype="f32")
        lr = torch.tensor([0.1, 0.05], device=dev)
        out = eng.unrolled_match(img.to(dev), txt.to(dev), lr[0:1], lr[1:2], th0i.to(dev), th0t.to(dev),
                                 tgi.to(dev), tgt.to(dev), perms=torch.stack(perms).to(dev),
                                 drop_masks=torch.stack(masks).to(dev), logit_scale=scale)
        torch.cuda.synchronize()
        e = dict(grand=abs(out["grand_loss"].item() - grand.item()) / abs(grand.item()),
                 ces=rel_err(out["contrastive"], torch.stack(info["contrastive"])),
                 g_img=rel_err(out["image_syn"], gi), g_txt=rel_err(out["text_syn"], gt_),
                 g_lri=abs(out["lr"][0].item() - gli.item()) / abs(gli.item()),
                 g_lrt=abs(out["lr"][1].item() - glt.item()) / abs(glt.item()))
        # rows never drawn by any step must have exactly zero gradient
        used = torch.zeros(nq, dtype=torch.bool); used[torch.cat(perms)] = True
        assert (out["image_syn"].cpu()[~used] == 0).all() and (out["text_syn"].cpu()[~used] == 0).all()
        report(f"unrolled_match subset batch {batch}/{nq}, dropout, scale={'lr_img' if scale is None else '1/0.07'}: "
               + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
        assert all(float(v) < 1e-3 for v in e.values()), e
        eng.close()


def test_stash_policy_recompute_matches_keep_all_and_the_oracle(report):
    """SURVEY 7.5 / VERDICT r1 item 7: with keep_steps < syn_steps the reverse sweep recomputes the
    forward pass + inner gradient of the steps whose activations were not kept.  nfnet_tiny, K=4,
    minibatch subsets + dropout masks, f32 mode: every policy must give the oracle's result (1e-3)
    and agree with keep-all to summation-order noise."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr
    nq, batch, size, d_txt, K = 6, 4, 64, 32, 4
    fi, ft = make_oracle("nfnet_tiny", d_txt, 33)
    feat = ft.module.fc.out_features
    img, txt = dr.synthetic_inputs(nq, size, d_txt, seed=12)
    g = torch.Generator().manual_seed(8)
    perms = [torch.randperm(nq, generator=g)[:batch] for _ in range(K)]
    masks = [(torch.rand(batch, feat, generator=g) >= 0.1).float() / 0.9 for _ in range(K)]
    th0i, th0t = fi.flat_param(), ft.flat_param()
    tgi = th0i + 2e-3 * torch.randn(th0i.shape, generator=g)
    tgt = th0t + 2e-3 * torch.randn(th0t.shape, generator=g)
    im, tx = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    lri = torch.tensor(0.1, requires_grad=True); lrt = torch.tensor(0.05, requires_grad=True)
    grand, info = dr.unrolled_match(fi, ft, im, tx, lri, lrt, th0i, th0t, tgi, tgt, perms, drop_masks=masks)
    gi, gt_, gli, glt = dr.outer_grads(grand, im, tx, lri, lrt)
    dev = "cuda"
    outs = {}
    for keep in (None, 0, 2, 3):
        eng = UnrollEngine("nfnet_tiny", batch=batch, num_queries=nq, image_size=size, d_txt=d_txt,
                           syn_steps=K, dtype="f32", keep_steps=keep)
        lr = torch.tensor([0.1, 0.05], device=dev)
        out = eng.unrolled_match(img.to(dev), txt.to(dev), lr[0:1], lr[1:2], th0i.to(dev), th0t.to(dev),
                                 tgi.to(dev), tgt.to(dev), perms=torch.stack(perms).to(dev),
                                 drop_masks=torch.stack(masks).to(dev))
        torch.cuda.synchronize()
        e = dict(grand=abs(out["grand_loss"].item() - grand.item()) / abs(grand.item()),
                 ces=rel_err(out["contrastive"], torch.stack(info["contrastive"])),
                 g_img=rel_err(out["image_syn"], gi), g_txt=rel_err(out["text_syn"], gt_),
                 g_lri=abs(out["lr"][0].item() - gli.item()) / abs(gli.item()),
                 g_lrt=abs(out["lr"][1].item() - glt.item()) / abs(glt.item()))
        report(f"stash policy keep_steps={keep} (slots {eng.num_slots}, ws {eng.workspace_bytes / 2**20:.0f} MiB): "
               + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
        assert all(float(v) < 1e-3 for v in e.values()), (keep, e)
        outs[keep] = {k: out[k].detach().clone() for k in ("image_syn", "text_syn", "lr")}
        eng.close()
    for keep in (0, 2, 3):
        d = max(rel_err(outs[keep][k], outs[None][k]) for k in outs[None])
        assert d < 1e-5, (keep, d)


def test_unrolled_match_with_fewer_steps_than_the_engine_was_built_for(report):
    """mdd_iter_args.syn_steps may be smaller than the engine's unroll depth (the workspace is planned for the
    maximum): K=4 engine, 2-step call, with and without the recompute policy, against the 2-step oracle."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr
    n, size, d_txt, Kmax, Ks = 4, 64, 32, 4, 2
    fi, ft = make_oracle("nfnet_tiny", d_txt, 41)
    img, txt = dr.synthetic_inputs(n, size, d_txt, seed=14)
    g = torch.Generator().manual_seed(9)
    perms = [torch.randperm(n, generator=g) for _ in range(Ks)]
    th0i, th0t = fi.flat_param(), ft.flat_param()
    tgi = th0i + 2e-3 * torch.randn(th0i.shape, generator=g)
    tgt = th0t + 2e-3 * torch.randn(th0t.shape, generator=g)
    im, tx = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    lri = torch.tensor(0.1, requires_grad=True); lrt = torch.tensor(0.1, requires_grad=True)
    grand, _ = dr.unrolled_match(fi, ft, im, tx, lri, lrt, th0i, th0t, tgi, tgt, perms)
    gi, gt_, gli, glt = dr.outer_grads(grand, im, tx, lri, lrt)
    dev = "cuda"
    for keep in (None, 1):
        eng = UnrollEngine("nfnet_tiny", batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=Kmax,
                           dtype="f32", keep_steps=keep)
        lr = torch.tensor([0.1, 0.1], device=dev)
        out = eng.unrolled_match(img.to(dev), txt.to(dev), lr[0:1], lr[1:2], th0i.to(dev), th0t.to(dev),
                                 tgi.to(dev), tgt.to(dev), perms=torch.stack(perms).to(dev), syn_steps=Ks)
        torch.cuda.synchronize()
        e = dict(grand=abs(out["grand_loss"].item() - grand.item()) / abs(grand.item()),
                 g_img=rel_err(out["image_syn"], gi), g_txt=rel_err(out["text_syn"], gt_),
                 g_lr=rel_err(out["lr"], torch.stack([gli, glt])))
        report(f"2-step call on a 4-step engine (keep_steps={keep}): " + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
        assert all(float(v) < 1e-3 for v in e.values()), (keep, e)
        eng.close()


def test_nfnet_l1_topology_matches_the_oracle(report):
    """BASELINE config 4's image encoder (the l0 recipe at depths (2,4,12,6), feat_mult 2 -> 3072 features;
    build-defined: timm 0.6.7 has no plain nfnet_l1 and the reference no working path for it) on a small
    instance: one unrolled step + outer gradients in f32 mode against the oracle's nfnet_l1."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr
    n, size, d_txt, K = 4, 64, 48, 1
    fi, ft = make_oracle("nfnet_l1", d_txt, 51)
    assert ft.module.fc.out_features == 3072
    img, txt = dr.synthetic_inputs(n, size, d_txt, seed=15)
    g = torch.Generator().manual_seed(10)
    perms = [torch.randperm(n, generator=g) for _ in range(K)]
    th0i, th0t = fi.flat_param(), ft.flat_param()
    tgi = th0i + 1e-3 * torch.randn(th0i.shape, generator=g)
    tgt = th0t + 1e-3 * torch.randn(th0t.shape, generator=g)
    im, tx = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    lri = torch.tensor(0.1, requires_grad=True); lrt = torch.tensor(0.1, requires_grad=True)
    grand, info = dr.unrolled_match(fi, ft, im, tx, lri, lrt, th0i, th0t, tgi, tgt, perms)
    gi, gt_, gli, glt = dr.outer_grads(grand, im, tx, lri, lrt)
    dev = "cuda"
    eng = UnrollEngine("nfnet_l1", batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K, dtype="f32")
    assert eng.P_img == th0i.numel() and eng.feature_dim == 3072
    lr = torch.tensor([0.1, 0.1], device=dev)
    out = eng.unrolled_match(img.to(dev), txt.to(dev), lr[0:1], lr[1:2], th0i.to(dev), th0t.to(dev),
                             tgi.to(dev), tgt.to(dev), perms=torch.stack(perms).to(dev))
    torch.cuda.synchronize()
    e = dict(grand=abs(out["grand_loss"].item() - grand.item()) / abs(grand.item()),
             ces=rel_err(out["contrastive"], torch.stack(info["contrastive"])),
             g_img=rel_err(out["image_syn"], gi), g_txt=rel_err(out["text_syn"], gt_),
             g_lr=rel_err(out["lr"], torch.stack([gli, glt])))
    report(f"nfnet_l1 ({eng.P_img} params, 3072 features) N={n} @{size}: " + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
    assert all(float(v) < 1e-3 for v in e.values()), e
    eng.close()


def _live_oracle_case(variant, nq, batch, size, d_txt, K, seed):
    from oracle import distill_ref as dr
    fi, ft = make_oracle(variant, d_txt, seed)
    img, txt = dr.synthetic_inputs(nq, size, d_txt, seed=seed + 3)
    g = torch.Generator().manual_seed(seed + 5)
    perms = [torch.randperm(nq, generator=g)[:batch] for _ in range(K)]
    th0i, th0t = fi.flat_param(), ft.flat_param()
    # expert displacement with the norm of the student's own K-step move (as oracle/gen_golden.py does): the
    # normalised matching loss is then O(1)-sensitive to what the kernels compute (a fixed 1e-3 noise swamps it)
    a, b = th0i.clone().requires_grad_(True), th0t.clone().requires_grad_(True)
    l0 = dr.contrastive_loss(fi(img[perms[0]], flat_param=a), ft(txt[perms[0]], flat_param=b), 0.1)
    gi0, gt0 = torch.autograd.grad(l0, [a, b])
    tgi = th0i + float(0.1 * K * gi0.norm() / th0i.numel() ** 0.5) * torch.randn(th0i.shape, generator=g)
    tgt = th0t + float(0.07 * K * gt0.norm() / th0t.numel() ** 0.5) * torch.randn(th0t.shape, generator=g)
    im, tx = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    lri = torch.tensor(0.1, requires_grad=True); lrt = torch.tensor(0.07, requires_grad=True)
    grand, info = dr.unrolled_match(fi, ft, im, tx, lri, lrt, th0i, th0t, tgi, tgt, perms)
    assert abs(float(grand.detach()) - 2.0) > 0.05, "the matching loss must depend on the student's move"
    gi, gt_, gli, glt = dr.outer_grads(grand, im, tx, lri, lrt)
    want = dict(grand=grand.detach(), ces=torch.stack(info["contrastive"]).detach(), g_img=gi, g_txt=gt_,
                g_lr=torch.stack([gli, glt]))
    return dict(img=img, txt=txt, perms=torch.stack(perms), th0i=th0i, th0t=th0t, tgi=tgi, tgt=tgt, want=want)


def _run_case(variant, nq, batch, size, d_txt, K, dtype, case):
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    dev = "cuda"
    eng = UnrollEngine(variant, batch=batch, num_queries=nq, image_size=size, d_txt=d_txt, syn_steps=K, dtype=dtype)
    lr = torch.tensor([0.1, 0.07], device=dev)
    out = eng.unrolled_match(case["img"].to(dev), case["txt"].to(dev), lr[0:1], lr[1:2], case["th0i"].to(dev),
                             case["th0t"].to(dev), case["tgi"].to(dev), case["tgt"].to(dev),
                             perms=case["perms"].to(dev))
    torch.cuda.synchronize()
    w = case["want"]
    e = dict(grand=abs(out["grand_loss"].item() - w["grand"].item()) / abs(w["grand"].item()),
             ces=rel_err(out["contrastive"], w["ces"]), g_img=rel_err(out["image_syn"], w["g_img"]),
             g_txt=rel_err(out["text_syn"], w["g_txt"]), g_lr=rel_err(out["lr"], w["g_lr"]))
    used = torch.zeros(nq, dtype=torch.bool); used[case["perms"].flatten()] = True
    assert (out["image_syn"].cpu()[~used] == 0).all() and (out["text_syn"].cpu()[~used] == 0).all()
    eng.close()
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("variant,nq,batch,size,d_txt,K", [
    ("nfnet_tiny", 2, 2, 32, 8, 1),       # the smallest the engine accepts: two pairs, one 1x1 final stage
    ("nfnet_tiny", 9, 7, 160, 24, 2),     # odd counts, 5x5 final stage, subsets that overlap between steps
    ("nfnet_l0", 5, 3, 96, 40, 2),        # full-width channels on ragged M: 1728 / 432 / 108 / 27 rows per stage
])
def test_ragged_shapes_against_the_live_oracle(variant, nq, batch, size, d_txt, K, report):
    """Edge shapes the goldens do not hold (their row counts are multiples of every tile): M-tile tails at every
    stage, stages smaller than one tile, a minibatch that is a strict subset of an odd number of pairs.  The
    oracle runs here on the host (seconds).  Bars: f32 mode 1e-4 (measured <= 5.4e-6), bf16x2 1e-3 (measured
    <= 7.2e-5 at these sizes) on every output."""
    case = _live_oracle_case(variant, nq, batch, size, d_txt, K, seed=61)
    for dtype, tol_s, tol_g in (("f32", 1e-4, 1e-4), ("bf16x2", 1e-3, 1e-3)):
        e = _run_case(variant, nq, batch, size, d_txt, K, dtype, case)
        report(f"ragged {variant} nq={nq} batch={batch} @{size} K={K} {dtype}: "
               + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
        assert float(e["grand"]) < tol_s and float(e["ces"]) < tol_s and float(e["g_lr"]) < tol_g, e
        assert float(e["g_img"]) < tol_g and float(e["g_txt"]) < tol_g, e
