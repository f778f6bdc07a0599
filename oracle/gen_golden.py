"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz.  Run in the build container (needs
/root/reference); the GPU box only ever sees the committed .npz files.

What is taken from the reference itself (imported / executed here, never copied into the repo):
  * `ReparamModule` -- imported from /root/reference/reparam_module.py (depends on torch only).
  * `ProjectionHead` -- the class definition is AST-extracted from /root/reference/networks.py
    :625-646 and executed against torch.nn (the module as a whole cannot be imported: its
    import-time `BertModel.from_pretrained` / `import clip` / `import timm` are unavailable).
The loop around them restates reference distill.py:509-598 (it is inline code in `main()` and
`distill.py` cannot be imported -- see SURVEY.md 8c).  Every golden is first cross-checked against
oracle/distill_ref.py (the restatement the tests use) and generation aborts on mismatch.

The image encoder is oracle/nfnet_ref.py (timm absent => PARITY UNPINNED for that part); it is
wrapped in the REAL ReparamModule here, so flatten order / view semantics are the reference's.
"""
import ast
import copy
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import distill_ref as dr  # noqa: E402
from oracle import nfnet_ref as nr  # noqa: E402


def load_reference_pieces():
    sys.path.insert(0, REF)
    import reparam_module  # the reference's own file
    src = open(os.path.join(REF, "networks.py")).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ProjectionHead"][0]
    ns = {"nn": nn, "torch": torch, "F": F}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "networks.py:ProjectionHead", "exec"), ns)
    return reparam_module.ReparamModule, ns["ProjectionHead"]


def ref_loop(img_net, txt_net, image_syn, text_syn, lr_img, lr_txt, th0_img, th0_txt, tgt_img,
             tgt_txt, perms, fixed_x=None):
    """reference distill.py:509-598 restated around the reference's ReparamModule objects.
    fixed_x: list of constant image features per step (text-only golden) or None."""
    img_p = [th0_img.clone().requires_grad_(True)] if img_net is not None else None
    txt_p = [th0_txt.clone().requires_grad_(True)]
    ces = []
    for k, idx in enumerate(perms):
        this_y = text_syn[idx]
        if img_net is not None:
            x = img_net(image_syn[idx], flat_param=img_p[-1])
        else:
            x = fixed_x[k]
        x = x / x.norm(dim=1, keepdim=True)
        this_y = txt_net(this_y, flat_param=txt_p[-1])
        this_y = this_y / this_y.norm(dim=1, keepdim=True)
        image_logits = lr_img * x.float() @ this_y.float().t()
        gt = torch.arange(len(image_logits)).type_as(image_logits).long()
        loss = (F.cross_entropy(image_logits, gt) + F.cross_entropy(image_logits.t(), gt)) / 2
        if img_net is not None:
            ig = torch.autograd.grad(loss, img_p[-1], create_graph=True)[0]
        tg = torch.autograd.grad(loss, txt_p[-1], create_graph=True)[0]
        ces.append(loss.item())
        if img_net is not None:
            img_p.append(img_p[-1] - lr_img * ig)
        txt_p.append(txt_p[-1] - lr_txt * tg)
    txt_loss = F.mse_loss(txt_p[-1], tgt_txt, reduction="sum") / \
        F.mse_loss(th0_txt, tgt_txt, reduction="sum")
    if img_net is not None:
        img_loss = F.mse_loss(img_p[-1], tgt_img, reduction="sum") / \
            F.mse_loss(th0_img, tgt_img, reduction="sum")
        grand = img_loss + txt_loss
    else:
        img_loss = torch.tensor(0.0)
        grand = txt_loss
    return grand, img_loss, txt_loss, ces, (img_p[-1] if img_p else None), txt_p[-1]


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def gen_text_only(ReparamModule, RefHead, path):
    """Pinned 100% by reference code: ReparamModule(ProjectionHead) + the loop, image features
    held constant.  Covers SURVEY 8a rows a4,a6,a7,a8(txt),a9,a10,a11(txt)."""
    torch.manual_seed(11)
    n, d_in, d_out, K = 6, 32, 288, 3   # d_out = feature dim of the nfnet_tiny engine
    head = RefHead(d_in, d_out, dropout=0.0)
    with torch.no_grad():
        head.layer_norm.weight.add_(0.1 * torch.randn(d_out))
        head.layer_norm.bias.add_(0.1 * torch.randn(d_out))
    head_state = copy.deepcopy(head.state_dict())   # ReparamModule strips the params below
    net = ReparamModule(head)
    net.train()
    th0 = net.flat_param.detach().clone()
    text_syn = (torch.randn(n, d_in) * 0.5253 - 0.0094).requires_grad_(True)
    lr_img = torch.tensor(0.7, requires_grad=True)   # doubles as logit scale (distill.py:548)
    lr_txt = torch.tensor(0.3, requires_grad=True)
    xs = [torch.randn(n, d_out) for _ in range(K)]
    # expert displacement with the norm of the student's own K-step move (loss ratio is O(1)-sensitive)
    a0 = th0.clone().requires_grad_(True)
    y0 = net(text_syn.detach(), flat_param=a0)
    y0 = y0 / y0.norm(dim=1, keepdim=True)
    x0 = xs[0] / xs[0].norm(dim=1, keepdim=True)
    lg = 0.7 * x0 @ y0.t()
    l0 = (F.cross_entropy(lg, torch.arange(n)) + F.cross_entropy(lg.t(), torch.arange(n))) / 2
    g0, = torch.autograd.grad(l0, a0)
    tgt = th0 + float(0.3 * K * g0.norm() / th0.numel() ** 0.5) * torch.randn_like(th0)
    perms = [torch.randperm(n) for _ in range(K)]
    grand, _, txt_loss, ces, _, thK = ref_loop(None, net, None, text_syn, lr_img, lr_txt, None,
                                              th0, None, tgt, perms, fixed_x=xs)
    g_txt, g_s, g_lr = torch.autograd.grad(grand, [text_syn, lr_img, lr_txt])

    # cross-check the restatement used by the tests
    oh = dr.ProjectionHead(d_in, d_out)
    oh.load_state_dict(head_state)
    fo = dr.FlatModule(oh)
    assert fo.param_numel == net.param_numel and torch.equal(fo.flat_param(), th0)
    assert [tuple(s) for s in net._param_shapes] == fo.shapes
    tp = [th0.clone().requires_grad_(True)]
    for k, idx in enumerate(perms):
        y = fo(text_syn[idx], flat_param=tp[-1])
        loss = dr.contrastive_loss(xs[k], y, lr_img)
        tg = torch.autograd.grad(loss, tp[-1], create_graph=True)[0]
        assert abs(loss.item() - ces[k]) < 1e-6
        tp.append(tp[-1] - lr_txt * tg)
    assert torch.allclose(tp[-1], thK, atol=1e-7)

    np.savez(path, n=n, d_in=d_in, d_out=d_out, K=K, theta0=f32(th0), target=f32(tgt),
             text_syn=f32(text_syn), lr_img=f32(lr_img), lr_txt=f32(lr_txt),
             xs=np.stack([f32(x) for x in xs]), perms=np.stack([p.numpy() for p in perms]),
             ces=np.array(ces, np.float32), grand=f32(grand), theta_K=f32(thK),
             g_text_syn=f32(g_txt), g_lr_img=f32(g_s), g_lr_txt=f32(g_lr))
    print("wrote", path, "grand", grand.item())


def build_pair(ReparamModule, RefHead, variant, d_txt, seed):
    torch.manual_seed(seed)
    enc = nr.ImageEncoder(variant)
    nr.randomize_like_trained(enc, seed + 1)
    head = RefHead(d_txt, enc.model.num_features, dropout=0.0)
    # independent copies for the restatement cross-check (ReparamModule strips the originals)
    o_enc = copy.deepcopy(enc)
    o_head = dr.ProjectionHead(d_txt, enc.model.num_features)
    o_head.load_state_dict(head.state_dict())
    img_net = ReparamModule(enc)
    txt_net = ReparamModule(head)
    img_net.train(), txt_net.train()
    return o_enc, o_head, img_net, txt_net


def gen_unroll(ReparamModule, RefHead, path, variant, n, size, d_txt, K, outer_its, full, seed,
               cross_check=True):
    """Full bi-trajectory path, `outer_its` consecutive outer iterations with the three
    SGD(momentum=0.5) optimisers (distill.py:233-241, 603-613)."""
    enc, head, img_net, txt_net = build_pair(ReparamModule, RefHead, variant, d_txt, seed)
    th0_img = img_net.flat_param.detach().clone()
    th0_txt = txt_net.flat_param.detach().clone()
    image_syn, text_syn = dr.synthetic_inputs(n, size, d_txt, seed=seed + 3)
    # expert displacement: isotropic noise with the SAME norm as the student's own K-step move, so the
    # normalised matching loss (distill.py:596-597) is O(1)-sensitive to the inner path.
    fi0, ft0 = dr.FlatModule(enc), dr.FlatModule(head)
    a = th0_img.clone().requires_grad_(True)
    b = th0_txt.clone().requires_grad_(True)
    l0 = dr.contrastive_loss(fi0(image_syn, flat_param=a), ft0(text_syn, flat_param=b), 0.1)
    gi0, gt0 = torch.autograd.grad(l0, [a, b])
    sig_img = float(0.1 * K * gi0.norm() / th0_img.numel() ** 0.5)
    sig_txt = float(0.1 * K * gt0.norm() / th0_txt.numel() ** 0.5)
    g = torch.Generator().manual_seed(seed + 2)
    tgt_img = th0_img + sig_img * torch.randn(th0_img.shape, generator=g)
    tgt_txt = th0_txt + sig_txt * torch.randn(th0_txt.shape, generator=g)
    image_syn.requires_grad_(True), text_syn.requires_grad_(True)
    lr_img = torch.tensor(0.1, requires_grad=True)
    lr_txt = torch.tensor(0.1, requires_grad=True)
    opt_img = torch.optim.SGD([image_syn], lr=1000.0, momentum=0.5)
    opt_txt = torch.optim.SGD([text_syn], lr=1000.0, momentum=0.5)
    opt_lr = torch.optim.SGD([lr_img, lr_txt], lr=1e-3, momentum=0.5)
    pg = torch.Generator().manual_seed(seed + 4)

    rec = dict(variant=variant, n=n, size=size, d_txt=d_txt, K=K, seed=seed, sig_img=sig_img,
               sig_txt=sig_txt)
    rec["image_syn0"] = f32(image_syn) if full else f32(image_syn[:, :, :4, :4])
    rec["text_syn0"] = f32(text_syn)
    if full:
        rec["theta0_img"], rec["theta0_txt"] = f32(th0_img), f32(th0_txt)
        rec["target_img"], rec["target_txt"] = f32(tgt_img), f32(tgt_txt)
    all_perms = []
    for it in range(outer_its):
        perms = [torch.randperm(n, generator=pg) for _ in range(K)]
        all_perms.append(np.stack([p.numpy() for p in perms]))
        grand, il, tl, ces, thKi, thKt = ref_loop(img_net, txt_net, image_syn, text_syn, lr_img,
                                                  lr_txt, th0_img, th0_txt, tgt_img, tgt_txt, perms)
        if it == 0 and cross_check:   # (skipped at N=100: a second double-backward graph does not fit in RAM;
            # the same restatement is cross-checked on the same topology by the c1 golden)
            fi, ft = dr.FlatModule(enc), dr.FlatModule(head)   # restatement used by the tests
            assert fi.param_numel == img_net.param_numel and torch.equal(fi.flat_param(), th0_img)
            assert [tuple(s) for s in img_net._param_shapes] == fi.shapes
            g2, aux = dr.unrolled_match(fi, ft, image_syn, text_syn, lr_img, lr_txt, th0_img,
                                        th0_txt, tgt_img, tgt_txt, perms)
            assert abs(g2.item() - grand.item()) < 1e-5 * abs(grand.item()), (g2, grand)
        opt_lr.zero_grad(), opt_img.zero_grad(), opt_txt.zero_grad()
        grand.backward()
        rec[f"it{it}_grand"] = f32(grand)
        rec[f"it{it}_img_loss"], rec[f"it{it}_txt_loss"] = f32(il), f32(tl)
        rec[f"it{it}_ces"] = np.array(ces, np.float32)
        rec[f"it{it}_g_lr_img"], rec[f"it{it}_g_lr_txt"] = f32(lr_img.grad), f32(lr_txt.grad)
        rec[f"it{it}_g_text_syn"] = f32(text_syn.grad)
        if full:
            rec[f"it{it}_g_image_syn"] = f32(image_syn.grad)
            rec[f"it{it}_theta_K_img_head"] = f32(thKi[:4096])
        else:
            rec[f"it{it}_g_image_syn_slice"] = f32(image_syn.grad[:, :, ::37, ::41])
            rec[f"it{it}_g_image_syn_norm"] = f32(image_syn.grad.norm())
        rec[f"it{it}_theta_K_img_sqdist"] = f32(((thKi - tgt_img) ** 2).sum())
        opt_lr.step(), opt_img.step(), opt_txt.step()
        rec[f"it{it}_lr_after"] = np.array([lr_img.item(), lr_txt.item()], np.float32)
        rec[f"it{it}_text_syn_after"] = f32(text_syn)
        if full:
            rec[f"it{it}_image_syn_after"] = f32(image_syn)
        print(f"  it{it}: grand={grand.item():.6f} ces={ces} |g_img|={image_syn.grad.norm().item():.3e}")
    rec["perms"] = np.stack(all_perms)
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_itm_eval(path):
    """Retrieval metrics golden: inputs + the output of the REFERENCE's own itm_eval (epoch.py:219-244,
    pure numpy), AST-extracted and executed on its own -- epoch.py as a module needs kornia/torchvision/
    networks and cannot be imported."""
    import contextlib
    import io
    src = open(os.path.join(REF, "epoch.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "itm_eval"][0]
    ns = {"np": np, "torch": torch}      # the definition carries a @torch.no_grad() decorator
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "epoch.py:itm_eval", "exec"), ns)
    ref_itm_eval = ns["itm_eval"]
    from oracle import retrieval_ref as rr
    rng = np.random.RandomState(7)
    n_img, per, d = 24, 5, 48
    n_txt = n_img * per
    img2txt = [list(range(i * per, (i + 1) * per)) for i in range(n_img)]
    txt2img = np.repeat(np.arange(n_img), per)
    img = rng.randn(n_img, d).astype(np.float32)
    txt = (0.22 * img[txt2img] + rng.randn(n_txt, d)).astype(np.float32)   # partly aligned: recalls in (0, 100)
    sim = rr.similarity(img, txt)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = ref_itm_eval(sim, sim.T.copy(), txt2img, img2txt)
    mine = rr.itm_eval(sim, sim.T.copy(), txt2img, img2txt)
    for k in ref:
        assert abs(ref[k] - mine[k]) < 1e-9, (k, ref[k], mine[k])
    r_i, r_t = rr.ranks(sim, sim.T.copy(), txt2img, img2txt)
    np.savez(path, img_feat=img, txt_feat=txt, txt2img=txt2img.astype(np.int32),
             img2txt=np.asarray(img2txt, dtype=np.int32), sim=sim, rank_i2t=r_i, rank_t2i=r_t,
             keys=np.array(sorted(ref)), values=np.array([ref[k] for k in sorted(ref)], dtype=np.float64))
    print("wrote", path, {k: round(v, 2) for k, v in ref.items()})


def gen_nearest_neighbor(path):
    """Caption-decode golden: the reference's own nearest_neighbor (distill.py:89-95, AST-extracted; it
    needs only numpy + sklearn's cosine_similarity, which is installed here) on a small bank."""
    from sklearn.metrics.pairwise import cosine_similarity
    src = open(os.path.join(REF, "distill.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "nearest_neighbor"][0]
    ns = {"np": np, "cosine_similarity": cosine_similarity}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "distill.py:nearest_neighbor", "exec"), ns)
    rng = np.random.RandomState(11)
    n, q, d = 400, 16, 64
    bank = rng.randn(n, d).astype(np.float32)
    pick = rng.permutation(n)[:q]
    query = (bank[pick] * rng.uniform(0.5, 2.0, size=(q, 1)) + 2.2 * rng.randn(q, d)).astype(np.float32)
    sentences = ["caption %d" % i for i in range(n)]
    got = ns["nearest_neighbor"](sentences, query, bank)
    from oracle import retrieval_ref as rr
    assert got == rr.nearest_neighbor(sentences, query, bank)
    idx = np.array([int(s.split()[1]) for s in got], dtype=np.int32)
    np.savez(path, bank=bank, query=query, index=idx)
    print("wrote", path, "hits on the planted rows:", int((idx == pick).sum()), "of", q)


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    ReparamModule, RefHead = load_reference_pieces()
    which = sys.argv[1:] or ["text", "tiny", "c1", "itm", "nn"]
    if "nn" in which:
        gen_nearest_neighbor(os.path.join(out, "nearest_neighbor_small.npz"))
    if "itm" in which:
        gen_itm_eval(os.path.join(out, "itm_eval_small.npz"))
    if "text" in which:
        gen_text_only(ReparamModule, RefHead, os.path.join(out, "text_only_unroll.npz"))
    if "tiny" in which:
        gen_unroll(ReparamModule, RefHead, os.path.join(out, "unroll_tiny.npz"), "nfnet_tiny",
                   n=4, size=64, d_txt=32, K=2, outer_its=2, full=True, seed=100)
    if "c2s" in which:  # BASELINE config 2's spatial scale (N=100 @224, NFNet-l0), one unrolled step
        gen_unroll(ReparamModule, RefHead, os.path.join(out, "unroll_c2s_scalars.npz"), "nfnet_l0",
                   n=100, size=224, d_txt=768, K=1, outer_its=1, full=False, seed=300, cross_check=False)
    if "k8" in which:  # BASELINE config 2's unroll depth on the full-width encoder: syn_steps=8, N=12 @224 (the same
        # number of image-steps in the double-backward graph as the N=100 x K=1 fixture)
        gen_unroll(ReparamModule, RefHead, os.path.join(out, "unroll_k8_scalars.npz"), "nfnet_l0",
                   n=12, size=224, d_txt=768, K=8, outer_its=1, full=False, seed=400, cross_check=False)
    if "c1" in which:  # BASELINE config 1: N=10, syn_steps=2, NFNet-l0 + 768-d text, fp32 CPU
        gen_unroll(ReparamModule, RefHead, os.path.join(out, "unroll_c1_scalars.npz"), "nfnet_l0",
                   n=10, size=224, d_txt=768, K=2, outer_its=1, full=False, seed=200)


if __name__ == "__main__":
    main()
