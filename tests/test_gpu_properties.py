"""Size-independent properties of the second-order path at BASELINE configs[1]'s FULL size (NFNet-l0, 100 pairs,
224x224, 768-d text): no CPU oracle finishes a double backward there (tests/test_gpu_c2.py holds the largest golden
that fits), but the calculus the four passes implement has identities that hold at any size:

  duality    <w, J u> = <J^T w, u>            tangent-forward (Dual conv_gemm / Dual elementwise kernels) against
                                              the first-order backward (dgrad + wgrad + weight-standardisation bwd)
  symmetry   <v, H_w u> = <u, H_w v>          H_w = d^2 <w, y(theta)> / d theta^2, produced by tangent-forward +
                                              tangent-backward: every Dual kernel of the reverse sweep
  linearity  H_w (a u) = a H_w u              the tangent passes are linear in the direction

J = d y / d theta of the image encoder (reference networks.py:678-682 under ReparamModule, distill.py:524) and of the
text projection (networks.py:639-646, distill.py:537); these are the operators `grand_loss.backward()` applies at
distill.py:606.  A kernel that drops a term, reads a stale stash or mis-indexes a tile breaks an identity by O(1);
rounding moves it by far less (the directions below make both sides of each identity sums of like-signed terms, so
per-contraction rounding averages out).  Bars = 4-100x the values measured on MI355X (the report line)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

#            duality  symmetry  linearity
# measured (MI355X, round 2): f32 1.1e-8 / 1.1e-8 / 9.3e-7 (nfnet_l0), 3.1e-8 / 3.7e-8 / 2.5e-6 (vit_b16);
# bf16x2 7.0e-8 / 9.6e-7 / 2.3e-6 (nfnet_l0), 9.8e-6 / 2.6e-6 / 1.4e-5 (vit_b16);
# bf16 nfnet_l0 N=100 1.5e-5 / 5.1e-5 / 1.6e-3, nfnet_l1 N=500 3.2e-6 / 4.8e-5 / 2.6e-3, vit_b16 1.3e-4 / 2.9e-4 / 6.4e-3
BARS = {"f32": (1e-6, 1e-6, 1e-5), "bf16x2": (5e-5, 5e-5, 1e-4), "bf16": (1e-3, 2e-3, 2e-2)}

def _dot(a, b):
    return float((a.double().flatten() @ b.double().flatten()).item())


def _rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


@pytest.mark.parametrize("variant,n,dtype", [
    ("nfnet_l0", 100, "f32"), ("nfnet_l0", 100, "bf16x2"), ("nfnet_l0", 100, "bf16"),
    # BASELINE configs[3]'s per-GPU shape: NFNet-l1, 500 pairs (1.57 M rows at 56^2, 150 GiB of activations for
    # one step + the tangent set).  Only bf16 fits one GPU, and no other test can check numbers at this size.
    ("nfnet_l1", 500, "bf16"),
    # BASELINE configs[4]: the ViT-B/16 image encoder at 100 pairs (19,700 token rows, 1200 attention matrices)
    ("vit_b16", 100, "f32"), ("vit_b16", 100, "bf16x2"), ("vit_b16", 100, "bf16"),
])
def test_image_encoder_identities_at_full_size(variant, n, dtype, report):
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from multimodal_dataset_distillation_amd.networks import synthetic_expert_params
    dev = torch.device("cuda", 0)
    size, d_txt = 224, 768
    eng = UnrollEngine(variant, batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=1, dtype=dtype,
                       device=dev)
    g = torch.Generator(device=dev).manual_seed(11)
    theta, theta_t = synthetic_expert_params(eng, seed=3, device=dev)
    x = torch.randn(n, 3, size, size, device=dev, generator=g)
    w = torch.randn(n, eng.feature_dim, device=dev, generator=g) / eng.feature_dim ** 0.5
    zeros = torch.zeros(n, eng.feature_dim, device=dev)
    # Directions are chosen so that neither side of an identity is a sum that cancels (with random u, v the inner
    # products are O(10) against |w||Ju| ~ 1e5 and their RELATIVE error measures luck, not kernels):
    # u along J^T w makes <J^T w, u> = c |J^T w|^2, v along H_w u makes <v, H_w u> = c' |H_w u|^2.
    y = eng.img_forward(0, theta, x)
    gth = eng.img_backward(0, theta, w, stash=True)                       # J^T w, backward signals stashed
    u = gth * (0.05 * float(theta.norm()) / float(gth.norm()))
    ju = eng.img_tangent_forward(0, theta, u)
    hu = eng.img_tangent_backward(0, theta, u, zeros)                     # H_w u
    v = hu * (0.05 * float(theta.norm()) / float(hu.norm()))
    jv = eng.img_tangent_forward(0, theta, v)
    hv = eng.img_tangent_backward(0, theta, v, zeros)
    u3 = 3.0 * u
    eng.img_tangent_forward(0, theta, u3)
    hu3 = eng.img_tangent_backward(0, theta, u3, zeros)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t).all()) for t in (y, gth, ju, hu, hv, hu3))
    dual = _rel(_dot(w, ju), _dot(gth, u))
    sym = _rel(_dot(v, hu), _dot(u, hv))
    lin = float((hu3 - 3.0 * hu).norm() / (3.0 * hu).norm())
    report(f"identities, image encoder {variant} N={n} @{size} {dtype}: duality {dual:.2e} symmetry {sym:.2e} "
           f"linearity {lin:.2e} (<w,Ju> {_dot(w, ju):.4e} <g,u> {_dot(gth, u):.4e} <v,Hu> {_dot(v, hu):.4e} <u,Hv> {_dot(u, hv):.4e})")
    assert float(ju.norm()) > 0 and float(hu.norm()) > 0
    bd, bs, bl = BARS[dtype]
    assert dual < bd and sym < bs and lin < bl, (dual, sym, lin)
    eng.close()
    del eng, y, gth, ju, hu, jv, hv, hu3
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_text_projection_identities_at_full_size(dtype, report):
    """The same identities for the text projection head with a dropout mask in place (the engine replays the
    caller's mask in backward and in both tangent passes)."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from multimodal_dataset_distillation_amd.networks import synthetic_expert_params
    dev = torch.device("cuda", 0)
    n, d_txt = 100, 768
    eng = UnrollEngine("nfnet_l0", batch=n, num_queries=n, image_size=32, d_txt=d_txt, syn_steps=1, dtype=dtype,
                       device=dev)
    g = torch.Generator(device=dev).manual_seed(12)
    _, theta = synthetic_expert_params(eng, seed=4, device=dev)
    x = torch.randn(n, d_txt, device=dev, generator=g) * 0.5
    mask = (torch.rand(n, eng.feature_dim, device=dev, generator=g) >= 0.1).float() / 0.9
    w = torch.randn(n, eng.feature_dim, device=dev, generator=g) / eng.feature_dim ** 0.5
    zeros = torch.zeros(n, eng.feature_dim, device=dev)
    eng.txt_forward(0, theta, x, None, mask)
    gth = eng.txt_backward(0, theta, w, stash=True)
    u = gth * (0.05 * float(theta.norm()) / float(gth.norm()))
    ju = eng.txt_tangent_forward(0, theta, u)
    hu = eng.txt_tangent_backward(0, theta, u, zeros)
    v = hu * (0.05 * float(theta.norm()) / float(hu.norm()))
    eng.txt_tangent_forward(0, theta, v)
    hv = eng.txt_tangent_backward(0, theta, v, zeros)
    torch.cuda.synchronize()
    dual = _rel(_dot(w, ju), _dot(gth, u))
    sym = _rel(_dot(v, hu), _dot(u, hv))
    report(f"identities, text projection N={n} {dtype}: duality {dual:.2e} symmetry {sym:.2e}")
    assert dual < 1e-6 and sym < 1e-6, (dual, sym)        # the head computes in fp32 in every mode (measured <= 1.5e-8)
    eng.close()


@pytest.mark.parametrize("variant,n,K,dtype,keep_steps", [
    ("nfnet_l0", 100, 8, "f32", None), ("nfnet_l0", 100, 8, "bf16x2", None),
    # BASELINE configs[3] per GPU: NFNet-l1, 500 pairs, syn_steps=16, bf16, every step's activations recomputed in
    # the reverse sweep (keep_steps=0, 157.6 GiB) -- the only end-to-end numerical check that reaches this size
    ("nfnet_l1", 500, 16, "bf16", 0),
    # BASELINE configs[4] per GPU: ViT-B/16, 100 pairs, syn_steps=8, bf16 (163 GiB)
    ("vit_b16", 100, 8, "bf16", None),
])
def test_outer_gradient_is_the_derivative_of_the_reported_loss_at_full_size(variant, n, K, dtype, keep_steps, report):
    """The whole outer iteration of BASELINE configs[1] (100 pairs, syn_steps=8, NFNet-l0 @224; reference
    distill.py:509-606): the gradients `grand_loss.backward()` would leave on image_syn / text_syn / syn_lr must be
    the derivative of the grand loss THE SAME CALL reports.  Central differences of the reported loss along the
    gradient directions (non-cancelling: the directional derivative is |g|) and in syn_lr_img, against the returned
    gradients -- end to end through eight unrolled steps and the reverse sweep, at the size no oracle reaches."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from multimodal_dataset_distillation_amd.networks import (student_move_normalised_targets,
                                                               synthetic_expert_params)
    dev = torch.device("cuda", 0)
    size, d_txt = 224, 768
    g = torch.Generator().manual_seed(0)
    mean = torch.tensor([-0.0626, -0.0221, 0.0680]).view(1, 3, 1, 1)
    std = torch.tensor([1.0451, 1.0752, 1.0539]).view(1, 3, 1, 1)
    image_syn = (torch.randn(n, 3, size, size, generator=g) * std + mean).to(dev)
    text_syn = (torch.randn(n, d_txt, generator=g) * 0.5253 - 0.0094).to(dev)
    lr = torch.tensor([0.1, 0.1], device=dev)
    perms = torch.stack([torch.randperm(n, generator=g) for _ in range(K)]).to(dev)
    eng = UnrollEngine(variant, batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K, dtype=dtype,
                       device=dev, keep_steps=keep_steps)
    th = synthetic_expert_params(eng, seed=100, device=dev)
    gt = torch.Generator(device=dev).manual_seed(200)
    tgi, tgt = student_move_normalised_targets(eng, th[0], th[1], image_syn, text_syn, lr, K, gt)[:2]

    def loss_at(img, txt, lrv):
        o = eng.unrolled_match(img, txt, lrv[0:1], lrv[1:2], th[0], th[1], tgi, tgt, perms=perms)
        torch.cuda.synchronize()
        return o

    o = loss_at(image_syn, text_syn, lr)
    L0 = float(o["grand_loss"])
    g_img, g_txt, g_lr = o["image_syn"].clone(), o["text_syn"].clone(), o["lr"].clone()
    assert all(bool(torch.isfinite(t).all()) for t in (g_img, g_txt, g_lr))
    rows = {}
    # Step size: the loss moves by ~4 % per side.  The reported loss is ROUGH at the 1e-4 level as a function of
    # its inputs (fp32 rounding of the chain, amplified ~800x by this instance -- DESIGN.md section 5: noise images
    # on an untrained student; a scan over step sizes shows difference-quotient errors of either sign, +-1e-4
    # absolute in L+ - L-, down to the smallest steps, the same in f32 and bf16x2 mode and also at C1 where the
    # engine matches the CPU oracle to 1e-6), so smaller steps measure that roughness, not the gradient.
    target = 8e-2 * abs(L0)
    for name, grad in (("image_syn", g_img), ("text_syn", g_txt), ("syn_lr_img", None)):
        if grad is not None:
            d = grad / grad.norm()
            analytic = float(grad.norm())
        else:
            analytic = float(g_lr[0])
        eps = target / (2.0 * abs(analytic))

        def central(e):
            if name == "image_syn":
                Lp = float(loss_at(image_syn + e * d, text_syn, lr)["grand_loss"])
                Lm = float(loss_at(image_syn - e * d, text_syn, lr)["grand_loss"])
            elif name == "text_syn":
                Lp = float(loss_at(image_syn, text_syn + e * d, lr)["grand_loss"])
                Lm = float(loss_at(image_syn, text_syn - e * d, lr)["grand_loss"])
            else:
                e2 = torch.tensor([e, 0.0], device=dev)
                Lp = float(loss_at(image_syn, text_syn, lr + e2)["grand_loss"])
                Lm = float(loss_at(image_syn, text_syn, lr - e2)["grand_loss"])
            return (Lp - Lm) / (2.0 * e)
        fd = central(eps)
        rows[name] = (analytic, fd, abs(fd - analytic) / abs(analytic), eps)
    report(f"finite differences of the reported grand loss, {variant} N={n} K={K} {dtype} keep_steps={keep_steps} (L = {L0:.6f}): "
           + " | ".join(f"{k}: analytic {a:.5e} fd {f:.5e} rel {r:.2e} (eps {e:.2e})" for k, (a, f, r, e) in rows.items()))
    assert all(r < FD_BAR[dtype] for (_, _, r, _) in rows.values()), rows
    eng.close()
    del eng
    torch.cuda.empty_cache()


FD_BAR = {"f32": 5e-3, "bf16x2": 5e-3, "bf16": 1e-2}      # measured: image 1.1e-4 / 5.2e-5, text 1.5e-3 (the cubic term: same in both modes), lr 5.1e-4; bf16 at configs[3] size: 5.1e-4 / 1.2e-4 / 1.4e-3
