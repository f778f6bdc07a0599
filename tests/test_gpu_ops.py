"""GPU parity of the single-op C-ABI entry points (implicit-GEMM conv forward / dgrad / wgrad)
against torch CPU fp32 convolution.  f32 mode (exact-fp32 MFMA) must agree to 1e-4 relative
(norm-wise; summation order differs), bf16x2 mode (fp32 storage, operands split into hi+lo bf16: 16
significand bits) to 1e-4 as well, bf16 mode to 2e-2 against the fp32 result on bf16-rounded operands."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # nimg, h, cin, cout, k, stride, groups          (pad = ((s-1)+(k-1))//2 as timm)
    (2, 16, 8, 16, 3, 2, 1),      # stem conv1 shape class (cin padded to 8, N=16)
    (2, 16, 16, 32, 3, 1, 1),     # stem conv2 (K=144 not a multiple of the K tile)
    (3, 12, 64, 64, 3, 1, 1),     # group width 64, one group
    (2, 12, 128, 128, 3, 2, 2),   # grouped, strided
    (2, 14, 384, 384, 3, 1, 6),   # stage-2 grouped 3x3
    (3, 14, 256, 64, 1, 1, 1),    # 1x1 reduce
    (3, 7, 96, 160, 1, 1, 1),     # 1x1, ragged M (147 rows), N=160 -> partial N tile
    (2, 9, 32, 48, 3, 2, 1),      # odd spatial size with stride 2
    (5, 1, 64, 40, 1, 1, 1),      # M = 5 (linear-layer-like)
    # 3x3 stride-1 convs with 64-channel groups: the LDS-slab path (MODE 4) in bf16 -- tiles that straddle
    # image boundaries at every feature-map width of NFNet-l0 (7, 14, 28, 56), ragged last tile
    (11, 7, 128, 128, 3, 1, 2),
    (5, 14, 192, 192, 3, 1, 3),
    (3, 28, 128, 128, 3, 1, 2),
    (2, 56, 64, 64, 3, 1, 1),
    # wide pointwise layer with >= 8192 rows: in bf16 the 256 x 256 pipelined kernels (k_gemm_pipe forward and data
    # gradient, k_wgrad_pipe); 8450 rows and 832 channels are not multiples of the tile
    (2, 65, 768, 832, 1, 1, 1),
]


def pack_fwd(w, groups):
    cout, cin_g, k, _ = w.shape
    return w.permute(0, 2, 3, 1).reshape(cout, k * k, cin_g).contiguous()


def pack_dgrad(w, groups):
    cout, cin_g, k, _ = w.shape
    cout_g = cout // groups
    return (w.view(groups, cout_g, cin_g, k, k).permute(0, 2, 3, 4, 1)
            .reshape(groups, cin_g, k * k, cout_g).contiguous())


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", ["f32", "bf16x2", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_ops(case, dtype, report):
    from multimodal_dataset_distillation_amd import _lib
    lib = _lib.load()
    nimg, h, cin, cout, k, stride, groups = case
    pad = ((stride - 1) + (k - 1)) // 2
    torch.manual_seed(hash(case) % 1000)
    x = torch.randn(nimg, cin, h, h)
    w = torch.randn(cout, cin // groups, k, k) / (cin // groups * k * k) ** 0.5
    b = torch.randn(cout)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    code = {"f32": 0, "bf16": 1, "bf16x2": 2}[dtype]
    tol = 2e-2 if dtype == "bf16" else 1e-4
    if dtype == "bf16":  # reference on the bf16-rounded operands: isolates accumulation error
        x, w = x.bfloat16().float(), w.bfloat16().float()
    x.requires_grad_(True), w.requires_grad_(True)
    y = F.conv2d(x, w, b, stride, pad, 1, groups)
    dy = torch.randn_like(y)
    if dtype == "bf16":
        dy = dy.bfloat16().float()
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    gb = dy.sum((0, 2, 3))
    ho = y.shape[2]
    dev = "cuda"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    xd = nhwc(x.detach()).to(dev, tdt)
    wf = pack_fwd(w.detach(), groups).to(dev, tdt)
    wt = pack_dgrad(w.detach(), groups).to(dev, tdt)
    bd = b.to(dev)
    # forward
    yo = torch.full((nimg, ho, ho, cout), float("nan"), device=dev, dtype=tdt)
    _lib.check(lib.mdd_op_conv2d(code, 0, nimg, h, h, cin, cout, k, stride, pad, groups, P(xd), P(wf),
                                 P(bd), P(yo), st))
    e_f = rel_err(yo.float().cpu().permute(0, 3, 1, 2), y)
    # dgrad
    dyd = nhwc(dy).to(dev, tdt)
    dxo = torch.full((nimg, h, h, cin), float("nan"), device=dev, dtype=tdt)
    _lib.check(lib.mdd_op_conv2d(code, 1, nimg, h, h, cin, cout, k, stride, pad, groups, P(dyd), P(wt),
                                 None, P(dxo), st))
    e_d = rel_err(dxo.float().cpu().permute(0, 3, 1, 2), gx)
    # wgrad (+ bias grad)
    dwo = torch.zeros(cout, k * k, cin // groups, device=dev)
    dbo = torch.zeros(cout, device=dev)
    _lib.check(lib.mdd_op_conv2d_wgrad(code, nimg, h, h, cin, cout, k, stride, pad, groups, P(dyd),
                                       P(xd), P(dwo), P(dbo), st))
    torch.cuda.synchronize()
    gw_o = dwo.cpu().view(cout, k * k, cin // groups).permute(0, 2, 1).reshape(cout, cin // groups, k, k)
    e_w = rel_err(gw_o, gw)
    e_b = rel_err(dbo.cpu(), gb)
    report(f"conv_ops {dtype} {case}: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e} bias {e_b:.2e}")
    assert e_f < tol and e_d < tol and e_w < tol and e_b < tol


@pytest.mark.parametrize("pairs", [1, 2])
@pytest.mark.parametrize("workspace", [False, True])
@pytest.mark.parametrize("shape", [(8274, 768, 768), (8200, 776, 1032), (9000, 1024, 768), (8193, 768, 1032)])
def test_wide_pointwise_wgrad(shape, workspace, pairs, report):
    """Weight gradient of wide pointwise layers in bf16 (the ViT linears: csrc/conv_wgrad.hip k_wgrad_pipe, 256 x 256
    tiles, LDS-DMA ring): one / two operand pairs, split-M workspace or atomics, channel counts that are not a
    multiple of the tile, a pixel count that is not a multiple of the K-tile.  Reference: fp32 matmul of the same
    bf16 operands (only the summation order differs: 1e-5)."""
    from multimodal_dataset_distillation_amd import _lib
    lib = _lib.load()
    M, cin, cout = shape
    dev = "cuda"
    torch.manual_seed(M + cin + pairs)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = [torch.randn(M, cin, device=dev).bfloat16() for _ in range(pairs)]
    dy = [torch.randn(M, cout, device=dev).bfloat16() for _ in range(pairs)]
    ws = torch.empty(12 * cin * cout, device=dev) if workspace else None
    dw = torch.full((cout, cin), float("nan"), device=dev) if workspace else torch.zeros(cout, cin, device=dev)
    db = torch.zeros(cout, device=dev)
    _lib.check(lib.mdd_op_conv2d_wgrad2(1, M, 1, 1, cin, cout, 1, 1, 0, 1, P(dy[0]), P(x[0]),
                                        P(dy[1]) if pairs == 2 else None, P(x[1]) if pairs == 2 else None,
                                        P(dw), P(db), P(ws), ws.numel() if workspace else 0, st))
    torch.cuda.synchronize()
    ref = sum(d.float().t() @ a.float() for d, a in zip(dy, x))
    rb = dy[0].float().sum(0)
    e_w, e_b = rel_err(dw.cpu(), ref.cpu()), rel_err(db.cpu(), rb.cpu())
    report(f"wide pointwise wgrad bf16 {shape} pairs {pairs} workspace {workspace}: dW {e_w:.2e} bias {e_b:.2e}")
    assert e_w < 1e-5 and e_b < 1e-5


@pytest.mark.parametrize("pairs", [1, 2])
@pytest.mark.parametrize("workspace", [False, True])
@pytest.mark.parametrize("shape", [(5, 14, 192), (3, 28, 128), (11, 7, 128), (2, 56, 64)])
def test_grouped_3x3_wgrad_pairs(shape, workspace, pairs, report):
    """Weight gradient of the grouped 3x3 stride-1 convolutions with 64-channel groups in bf16 (csrc/conv_wgrad.hip
    k_wgrad_slab: the x rows of all nine taps from one LDS ring): the engine's form -- a second operand pair (tangent pass),
    the split-M workspace -- at every feature-map width of NFNet-l0, chunks that straddle image boundaries.  Reference:
    torch's conv2d weight gradient in fp32 on the same bf16 operands."""
    from multimodal_dataset_distillation_amd import _lib
    lib = _lib.load()
    nimg, h, ch = shape
    groups = ch // 64
    dev = "cuda"
    torch.manual_seed(nimg * h + pairs)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xs = [torch.randn(nimg, ch, h, h).bfloat16().float() for _ in range(pairs)]
    dys = [torch.randn(nimg, ch, h, h).bfloat16().float() for _ in range(pairs)]
    ref = torch.zeros(ch, 64, 3, 3)
    for x, dy in zip(xs, dys):
        w = torch.zeros(ch, 64, 3, 3, requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv2d(x, w, None, 1, 1, 1, groups), [w], dy)
        ref += gw
    rb = dys[0].sum((0, 2, 3))
    xd = [nhwc(x).to(dev, torch.bfloat16) for x in xs]
    dyd = [nhwc(d).to(dev, torch.bfloat16) for d in dys]
    ws = torch.empty(40 * ch * 576, device=dev) if workspace else None
    dw = torch.full((ch, 9, 64), float("nan"), device=dev) if workspace else torch.zeros(ch, 9, 64, device=dev)
    db = torch.zeros(ch, device=dev)
    _lib.check(lib.mdd_op_conv2d_wgrad2(1, nimg, h, h, ch, ch, 3, 1, 1, groups, P(dyd[0]), P(xd[0]),
                                        P(dyd[1]) if pairs == 2 else None, P(xd[1]) if pairs == 2 else None,
                                        P(dw), P(db), P(ws), ws.numel() if workspace else 0, st))
    torch.cuda.synchronize()
    got = dw.cpu().view(ch, 9, 64).permute(0, 2, 1).reshape(ch, 64, 3, 3)
    e_w, e_b = rel_err(got, ref), rel_err(db.cpu(), rb)
    report(f"grouped 3x3 wgrad bf16 {shape} pairs {pairs} workspace {workspace}: dW {e_w:.2e} bias {e_b:.2e}")
    assert e_w < 1e-5 and e_b < 1e-5


def test_flat_ops(report):
    from multimodal_dataset_distillation_amd import _lib
    lib = _lib.load()
    dev = "cuda"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    for n in (1, 3, 4, 1027, 1 << 20, (1 << 22) + 3):
        torch.manual_seed(n)
        x, g = torch.randn(n), torch.randn(n)
        lr = torch.tensor([0.37])
        out = torch.empty(n, device=dev)
        xd, gd, lrd = x.to(dev), g.to(dev), lr.to(dev)   # keep alive: the ABI takes raw pointers
        _lib.check(lib.mdd_flat_axpy(P(out), P(xd), P(gd), P(lrd), -1.0, n, st))
        assert rel_err(out, x - 0.37 * g) < 1e-6
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        _lib.check(lib.mdd_flat_sqdist(P(xd), P(gd), P(acc), n, st))
        ref = ((x.double() - g.double()) ** 2).sum()
        assert abs(acc.item() - ref.item()) <= 1e-6 * ref.item() + 1e-12
        p, buf = x.clone().to(dev), torch.zeros(n, device=dev)
        _lib.check(lib.mdd_flat_sgd_momentum(P(p), P(gd), P(buf), 1000.0, 0.5, 1, n, st))
        _lib.check(lib.mdd_flat_sgd_momentum(P(p), P(gd), P(buf), 1000.0, 0.5, 0, n, st))
        ref_p = x - 1000.0 * g - 1000.0 * (0.5 * g + g)
        assert rel_err(p, ref_p) < 1e-6
        # guarded step: a no-op (parameters AND momentum buffer) when the device flag is set
        flag = torch.ones(1, device=dev)
        p_before, buf_before = p.clone(), buf.clone()
        _lib.check(lib.mdd_flat_sgd_momentum_guarded(P(p), P(gd), P(buf), 1000.0, 0.5, 0, n, P(flag), st))
        assert torch.equal(p, p_before) and torch.equal(buf, buf_before)
        flag.zero_()
        _lib.check(lib.mdd_flat_sgd_momentum_guarded(P(p), P(gd), P(buf), 1000.0, 0.5, 0, n, P(flag), st))
        assert rel_err(p, ref_p - 1000.0 * (0.5 * 1.5 * g + g)) < 1e-6
    report("flat_ops ok")
