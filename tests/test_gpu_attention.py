"""Fused multi-head attention of the bf16 ViT path (csrc/attn.hip, `mdd_op_attention`) against plain PyTorch fp32 on
the same bf16-rounded inputs: forward, tangent of the forward (torch.func.jvp), backward (vjp) and the tangent of the
backward (jvp of the vjp) -- the four passes of the unrolled loop (reference distill.py:524, 562-567, 606 through
timm's Attention, networks.py:668).  Tolerance: bf16 operands (the probabilities and score gradients enter the second
product rounded to bf16, as in the unfused path), so ~1e-2 relative on every tensor."""
import ctypes as C

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_attention(qkv, heads):
    n, t, _ = qkv.shape
    d = qkv.shape[2] // 3
    hd = d // heads
    q, k, v = qkv.reshape(n, t, 3, heads, hd).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    return (p @ v).transpose(1, 2).reshape(n, t, d)


@pytest.mark.parametrize("n,tokens,heads", [(3, 197, 3), (2, 50, 2), (2, 224, 1)])
def test_fused_attention_all_passes(n, tokens, heads, report):
    from multimodal_dataset_distillation_amd import _lib
    lib = _lib.load()
    d = heads * 64
    g = torch.Generator().manual_seed(tokens)
    bf = lambda x: x.to(torch.bfloat16)
    qkv = bf(torch.randn(n, tokens, 3 * d, generator=g) * 1.5)
    qkv_t = bf(torch.randn(n, tokens, 3 * d, generator=g))
    dout = bf(torch.randn(n, tokens, d, generator=g))
    dout_t = bf(torch.randn(n, tokens, d, generator=g))
    f = lambda x: ref_attention(x, heads)
    q32, qt32, do32, dot32 = qkv.float(), qkv_t.float(), dout.float(), dout_t.float()
    o_ref, ot_ref = torch.func.jvp(f, (q32,), (qt32,))

    def grad(x, w):
        _, pull = torch.func.vjp(f, x)
        return pull(w)[0]
    dq_ref, dqt_ref = torch.func.jvp(grad, (q32, do32), (qt32, dot32))

    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    scale = 64 ** -0.5
    qd, qtd, dod, dotd = qkv.to(DEV), qkv_t.to(DEV), dout.to(DEV), dout_t.to(DEV)
    st = lambda: torch.zeros(n, heads, tokens, device=DEV)
    m, l, r, D, Dt, Dscr = st(), st(), st(), st(), st(), st()
    o = torch.zeros(n, tokens, d, device=DEV, dtype=torch.bfloat16)
    ot = torch.zeros_like(o)
    dq = torch.zeros(n, tokens, 3 * d, device=DEV, dtype=torch.bfloat16)
    dqt = torch.zeros_like(dq)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(mode, out, r_tan=0, accum=0, skip0=0, Dbuf=None):
        _lib.check(lib.mdd_op_attention(mode, n, tokens, heads, scale, P(qd), P(qtd), P(dod), P(dotd), P(o), P(ot), P(out), P(m), P(l),
                                        P(r), P(Dbuf if Dbuf is not None else D), P(Dt), r_tan, accum, skip0, s))
    call(0, o)                       # forward: O, statistics m, l
    call(1, ot)                      # tangent forward: O_t, statistic r
    call(2, dq)                      # backward: dQ (+ D), then dV and dK
    call(4, dq)
    call(3, dqt)                     # tangent backward: dS_t K (+ D_t) ...
    call(2, dqt, r_tan=1, accum=1, Dbuf=Dscr)   # ... + dS K_t
    call(5, dqt)                     # dV_t
    call(6, dqt)                     # dS_t^T Q ...
    call(4, dqt, r_tan=1, accum=1, skip0=1)     # ... + dS^T Q_t
    torch.cuda.synchronize()
    e = dict(O=rel_err(o.float(), o_ref), O_t=rel_err(ot.float(), ot_ref), dqkv=rel_err(dq.float(), dq_ref),
             dqkv_t=rel_err(dqt.float(), dqt_ref))
    for name, a, b in (("dq", dq, dq_ref), ("dq_t", dqt, dqt_ref)):
        for i, comp in enumerate("qkv"):
            e[name + "." + comp] = rel_err(a.float()[..., i * d:(i + 1) * d], b[..., i * d:(i + 1) * d])
    report(f"fused attention n={n} tokens={tokens} heads={heads}: " + " ".join(f"{k} {v:.1e}" for k, v in e.items()))
    assert all(v < 2.5e-2 for v in e.values()), e
    assert torch.equal(Dscr, D)      # the second dQ launch recomputes the same D
