"""Host-logic tests of the product ReparamModule (generic split/view path runs on CPU; the native
HIP path is covered by the -m gpu tests)."""
import os
import sys

import pytest
import torch
import torch.nn as nn

REF = "/root/reference"


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(4, 6)
        self.b = nn.Linear(6, 3)
        self.b.bias = self.a.bias if False else self.b.bias

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def test_flatten_order_views_and_double_backward():
    from multimodal_dataset_distillation_amd.reparam_module import ReparamModule
    torch.manual_seed(0)
    m = Toy()
    ref_params = [p.detach().clone() for p in m.parameters()]
    rm = ReparamModule(m)
    assert rm.param_numel == sum(p.numel() for p in ref_params)
    assert rm._param_infos == (("module.a", "weight"), ("module.a", "bias"), ("module.b", "weight"),
                               ("module.b", "bias"))
    assert torch.equal(rm.flat_param.detach(), torch.cat([p.reshape(-1) for p in ref_params]))
    assert list(rm.parameters())[0] is rm.flat_param and len(list(rm.parameters())) == 1
    x = torch.randn(5, 4, requires_grad=True)
    th = rm.flat_param.detach().clone().requires_grad_(True)
    y = rm(x, flat_param=th.unsqueeze(0))          # DataParallel-style [1,P] row is squeezed
    g, = torch.autograd.grad(y.pow(2).sum(), th, create_graph=True)
    th2 = th - 0.1 * g
    loss = (rm(x, flat_param=th2) ** 2).sum()
    gx, = torch.autograd.grad(loss, x)
    assert gx.shape == x.shape and torch.isfinite(gx).all()
    # flat_param=None falls back to the module's own parameter (the reference crashes here)
    assert torch.allclose(rm(x), rm(x, flat_param=rm.flat_param))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_matches_reference_reparam_module():
    from multimodal_dataset_distillation_amd.reparam_module import ReparamModule
    sys.path.insert(0, REF)
    import reparam_module as ref
    torch.manual_seed(1)
    a, b = Toy(), Toy()
    b.load_state_dict(a.state_dict())
    mine, theirs = ReparamModule(a), ref.ReparamModule(b)
    assert mine._param_infos == theirs._param_infos
    assert mine._param_numels == theirs._param_numels
    assert tuple(mine._param_shapes) == tuple(theirs._param_shapes)
    th = torch.randn(mine.param_numel, requires_grad=True)
    x = torch.randn(3, 4)
    y1, y2 = mine(x, flat_param=th), theirs(x, flat_param=th)
    assert torch.allclose(y1, y2, atol=1e-7)
    g1, = torch.autograd.grad(y1.sum(), th, create_graph=True)
    g2, = torch.autograd.grad(y2.sum(), th, create_graph=True)
    assert torch.allclose(g1, g2, atol=1e-7)


def test_expert_buffer_roundtrip(tmp_path):
    from multimodal_dataset_distillation_amd import expert_buffer as eb
    shapes = [(4, 3, 3, 3), (4,), (4, 1, 1, 1), (2, 4)]
    P = sum(int(torch.tensor(s).prod()) for s in shapes)
    flat = torch.randn(2, 3, P)
    path = str(tmp_path / "img_replay_buffer_0.pt")
    eb.save_expert_file(path, flat, shapes)
    raw = torch.load(path, weights_only=True)          # reference format: list[expert][epoch][tensor]
    assert len(raw) == 2 and len(raw[0]) == 3 and [tuple(t.shape) for t in raw[0][0]] == shapes
    back = eb.load_expert_file(path, expect_numel=P, expect_shapes=shapes)
    assert torch.equal(back, flat)
    with pytest.raises(ValueError):
        eb.load_expert_file(path, expect_numel=P + 1)
    (tmp_path / "txt_replay_buffer_0.pt").write_bytes(b"")
    assert eb.list_expert_files(str(tmp_path))[0] == [path]
