"""Weight gradient of wide pointwise layers (the ViT linears; csrc/conv_wgrad.hip k_wgrad_pipe) through the op-level
C ABI: error against an fp32 matmul of the same bf16 operands and time per launch, with the engine's split-M workspace,
one and two operand pairs."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_dataset_distillation_amd import _lib
_v = os.environ.get("WG_VARIANT")
lib = _lib.load(variant=os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "libmdd_hip.%s.so" % _v) if _v else None)
dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
ws = torch.empty(16 * 3072 * 768, device=dev)
shapes = [(8274, 768, 768), (8200, 776, 1032), (19700, 768, 3072), (19700, 3072, 768), (19700, 768, 2304), (19700, 768, 768),
          (19600, 1536, 384), (19600, 384, 1536), (4900, 1536, 2304)]
for (M, cin, cout) in shapes:
    for two in (False, True):
        x = torch.randn(M, 1, 1, cin, device=dev).bfloat16()
        dy = torch.randn(M, 1, 1, cout, device=dev).bfloat16()
        x2 = torch.randn(M, 1, 1, cin, device=dev).bfloat16() if two else None
        dy2 = torch.randn(M, 1, 1, cout, device=dev).bfloat16() if two else None
        dw = torch.full((cout, cin), float("nan"), device=dev); db = torch.zeros(cout, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def wg(): _lib.check(lib.mdd_op_conv2d_wgrad2(1, M, 1, 1, cin, cout, 1, 1, 0, 1, P(dy), P(x), P(dy2), P(x2), P(dw), P(db),
                                                      P(ws), ws.numel(), st))
        wg(); torch.cuda.synchronize()
        ref = dy.view(M, cout).float().t() @ x.view(M, cin).float()
        if two: ref += dy2.view(M, cout).float().t() @ x2.view(M, cin).float()
        rb = dy.view(M, cout).float().sum(0)
        e = ((dw - ref).norm() / ref.norm()).item(); eb = ((db - rb).norm() / rb.norm()).item()
        for _ in range(3): wg()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): wg()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * M * cin * cout * (2 if two else 1)
        print(f"M {M} {cin}->{cout} pairs {2 if two else 1}: err {e:.2e} bias {eb:.2e}  {us:7.1f} us {fl / us / 1e6:6.0f} TF/s (with the combine kernel)", flush=True)
