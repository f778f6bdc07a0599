"""Weight gradient of the grouped 3x3 stride-1 convolutions (64-channel groups) through the op-level C ABI: time per launch
with the engine's split-M workspace, one and two operand pairs (WG_VARIANT selects a variant build)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_dataset_distillation_amd import _lib
_v = os.environ.get("WG_VARIANT")
lib = _lib.load(variant=os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "libmdd_hip.%s.so" % _v) if _v else None)
dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
ws = torch.empty(64 << 20, device=dev)
for (n, h, ch) in [(100, 56, 64), (100, 28, 128), (100, 14, 384), (100, 7, 384)]:
    g = ch // 64
    for two in (False, True):
        x = torch.randn(n, h, h, ch, device=dev).bfloat16()
        dy = torch.randn(n, h, h, ch, device=dev).bfloat16()
        x2 = torch.randn_like(x) if two else None
        dy2 = torch.randn_like(dy) if two else None
        dw = torch.empty(ch, 9, 64, device=dev); db = torch.zeros(ch, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def wg(): _lib.check(lib.mdd_op_conv2d_wgrad2(1, n, h, h, ch, ch, 3, 1, 1, g, P(dy), P(x), P(dy2), P(x2), P(dw), P(db), P(ws), ws.numel(), st))
        for _ in range(3): wg()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): wg()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * n * h * h * ch * 64 * 9 * (2 if two else 1)
        print(f"{n} x {h}^2 x {ch} (g{g}) pairs {2 if two else 1}: {us:7.1f} us {fl / us / 1e6:6.0f} TF/s (with the combine kernel)", flush=True)
