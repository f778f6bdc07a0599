"""The ViT-B/16 linear shapes (19,700 token rows) on k_conv_gemm (forward) and k_conv_wgrad, standalone, through the
op-level C ABI: TFLOP/s per launch with nothing else on the GPU."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_dataset_distillation_amd import _lib
lib = _lib.load()
dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr())
M = 19700
for (cin, cout) in [(768, 3072), (3072, 768), (768, 2304), (768, 768)]:
    x = torch.randn(M, 1, 1, cin, device=dev).bfloat16()
    dy = torch.randn(M, 1, 1, cout, device=dev).bfloat16()
    w = (torch.randn(cout, 1, cin, device=dev) * 0.02).bfloat16()
    b = torch.zeros(cout, device=dev)
    y = torch.empty(M, 1, 1, cout, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(cout, cin, device=dev); db = torch.zeros(cout, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def fwd(): _lib.check(lib.mdd_op_conv2d(1, 0, M, 1, 1, cin, cout, 1, 1, 0, 1, P(x), P(w), P(b), P(y), st))
    def wg(): _lib.check(lib.mdd_op_conv2d_wgrad(1, M, 1, 1, cin, cout, 1, 1, 0, 1, P(dy), P(x), P(dw), P(db), st))
    for name, f in (("forward", fwd), ("wgrad", wg)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:8s} {cin:5d} -> {cout:5d}: {us:7.1f} us  {2.0 * M * cin * cout / us / 1e6:6.0f} TFLOP/s", flush=True)
