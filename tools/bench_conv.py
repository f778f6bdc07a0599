"""Micro-benchmark of the conv_gemm kernel on single layer shapes through the op-level C ABI."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_dataset_distillation_amd import _lib
if len(sys.argv) > 1:            # tools/bench_conv.py <library.so>: time another build (A/B)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.load()
dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [  # nimg, h, cin, cout, k, stride, groups
    (100, 14, 384, 1536, 1, 1, 1), (100, 14, 1536, 384, 1, 1, 1), (100, 14, 384, 384, 3, 1, 6),
    (100, 56, 64, 256, 1, 1, 1), (100, 112, 32, 64, 3, 1, 1),
    (100, 28, 128, 128, 3, 1, 2), (100, 56, 64, 64, 3, 1, 1), (100, 7, 384, 384, 3, 1, 6),
]
for (n, h, cin, cout, k, s, g) in SHAPES:
    pad = ((s - 1) + (k - 1)) // 2
    ho = (h + 2 * pad - k) // s + 1
    x = torch.randn(n, h, h, cin, device=dev).bfloat16()
    w = (torch.randn(cout, k * k, cin // g, device=dev) * 0.05).bfloat16()
    b = torch.randn(cout, device=dev)
    y = torch.empty(n, ho, ho, cout, device=dev, dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        _lib.check(lib.mdd_op_conv2d(1, 0, n, h, h, cin, cout, k, s, pad, g, P(x), P(w), P(b), P(y), st))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 50
    for _ in range(R): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    flops = 2.0 * n * ho * ho * cout * (cin // g) * k * k
    byts = (x.numel() + w.numel() + y.numel()) * 2
    print("lib=%s dbg=%s %s: %.1f us  %.0f TF/s  %.0f GB/s" % (os.path.basename(_lib.LIB_PATH), os.environ.get("MDD_DBG", "0"), (n, h, cin, cout, k, s, g), us, flops / us / 1e6, byts / us / 1e3))
