#!/usr/bin/env python3
"""Time the fused attention modes standalone at configs[4]'s shape (100 images, 12 heads, 197 tokens):
    python tools/micro/attn_bench.py [lib.so ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(lib_path):
    import torch
    from multimodal_dataset_distillation_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib_path)
    lib = _lib.load()
    n, T, H = 100, 197, 12
    d = H * 64
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(n, T, 3 * d, device=dev, generator=g).bfloat16()
    qkv_t = torch.randn(n, T, 3 * d, device=dev, generator=g).bfloat16()
    do = torch.randn(n, T, d, device=dev, generator=g).bfloat16()
    do_t = torch.randn(n, T, d, device=dev, generator=g).bfloat16()
    o = torch.zeros(n, T, d, device=dev, dtype=torch.bfloat16)
    dq = torch.zeros(n, T, 3 * d, device=dev, dtype=torch.bfloat16)
    st = [torch.zeros(n, H, T, device=dev) for _ in range(5)]
    P = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for mode in range(7):
        out = o if mode <= 1 else dq
        call = lambda: _lib.check(lib.mdd_op_attention(mode, n, T, H, 0.125, P(qkv), P(qkv_t), P(do), P(do_t), P(o), P(o), P(out),
                                                      P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), 0, 0, 0, s))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call()
        b.record()
        torch.cuda.synchronize()
        res[mode] = round(a.elapsed_time(b) / 20 * 1e3, 1)
    print(os.path.basename(lib_path), "us per launch by mode:", res, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "multimodal_dataset_distillation_amd", "libmdd_hip.so")]
        for lib in libs:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", lib])
