"""Experiment: does running a dependent chain of conv_gemm launches as S independent sample-range chains on S
HIP streams hide the per-launch fill/drain?  One 14^2 NFNet-l0 stage (6 blocks x [1x1 1536->384, 3x3 g6, 3x3 g6,
1x1 384->1536]) on 100 samples, through the op-level C ABI; S = 1, 2, 4.  Same total work in every row."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_dataset_distillation_amd import _lib
lib = _lib.load()
dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr())
N, H = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 14
LAYERS = [(1536, 384, 1, 1), (384, 384, 3, 6), (384, 384, 3, 6), (384, 1536, 1, 1)] * 6


def make(n):
    bufs = {c: [torch.randn(n, H, H, c, device=dev).bfloat16() for _ in range(2)] for c in (384, 1536)}
    ws = [((torch.randn(co, k * k, ci // g, device=dev) * 0.02).bfloat16(), torch.zeros(co, device=dev))
          for (ci, co, k, g) in LAYERS]
    return bufs, ws


def chain(n, bufs, ws, stream):
    st = C.c_void_p(stream.cuda_stream)
    flip = {384: 0, 1536: 0}
    x = bufs[1536][0]
    for (ci, co, k, g), (w, b) in zip(LAYERS, ws):
        flip[co] ^= 1
        y = bufs[co][flip[co]]
        _lib.check(lib.mdd_op_conv2d(1, 0, n, H, H, ci, co, k, 1, k // 2, g, P(x), P(w), P(b), P(y), st))
        x = y


for S in (1, 2, 4):
    n = N // S
    parts = [make(n) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run():
        for (bufs, ws), st in zip(parts, streams):
            chain(n, bufs, ws, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for st in streams: st.wait_stream(torch.cuda.current_stream())
    for _ in range(R): run()
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    print("H=%d streams=%d samples/stream=%d: %.1f us per stage pass (%d launches)" %
          (H, S, n, e0.elapsed_time(e1) / R * 1e3, len(LAYERS) * S))
