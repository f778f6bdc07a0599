"""Reference point for the implicit-GEMM core: the same pointwise-conv GEMM shapes through torch.matmul
(hipBLASLt / rocBLAS bf16, plain store, no fused epilogue)."""
import torch
dev = "cuda"
SHAPES = [(19600, 384, 1536), (19600, 1536, 384), (19600, 768, 1536), (19600, 3072, 384),
          (313600, 64, 256), (313600, 128, 256), (313600, 256, 128), (78400, 512, 128), (78400, 128, 512),
          (4900, 1536, 384), (4900, 1536, 2304)]
for (M, K, N) in SHAPES:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 50
    for _ in range(R):
        torch.matmul(a, b.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    fl = 2.0 * M * K * N
    by = (M * K + N * K + M * N) * 2
    print("M %7d K %5d N %5d: %7.1f us  %6.0f TF/s  %5.0f GB/s" % (M, K, N, us, fl / us / 1e6, by / us / 1e3))
