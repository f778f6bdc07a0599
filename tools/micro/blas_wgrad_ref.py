"""Reference point for the pointwise weight-gradient contraction dW[co,ci] = sum_m dy[m,co] x[m,ci]
(reduction over the M = N*H*W pixels) through torch.matmul (hipBLASLt, bf16 in / bf16 out)."""
import torch
dev = "cuda"
for (M, CO, CI) in [(19600, 1536, 384), (19600, 384, 1536), (313600, 256, 64), (313600, 64, 128), (78400, 512, 128), (4900, 1536, 384), (4900, 2304, 1536)]:
    dy = torch.randn(M, CO, device=dev).bfloat16()
    x = torch.randn(M, CI, device=dev).bfloat16()
    out = torch.empty(CO, CI, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(dy.t(), x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 50
    for _ in range(R):
        torch.matmul(dy.t(), x, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    print("M %7d co %5d ci %5d: %7.1f us  %6.0f TF/s  %5.0f GB/s" % (M, CO, CI, us, 2.0 * M * CO * CI / us / 1e6, (M * (CO + CI)) * 2 / us / 1e3))
