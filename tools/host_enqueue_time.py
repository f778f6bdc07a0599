"""How long the host thread takes to ENQUEUE one outer iteration (mdd_unrolled_match returns when
everything is queued) next to how long the GPU takes to run it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_dataset_distillation_amd.engine import UnrollEngine
from multimodal_dataset_distillation_amd.networks import synthetic_expert_params

n, K, size, d_txt = 100, 8, 224, 768
dev = torch.device("cuda")
eng = UnrollEngine("nfnet_l0", batch=n, image_size=size, d_txt=d_txt, syn_steps=K, dtype="bf16")
g = torch.Generator().manual_seed(0)
image_syn = torch.randn(n, 3, size, size, generator=g).to(dev)
text_syn = (torch.randn(n, d_txt, generator=g) * 0.5).to(dev)
lr = torch.tensor([0.1, 0.1], device=dev)
a, b = synthetic_expert_params(eng, seed=100, device=dev)
tg = (a + 1e-3 * torch.randn_like(a), b + 1e-3 * torch.randn_like(b))
perms = torch.stack([torch.randperm(n) for _ in range(K)]).to(dev)
out = None
for _ in range(2):
    out = eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], a, b, *tg, perms=perms, out=out)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    out = eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], a, b, *tg, perms=perms, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.1f ms, GPU done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
