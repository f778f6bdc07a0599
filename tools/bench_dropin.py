#!/usr/bin/env python3
"""Throughput of the DROP-IN surface (INTEGRATION.md option (a)): the reference's loop verbatim
(distill.py:509-606: ReparamModule forward, autograd.grad(create_graph=True), backward) over the HIP ops,
beside the fused engine call (option (b)) on the same inputs.   python tools/bench_dropin.py [c1|c2] [dtype]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from multimodal_dataset_distillation_amd import networks as nw
    from multimodal_dataset_distillation_amd.distill import reference_loop_iteration
    from multimodal_dataset_distillation_amd.reparam_module import ReparamModule
    wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    n, K = {"c1": (10, 2), "c2": (100, 8)}[wl]
    size, d_txt, dev = 224, 768, "cuda"
    enc = nw.ImageEncoder(variant="nfnet_l0", dtype=dtype)
    enc.syn_steps, enc.d_txt = K, d_txt
    head = nw.ProjectionHead(d_txt, enc.num_features, dropout=0.1, image_size=size, variant="nfnet_l0",
                             syn_steps=K, dtype=dtype)
    img_net, txt_net = ReparamModule(enc).to(dev), ReparamModule(head).to(dev)
    img_net.train(), txt_net.train()
    eng = nw.get_engine("nfnet_l0", n, size, d_txt, K, dtype, torch.device(dev))
    th0i, th0t = nw.synthetic_expert_params(eng, 100, device=dev)
    g = torch.Generator().manual_seed(0)
    image_syn = torch.randn(n, 3, size, size, generator=g).to(dev)
    text_syn = (torch.randn(n, d_txt, generator=g) * 0.5253).to(dev)
    lr = torch.tensor([0.1, 0.1], device=dev)
    gt = torch.Generator(device=dev).manual_seed(200)
    tgi, tgt, _, _ = nw.student_move_normalised_targets(eng, th0i, th0t, image_syn, text_syn, lr, K, gt)
    perms = [torch.randperm(n, generator=g).to(dev) for _ in range(K)]

    def dropin():
        img_r = image_syn.detach().requires_grad_(True)
        txt_r = text_syn.detach().requires_grad_(True)
        lri = lr[0].detach().clone().requires_grad_(True)
        lrt = lr[1].detach().clone().requires_grad_(True)
        grand, il, tl, ces = reference_loop_iteration(img_net, txt_net, img_r, txt_r, lri, lrt, th0i, th0t, tgi, tgt, perms)
        gi, gt_, gli, glt = torch.autograd.grad(grand, [img_r, txt_r, lri, lrt])
        return grand.detach(), gi

    def fused():
        o = eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=torch.stack(perms))
        return o["grand_loss"], o["image_syn"]

    res = {}
    for name, fn, reps in (("dropin_autograd", dropin, 4), ("fused_engine", fused, 6)):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps
        print("%s %s %s: %.2f ms/iter = %.2f it/s  grand %.5f |g_img| %.4e"
              % (wl, dtype, name, res[name] * 1e3, 1.0 / res[name], float(out[0]), float(out[1].norm())), flush=True)


if __name__ == "__main__":
    main()
