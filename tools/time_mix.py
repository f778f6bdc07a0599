#!/usr/bin/env python3
"""Time mixed-precision pass assignments of the fp32-storage engine on the benched workload (C2):
    python tools/time_mix.py xxxx xxxh xxhh xhhh        (x = split bf16, h = one bf16, f = exact fp32; order F B T-F T-B)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_dataset_distillation_amd.engine import UnrollEngine  # noqa: E402
from multimodal_dataset_distillation_amd.networks import student_move_normalised_targets, synthetic_expert_params  # noqa: E402

CODE = {"x": 1, "h": 2, "f": 3, "g": 4}


def main():
    n, K, size, d_txt = 100, 8, 224, 768
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    image_syn = torch.randn(n, 3, size, size, generator=g).to(dev)
    text_syn = (torch.randn(n, d_txt, generator=g) * 0.5253).to(dev)
    lr = torch.tensor([0.1, 0.1], device=dev)
    perms = torch.stack([torch.randperm(n, generator=g) for _ in range(K)]).to(dev)
    ref = None
    for mix in ["ffff"] + sys.argv[1:]:
        eng = UnrollEngine("nfnet_l0", batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K,
                           dtype="bf16x2" if mix != "ffff" else "f32", device=dev)
        if mix != "ffff":
            eng.set_pass_precision(*[CODE[c] for c in mix])
        if ref is None:
            th0i, th0t = synthetic_expert_params(eng, seed=100, device=dev)
            gt = torch.Generator(device=dev).manual_seed(200)
            tgi, tgt, _, _ = student_move_normalised_targets(eng, th0i, th0t, image_syn, text_syn, lr, K, gt)
        out = None
        for _ in range(2):
            out = eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        cur = {k: out[k].double().cpu().clone() for k in ("image_syn", "text_syn", "lr")}
        cur["grand"] = out["grand_loss"].double().cpu().clone()
        if ref is None:
            ref = cur
        rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
        print(json.dumps({"mix": mix, "ms_per_iter": round(ms, 1), "it_per_s": round(1e3 / ms, 2),
                          "vs_f32": {k: rel(cur[k], ref[k]) for k in ("grand", "image_syn", "text_syn", "lr")}}), flush=True)
        eng.close()
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
