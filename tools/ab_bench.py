#!/usr/bin/env python3
"""A/B timing of library builds on the benched workload (C2 iteration, no SGD / self-check):
    python tools/ab_bench.py [--dtype bf16] [--steps 6] libA.so libB.so ...
Each library runs in its own child process (the C-ABI library is loaded once per process); prints
ms/iteration and the per-class HIP-event times of one instrumented iteration."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(lib_path, dtype, steps, workload, own_stream=False):
    import torch
    if own_stream:
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1 if own_stream == "high" else 0))
    from multimodal_dataset_distillation_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib_path)
    os.environ.pop("MDD_HIP_LIB", None)
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from multimodal_dataset_distillation_amd.networks import (student_move_normalised_targets,
                                                               synthetic_expert_params)
    variant, n, K, size, d_txt = {"c2": ("nfnet_l0", 100, 8, 224, 768), "c1": ("nfnet_l0", 10, 2, 224, 768),
                                 "c5": ("vit_b16", 100, 8, 224, 512)}[workload]
    dev = torch.device("cuda", 0)
    eng = UnrollEngine(variant, batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K, dtype=dtype, device=dev)
    lib = _lib.load()
    if os.environ.get("AB_NO_PIPE"):          # A/B of the 256 x 256 pipelined kernels inside one build
        lib.mdd_set_pipe_kernels(0)
    g = torch.Generator().manual_seed(0)
    image_syn = torch.randn(n, 3, size, size, generator=g).to(dev)
    text_syn = (torch.randn(n, d_txt, generator=g) * 0.5253).to(dev)
    lr = torch.tensor([0.1, 0.1], device=dev)
    th0i, th0t = synthetic_expert_params(eng, seed=100, device=dev)
    gt = torch.Generator(device=dev).manual_seed(200)
    tgi, tgt, _, _ = student_move_normalised_targets(eng, th0i, th0t, image_syn, text_syn, lr, K, gt)
    perms = torch.stack([torch.randperm(n, generator=g) for _ in range(K)]).to(dev)
    out = None
    for _ in range(2):
        out = eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    lib.mdd_engine_profile(eng.h, 1)
    eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms, out=out)
    torch.cuda.synchronize()
    buf = (C.c_double * 4)()
    kinds = []
    for k in range(5):
        if lib.mdd_engine_profile_read(eng.h, k, buf) != 0:     # an older build without this class
            kinds.append(None)
            continue
        kinds.append(round(buf[1], 2))
    lib.mdd_engine_profile(eng.h, 0)
    print(json.dumps({"lib": os.path.basename(lib_path), "dtype": dtype, "ms_per_iter": round(ms, 2),
                      "grand": float(out["grand_loss"]), "gnorm": float(out["image_syn"].norm()),
                      "class_ms[128x32,256x64,128x128,wgrad,wgrad_reduce]": kinds}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", default=None)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--stream", nargs="?", const="normal", default=None, choices=["normal", "high"],
                    help="run on a non-default HIP stream (graph-capture builds); 'high' = a high-priority stream")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.one:
        one(a.one, a.dtype, a.steps, a.workload, a.stream)
        return
    for lib in a.libs:
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--one", lib, "--dtype", a.dtype,
                              "--steps", str(a.steps), "--workload", a.workload] + (["--stream", a.stream] if a.stream else []))
        if rc:
            print("FAILED", lib, rc, flush=True)


if __name__ == "__main__":
    main()
