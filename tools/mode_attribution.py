#!/usr/bin/env python3
"""Per-pass error attribution of the reduced-precision modes.

An fp32-storage engine runs each of the four image-encoder passes -- F (forward), B (inner gradient), T-F and T-B
(the tangent passes of the reverse sweep) -- with either split-bf16 operands ("x": hi + lo, 16 significand bits) or
one bf16 per operand ("h": what the bf16 mode's matrix cores see).  For every combination the three north-star outputs
(matching loss, image_syn / text_syn after the reference's SGD(lr=1000) step) are compared with the CPU oracle's
goldens: tests/golden/unroll_c2s_scalars.npz (N=100 @224, one step) and unroll_k8_scalars.npz (N=12 @224, eight steps).
    python tools/mode_attribution.py [--golden c2s k8] [--time]        (on the GPU box)
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


CODE = {"x": 1, "h": 2, "f": 3, "g": 4}


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load_case(name):
    from oracle import distill_ref as dr, nfnet_ref as nr
    g = np.load(os.path.join(ROOT, "tests", "golden", "unroll_%s_scalars.npz" % name))
    n, size, d_txt, K, seed = int(g["n"]), int(g["size"]), int(g["d_txt"]), int(g["K"]), int(g["seed"])
    torch.manual_seed(seed)
    enc = nr.ImageEncoder(str(g["variant"]))
    nr.randomize_like_trained(enc, seed + 1)
    head = dr.ProjectionHead(d_txt, enc.model.num_features)
    th0i, th0t = dr.FlatModule(enc).flat_param(), dr.FlatModule(head).flat_param()
    gen = torch.Generator().manual_seed(seed + 2)
    tgi = th0i + float(g["sig_img"]) * torch.randn(th0i.shape, generator=gen)
    tgt = th0t + float(g["sig_txt"]) * torch.randn(th0t.shape, generator=gen)
    img, txt = dr.synthetic_inputs(n, size, d_txt, seed=seed + 3)
    return dict(g=g, n=n, size=size, d_txt=d_txt, K=K, variant=str(g["variant"]), th0i=th0i, th0t=th0t, tgi=tgi,
                tgt=tgt, img=img, txt=txt)


def run(case, dtype, passes=None, timeit=False):
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    dev = "cuda"
    g = case["g"]
    eng = UnrollEngine(case["variant"], batch=case["n"], num_queries=case["n"], image_size=case["size"],
                       d_txt=case["d_txt"], syn_steps=case["K"], dtype=dtype)
    if passes is not None:
        eng.set_pass_precision(*passes)
    lr = torch.tensor([0.1, 0.1], device=dev)
    args = (case["img"].to(dev), case["txt"].to(dev), lr[0:1], lr[1:2], case["th0i"].to(dev), case["th0t"].to(dev),
            case["tgi"].to(dev), case["tgt"].to(dev))
    perms = torch.from_numpy(g["perms"][0]).to(dev)
    out = eng.unrolled_match(*args, perms=perms)
    torch.cuda.synchronize()
    ms = None
    if timeit:
        t0 = time.perf_counter()
        for _ in range(3):
            eng.unrolled_match(*args, perms=perms, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
    gi = out["image_syn"].cpu()
    sl = (slice(None), slice(None), slice(None, None, 37), slice(None, None, 41))
    img, txt = case["img"], case["txt"]
    ref_slice = torch.from_numpy(g["it0_g_image_syn_slice"])
    e = dict(grand=abs(out["grand_loss"].item() - float(g["it0_grand"])) / abs(float(g["it0_grand"])),
             image_syn_after=rel(img[sl] - 1000.0 * gi[sl], img[sl] - 1000.0 * ref_slice),
             text_syn_after=rel(txt - 1000.0 * out["text_syn"].cpu(), torch.from_numpy(g["it0_text_syn_after"])),
             g_img=rel(gi[sl], ref_slice), g_txt=rel(out["text_syn"], torch.from_numpy(g["it0_g_text_syn"])))
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return e, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--golden", nargs="+", default=["c2s", "k8"])
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--combos", nargs="*", default=None,
                    help="pass assignments F B T-F T-B, e.g. xxxh gggg (x split bf16, h one bf16, g one fp16, f exact fp32); "
                         "default: all 16 x/h combinations")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mode_attribution.json"))
    a = ap.parse_args()
    rows = []
    for name in a.golden:
        case = load_case(name)
        for dtype in ("f32", "bf16"):
            e, ms = run(case, dtype, None, a.time)
            rows.append(dict(golden=name, mode=dtype, ms=ms, **e))
            print(json.dumps(rows[-1]), flush=True)
        combos = a.combos if a.combos else ["".join(c) for c in itertools.product("xh", repeat=4)]
        for combo in combos:                                   # F, B, T-F, T-B
            passes = [CODE[c] for c in combo]
            e, ms = run(case, "bf16x2", passes, a.time)
            rows.append(dict(golden=name, mode="".join(combo), ms=ms, **e))
            print(json.dumps(rows[-1]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)
    # markdown table
    print("\n| golden | F B T-F T-B | grand | image_syn_after | text_syn_after | g_img | g_txt | ms |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.1e | %.1e | %.1e | %.1e | %.1e | %s |" % (
            r["golden"], r["mode"], r["grand"], r["image_syn_after"], r["text_syn_after"], r["g_img"], r["g_txt"],
            "-" if r["ms"] is None else "%.0f" % r["ms"]))


if __name__ == "__main__":
    main()
