#!/usr/bin/env python3
"""Benchmark of the hot path: distillation iterations/sec (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one outer distillation iteration on one expert trajectory (reference
distill.py:466-613 minus eval/logging/student rebuild): syn_steps unrolled student steps
(NFNet-l0 + text projection forward, contrastive loss, inner gradient, theta update), the
normalised trajectory-matching loss, the outer gradient w.r.t. (image_syn, text_syn, lr_img, lr_txt)
and the three SGD-momentum updates.  Workload at N=1: BASELINE config 2 (100 synthetic pairs,
syn_steps=8, NFNet-l0 + 768-d text, 224x224, bf16), synthetic inputs + synthetic expert snapshots
resident in HBM.  Multi-GPU (SURVEY 8e mode A): one expert replica per rank, full synthetic set
replicated, ONE RCCL all-reduce of [d image_syn | d text_syn | d lr] per step, identical SGD step on
every rank; value = expert-iterations/sec over all ranks (weak scaling).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (variant, pairs, syn_steps, image, d_txt)
    "c2": ("nfnet_l0", 100, 8, 224, 768),     # BASELINE.json configs[1] -- the metric's config
    "c1": ("nfnet_l0", 10, 2, 224, 768),      # configs[0] (reference's CPU-runnable case)
    "c4": ("nfnet_l1", 500, 16, 224, 768),    # configs[3] per GPU (mode A); needs --keep-steps 0|1 to fit 288 GB
    "c5": ("vit_b16", 100, 8, 224, 512),      # configs[4] per GPU: ViT-B/16 image encoder + 512-d CLIP text embeddings
    "tiny": ("nfnet_tiny", 4, 2, 64, 32),     # plumbing
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
KIND_NAMES = ["conv_gemm<128x32>", "conv_gemm<256x64>", "conv_gemm<128x128>", "conv_wgrad", "k_wgrad_reduce"]
TRAFFIC_FILES = ["traffic_r03.json", "traffic_r02.json", "traffic_r01.json"]      # newest first
# Budget for the post-timing self-check: relative error of the benched mode's outer gradients against ONE
# f32-mode (exact-fp32 MFMA) iteration on the same inputs.  ~2x the errors measured on MI355X.
SELFCHECK_BUDGET = {"bf16": dict(grand=5e-3, g_img=1e-1, g_txt=9e-2, g_lr=2e-2),
                    "bf16x2": dict(grand=1e-4, g_img=3e-3, g_txt=3e-3, g_lr=1e-3),
                    "f32": dict(grand=1e-5, g_img=1e-4, g_txt=1e-4, g_lr=1e-4)}


def algorithmic_flops_per_iter(n, syn_steps, variant="nfnet_l0"):
    """9 contractions of forward size per conv/linear per step."""
    if variant == "vit_b16":
        T, D, L = 197, 768, 12     # per image: patch embedding + L x (qkv, proj, fc1, fc2, q k^T, p v) + 512->768 head
        img = 196 * D * D + L * (T * (3 * D * D + D * D + 8 * D * D) + 2 * T * T * D)
        macs_fwd = n * (img + 512 * D + D * D) + n * n * D
        return syn_steps * 9 * 2 * macs_fwd
    macs_fwd = n * (4.2419e9 + 7.08e6) + n * n * 2304
    return syn_steps * 9 * 2 * macs_fwd


def cpu_baseline(workload, seconds_budget=30.0):
    """The oracle (oracle/distill_ref.py, torch CPU fp32 autograd) timed on the host cores.
    SURVEY 8d: BASELINE config 1 (N=10, syn_steps=2, NFNet-l0 @224) is timed IN FULL -- one complete
    outer iteration, no scaling (`c1_full_*`).  The figure for the benched workload (config 2: 100 pairs
    x 8 steps does not fit a bounded CPU sample) is that measurement scaled linearly by pairs*syn_steps
    and is labelled as an extrapolation."""
    from oracle import distill_ref as dr, nfnet_ref as nr
    variant, n, K, size, d_txt = WORKLOADS[workload]
    sn, sK = (min(n, 10), min(K, 2)) if variant in ("nfnet_l0", "vit_b16") else (n, K)
    torch.manual_seed(0)
    if variant.startswith("vit"):
        from oracle import vit_ref as vr
        enc = vr.ImageEncoder(variant)
        vr.randomize_like_trained(enc, 1)
    else:
        enc = nr.ImageEncoder(variant)
        nr.randomize_like_trained(enc, 1)
    fi = dr.FlatModule(enc)
    ft = dr.FlatModule(dr.ProjectionHead(d_txt, enc.model.num_features))
    img, txt = dr.synthetic_inputs(sn, size, d_txt, seed=3)
    img.requires_grad_(True), txt.requires_grad_(True)
    lri, lrt = torch.tensor(0.1, requires_grad=True), torch.tensor(0.1, requires_grad=True)
    th0i, th0t = fi.flat_param(), ft.flat_param()
    tgi, tgt = th0i + 1e-3 * torch.randn_like(th0i), th0t + 1e-3 * torch.randn_like(th0t)
    perms = [torch.randperm(sn) for _ in range(sK)]
    t0 = time.time()
    reps = 0
    while True:
        grand, _ = dr.unrolled_match(fi, ft, img, txt, lri, lrt, th0i, th0t, tgi, tgt, perms)
        dr.outer_grads(grand, img, txt, lri, lrt)
        reps += 1
        if time.time() - t0 > seconds_budget * 0.5 or reps >= 2:
            break
    dt = (time.time() - t0) / reps
    scale = (n * K) / float(sn * sK)
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
    except OSError:
        pass
    out = {"value": 1.0 / (dt * scale), "unit": "iters/sec", "cores": torch.get_num_threads(), "cpu_model": cpu_model,
           "kind": "port", "extrapolated": scale != 1.0, "scale_factor": scale,
           "sample_iters_per_sec": 1.0 / dt, "sample_seconds_per_iter": dt, "sample_reps": reps,
           "sample": "oracle/distill_ref.py torch-CPU fp32: %d pairs x %d syn_step(s) of %s @%d timed in "
                     "full (%.1f s per outer iteration, %d rep(s))%s"
                     % (sn, sK, variant, size, dt, reps,
                        "" if scale == 1.0 else "; `value` = that x1/%.0f (linear in pairs*syn_steps): an "
                        "EXTRAPOLATION to the benched workload, not a measurement" % scale)}
    if variant == "nfnet_l0" and (sn, sK) == (10, 2):
        out["c1_full_iters_per_sec"] = 1.0 / dt      # BASELINE configs[0], measured, unscaled
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bf16x2", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None,
                    help="CSV path: one row per contraction launch of the instrumented iteration")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo lets the N>1 code path "
                         "be rehearsed with several ranks sharing one GPU)")
    ap.add_argument("--collective", default="torch", choices=["torch", "library"],
                    help="who issues the per-step all-reduce at N>1: torch.distributed (default) or libmdd_hip.so's "
                         "own RCCL communicator (mdd_allreduce_syn_grads; needs GPUs RCCL can pair, one per rank)")
    ap.add_argument("--keep-steps", type=int, default=None,
                    help="activation stash policy (engine keep_steps): default keeps every step")
    ap.add_argument("--no-selfcheck", action="store_true",
                    help="skip the post-timing f32-mode gradient check (tools/profile runs)")
    ap.add_argument("--expert-source", default="resident", choices=["resident", "host", "hbm"],
                    help="where each iteration's expert snapshot pair (theta_start, theta_target: reference "
                         "distill.py:450-476) comes from: 'resident' = pointers into HBM (default); 'host' = a pool of "
                         "pairs in pinned host memory, the NEXT iteration's pair prefetched over PCIe on a copy stream "
                         "while this iteration runs (BASELINE configs[3]: expert buffer streaming); 'hbm' = copied out "
                         "of a resident [E, 2, P] pool on the copy stream")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default C2 run only: skip the short configs[3] / configs[4] child runs reported under "
                         "`other_workloads`")
    args = ap.parse_args()

    # A stray MDD_* variable must never shape the headline number: the product library ignores all of
    # them (work-skipping switches exist only in -DMDD_DEBUG_SWITCHES experiment builds, and
    # MDD_HIP_LIB would swap the library) -- refuse to run rather than record a doubtful line.
    stray = sorted(k for k in os.environ if k.startswith("MDD_"))
    if stray:
        sys.stderr.write("bench.py: refusing to run with MDD_* environment variables set: %s\n" % stray)
        sys.exit(3)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called without a launcher: start one rank per GPU as CHILD processes (nothing has touched the
        # GPU in this process yet) and hand their exit code back
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes a version banner to the process's
    # stdout when a communicator is created (seen with `--collective library`; torch's own nccl backend initialises one
    # lazily at N > 1).  From here on file descriptor 1 points at stderr; the result line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from multimodal_dataset_distillation_amd import _lib
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from multimodal_dataset_distillation_amd.networks import synthetic_expert_params

    variant, n, K, size, d_txt = WORKLOADS[args.workload]
    eng = UnrollEngine(variant, batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K,
                       dtype=args.dtype, device=dev, keep_steps=args.keep_steps)
    lib = _lib.load()
    fd_check = args.workload in ("c4", "c5") and not args.no_selfcheck
    if args.workload in ("c4", "c5"):
        args.no_selfcheck = True     # an f32-mode engine of this size does not fit one GPU: the self-check is by
                                     # central differences of the reported loss instead (fd_check below)

    # ---- synthetic inputs: identical on every rank
    g = torch.Generator().manual_seed(0)
    mean = torch.tensor([-0.0626, -0.0221, 0.0680]).view(1, 3, 1, 1)
    std = torch.tensor([1.0451, 1.0752, 1.0539]).view(1, 3, 1, 1)
    image_syn = (torch.randn(n, 3, size, size, generator=g) * std + mean).to(dev)
    text_syn = (torch.randn(n, d_txt, generator=g) * 0.5253 - 0.0094).to(dev)
    lr = torch.tensor([0.1, 0.1], device=dev)            # syn_lr_img, syn_lr_txt (distill.py:647-648)
    # ---- synthetic expert trajectory of THIS rank, resident in HBM: start snapshot + target
    th0i, th0t = synthetic_expert_params(eng, seed=100 + rank, device=dev)
    gt = torch.Generator(device=dev).manual_seed(200 + rank)
    # expert displacement theta* - theta0: isotropic, with the norm of the student's own syn_steps-step move
    # (lr * K * |g0|), as oracle/gen_golden.py does -- so the normalised matching loss (distill.py:596-597)
    # and every outer gradient depend O(1) on what the conv kernels compute (a 1e-3 displacement swamps the
    # student's move: grand_loss = 2 whatever the kernels do).
    from multimodal_dataset_distillation_amd.networks import student_move_normalised_targets
    tgi, tgt, sig_i, sig_t = student_move_normalised_targets(eng, th0i, th0t, image_syn, text_syn, lr, K, gt)
    # ---- expert snapshot source (distill.py:450-476).  'host' / 'hbm': a pool of E pairs [theta_start | theta_target]
    # (image and text networks back to back); every iteration uses pair (i mod E) out of a two-deep device buffer that
    # the copy stream fills one iteration ahead -- the timed loop then contains the reference's per-iteration expert
    # staging (its 478 MB H2D at configs[1], 576 MB at configs[3]) overlapped with the previous iteration.
    xs = None
    if args.expert_source != "resident":
        Pi, Pt = th0i.numel(), th0t.numel()
        E = 3
        pair = torch.cat([th0i, th0t, tgi, tgt])
        pool = torch.empty(E, pair.numel(), device="cpu" if args.expert_source == "host" else dev)
        if args.expert_source == "host":
            pool = pool.pin_memory()
        for e in range(E):
            pool[e].copy_(pair)        # same expert in every slot: the arithmetic (and the self-check) is unchanged
        xs = dict(pool=pool, E=E, buf=[torch.empty_like(pair, device=dev) for _ in range(2)],
                  stream=torch.cuda.Stream(device=dev), ready=[None, None], ev=[], bytes=pair.numel() * 4, it=0,
                  split=(Pi, Pt, Pi, Pt))

        def prefetch(i):
            slot = i & 1
            with torch.cuda.stream(xs["stream"]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                xs["buf"][slot].copy_(xs["pool"][i % xs["E"]], non_blocking=True)
                b.record()
            xs["ready"][slot] = b
            xs["ev"].append((a, b))
        xs["prefetch"] = prefetch
        prefetch(0)
    # minibatch permutations (distill.py:510-511) are drawn ON THE DEVICE: a host randperm + pageable H2D copy is
    # stream-ordered behind the previous iteration and made the host wait for it before it could enqueue the
    # next one (4-5 ms of idle GPU per step).  Same seed -> same permutations on every rank.
    pg = torch.Generator(device=dev).manual_seed(3)
    n_img, n_txt = image_syn.numel(), text_syn.numel()
    from multimodal_dataset_distillation_amd import parallel as par
    flat, views = par.fused_grad_buffer(image_syn, text_syn)   # one fused all-reduce buffer (+ NaN flag)
    flat_grad = views["grads"]
    out = dict(image_syn=views["image_syn"], text_syn=views["text_syn"], lr=views["lr"],
               losses=torch.zeros(3 + K, device=dev))
    mom = torch.zeros_like(flat_grad)
    # --collective library also at N=1: the step then still goes through mdd_comm_create / mdd_allreduce_syn_grads
    # (a one-rank communicator), so the same command line exercises the library-owned RCCL path at any N
    libcoll = par.LibraryCollective(dev) if args.collective == "library" else None
    SGD_LR_SCALE = 1e-6
    ar_events = []          # (start, end) HIP events around each all-reduce (current stream)
    params = [(image_syn, 0, n_img, 1000.0), (text_syn, n_img, n_txt, 1000.0),
              (lr, n_img + n_txt, 2, 1e-3)]                  # distill.py:233-241
    P = lambda t: C.c_void_p(t.data_ptr())
    step_no = [0]

    def one_step():
        perms = torch.stack([torch.randperm(n, generator=pg, device=dev) for _ in range(K)])
        a0i, a0t, ati, att = th0i, th0t, tgi, tgt
        if xs is not None:
            i = xs["it"]
            xs["it"] += 1
            slot = i & 1
            # the copy stream may only overwrite the OTHER slot once the iteration that used it has been enqueued:
            # it waits for everything on the compute stream so far, then fetches pair i+1
            xs["stream"].wait_stream(torch.cuda.current_stream())
            xs["prefetch"](i + 1)
            torch.cuda.current_stream().wait_event(xs["ready"][slot])
            a0i, a0t, ati, att = xs["buf"][slot].split(xs["split"])
        eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], a0i, a0t, ati, att, perms=perms,
                           out=out)
        if world > 1 or libcoll is not None:
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            if libcoll is not None:
                libcoll.all_reduce_mean_(flat)   # the library's own RCCL communicator, this stream
            else:
                dist.all_reduce(flat)        # RCCL; gradients + the collective NaN flag in one message
                flat.div_(world)
            eb.record()
            ar_events.append((ea, eb))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for p, off, cnt, lrv in params:
            _lib.check(lib.mdd_flat_sgd_momentum(P(p), P(flat_grad[off:off + cnt]), P(mom[off:off + cnt]),
                                                 lrv * SGD_LR_SCALE, 0.5, 1 if step_no[0] == 0 else 0, cnt, st))
        step_no[0] += 1

    # NOTE: the SGD learning rates are scaled by 1e-6 here: with SYNTHETIC (untrained) experts the
    # reference's lr=1000 blows the pixels up within a few steps (NaN => the reference breaks out of
    # its loop, distill.py:599); the arithmetic per step is identical.
    for _ in range(args.warmup):
        one_step()
    ar_events.clear()
    if xs is not None:
        xs["ev"].clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    losses = out["losses"].cpu().tolist()
    # gradient norms of the last timed step (of THIS rank's reduced buffer) -- what the kernels produced
    gnorm = dict(grad_image_syn=float(out["image_syn"].norm()), grad_text_syn=float(out["text_syn"].norm()),
                 grad_lr=[float(v) for v in out["lr"].tolist()])
    rank_losses, ar_ms = [losses[0]], None
    if world == 1 and ar_events:
        ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))
    if world > 1:
        gl = [torch.zeros(1, device=dev) for _ in range(world)]
        dist.all_gather(gl, torch.tensor([losses[0]], device=dev))
        rank_losses = [float(t.item()) for t in gl]
        ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))

    result = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.steps / dt
        flops_iter = algorithmic_flops_per_iter(n, K, variant) if variant in ("nfnet_l0", "vit_b16") else None
        result = {
            "metric": "distillation iters/sec (%d syn pairs, syn_steps=%d)" % (n, K), "value": value,
            "unit": "iters/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Flickr30K-shaped, %d synthetic pairs, "
                                   "syn_steps=%d, %s + text projection 768->2304, %dx%d images"
                                   % (n, K, variant, size, size) if args.workload == "c2"
                       else ("BASELINE configs[3] per GPU: COCO-shaped, %d synthetic pairs, syn_steps=%d, %s + text "
                             "projection, %dx%d images, activations recomputed per keep_steps"
                             % (n, K, variant, size, size)) if args.workload == "c4"
                       else ("BASELINE configs[4] per GPU: %d synthetic pairs, syn_steps=%d, ViT-B/16 image encoder "
                             "(%s) + text projection 512->768, %dx%d images" % (n, K, variant, size, size))
                       if args.workload == "c5"
                       else args.workload,
                       "global_batch": n, "syn_steps": K,
                       "keep_steps": eng.keep_steps, "workspace_gib": eng.workspace_bytes / 2**30,
                       "parallelism": "expert-replica x%d (1 all-reduce/step)" % world,
                       "collective_backend": args.dist_backend if world > 1 else None,
                       "collective_issued_by": (args.collective if (world > 1 or libcoll is not None) else None)},
            "grand_loss": losses[0], "grand_loss_per_rank": rank_losses,
            "grad_norms": gnorm,
            "sgd_lr_scale": SGD_LR_SCALE,
            "sgd_lr_note": "the three SGD(momentum 0.5) steps run with lr x %g: with synthetic (untrained) "
                           "experts the reference's lr=1000 drives the pixels to NaN within a few steps "
                           "(distill.py:599 would break); same kernels, same bytes" % SGD_LR_SCALE,
            "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
            "allreduce_ms_per_step": ar_ms,
            "allreduce_bytes": flat.numel() * 4 if world > 1 else 0,
        }
        if xs is not None:
            cms = [a.elapsed_time(b) for a, b in xs["ev"][:args.steps]]
            result["expert_streaming"] = {
                "source": args.expert_source, "bytes_per_iter": xs["bytes"],
                "copy_ms_per_iter": sum(cms) / max(1, len(cms)),
                "streaming_gbps": xs["bytes"] / 1e9 / (sum(cms) / max(1, len(cms)) * 1e-3),
                "note": "pair i+1 is fetched on a copy stream while iteration i runs; the compute stream waits for "
                        "the copy event of its own pair only"}
        if flops_iter:
            result["algorithmic_tflops_per_iter"] = flops_iter / 1e12
            result["mfma_util_pct"] = 100.0 * flops_iter * value / world / (
                (PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS) * 1e12)

    # ---- roofline of the dominant kernel: one extra, HIP-event-instrumented iteration (rank 0)
    if not args.no_roofline:
        # every rank runs the extra iteration (it contains the all-reduce); only rank 0 records
        if rank == 0:
            lib.mdd_engine_profile(eng.h, 1)
        one_step()
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        kinds = []
        buf = (C.c_double * 4)()
        for k in range(len(KIND_NAMES)):
            _lib.check(lib.mdd_engine_profile_read(eng.h, k, buf))
            kinds.append(dict(kernel=KIND_NAMES[k], launches=int(buf[0]), ms=buf[1], flops=buf[2],
                              bytes=buf[3]))
        if args.dump_launches:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_launches)), exist_ok=True)
            _lib.check(lib.mdd_engine_profile_dump(eng.h, args.dump_launches.encode()))
        lib.mdd_engine_profile(eng.h, 0)
        # dominant kernel class = most time per iteration.  Its roofline is the one its ALGORITHMIC
        # arithmetic intensity selects: below the ridge (peak flops / peak HBM bytes) it is HBM-bound.
        dom = max((d for d in kinds if d["flops"] > 0), key=lambda d: d["ms"])
        peak_tf = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS   # bf16x2 issues bf16 MFMAs
        secs = dom["ms"] * 1e-3
        tfs = dom["flops"] / secs / 1e12 if secs > 0 else 0.0
        gbs = dom["bytes"] / secs / 1e9 if secs > 0 else 0.0
        hbm_bound = dom["bytes"] > 0 and dom["flops"] / dom["bytes"] < peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        # HBM traffic per launch comes from SEPARATE rocprofv3 --pmc passes (tools/pmc_traffic.py; the
        # counters cannot be read from inside this process): it is a recorded figure, not measured in this
        # run -- the source file and the commit it was collected at are named beside it.
        traffic, traffic_source = None, None
        for name in (TRAFFIC_FILES if args.workload == "c2" else []):   # the PMC passes were taken on configs[1]
            tf = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tf):
                try:
                    tj = json.load(open(tf))
                    traffic = tj.get(dom["kernel"])
                    traffic_source = {"file": "profiles/" + name, "commit": tj.get("_commit"),
                                      "measured_in_this_run": False}
                except Exception:
                    traffic = None
                break
        result["roofline"] = {"bound": "hbm" if hbm_bound else "mfma",
                              "achieved": gbs if hbm_bound else tfs,
                              "peak": PEAK_HBM_GBS if hbm_bound else peak_tf,
                              "unit": "GB/s" if hbm_bound else "TFLOP/s",
                              "frac": (gbs / PEAK_HBM_GBS) if hbm_bound else (tfs / peak_tf),
                              "traffic": traffic, "traffic_source": traffic_source, "kernel": dom["kernel"],
                              "avg_launch_ms": dom["ms"] / max(1, dom["launches"]),
                              "launches_per_iter": dom["launches"],
                              "algorithmic_bytes_per_launch": dom["bytes"] / max(1, dom["launches"]),
                              "algorithmic_tflops_per_s": tfs,
                              "flops_per_byte": dom["flops"] / max(dom["bytes"], 1.0)}
        result["kernels"] = [dict(kernel=d["kernel"], launches=d["launches"], ms=round(d["ms"], 3),
                                  tflops=round(d["flops"] / max(d["ms"], 1e-9) / 1e9, 2),
                                  gbps=round(d["bytes"] / max(d["ms"], 1e-9) / 1e6, 1)) for d in kinds]
    # ---- self-check (outside the timed region, rank 0, N=1): the SAME iteration (fixed inputs and
    # permutations) once more in the benched mode and once in f32 mode (exact-fp32 MFMA, parity-grade:
    # it matches the CPU oracle's goldens to ~1e-6); the run FAILS when a gradient is non-finite or off by
    # more than the measured budget -- a broken or skipped kernel cannot print a healthy line.
    failed = None
    if rank == 0 and world == 1 and not args.no_selfcheck:
        cperms = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(77 + k))
                              for k in range(K)]).to(dev)
        img_c, txt_c = image_syn.clone(), text_syn.clone()

        def run_check(engine):
            o = engine.unrolled_match(img_c, txt_c, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=cperms)
            torch.cuda.synchronize()
            return dict(grand=o["grand_loss"].double().cpu(), g_img=o["image_syn"].double().cpu(),
                        g_txt=o["text_syn"].double().cpu(), g_lr=o["lr"].double().cpu(),
                        ces=o["contrastive"].double().cpu())
        got = run_check(eng)
        eng.close()
        del eng
        torch.cuda.empty_cache()
        eng32 = UnrollEngine(variant, batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K,
                             dtype="f32", device=dev)
        ref = run_check(eng32)
        eng32.close()
        del eng32
        torch.cuda.empty_cache()
        # the parity-grade fast mode beside the benched one: bf16x2 = fp32 storage, every
        # contraction operand split into hi+lo bf16; same iteration, timed over 2 runs after 1 warm-up
        par = None
        if args.dtype == "bf16" and args.workload == "c2":
            engx = UnrollEngine(variant, batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=K,
                                dtype="bf16x2", device=dev)
            gx = run_check(engx)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                engx.unrolled_match(img_c, txt_c, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=cperms)
            torch.cuda.synchronize()
            tx = (time.perf_counter() - t1) / 2
            engx.close()
            par = {"dtype": "bf16x2", "iters_per_sec": 1.0 / tx, "ms_per_iter": tx * 1e3,
                   "rel_err_vs_f32": {k: float((gx[k] - ref[k]).norm() / (ref[k].norm() + 1e-300))
                                      for k in ("grand", "g_img", "g_txt", "g_lr")}}
        rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
        errs = {k: rel(got[k], ref[k]) for k in ("grand", "g_img", "g_txt", "g_lr")}
        finite = all(bool(torch.isfinite(got[k]).all()) for k in got)
        budget = SELFCHECK_BUDGET[args.dtype]
        ok = finite and all(errs[k] <= budget[k] for k in budget)
        result["selfcheck"] = {"against": "one f32-mode iteration, same inputs", "rel_err": errs,
                               "budget": budget, "finite": finite, "pass": ok,
                               "grand_loss_checked": float(got["grand"]), "grand_loss_f32": float(ref["grand"]),
                               "norms_f32": {k: float(ref[k].norm()) for k in ("g_img", "g_txt", "g_lr")}}
        if par:
            result["parity_mode"] = par
        if not ok:
            failed = "self-check failed: %s (budget %s, finite=%s)" % (errs, budget, finite)
    if rank == 0 and world == 1 and fd_check:
        # No second engine fits beside this one: check the returned gradients against central differences of the
        # grand loss the SAME call reports, along the gradient directions and in syn_lr_img (6 more iterations;
        # steps that move the loss by ~4 % per side -- tests/test_gpu_properties.py explains the step size).
        cperms = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(77 + k))
                              for k in range(K)]).to(dev)
        img_c, txt_c, lr_c = image_syn.clone(), text_syn.clone(), lr.clone()

        def loss_at(im, tx, lrv):
            o = eng.unrolled_match(im, tx, lrv[0:1], lrv[1:2], th0i, th0t, tgi, tgt, perms=cperms)
            torch.cuda.synchronize()
            return o
        o0 = loss_at(img_c, txt_c, lr_c)
        L0 = float(o0["grand_loss"])
        g_img, g_txt, g_lr0 = o0["image_syn"].clone(), o0["text_syn"].clone(), float(o0["lr"][0])
        finite = bool(torch.isfinite(g_img).all() and torch.isfinite(g_txt).all()) and L0 == L0
        errs = {}
        if finite:
            for name, grad in (("g_img", g_img), ("g_txt", g_txt), ("g_lr", None)):
                a = float(grad.norm()) if grad is not None else g_lr0
                e = 0.08 * abs(L0) / (2.0 * abs(a))
                if name == "g_img":
                    d = grad / grad.norm()
                    Lp, Lm = loss_at(img_c + e * d, txt_c, lr_c), loss_at(img_c - e * d, txt_c, lr_c)
                elif name == "g_txt":
                    d = grad / grad.norm()
                    Lp, Lm = loss_at(img_c, txt_c + e * d, lr_c), loss_at(img_c, txt_c - e * d, lr_c)
                else:
                    e2 = torch.tensor([e, 0.0], device=dev)
                    Lp, Lm = loss_at(img_c, txt_c, lr_c + e2), loss_at(img_c, txt_c, lr_c - e2)
                fd = (float(Lp["grand_loss"]) - float(Lm["grand_loss"])) / (2.0 * e)
                errs[name] = abs(fd - a) / abs(a)
        budget = {"g_img": 1e-2, "g_txt": 1e-2, "g_lr": 1e-2}       # measured 5e-4 / 8e-4 / 9e-4 (bf16)
        ok = finite and all(errs[k] <= budget[k] for k in budget)
        result["selfcheck"] = {"against": "central differences of the reported grand loss (no f32 engine fits)",
                               "rel_err": errs, "budget": budget, "finite": finite, "pass": ok,
                               "grand_loss_checked": L0}
        if not ok:
            failed = "self-check failed: %s (budget %s, finite=%s)" % (errs, budget, finite)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if (rank == 0 and world == 1 and args.workload == "c2" and args.dtype == "bf16" and not args.no_other_workloads
            and not args.no_selfcheck):
        # BASELINE configs[3] / configs[4] at their per-GPU shapes, each in a CHILD process after this one has freed
        # its engines (outside every timed region of the C2 line): short runs with their own finite-difference
        # self-checks, reported compactly.  configs[3] streams its expert pairs from pinned host memory.
        import subprocess
        try:
            eng.close()
        except Exception:
            pass
        torch.cuda.empty_cache()
        others = {}
        for name, extra in (("c4", ["--keep-steps", "0", "--steps", "1", "--warmup", "1", "--expert-source", "host"]),
                            ("c5", ["--steps", "4", "--warmup", "1"])):
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--no-cpu-baseline", "--no-roofline",
                   "--no-other-workloads"] + extra
            t1 = time.time()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    others[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
                    continue
                j = json.loads(line[-1])
                others[name] = {k: j.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup",
                                                       "dtype", "mfma_util_pct", "grand_loss", "selfcheck",
                                                       "expert_streaming")}
                others[name]["config"] = j["config"]
                others[name]["wall_s"] = time.time() - t1
            except Exception as ex:      # a child that hangs or dies must not take the C2 line with it
                others[name] = {"error": repr(ex)[:400]}
        result["other_workloads"] = others
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if libcoll is not None:
        libcoll.close()       # ncclCommDestroy before torch's process group goes away
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: " + failed + "\n")
        sys.exit(4)


if __name__ == "__main__":
    main()
