"""Stage-2 driver: vision-language dataset distillation by bi-trajectory matching, MI355X engine.

Same command line as the reference's distill.py (flags of reference distill.py:625-679, unknown
flags tolerated as at :680-682) plus a few additive ones (`--engine`, `--compute_dtype`,
`--pix_init/--txt_init`, `--synthetic_experts`, `--save_dir`).  The outer loop mirrors
reference distill.py:288-620 for the HOT PATH rows of SURVEY 8a:

  expert rotation + start-epoch sampling   distill.py:450-470   (buffers resident in HBM)
  syn_steps unrolled student training      distill.py:509-583   engine.unrolled_match (fused) or
                                                                the reference loop verbatim through
                                                                ReparamModule + torch.autograd
  trajectory-matching loss, NaN break      distill.py:584-600
  outer backward + 3x SGD(momentum=0.5)    distill.py:603-613

Out of scope here (SURVEY 8f "next"): evaluation of the synthetic set, wandb logging, image grids,
real-data initialisation (needs the dataset + a frozen BERT; offline neither exists) -- the
synthetic set is initialised with upstream's noise init (distill_original.py:138-148).
The student is built ONCE (the reference rebuilds timm+BERT every iteration, distill.py:440).
Multi-GPU: launch with torch.distributed.run; each rank matches its own expert trajectory and the
synthetic-set gradient is averaged with one RCCL all-reduce per iteration (SURVEY 8e mode A).
"""
import argparse
import ctypes as C
import datetime
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F


def build_parser():
    p = argparse.ArgumentParser(description="Parameter Processing")
    # ---- reference distill.py:625-679 (same names, types, defaults)
    p.add_argument("--distributed", action="store_true")
    p.add_argument("--max_files", type=int, default=1)
    p.add_argument("--dataset", type=str, default="roco", choices=["roco", "coco", "flickr"])
    p.add_argument("--num_queries", type=int, default=100)
    p.add_argument("--lr_img", type=float, default=1000)
    p.add_argument("--lr_txt", type=float, default=1000)
    p.add_argument("--lr_lr", type=float, default=1e-03)
    p.add_argument("--Iteration", type=int, default=50000)
    p.add_argument("--eval_it", type=int, default=50)
    p.add_argument("--num_eval", type=int, default=5)
    p.add_argument("--epoch_eval_train", type=int, default=1)
    p.add_argument("--syn_steps", type=int, default=20)
    p.add_argument("--mini_batch_size", type=int, default=100)
    p.add_argument("--max_start_epoch", type=int, default=25)
    p.add_argument("--expert_epochs", type=int, default=3)
    p.add_argument("--ipc", type=int, default=1)
    p.add_argument("--force_save", action="store_true")
    p.add_argument("--draw", type=bool, default=True)
    p.add_argument("--transfer", type=bool, default=False)
    p.add_argument("--std", type=bool, default=False)
    p.add_argument("--disable_wandb", action="store_true")
    p.add_argument("--num_experts", type=int, default=100)
    p.add_argument("--lr_teacher_img", type=float, default=0.1)
    p.add_argument("--lr_teacher_txt", type=float, default=0.1)
    p.add_argument("--batch_train", type=int, default=128)
    p.add_argument("--dsa", type=str, default="True", choices=["True", "False"])
    p.add_argument("--dsa_strategy", type=str, default="color_crop_cutout_flip_scale_rotate")
    p.add_argument("--data_path", type=str, default="/kaggle/input/roco-dataset/")
    p.add_argument("--buffer_path", type=str, default="/kaggle/working")
    p.add_argument("--train_epochs", type=int, default=50)
    p.add_argument("--zca", action="store_true")
    p.add_argument("--decay", action="store_true")
    p.add_argument("--mom", type=float, default=0)
    p.add_argument("--l2", type=float, default=0)
    p.add_argument("--save_interval", type=int, default=10)
    p.add_argument("--name", type=str, default=datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"))
    p.add_argument("--text_pretrained", type=bool, default=True)
    p.add_argument("--image_pretrained", type=bool, default=True)
    p.add_argument("--text_trainable", type=bool, default=False)
    p.add_argument("--image_trainable", type=bool, default=True)
    p.add_argument("--batch_size_train", type=int, default=128)
    p.add_argument("--batch_size_test", type=int, default=128)
    p.add_argument("--image_root", type=str, default="")
    p.add_argument("--ann_root", type=str, default="")
    p.add_argument("--image_size", type=int, default=224)
    p.add_argument("--k_test", type=int, default=128)
    p.add_argument("--load_npy", type=bool, default=False)
    p.add_argument("--image_encoder", type=str, default="nfnet")
    p.add_argument("--text_encoder", type=str, default="bert", choices=["bert", "clip"])
    p.add_argument("--margin", default=0.2, type=float)
    p.add_argument("--measure", default="cosine")
    p.add_argument("--max_violation", action="store_true")
    p.add_argument("--only_has_image_projection", type=bool, default=False)
    p.add_argument("--grounding", type=bool, default=False)
    p.add_argument("--distill", type=bool, default=False)
    # ---- additive
    p.add_argument("--engine", default="fused", choices=["fused", "autograd"],
                   help="fused: one C-ABI call per iteration; autograd: the reference loop verbatim "
                        "through ReparamModule + torch.autograd over the HIP ops")
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--pix_init", default="noise", choices=["noise", "real"])   # distill_original.py:138
    p.add_argument("--txt_init", default="noise", choices=["noise", "real"])   # distill_original.py:146
    p.add_argument("--logit_scale", type=float, default=None,
                   help="constant logit scale (upstream distill_original.py:430); default = the "
                        "fork's behaviour: syn_lr_img doubles as the scale (distill.py:548)")
    p.add_argument("--synthetic_experts", type=int, nargs=2, metavar=("E", "T"), default=None,
                   help="generate E synthetic expert trajectories of T snapshots instead of reading "
                        "--buffer_path (no dataset/pretrained weights exist offline)")
    p.add_argument("--save_dir", type=str, default=None)
    p.add_argument("--seed", type=int, default=0)
    return p


def nearest_neighbor(sentences, query_embeddings, database_embeddings, device="cuda", return_index=False):
    """reference distill.py:89-95 (decode of `text_syn` at :244 / :374): for each query embedding the
    sentence whose embedding is most cosine-similar.  One `mdd_nearest_neighbor` launch chain on the GPU
    (norms, [Q x N] cosine GEMM, row argmax) instead of Q sklearn calls over the whole bank."""
    from . import _lib
    lib = _lib.load()
    q = torch.as_tensor(np.asarray(query_embeddings) if not torch.is_tensor(query_embeddings) else query_embeddings)
    b = torch.as_tensor(np.asarray(database_embeddings) if not torch.is_tensor(database_embeddings) else database_embeddings)
    q, b = q.detach().float().to(device).contiguous(), b.detach().float().to(device).contiguous()
    if q.dim() != 2 or b.dim() != 2 or q.shape[1] != b.shape[1]:
        raise ValueError("nearest_neighbor: [Q,D] queries and [N,D] database expected")
    if sentences is not None and len(sentences) != b.shape[0]:
        raise ValueError("nearest_neighbor: one sentence per database row expected")
    scores = torch.empty(q.shape[0], b.shape[0], device=device)
    norms = torch.empty(q.shape[0] + b.shape[0], device=device)
    idx = torch.empty(q.shape[0], dtype=torch.int32, device=device)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.mdd_nearest_neighbor(P(q), P(b), q.shape[0], b.shape[0], q.shape[1], P(scores), P(norms),
                                        P(idx), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ids = idx.cpu().tolist()
    if return_index:
        return ids
    return [sentences[i] for i in ids]


def init_synthetic_set(args, d_txt, device):
    """upstream noise init (distill_original.py:138-148)."""
    if args.pix_init != "noise" or args.txt_init != "noise":
        raise NotImplementedError(
            "real-pair initialisation needs the dataset and a frozen BERT text encoder "
            "(reference distill.py:97-105); neither is part of the MI355X hot path -- use noise init")
    mean = torch.tensor([-0.0626, -0.0221, 0.0680]).view(1, 3, 1, 1)
    std = torch.tensor([1.0451, 1.0752, 1.0539]).view(1, 3, 1, 1)
    image_syn = torch.randn([args.num_queries, 3, args.image_size, args.image_size]) * std + mean
    text_syn = torch.normal(mean=-0.0094, std=0.5253, size=(args.num_queries, d_txt))
    return image_syn.to(device).contiguous(), text_syn.float().to(device).contiguous()


def reference_loop_iteration(img_net, txt_net, image_syn, text_syn, syn_lr_img, syn_lr_txt, th0_img,
                             th0_txt, tgt_img, tgt_txt, perms, logit_scale=None):
    """reference distill.py:509-598, verbatim semantics, on ReparamModule-wrapped HIP students."""
    img_p = [th0_img.detach().clone().requires_grad_(True)]
    txt_p = [th0_txt.detach().clone().requires_grad_(True)]
    ces = []
    for idx in perms:
        x = img_net(image_syn[idx], flat_param=img_p[-1])
        x = x / x.norm(dim=1, keepdim=True)
        y = txt_net(text_syn[idx], flat_param=txt_p[-1])
        y = y / y.norm(dim=1, keepdim=True)
        scale = syn_lr_img if logit_scale is None else logit_scale
        logits = scale * x.float() @ y.float().t()
        gt = torch.arange(len(logits), device=logits.device)
        loss = (F.cross_entropy(logits, gt) + F.cross_entropy(logits.t(), gt)) / 2
        ig = torch.autograd.grad(loss, img_p[-1], create_graph=True)[0]
        tg = torch.autograd.grad(loss, txt_p[-1], create_graph=True)[0]
        ces.append(loss.detach())
        img_p.append(img_p[-1] - syn_lr_img * ig)
        txt_p.append(txt_p[-1] - syn_lr_txt * tg)
    img_loss = F.mse_loss(img_p[-1], tgt_img, reduction="sum") / F.mse_loss(th0_img, tgt_img, reduction="sum")
    txt_loss = F.mse_loss(txt_p[-1], tgt_txt, reduction="sum") / F.mse_loss(th0_txt, tgt_txt, reduction="sum")
    return img_loss + txt_loss, img_loss, txt_loss, ces


def main(args):
    from . import _lib
    from .engine import UnrollEngine
    from .expert_buffer import ExpertBuffer, list_expert_files, shuffle_files, synthetic_buffer
    from .networks import CLIPModel_full, VARIANTS
    from .reparam_module import ReparamModule

    if not torch.cuda.is_available():
        raise RuntimeError("distill.py needs an MI355X; the engine has no CPU path")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    args.device = str(device)
    torch.manual_seed(args.seed)
    # world > 1 without --distributed: mode A (every rank its own expert, gradients averaged);
    # with --distributed: mode B = the reference flag's meaning (nn.DataParallel, distill.py:443-445):
    # every rank a chunk of each minibatch, SAME expert / start epoch / permutations everywhere
    mode_b = bool(args.distributed) and world > 1
    np.random.seed(args.seed + (0 if mode_b else rank))
    if args.image_encoder not in VARIANTS:
        raise NotImplementedError("hot path encoders: %s" % sorted(VARIANTS))
    variant = VARIANTS[args.image_encoder]
    d_txt = 768 if args.text_encoder == "bert" else 512
    batch = min(args.mini_batch_size, args.num_queries)
    if mode_b and batch % world:
        raise ValueError("--distributed: mini_batch_size %d is not divisible by %d ranks" % (batch, world))
    if mode_b and args.engine != "fused":
        raise NotImplementedError("--distributed sample sharding runs on the fused engine passes")

    eng = UnrollEngine(variant, batch=batch // world if mode_b else batch, num_queries=args.num_queries,
                       image_size=args.image_size,
                       d_txt=d_txt, syn_steps=args.syn_steps, dtype=args.compute_dtype, device=device)
    lib = _lib.load()
    image_syn, text_syn = init_synthetic_set(args, d_txt, device)
    lr = torch.tensor([args.lr_teacher_img, args.lr_teacher_txt], device=device)  # syn_lr_img/txt
    n_img, n_txt = image_syn.numel(), text_syn.numel()
    grads = torch.zeros(n_img + n_txt + 2, device=device)
    mom = torch.zeros_like(grads)
    out = dict(image_syn=grads[:n_img].view_as(image_syn), text_syn=grads[n_img:n_img + n_txt].view_as(text_syn),
               lr=grads[n_img + n_txt:], losses=torch.zeros(3 + args.syn_steps, device=device))

    # ---- expert buffers (distill.py:255-283)
    if args.synthetic_experts:
        E, T = args.synthetic_experts
        buf = synthetic_buffer(eng, E, T, seed=args.seed + (0 if mode_b else rank), device=device)
        files = None
    else:
        img_files, txt_files = list_expert_files(args.buffer_path)
        if not img_files:
            raise FileNotFoundError("no img_replay_buffer_*.pt under %s (use --synthetic_experts E T "
                                    "for a synthetic run)" % args.buffer_path)
        img_files, txt_files = shuffle_files(img_files, txt_files)
        files = (img_files, txt_files)
        print("loading file {}".format(img_files[0]))
        buf = ExpertBuffer.from_files(img_files[0], txt_files[0], eng.P_img, eng.P_txt, device)
    file_idx = expert_idx = 0

    student = img_net = txt_net = None
    if args.engine == "autograd":
        args.compute_dtype = args.compute_dtype
        student = CLIPModel_full(args)
        img_net = ReparamModule(student.image_encoder).to(device)
        txt_net = ReparamModule(student.text_projection).to(device)
        img_net.train(), txt_net.train()
        txt_net.module.p = 0.1

    P = lambda t: C.c_void_p(t.data_ptr())
    first_step = True
    t_start = time.time()
    for it in range(args.Iteration + 1):
        # ---- expert rotation (distill.py:450-465) and trajectory segment (:466-470)
        e_idx = expert_idx
        expert_idx += 1
        if expert_idx == buf.num_experts:
            expert_idx = 0
            if files is not None:
                file_idx += 1
                if file_idx == len(files[0]):
                    file_idx = 0
                    files = shuffle_files(*files)
                if args.max_files != 1:
                    print("loading file {}".format(files[0][file_idx]))
                    buf = ExpertBuffer.from_files(files[0][file_idx], files[1][file_idx], eng.P_img,
                                                  eng.P_txt, device)
        start_epoch = np.random.randint(0, args.max_start_epoch)
        th0i, th0t, tgi, tgt = buf.pick(e_idx, start_epoch, args.expert_epochs)
        perms = torch.stack([torch.randperm(args.num_queries)[:batch] for _ in range(args.syn_steps)]).to(device)

        if mode_b:
            from . import parallel as par
            gm = torch.Generator(device=device).manual_seed(args.seed * 1_000_003 + it)   # same masks on every rank
            masks = (torch.rand(args.syn_steps, batch, eng.feature_dim, device=device, generator=gm) >= 0.1).float() / 0.9
            ob = par.run_collectives(par.sharded_unrolled_match(
                eng, rank, world, image_syn, text_syn, lr, th0i, th0t, tgi, tgt, perms, drop_masks=masks,
                logit_scale=args.logit_scale))
            out["image_syn"].copy_(ob["image_syn"]), out["text_syn"].copy_(ob["text_syn"]), out["lr"].copy_(ob["lr"])
            losses = torch.cat([torch.stack([ob["grand_loss"], ob["img_loss"], ob["txt_loss"]]), ob["contrastive"]])
        elif args.engine == "fused":
            masks = None
            if True:  # the student text projection is in train mode (distill.py:446-447): Dropout(0.1)
                masks = (torch.rand(args.syn_steps, batch, eng.feature_dim, device=device) >= 0.1).float() / 0.9
            eng.unrolled_match(image_syn, text_syn, lr[0:1], lr[1:2], th0i, th0t, tgi, tgt, perms=perms,
                               drop_masks=masks, logit_scale=args.logit_scale, out=out)
            losses = out["losses"]
        else:
            img_r = image_syn.detach().requires_grad_(True)
            txt_r = text_syn.detach().requires_grad_(True)
            lri = lr[0].detach().clone().requires_grad_(True)
            lrt = lr[1].detach().clone().requires_grad_(True)
            grand, il, tl, ces = reference_loop_iteration(img_net, txt_net, img_r, txt_r, lri, lrt, th0i, th0t,
                                                          tgi, tgt, list(perms), args.logit_scale)
            gi, gt_, gli, glt = torch.autograd.grad(grand, [img_r, txt_r, lri, lrt])
            out["image_syn"].copy_(gi), out["text_syn"].copy_(gt_)
            out["lr"].copy_(torch.stack([gli, glt]))
            losses = torch.stack([grand.detach(), il.detach(), tl.detach()] + ces)
        if world > 1 and not mode_b:
            dist.all_reduce(grads)
            grads.div_(world)
        # NaN -> leave the loop (distill.py:599-600); one host sync per iteration
        lh = losses[:3].tolist()
        if math.isnan(lh[1]):
            print("img_param_loss is NaN at iteration %d: stopping (reference distill.py:599)" % it)
            break
        # ---- optimizer_lr / optimizer_img / optimizer_txt .step() (distill.py:233-241, 611-613)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for p, off, cnt, lrv in ((image_syn, 0, n_img, args.lr_img), (text_syn, n_img, n_txt, args.lr_txt),
                                 (lr, n_img + n_txt, 2, args.lr_lr)):
            _lib.check(lib.mdd_flat_sgd_momentum(P(p), P(grads[off:off + cnt]), P(mom[off:off + cnt]),
                                                 float(lrv), 0.5, 1 if first_step else 0, cnt, st))
        first_step = False
        if it % 10 == 0 and rank == 0:
            print("%s iter = %04d, loss = %.4f (img %.4f txt %.4f) start_epoch=%d  %.2f it/s"
                  % (time.strftime("[%Y-%m-%d %H:%M:%S]"), it, lh[0], lh[1], lh[2], start_epoch,
                     (it + 1) / (time.time() - t_start)))
        if args.save_dir and rank == 0 and (it % max(1, args.save_interval * 10) == 0 or it == args.Iteration):
            os.makedirs(args.save_dir, exist_ok=True)
            torch.save({"image_syn": image_syn.cpu(), "text_syn": text_syn.cpu(), "syn_lr": lr.cpu(), "it": it},
                       os.path.join(args.save_dir, "distilled_%s.pt" % args.dataset))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return image_syn, text_syn, lr


if __name__ == "__main__":
    parser = build_parser()
    a, unknown = parser.parse_known_args()
    if unknown:
        print("Warning: Ignoring unknown arguments:", unknown)   # reference distill.py:680-682
    main(a)
