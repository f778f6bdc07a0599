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

Either side of the hot path, wired to the on-disk inputs the reference uses:
  real-pair initialisation   distill.py:97-105, 228   --txt_init real reads the train-caption cache
                             {dataset}_{text_encoder}_train_text_embed.npz (key bert_test_embed,
                             utils.py:885); --pix_init real additionally needs --train_images (a tensor
                             file of the training images in annotation order: the dataset itself is
                             not available offline).  Default: upstream's noise init
                             (distill_original.py:138-148).
  caption decode             distill.py:89-95, 244    nearest_neighbor() on the same cache
  evaluation every eval_it   distill.py:293-357       epoch.evaluate_synset when --eval_data is given
Not reproduced: wandb logging, image grids.
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
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "bf16x2", "f32"])
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
    p.add_argument("--keep_steps", type=int, default=None,
                   help="activation stash policy: keep the activations of the first KEEP inner steps in "
                        "HBM, recompute the others during the outer backward (default: keep all)")
    p.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                   help="torch.distributed backend under a launcher (nccl = RCCL; gloo lets several "
                        "ranks share one GPU in tests)")
    p.add_argument("--collective", default="torch", choices=["torch", "library"],
                   help="mode A's all-reduce issued by torch.distributed (default) or by libmdd_hip.so's own "
                        "RCCL communicator (mdd_allreduce_syn_grads)")
    p.add_argument("--embed_dir", type=str, default=".",
                   help="directory of the {dataset}_{text_encoder}_{text,train_text}_embed.npz caches "
                        "(reference utils.py:885 reads them from the working directory)")
    p.add_argument("--train_images", type=str, default=None,
                   help="tensor file (.pt/.npy/.npz key 'images') [M,3,S,S] of the training images in "
                        "annotation order, for --pix_init real")
    p.add_argument("--train_sentences", type=str, default=None,
                   help="text file, one training caption per line (annotation order), for the caption decode")
    p.add_argument("--eval_data", type=str, default=None,
                   help=".npz with the held-out retrieval set (images, txt2img[, bert_test_embed]); "
                        "enables evaluate_synset every --eval_it iterations")
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


def init_synthetic_set(args, d_txt, device, train_caption_embed=None):
    """Synthetic-set initialisation.
    'real' (the fork, distill.py:97-105 + :228): n random training pairs -- images from --train_images,
    text = the frozen encoder's embedding of the pair's caption = row i of the train-caption cache.
    'noise' (upstream, distill_original.py:138-148): per-channel Gaussian pixels, N(-0.0094, 0.5253) text.
    The two sides can be mixed (upstream has separate --pix_init / --txt_init flags)."""
    from .embed_cache import get_images_texts, load_tensor_file
    image_syn = text_syn = None
    if args.pix_init == "real" or args.txt_init == "real":
        if train_caption_embed is None:
            raise FileNotFoundError("real initialisation needs the train-caption embedding cache")
        train_images = None
        if args.pix_init == "real":
            if not args.train_images:
                raise FileNotFoundError(
                    "--pix_init real needs --train_images (tensor file of the training images in annotation "
                    "order); the dataset loader itself is outside the MI355X path")
            train_images = load_tensor_file(args.train_images, key="images")
            if tuple(train_images.shape[1:]) != (3, args.image_size, args.image_size):
                raise ValueError("--train_images: expected [M,3,%d,%d], got %s"
                                 % (args.image_size, args.image_size, tuple(train_images.shape)))
        if train_caption_embed.shape[1] != d_txt:
            raise ValueError("train caption embeddings are %d-d, the text encoder %r gives %d"
                             % (train_caption_embed.shape[1], args.text_encoder, d_txt))
        img_r, txt_r, _ = get_images_texts(args.num_queries, train_images, train_caption_embed)
        if args.pix_init == "real":
            image_syn = img_r
        if args.txt_init == "real":
            text_syn = txt_r
    if image_syn is None:
        mean = torch.tensor([-0.0626, -0.0221, 0.0680]).view(1, 3, 1, 1)
        std = torch.tensor([1.0451, 1.0752, 1.0539]).view(1, 3, 1, 1)
        image_syn = torch.randn([args.num_queries, 3, args.image_size, args.image_size]) * std + mean
    if text_syn is None:
        text_syn = torch.normal(mean=-0.0094, std=0.5253, size=(args.num_queries, d_txt))
    return image_syn.float().to(device).contiguous(), text_syn.float().to(device).contiguous()


def run_evaluation(args, it, image_syn, text_syn, syn_lr_img, eval_set, eval_eng_box, variant, d_txt, device):
    """The evaluation block of reference distill.py:293-357: num_eval fresh students trained on the
    current synthetic set (evaluate_synset), retrieval recalls printed per run and, with --std, as
    mean/std.  eval_set = (test_images, bert_test_embed, img2txt, txt2img) on `device`."""
    from .engine import UnrollEngine
    from .epoch import evaluate_synset
    print("-------------------------\nEvaluation")
    print("image_model_train = %s, text_model_train = %s, iteration = %d" % (args.image_encoder, args.text_encoder, it))
    if eval_eng_box[0] is None:
        nb = min(args.batch_train, args.num_queries)
        eval_eng_box[0] = UnrollEngine(variant, batch=nb, num_queries=args.num_queries, image_size=args.image_size,
                                       d_txt=d_txt, syn_steps=1, dtype=args.compute_dtype, device=device)
    eng_e = eval_eng_box[0]
    test_images, test_embed, img2txt, txt2img = eval_set
    keys = ("img_r1", "img_r5", "img_r10", "img_r_mean", "txt_r1", "txt_r5", "txt_r10", "txt_r_mean", "r_mean")
    runs = []
    for it_eval in range(args.num_eval):
        lr_net = float(syn_lr_img)                      # args.lr_net = syn_lr_img.item()  (distill.py:312)
        _, _, res = evaluate_synset(eng_e, image_syn.detach().clone(), text_syn.detach().clone(), test_images,
                                    test_embed, img2txt, txt2img, lr_net=lr_net,
                                    epoch_eval_train=args.epoch_eval_train, seed=args.seed * 1009 + it * 31 + it_eval)
        print("Evaluate_%02d: Img R@1 = %.4f, Img R@5 = %.4f, Img R@10 = %.4f, Img R@Mean = %.4f, "
              "Txt R@1 = %.4f, Txt R@5 = %.4f, Txt R@10 = %.4f, Txt R@Mean = %.4f, R@Mean = %.4f"
              % ((it_eval,) + tuple(res[k] for k in keys)))
        runs.append(res)
    summary = {k: (float(np.mean([r[k] for r in runs])), float(np.std([r[k] for r in runs]))) for k in keys}
    if args.std:
        print("  ".join("Mean/%s = %.4f Std/%s = %.4f" % (k, summary[k][0], k, summary[k][1]) for k in keys))
    return summary


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
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        # Evaluation (every --eval_it iterations) runs on rank 0 while the other ranks already wait in the
        # iteration's all-reduce: the process-group timeout must cover a whole evaluate_synset sweep
        # (num_eval x (epoch_eval_train + 1) epochs), not the 10-minute default.
        import datetime
        tmo = datetime.timedelta(minutes=float(os.environ.get("DISTILL_DIST_TIMEOUT_MIN", "240")))
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)   # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend, timeout=tmo)   # gloo: several ranks sharing one GPU (tests)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    args.device = str(device)
    torch.manual_seed(args.seed)
    # world > 1 without --distributed: mode A (every rank its own expert, gradients averaged);
    # with --distributed: mode B = the reference flag's meaning (nn.DataParallel, distill.py:443-445):
    # every rank a chunk of each minibatch, SAME expert / start epoch / permutations everywhere
    mode_b = bool(args.distributed) and world > 1
    # The synthetic set is REPLICATED in both modes: its initial draw (real pairs: np.random.permutation inside
    # get_images_texts, reference distill.py:97-105) must be the same on every rank, so numpy is seeded with the
    # bare seed until the set exists and only then per rank (mode A: every rank its own start epochs).
    np.random.seed(args.seed)
    stray = _lib.stray_env()
    if stray and rank == 0:
        print("note: MDD_* environment variables are set and ignored by the product library:", stray)
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
                       d_txt=d_txt, syn_steps=args.syn_steps, dtype=args.compute_dtype, device=device,
                       keep_steps=args.keep_steps)
    lib = _lib.load()
    from .embed_cache import load_embed_cache, load_eval_data
    # ---- on-disk inputs around the loop (distill.py:221-229): caption-embedding caches
    train_caption_embed = train_sentences = None
    need_train_cache = args.pix_init == "real" or args.txt_init == "real" or args.train_sentences
    if need_train_cache:
        train_caption_embed = load_embed_cache(args, "train_text")
    if args.train_sentences:
        with open(args.train_sentences, "r", encoding="utf-8") as f:
            train_sentences = [ln.rstrip("\n") for ln in f]
        if len(train_sentences) != train_caption_embed.shape[0]:
            raise ValueError("--train_sentences has %d lines, the train-caption cache %d rows"
                             % (len(train_sentences), train_caption_embed.shape[0]))
    eval_set, eval_eng_box = None, [None]
    if args.eval_data:
        ti, te, i2t, t2i = load_eval_data(args.eval_data, args)
        eval_set = (ti.to(device).contiguous(), te.to(device).contiguous(), i2t, t2i)
    eval_it_pool = set(np.arange(0, args.Iteration + 1, max(1, args.eval_it)).tolist())
    image_syn, text_syn = init_synthetic_set(args, d_txt, device, train_caption_embed)
    if world > 1:      # one synthetic set, whatever a rank's RNG state was: rank 0's draw wins
        dist.broadcast(image_syn, 0), dist.broadcast(text_syn, 0)
    np.random.seed(args.seed + (0 if mode_b else rank))
    if train_sentences is not None and rank == 0:      # distill.py:244: decode of the initial text_syn
        sl = nearest_neighbor(train_sentences, text_syn, train_caption_embed, device=device)
        print("original_sentence_list:", " | ".join(sl[:5]), "..." if len(sl) > 5 else "")
    lr = torch.tensor([args.lr_teacher_img, args.lr_teacher_txt], device=device)  # syn_lr_img/txt
    n_img, n_txt = image_syn.numel(), text_syn.numel()
    from . import parallel as par
    # [d image_syn | d text_syn | d lr (2) | NaN flag]: ONE buffer = one all-reduce per iteration in mode A
    flat, views = par.fused_grad_buffer(image_syn, text_syn)
    grads = views["grads"]
    mom = torch.zeros_like(grads)
    out = dict(image_syn=views["image_syn"], text_syn=views["text_syn"], lr=views["lr"],
               losses=torch.zeros(3 + args.syn_steps, device=device))

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
    stopper = par.DeferredStop(device)
    coll = par.LibraryCollective(device) if (world > 1 and not mode_b and args.collective == "library") else None
    nan_at = None
    t_start = time.time()
    for it in range(args.Iteration + 1):
        # ---- evaluation block (distill.py:293-357), when a held-out set was supplied
        if eval_set is not None and it in eval_it_pool and rank == 0:
            run_evaluation(args, it, image_syn, text_syn, lr[0].item(), eval_set, eval_eng_box, variant, d_txt, device)
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
        # NaN -> leave the loop (distill.py:599-600).  The decision is COLLECTIVE (the flag rides in the
        # all-reduced buffer: every rank sees the same value) and taken ON THE DEVICE: the optimiser steps
        # below are guarded by the sticky flag, so nothing is applied from the NaN iteration on, and the host
        # reads the flag one iteration late, after the next iteration has been enqueued -- no host
        # synchronisation between iterations.
        prev = stopper.update(flat, views, losses[:3], reduce=world > 1 and not mode_b, group=coll, iteration=it)
        # ---- optimizer_lr / optimizer_img / optimizer_txt .step() (distill.py:233-241, 611-613)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for p, off, cnt, lrv in ((image_syn, 0, n_img, args.lr_img), (text_syn, n_img, n_txt, args.lr_txt),
                                 (lr, n_img + n_txt, 2, args.lr_lr)):
            _lib.check(lib.mdd_flat_sgd_momentum_guarded(P(p), P(grads[off:off + cnt]), P(mom[off:off + cnt]),
                                                         float(lrv), 0.5, 1 if first_step else 0, cnt,
                                                         P(stopper.sticky), st))
        first_step = False
        if prev is not None:
            stop, pit, lh = stopper.read(prev)
            if stop:
                nan_at = pit
                break
            if pit % 10 == 0 and rank == 0:
                print("%s iter = %04d, loss = %.4f (img %.4f txt %.4f)  %.2f it/s"
                      % (time.strftime("[%Y-%m-%d %H:%M:%S]"), pit, lh[0], lh[1], lh[2],
                         (pit + 1) / (time.time() - t_start)))
        if args.save_dir and rank == 0 and (it % max(1, args.save_interval * 10) == 0 or it == args.Iteration):
            os.makedirs(args.save_dir, exist_ok=True)
            torch.save({"image_syn": image_syn.cpu(), "text_syn": text_syn.cpu(), "syn_lr": lr.cpu(), "it": it},
                       os.path.join(args.save_dir, "distilled_%s.pt" % args.dataset))
    if nan_at is None:
        stop, pit, _ = stopper.last()
        nan_at = pit if stop else None
    if nan_at is not None:
        print("img_param_loss is NaN at iteration %d: stopping (reference distill.py:599); the synthetic set "
              "is the one before that iteration's update" % nan_at)
    if coll is not None:
        coll.close()      # ncclCommDestroy before torch's own process group goes away
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
