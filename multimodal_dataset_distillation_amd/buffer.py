"""Stage-1 driver surface: expert trajectories for the distillation hot path (reference buffer.py).

Accepts the reference's command line (buffer.py:118-161).  What it produces is the hot path's
on-disk INPUT format, unchanged: `{buffer_path}/{dataset}/{image_encoder}/{text_encoder}/
img_replay_buffer_{n}.pt` / `txt_replay_buffer_{n}.pt` = torch.save(list[expert] of
list[epoch] of list[Tensor]) (buffer.py:67-68, 94-95, 104-112).

Expert training itself is first-order only and is NOT the hot path.  It reuses
the hot path's kernels: one training step = engine forward (F) + inner gradient (B) + the flat SGD
step with the reference's optimiser semantics -- torch.optim.SGD(lr, momentum=--mom,
weight_decay=--l2) per network (buffer.py:59-60), rebuilt with lr*0.1 after epoch
train_epochs//2+1 under --decay (buffer.py:97-101; the reference multiplies an undefined `lr`
there -- the intent, both teacher learning rates x0.1 and fresh momentum buffers, is what runs
here) -- and the reference's fixed logit scale 1/0.07 (networks.py:878).
Data sources: `--train_images FILE` (tensor file [M,3,S,S] of the training images in annotation order)
+ the train-caption embedding cache `{dataset}_{text_encoder}_train_text_embed.npz` (reference
utils.py:885; with precomputed embeddings the reference's `epoch` takes `caption` as a tensor,
networks.py:866-867): real (image, caption) pairs, one shuffled pass per epoch like the reference's
DataLoader, optional per-epoch retrieval metrics on `--eval_data` (buffer.py:74-75); or
`--synthetic_data STEPS` (random pairs) to exercise the pipeline without a dataset.  Decoding JPEGs,
augmentation and running BERT are outside the MI355X path.
"""
import argparse
import datetime
import os

import torch


def build_parser():
    p = argparse.ArgumentParser(description="Parameter Processing")
    p.add_argument("--dataset", type=str, default="flickr", choices=["flickr", "coco", "roco"])
    p.add_argument("--num_experts", type=int, default=100)
    p.add_argument("--lr_teacher_img", type=float, default=0.1)
    p.add_argument("--lr_teacher_txt", type=float, default=0.1)
    p.add_argument("--batch_train", type=int, default=128)
    p.add_argument("--dsa", type=str, default="True", choices=["True", "False"])
    p.add_argument("--dsa_strategy", type=str, default="color_crop_cutout_flip_scale_rotate")
    p.add_argument("--data_path", type=str, default="./data/Flickr30k/")
    p.add_argument("--buffer_path", type=str, default="./buffers")
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
    # additive
    p.add_argument("--synthetic_data", type=int, default=0, metavar="STEPS_PER_EPOCH",
                   help="train on random pairs, this many steps per epoch (no dataset offline)")
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "bf16x2", "f32"])
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--train_images", type=str, default=None,
                   help="tensor file (.pt/.npy/.npz key 'images') [M,3,S,S]: training images, annotation order")
    p.add_argument("--embed_dir", type=str, default=".",
                   help="directory of {dataset}_{text_encoder}_train_text_embed.npz (reference utils.py:885)")
    p.add_argument("--eval_data", type=str, default=None,
                   help=".npz held-out retrieval set (images, txt2img[, bert_test_embed]): metrics per epoch")
    return p


def main(args):
    from .engine import UnrollEngine
    from .expert_buffer import save_expert_file
    from .networks import VARIANTS, synthetic_expert_params

    if not torch.cuda.is_available():
        raise RuntimeError("buffer.py needs an MI355X; the engine has no CPU path")
    if not args.synthetic_data and not args.train_images:
        raise NotImplementedError(
            "no data source: pass --train_images FILE (+ the train-caption embedding cache in --embed_dir) "
            "for real pairs, or --synthetic_data STEPS to exercise the pipeline on random pairs")
    device = torch.device("cuda")
    variant = VARIANTS[args.image_encoder]
    d_txt = 768 if args.text_encoder == "bert" else 512
    n = args.batch_train
    eng = UnrollEngine(variant, batch=n, num_queries=n, image_size=args.image_size, d_txt=d_txt,
                       syn_steps=1, dtype=args.compute_dtype, device=device)
    save_dir = os.path.join(args.buffer_path, args.dataset, args.image_encoder, args.text_encoder)
    os.makedirs(save_dir, exist_ok=True)
    shapes_i = [s for _, s, _ in eng.param_table("img")]
    shapes_t = [s for _, s, _ in eng.param_table("txt")]
    g = torch.Generator(device=device).manual_seed(args.seed)
    from .optim import sgd_step
    real = eval_set = eng_tail = None
    if args.train_images:
        from . import epoch as ep
        from .embed_cache import load_embed_cache, load_eval_data, load_tensor_file
        imgs = load_tensor_file(args.train_images, key="images").float()
        emb = load_embed_cache(args, "train_text")
        if imgs.shape[0] != emb.shape[0] or tuple(imgs.shape[1:]) != (3, args.image_size, args.image_size) \
                or emb.shape[1] != d_txt:
            raise ValueError("train images %s / caption embeddings %s do not form [M,3,%d,%d] x [M,%d] pairs"
                             % (tuple(imgs.shape), tuple(emb.shape), args.image_size, args.image_size, d_txt))
        real = (imgs.to(device).contiguous(), emb.to(device).contiguous())
        tail = imgs.shape[0] % n
        if tail >= 2:      # the reference's DataLoader also trains on the ragged last batch
            eng_tail = UnrollEngine(variant, batch=tail, num_queries=tail, image_size=args.image_size, d_txt=d_txt,
                                    syn_steps=1, dtype=args.compute_dtype, device=device)
        if args.eval_data:
            ti, te, i2t, t2i = load_eval_data(args.eval_data, args)
            eval_set = (ti.to(device).contiguous(), te.to(device).contiguous(), i2t, t2i)
        cpu_gen = torch.Generator().manual_seed(args.seed)
    for it in range(args.num_experts):
        th_i, th_t = synthetic_expert_params(eng, args.seed * 100003 + it, device=device)
        snaps_i, snaps_t = [th_i.cpu()], [th_t.cpu()]          # buffer.py:67-68
        # teacher_optim_img / teacher_optim_txt (buffer.py:59-60)
        lr_i, lr_t = float(args.lr_teacher_img), float(args.lr_teacher_txt)
        mom_i, mom_t = torch.zeros_like(th_i), torch.zeros_like(th_t)
        first = True
        lr_schedule = [args.train_epochs // 2 + 1]             # buffer.py:70
        for e in range(args.train_epochs):
            if real is not None:     # reference buffer.py:73-92: one pass over the training pairs, then metrics
                train_loss, train_acc, first = ep.train_epoch(
                    eng, th_i, th_t, mom_i, mom_t, real[0], real[1], lr_i, n, first, momentum=args.mom,
                    weight_decay=args.l2, generator=cpu_gen, lr_txt=lr_t, eng_tail=eng_tail, with_accuracy=True)
                msg = "Itr: {}\tEpoch: {}\tTrain Loss: {:.4f}\tTrain Acc: {:.4f}".format(it, e, train_loss, train_acc)
                if eval_set is not None:
                    r = ep.epoch_test_metrics(eng, th_i, th_t, *eval_set)
                    msg += "\tImg R@1: {:.2f}\tR@5: {:.2f}\tR@10: {:.2f}\tTxt R@1: {:.2f}\tR@5: {:.2f}\tR@10: {:.2f}".format(
                        r["img_r1"], r["img_r5"], r["img_r10"], r["txt_r1"], r["txt_r5"], r["txt_r10"])
                print(msg)
            for _ in range(args.synthetic_data if real is None else 0):
                img = torch.randn(n, 3, args.image_size, args.image_size, device=device, generator=g)
                txt = torch.randn(n, d_txt, device=device, generator=g) * 0.5253
                x = eng.img_forward(0, th_i, img)
                # the text projection trains with Dropout(0.1) (networks.py:625-646, model.train())
                mask = (torch.rand(n, eng.feature_dim, device=device, generator=g) >= 0.1).float() / 0.9
                y = eng.txt_forward(0, th_t, txt, drop_mask=mask)
                _, xb, yb, _ = eng.contrastive(x, y, 1.0 / 0.07)   # networks.py:878
                gi = eng.img_backward(0, th_i, xb)
                gt = eng.txt_backward(0, th_t, yb)
                sgd_step(th_i, gi, mom_i, lr_i, args.mom, args.l2, first)
                sgd_step(th_t, gt, mom_t, lr_t, args.mom, args.l2, first)
                first = False
            snaps_i.append(th_i.cpu()), snaps_t.append(th_t.cpu())   # buffer.py:94-95
            if e in lr_schedule and args.decay:                      # buffer.py:97-101
                lr_i *= 0.1
                lr_t *= 0.1
                first = True                                         # rebuilt optimisers: fresh momentum buffers
        k = 0
        while os.path.exists(os.path.join(save_dir, "img_replay_buffer_%d.pt" % k)):
            k += 1
        print("Saving {}".format(os.path.join(save_dir, "img_replay_buffer_%d.pt" % k)))
        save_expert_file(os.path.join(save_dir, "img_replay_buffer_%d.pt" % k),
                         torch.stack(snaps_i)[None], shapes_i)
        save_expert_file(os.path.join(save_dir, "txt_replay_buffer_%d.pt" % k),
                         torch.stack(snaps_t)[None], shapes_t)
    eng.close()
    if eng_tail is not None:
        eng_tail.close()


if __name__ == "__main__":
    main(build_parser().parse_args())
