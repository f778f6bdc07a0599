"""Synthetic-set evaluation: train a fresh student on (image_syn, text_syn), then
image<->text retrieval recall@K on a held-out set -- reference epoch.py (`epoch` :59-98,
`epoch_test_metrics` :103-216, `itm_eval` :219-244, `evaluate_synset` :348-397).

Every FLOP runs in libmdd_hip.so: the training step is the engine's first-order forward + inner gradient
(the same kernels as the hot path and as buffer.py), the metrics are `mdd_retrieval_ranks` (similarity
GEMM + rank counts on the device; the reference argsorts one row at a time on the host).
Held-out data (test images, frozen-BERT caption embeddings, img2txt / txt2img maps) come from the
caller: datasets and BERT are outside the MI355X path.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

LOGIT_SCALE = float(np.exp(np.log(1 / 0.07)))   # reference epoch.py:106-107, networks.py:878


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def retrieval_ranks(img_feat, txt_feat, img2txt, txt2img, scale=LOGIT_SCALE):
    """0-based ranks (rank_i2t [n_img], rank_t2i [n_txt]) and the scaled similarity matrix.
    img_feat / txt_feat: un-normalised fp32 device tensors; img2txt: list of lists of caption ids;
    txt2img: sequence of image ids."""
    lib = _lib.load()
    dev = img_feat.device
    img_feat, txt_feat = img_feat.float().contiguous(), txt_feat.float().contiguous()
    b, d = img_feat.shape
    n = txt_feat.shape[0]
    if txt_feat.shape[1] != d or len(img2txt) != b or len(txt2img) != n:
        raise ValueError("retrieval_ranks: inconsistent shapes")
    off = np.zeros(b + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(g) for g in img2txt])
    if (np.diff(off) == 0).any():
        raise ValueError("every image needs at least one ground-truth caption")
    idx = np.concatenate([np.asarray(g, dtype=np.int32) for g in img2txt])
    t2i = np.asarray(txt2img, dtype=np.int32)
    if idx.min() < 0 or idx.max() >= n or t2i.min() < 0 or t2i.max() >= b:
        raise ValueError("ground-truth index out of range")
    d_off, d_idx, d_t2i = (torch.from_numpy(a).to(dev) for a in (off, idx, t2i))
    scores = torch.empty(b, n, device=dev)
    norm_ws = torch.empty(b + n, device=dev)
    r_i = torch.empty(b, dtype=torch.int32, device=dev)
    r_t = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.mdd_retrieval_ranks(_ptr(img_feat), _ptr(txt_feat), _ptr(d_off), _ptr(d_idx), _ptr(d_t2i),
                                       b, n, d, float(scale), _ptr(scores), _ptr(norm_ws), _ptr(r_i),
                                       _ptr(r_t), _stream()))
    return r_i, r_t, scores


def recalls(rank_i2t, rank_t2i):
    """reference itm_eval epoch.py:226-244 on precomputed ranks."""
    ri, rt = rank_i2t.cpu().numpy(), rank_t2i.cpu().numpy()
    tr = [100.0 * (ri < k).sum() / len(ri) for k in (1, 5, 10)]
    ir = [100.0 * (rt < k).sum() / len(rt) for k in (1, 5, 10)]
    return {"txt_r1": tr[0], "txt_r5": tr[1], "txt_r10": tr[2], "txt_r_mean": sum(tr) / 3,
            "img_r1": ir[0], "img_r5": ir[1], "img_r10": ir[2], "img_r_mean": sum(ir) / 3,
            "r_mean": (sum(tr) + sum(ir)) / 6}


def _embed_images(eng, theta_img, images):
    """image_encoder over a test set, engine.batch images at a time (last batch padded)."""
    n, B = images.shape[0], eng.batch
    feats = []
    for s in range(0, n, B):
        chunk = images[s:s + B]
        if chunk.shape[0] < B:
            chunk = torch.cat([chunk, chunk[:1].expand(B - chunk.shape[0], -1, -1, -1)], 0)
        feats.append(eng.img_forward(0, theta_img, chunk.contiguous()).clone())
    return torch.cat(feats, 0)[:n]


def _embed_texts(eng, theta_txt, text_embeds):
    n, B = text_embeds.shape[0], eng.batch
    feats = []
    for s in range(0, n, B):
        chunk = text_embeds[s:s + B]
        if chunk.shape[0] < B:
            chunk = torch.cat([chunk, chunk[:1].expand(B - chunk.shape[0], -1)], 0)
        feats.append(eng.txt_forward(0, theta_txt, chunk.contiguous(), drop_mask=None).clone())   # eval: no dropout
    return torch.cat(feats, 0)[:n]


@torch.no_grad()
def epoch_test_metrics(eng, theta_img, theta_txt, test_images, text_embeds, img2txt, txt2img):
    """reference epoch.py:103-216 (model.eval(): the text projection's dropout is off)."""
    img_feat = _embed_images(eng, theta_img, test_images)
    txt_feat = _embed_texts(eng, theta_txt, text_embeds)
    r_i, r_t, _ = retrieval_ranks(img_feat, txt_feat, img2txt, txt2img)
    return recalls(r_i, r_t)


def _train_accuracy(x, y):
    """acc of reference CLIPModel_full.forward (networks.py:883-886): argmax hits in both directions / 2.
    Logging only (a [B,B] product on B <= a few hundred rows); device tensor, no sync."""
    xn, yn = x / x.norm(dim=1, keepdim=True), y / y.norm(dim=1, keepdim=True)
    logits = xn @ yn.t()
    gt = torch.arange(logits.shape[0], device=logits.device)
    return ((logits.argmax(1) == gt).sum() + (logits.argmax(0) == gt).sum()).float() / 2


def train_epoch(eng, theta_img, theta_txt, mom_img, mom_txt, images, texts, lr, batch, first,
                momentum=0.9, weight_decay=5e-4, generator=None, lr_txt=None, eng_tail=None,
                with_accuracy=False):
    """One epoch of reference `epoch` (epoch.py:59-98): shuffled mini-batches of (image, caption embedding)
    pairs, forward + fixed-scale (1/0.07) contrastive loss + backward + one SGD step per network
    (`optim.sgd_step` = torch.optim.SGD(lr, momentum, weight_decay)); in place on theta_*/mom_*.
    Defaults are evaluate_synset's optimisers (epoch.py:361-362); buffer.py passes --mom/--l2 and the two
    teacher learning rates.  The engine's batch is fixed: a ragged last batch runs on `eng_tail` (an
    engine of that batch size) when one is given and is dropped otherwise.  One host sync per epoch.
    Returns (mean loss, first-step flag) or (mean loss, mean accuracy, first-step flag)."""
    from .optim import sgd_step
    n = images.shape[0]
    perm = torch.randperm(n, generator=generator).to(images.device)
    lr_t = lr if lr_txt is None else lr_txt
    loss_sum = torch.zeros(1, device=images.device)
    acc_sum = torch.zeros(1, device=images.device)
    seen = 0
    for s in range(0, n, batch):
        idx = perm[s:s + batch]
        e = eng
        if idx.numel() < batch:
            if eng_tail is None or eng_tail.batch != idx.numel():
                break                                   # ragged tail without a matching engine: dropped
            e = eng_tail
        b = idx.numel()
        x = e.img_forward(0, theta_img, images, idx=idx)
        mask = (torch.rand(b, e.feature_dim, device=images.device) >= 0.1).float() / 0.9
        y = e.txt_forward(0, theta_txt, texts, idx=idx, drop_mask=mask)
        loss, xb, yb, _ = e.contrastive(x, y, LOGIT_SCALE)
        gi = e.img_backward(0, theta_img, xb)
        gt = e.txt_backward(0, theta_txt, yb)
        sgd_step(theta_img, gi, mom_img, lr, momentum, weight_decay, first)
        sgd_step(theta_txt, gt, mom_txt, lr_t, momentum, weight_decay, first)
        first = False
        loss_sum += loss * b
        if with_accuracy:
            acc_sum += _train_accuracy(x, y)
        seen += b
    mean_loss = float(loss_sum.item()) / max(1, seen)
    if with_accuracy:
        return mean_loss, float(acc_sum.item()) / max(1, seen), first
    return mean_loss, first


def evaluate_synset(eng, image_syn, text_syn, test_images, text_embeds, img2txt, txt2img, lr_net=0.1,
                    epoch_eval_train=1, seed=0):
    """reference evaluate_synset (epoch.py:348-397): a FRESH student trained on the synthetic pairs for
    epoch_eval_train+1 epochs, then retrieval metrics.  eng.batch must divide into the synthetic set."""
    from .networks import synthetic_expert_params
    dev = image_syn.device
    theta_img, theta_txt = synthetic_expert_params(eng, seed, device=dev)
    mom_img, mom_txt = torch.zeros_like(theta_img), torch.zeros_like(theta_txt)
    g = torch.Generator().manual_seed(seed)
    first, losses = True, []
    for ep in range(int(epoch_eval_train) + 1):
        l, first = train_epoch(eng, theta_img, theta_txt, mom_img, mom_txt, image_syn, text_syn, lr_net,
                               eng.batch, first, generator=g)
        if math.isnan(l):
            break
        losses.append(l)
    res = epoch_test_metrics(eng, theta_img, theta_txt, test_images, text_embeds, img2txt, txt2img)
    return (theta_img, theta_txt), losses, res
