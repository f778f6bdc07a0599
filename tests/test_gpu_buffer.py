"""GPU tests of the rows either side of the hot path: one first-order expert
training step (reference buffer.py:73 -> epoch.py:59-98 -> networks.py:845-889: fixed logit scale
1/0.07, SGD without momentum) against the CPU oracle, and the expert-buffer file format written by
the stage-1 driver and read back by the stage-2 driver."""
import os

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_expert_training_step_matches_oracle(report):
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr
    from test_gpu_engine import make_oracle
    n, size, d_txt, lr = 6, 64, 32, 0.1
    fi, ft = make_oracle("nfnet_tiny", d_txt, 11)
    img, txt = dr.synthetic_inputs(n, size, d_txt, seed=3)
    thi = fi.flat_param().clone().requires_grad_(True)
    tht = ft.flat_param().clone().requires_grad_(True)
    feat = ft.module.fc.out_features
    mask = ((torch.rand(n, feat, generator=torch.Generator().manual_seed(4)) >= 0.1).float() / 0.9)
    loss = dr.contrastive_loss(fi(img, flat_param=thi), ft(txt, flat_param=tht, drop_mask=mask), 1.0 / 0.07)
    gi, gt = torch.autograd.grad(loss, [thi, tht])
    ref_i, ref_t = (thi - lr * gi).detach(), (tht - lr * gt).detach()

    dev = "cuda"
    eng = UnrollEngine("nfnet_tiny", batch=n, num_queries=n, image_size=size, d_txt=d_txt, syn_steps=1,
                       dtype="f32")
    th_i, th_t = thi.detach().to(dev), tht.detach().to(dev)
    x = eng.img_forward(0, th_i, img.to(dev))
    y = eng.txt_forward(0, th_t, txt.to(dev), drop_mask=mask.to(dev))
    l, xb, yb, _ = eng.contrastive(x, y, 1.0 / 0.07)
    g_i = eng.img_backward(0, th_i, xb)
    g_t = eng.txt_backward(0, th_t, yb)
    new_i, new_t = th_i - lr * g_i, th_t - lr * g_t
    torch.cuda.synchronize()
    e = dict(loss=abs(l.item() - loss.item()) / abs(loss.item()),
             g_img=rel_err(g_i.cpu(), gi), g_txt=rel_err(g_t.cpu(), gt),
             th_img=rel_err(new_i.cpu(), ref_i), th_txt=rel_err(new_t.cpu(), ref_t))
    report("expert training step (scale 1/0.07, dropout mask) vs oracle: "
           + " ".join(f"{k} {float(v):.2e}" for k, v in e.items()))
    assert all(float(v) < 1e-3 for v in e.values()), e
    eng.close()


def test_buffer_cli_writes_reference_format_and_distill_reads_it(report, tmp_path):
    from multimodal_dataset_distillation_amd import buffer, distill, networks as nw
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    bdir = str(tmp_path / "buffers")
    args = buffer.build_parser().parse_args(
        ["--dataset", "flickr", "--num_experts", "2", "--train_epochs", "3", "--batch_train", "4",
         "--image_size", "64", "--image_encoder", "nfnet_tiny", "--synthetic_data", "1",
         "--compute_dtype", "f32", "--buffer_path", bdir])
    buffer.main(args)
    d = os.path.join(bdir, "flickr", "nfnet_tiny", "bert")
    files = sorted(os.listdir(d))
    assert files == ["img_replay_buffer_0.pt", "img_replay_buffer_1.pt", "txt_replay_buffer_0.pt",
                     "txt_replay_buffer_1.pt"]
    # the reference's nested-list layout (buffer.py:104-112), readable without unpickling code
    traj = torch.load(os.path.join(d, "img_replay_buffer_0.pt"), map_location="cpu", weights_only=True)
    assert len(traj) == 1 and len(traj[0]) == 4               # 1 expert per file, init + 3 epochs
    eng = UnrollEngine("nfnet_tiny", batch=4, num_queries=4, image_size=64, d_txt=768, syn_steps=1,
                       dtype="f32")
    assert [tuple(t.shape) for t in traj[0][0]] == [tuple(s) for _, s, _ in eng.param_table("img")]
    assert all(t.dtype == torch.float32 for t in traj[0][0])
    moved = sum(float((a - b).abs().sum()) for a, b in zip(traj[0][0], traj[0][3]))
    assert moved > 0                                           # the experts did train
    eng.close()
    # stage 2 consumes the directory exactly as the reference does (distill.py:255-283, 450-476)
    a, _ = distill.build_parser().parse_known_args(
        ["--image_encoder", "nfnet_tiny", "--num_queries", "4", "--mini_batch_size", "4", "--syn_steps", "2",
         "--expert_epochs", "1", "--max_start_epoch", "2", "--Iteration", "3", "--image_size", "64",
         "--lr_img", "0.5", "--lr_txt", "0.5", "--lr_lr", "1e-5", "--compute_dtype", "f32",
         "--buffer_path", d, "--max_files", "2"])
    img, txt, lr = distill.main(a)
    assert torch.isfinite(img).all() and torch.isfinite(txt).all() and torch.isfinite(lr).all()
    report(f"buffer.py -> {len(files)} files -> distill.py: |image_syn| {img.norm().item():.3f} lr {lr.tolist()}")
    nw.release_engines()


def test_stage1_sgd_momentum_weight_decay_and_decay_schedule(report):
    """reference buffer.py:59-60: torch.optim.SGD(lr, momentum=--mom, weight_decay=--l2) per network, and
    :97-101 (--decay): lr x0.1 with rebuilt optimisers.  Three steps on fixed flat vectors through
    optim.sgd_step against torch.optim.SGD itself (CPU), the second pair of steps after a 'decay'."""
    from multimodal_dataset_distillation_amd.optim import sgd_step
    torch.manual_seed(0)
    n, lr, mom, wd = 100003, 0.1, 0.9, 5e-4
    th = torch.randn(n)
    grads = [torch.randn(n) for _ in range(4)]
    p = th.clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=lr, momentum=mom, weight_decay=wd)
    dev = "cuda"
    t, buf, first, cur_lr = th.to(dev), torch.zeros(n, device=dev), True, lr
    for i, g in enumerate(grads):
        if i == 2:   # --decay: the reference rebuilds the optimiser with lr*0.1 (fresh momentum buffer)
            opt = torch.optim.SGD([p], lr=lr * 0.1, momentum=mom, weight_decay=wd)
            cur_lr, first = lr * 0.1, True
        p.grad = g.clone()
        opt.step()
        sgd_step(t, g.to(dev), buf, cur_lr, mom, wd, first)
        first = False
        e = rel_err(t.cpu(), p.detach())
        assert e < 1e-6, (i, e)
    # momentum 0 / wd 0 (the reference defaults) == plain theta -= lr * g
    t2, b2 = th.to(dev), torch.zeros(n, device=dev)
    sgd_step(t2, grads[0].to(dev), b2, lr, 0.0, 0.0, True)
    assert rel_err(t2.cpu(), th - lr * grads[0]) < 1e-6
    report("stage-1 SGD(momentum, weight_decay) + decay rebuild == torch.optim.SGD")


def test_buffer_cli_honours_mom_l2_decay(report, tmp_path):
    from multimodal_dataset_distillation_amd import buffer
    outs = {}
    for tag, extra in (("plain", []), ("mom", ["--mom", "0.9", "--l2", "0.01", "--decay"])):
        bdir = str(tmp_path / tag)
        args = buffer.build_parser().parse_args(
            ["--dataset", "flickr", "--num_experts", "1", "--train_epochs", "4", "--batch_train", "4",
             "--image_size", "64", "--image_encoder", "nfnet_tiny", "--synthetic_data", "1",
             "--compute_dtype", "f32", "--buffer_path", bdir, "--seed", "3"] + extra)
        buffer.main(args)
        d = os.path.join(bdir, "flickr", "nfnet_tiny", "bert")
        traj = torch.load(os.path.join(d, "txt_replay_buffer_0.pt"), map_location="cpu", weights_only=True)
        outs[tag] = torch.cat([t.reshape(-1) for t in traj[0][-1]])
        assert torch.isfinite(outs[tag]).all()
    diff = rel_err(outs["mom"], outs["plain"])
    report(f"buffer.py --mom 0.9 --l2 0.01 --decay vs defaults: final expert differs by {diff:.2e}")
    assert diff > 1e-4        # the flags change the trajectory (round 1 dropped them silently)


def test_buffer_cli_trains_on_real_pairs_from_tensor_files(report, tmp_path, capsys):
    """SURVEY 8f rank 2 'next': expert training on real (image, caption-embedding) pairs -- training
    images from a tensor file, captions from the reference's train-caption cache (utils.py:885), the ragged
    last batch trained like the reference's DataLoader does, per-epoch retrieval metrics on a held-out set."""
    import numpy as np
    from multimodal_dataset_distillation_amd import buffer
    rs = np.random.RandomState(1)
    m = 10
    torch.save(torch.from_numpy(rs.randn(m, 3, 64, 64).astype(np.float32)), os.path.join(tmp_path, "train.pt"))
    np.savez(os.path.join(tmp_path, "flickr_bert_train_text_embed.npz"),
             bert_test_embed=rs.randn(m, 768).astype(np.float32))
    np.savez(os.path.join(tmp_path, "eval.npz"), images=rs.randn(4, 3, 64, 64).astype(np.float32),
             txt2img=np.array([0, 1, 2, 3, 0, 1]), bert_test_embed=rs.randn(6, 768).astype(np.float32))
    bdir = str(tmp_path / "buffers")
    args = buffer.build_parser().parse_args(
        ["--dataset", "flickr", "--num_experts", "1", "--train_epochs", "2", "--batch_train", "4",
         "--image_size", "64", "--image_encoder", "nfnet_tiny", "--compute_dtype", "f32", "--buffer_path", bdir,
         "--train_images", os.path.join(tmp_path, "train.pt"), "--embed_dir", str(tmp_path),
         "--eval_data", os.path.join(tmp_path, "eval.npz"), "--mom", "0.5", "--lr_teacher_img", "0.01",
         "--lr_teacher_txt", "0.01"])
    buffer.main(args)
    out = capsys.readouterr().out
    assert out.count("Train Loss:") == 2 and "Img R@1:" in out
    d = os.path.join(bdir, "flickr", "nfnet_tiny", "bert")
    traj = torch.load(os.path.join(d, "img_replay_buffer_0.pt"), map_location="cpu", weights_only=True)
    assert len(traj[0]) == 3 and all(torch.isfinite(t).all() for t in traj[0][-1])
    assert sum(float((a - b).abs().sum()) for a, b in zip(traj[0][0], traj[0][-1])) > 0
    report("buffer.py on real pairs (tensor file + caption cache, ragged tail batch, eval per epoch): ok")
