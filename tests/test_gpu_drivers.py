"""GPU tests of the stage-2 driver's rows around the loop: real-pair
initialisation + caption decode from the embedding caches (reference distill.py:97-105, 228, 244), the
evaluation hook every eval_it (distill.py:293-357) and mode A under a launcher with the ENGINE in the loop
(two ranks sharing the one GPU of the test box over gloo)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _common(tmp_path, extra):
    from multimodal_dataset_distillation_amd import distill
    argv = ["--dataset", "flickr", "--image_encoder", "nfnet_tiny", "--num_queries", "4", "--mini_batch_size", "4",
            "--syn_steps", "2", "--expert_epochs", "1", "--max_start_epoch", "2", "--Iteration", "2",
            "--image_size", "64", "--lr_img", "0.5", "--lr_txt", "0.5", "--lr_lr", "1e-5",
            "--synthetic_experts", "3", "4", "--compute_dtype", "f32", "--embed_dir", str(tmp_path)] + extra
    args, _ = distill.build_parser().parse_known_args(argv)
    return distill, args


def test_real_init_decode_and_eval_hook(report, tmp_path, capsys):
    from multimodal_dataset_distillation_amd import networks as nw
    rs = np.random.RandomState(0)
    m, d = 24, 768
    train_emb = rs.randn(m, d).astype(np.float32)
    np.savez(os.path.join(tmp_path, "flickr_bert_train_text_embed.npz"), bert_test_embed=train_emb)
    train_img = torch.from_numpy(rs.randn(m, 3, 64, 64).astype(np.float32))
    torch.save(train_img, os.path.join(tmp_path, "train_images.pt"))
    with open(os.path.join(tmp_path, "captions.txt"), "w") as f:
        f.write("\n".join("caption number %d" % i for i in range(m)) + "\n")
    # held-out set: 6 images x 2 captions
    t2i = np.repeat(np.arange(6), 2)
    np.savez(os.path.join(tmp_path, "eval.npz"), images=rs.randn(6, 3, 64, 64).astype(np.float32), txt2img=t2i,
             bert_test_embed=rs.randn(12, d).astype(np.float32))
    distill, args = _common(tmp_path, ["--pix_init", "real", "--txt_init", "real",
                                       "--train_images", os.path.join(tmp_path, "train_images.pt"),
                                       "--train_sentences", os.path.join(tmp_path, "captions.txt"),
                                       "--eval_data", os.path.join(tmp_path, "eval.npz"), "--eval_it", "2",
                                       "--num_eval", "2", "--epoch_eval_train", "1", "--batch_train", "4", "--std", "1"])
    np.random.seed(11)
    want = np.random.permutation(m)[:4]
    np.random.seed(11)
    torch.manual_seed(0)
    # initial state: the drawn pairs, decoded back to their own captions
    img0, txt0 = distill.init_synthetic_set(args, d, "cuda", torch.from_numpy(train_emb))
    assert torch.equal(img0.cpu(), train_img[want]) and torch.equal(txt0.cpu(), torch.from_numpy(train_emb[want]))
    dec = distill.nearest_neighbor(["caption number %d" % i for i in range(m)], txt0, train_emb)
    assert dec == ["caption number %d" % i for i in want]
    np.random.seed(11)
    img, txt, lr = distill.main(args)
    out = capsys.readouterr().out
    assert out.count("Evaluate_00:") == 2 and out.count("Evaluate_01:") == 2     # iterations 0 and 2, num_eval 2
    assert "original_sentence_list:" in out and "Mean/img_r1" in out
    assert torch.isfinite(img).all() and torch.isfinite(txt).all()
    report("distill.py real init + caption decode + eval hook: ok")
    nw.release_engines()


@pytest.mark.parametrize("mode", ["A", "B", "A-real"])
def test_two_ranks_on_one_gpu_with_the_engine_in_the_loop(mode, report, tmp_path):
    """torch.distributed.run, 2 ranks, gloo (RCCL refuses two ranks on one device).
    mode A: every rank runs the fused engine on its own synthetic expert, the fused [grads | NaN flag] buffer
    is all-reduced; mode B (`--distributed`, the reference flag's meaning): every rank runs the unrolled loop
    on its half of each minibatch, features all-gathered and gradients all-reduced inside the loop.
    Both ranks must end with the identical synthetic set (saved by rank 0 and rank 1 separately).
    A-real: mode A with the fork's real-pair initialisation (reference distill.py:97-105, 228: a numpy permutation
    of the training pairs) and NO common torch seed in the rank script -- the drawn set must still be the same on
    every rank (it was drawn with a per-rank numpy seed and never broadcast)."""
    extra = ["--distributed"] if mode == "B" else []
    if mode == "A-real":
        rs = np.random.RandomState(3)
        np.savez(os.path.join(tmp_path, "flickr_bert_train_text_embed.npz"),
                 bert_test_embed=rs.randn(40, 768).astype(np.float32))
        torch.save(torch.from_numpy(rs.randn(40, 3, 64, 64).astype(np.float32)), os.path.join(tmp_path, "train_images.pt"))
        extra = ["--dataset", "flickr", "--embed_dir", str(tmp_path), "--pix_init", "real", "--txt_init", "real",
                 "--train_images", os.path.join(tmp_path, "train_images.pt")]
    script = os.path.join(tmp_path, "run_rank.py")
    with open(script, "w") as f:
        f.write(
            "import os, sys, torch\n"
            "sys.path.insert(0, %r)\n"
            "from multimodal_dataset_distillation_amd import distill\n"
            "argv = ['--image_encoder','nfnet_tiny','--num_queries','4','--mini_batch_size','4','--syn_steps','2',\n"
            "        '--expert_epochs','1','--max_start_epoch','2','--Iteration','2','--image_size','64',\n"
            "        '--lr_img','0.5','--lr_txt','0.5','--lr_lr','1e-5','--synthetic_experts','3','4',\n"
            "        '--compute_dtype','f32','--dist_backend','gloo','--seed','5'] + %r\n"
            "args, _ = distill.build_parser().parse_known_args(argv)\n"
            "torch.manual_seed(0 if %r != 'A-real' else 1000 + int(os.environ['RANK']))\n"
            "img, txt, lr = distill.main(args)\n"
            "torch.save({'img': img.cpu(), 'txt': txt.cpu(), 'lr': lr.cpu()}, os.path.join(%r, 'rank%%s.pt' %% os.environ['RANK']))\n"
            % (ROOT, extra, mode, str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port",
                        {"A": "29617", "B": "29618", "A-real": "29619"}[mode], script],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=True)
    b = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=True)
    for k in a:
        assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k       # identical update on every rank
    report("mode %s, 2 ranks x engine on one GPU (gloo): identical synthetic set on both ranks, |img| %.4f"
           % (mode, a["img"].norm().item()))


@pytest.mark.gpu
def test_library_rccl_allreduce_single_rank_and_through_the_stop_logic(report):
    """`mdd_allreduce_syn_grads` (include/mdd_hip.h): the library binds RCCL at run time, creates its
    own communicator and all-reduces in the caller's stream.  This pool has one GPU per box and RCCL refuses two
    ranks on one device, so the collective itself is exercised at world size 1 (sum over one rank = identity,
    the mean divides by 1): what is checked is the binding (dlopen of the process's librccl, the symbols, the
    unique-id hand-over, communicator init on the right device, stream ordering, teardown) and that the
    mode-A stop logic runs unchanged on top of it."""
    import ctypes as C
    from multimodal_dataset_distillation_amd import _lib, parallel as par
    dev = torch.device("cuda", 0)
    coll = par.LibraryCollective(dev)
    assert coll.world == 1 and _lib.load().mdd_comm_world(coll.h) == 1
    g = torch.Generator(device=dev).manual_seed(5)
    image_syn, text_syn = torch.randn(4, 3, 16, 16, device=dev, generator=g), torch.randn(4, 8, device=dev, generator=g)
    flat, views = par.fused_grad_buffer(image_syn, text_syn)
    flat[:-1].copy_(torch.randn(flat.numel() - 1, device=dev, generator=g))
    want = flat.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                 # the collective runs in the CALLER's current stream
        flat.mul_(2.0)
        coll.all_reduce_mean_(flat)
        flat.mul_(0.5)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(flat, want)
    # the NaN stop rides through the same buffer
    stop = par.DeferredStop(dev)
    losses = torch.tensor([float("nan"), 1.0, 1.0], device=dev)
    stop.update(flat, views, losses, reduce=True, group=coll, iteration=0)
    assert stop.last()[0] is True and float(views["nan_flag"]) == 1.0
    # argument errors come back through mdd_last_error, not as a crash
    with pytest.raises(RuntimeError, match="contiguous fp32"):
        coll.all_reduce_mean_(flat.double())
    h = C.c_void_p()
    uid = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
    rc = _lib.load().mdd_comm_create(C.c_void_p(uid.data_ptr()), 3, 2, 0, C.byref(h))   # rank >= world
    assert rc != 0 and b"invalid argument" in _lib.load().mdd_last_error()
    coll.close()
    coll.close()                                   # idempotent
    report("library RCCL all-reduce: world 1 identity ok, stop logic ok")
