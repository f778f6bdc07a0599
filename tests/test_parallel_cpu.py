"""world_size-2 gloo test of the N>1 path's host logic: fused gradient buffer, all-reduce mean,
expert assignment and rank-identical minibatch permutations / SGD state."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.distill_ref import SGDMomentum


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_dataset_distillation_amd import parallel as par
    torch.manual_seed(0)
    image_syn, text_syn = torch.randn(6, 3, 8, 8), torch.randn(6, 5)
    lr = torch.tensor([0.1, 0.1])
    flat, views = par.fused_grad_buffer(image_syn, text_syn)
    opt = SGDMomentum(0.01)
    state = torch.cat([image_syn.flatten(), text_syn.flatten(), lr])
    experts = []
    for it in range(3):
        e = par.expert_for_rank(it, rank, world, 5)
        experts.append(e)
        perms = par.shared_permutations(6, 4, 2, it)
        # stand-in for engine.unrolled_match: a rank/expert-specific gradient written through the views
        g = torch.Generator().manual_seed(1000 + e)
        views["image_syn"].copy_(torch.randn(image_syn.shape, generator=g))
        views["text_syn"].copy_(torch.randn(text_syn.shape, generator=g))
        views["lr"].copy_(torch.randn(2, generator=g))
        stop = par.reduce_gradients_and_stop_flag_(flat, views, torch.tensor([1.0, 0.5, 0.5]))
        assert not stop
        state = opt.step(state, views["grads"].clone())
    out[rank] = dict(state=state, experts=experts, perms=perms, flat=views["grads"].clone())
    dist.destroy_process_group()


def test_two_rank_gradient_averaging_and_identical_updates():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["state"], b["state"])          # identical synthetic set on every rank
    assert torch.equal(a["perms"], b["perms"])
    assert a["experts"] == [0, 2, 4] and b["experts"] == [1, 3, 0]
    # last reduced buffer is the mean of the two ranks' expert gradients
    def grad(e):
        g = torch.Generator().manual_seed(1000 + e)
        return torch.cat([torch.randn(6, 3, 8, 8, generator=g).flatten(),
                          torch.randn(6, 5, generator=g).flatten(), torch.randn(2, generator=g)])
    assert torch.allclose(a["flat"], 0.5 * (grad(4) + grad(0)), atol=1e-6)


def _worker_nan(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_dataset_distillation_amd import parallel as par
    image_syn, text_syn = torch.zeros(2, 3, 4, 4), torch.zeros(2, 5)
    flat, views = par.fused_grad_buffer(image_syn, text_syn)
    applied, stopped_at = 0, None
    for it in range(4):
        views["grads"].fill_(float(rank + 1))
        nan_here = rank == 1 and it == 2          # only rank 1's expert diverges, at iteration 2
        losses = torch.tensor([float("nan")] * 3 if nan_here else [1.0, 0.5, 0.5])
        if nan_here:
            views["grads"].fill_(float("nan"))
        if par.reduce_gradients_and_stop_flag_(flat, views, losses):
            stopped_at = it
            break
        assert torch.isfinite(views["grads"]).all()
        applied += 1
    dist.barrier()       # would hang / mismatch if one rank were still inside the loop's all-reduce
    out[rank] = (applied, stopped_at)
    dist.destroy_process_group()


def test_nan_break_is_collective():
    """a rank whose own expert yields NaN must not leave the loop alone -- every rank stops at
    the same iteration and none applies the NaN-poisoned averaged gradient."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_nan, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == (2, 2) and out[1] == (2, 2)


def _worker_deferred(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_dataset_distillation_amd import parallel as par
    image_syn, text_syn = torch.zeros(2, 3, 4, 4), torch.zeros(2, 5)
    flat, views = par.fused_grad_buffer(image_syn, text_syn)
    stopper = par.DeferredStop("cpu")
    state, applied, left_at, nan_seen = torch.zeros(1), [], None, None
    for it in range(6):
        views["grads"].fill_(1.0)
        nan_here = rank == 1 and it == 2
        losses = torch.tensor([float("nan")] * 3 if nan_here else [1.0 + it, 0.5, 0.5])
        if nan_here:
            views["grads"].fill_(float("nan"))
        prev = stopper.update(flat, views, losses, iteration=it)
        if stopper.sticky.item() == 0:          # what mdd_flat_sgd_momentum_guarded does on the device
            state += views["grads"][0]
            applied.append(it)
        if prev is not None:
            stop, pit, lh = stopper.read(prev)
            if stop:
                left_at, nan_seen = it, pit
                break
            assert lh[0] == 1.0 + pit or rank == 1      # rank 1's own losses may be the NaN ones later
    dist.barrier()
    out[rank] = (applied, left_at, nan_seen, float(state))
    dist.destroy_process_group()


def test_deferred_nan_stop_is_collective_and_freezes_the_state_at_the_nan_iteration():
    """distill.py's loop without a host sync per iteration: the reduced NaN flag turns every later optimiser
    step into a no-op on the device, the host notices one iteration later -- on every rank at the same
    iteration, with the state of iteration 1 (the last finite one) intact."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_deferred, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        applied, left_at, nan_seen, state = out[r]
        assert applied == [0, 1] and left_at == 3 and nan_seen == 2 and state == 2.0, out[r]


def _toy_shard(rank, world, base):
    """A stand-in for parallel.sharded_unrolled_match with the same exchange protocol: gather the ranks'
    feature rows, reduce a rank-specific gradient, twice, and return what it was sent."""
    x = base[rank * 2:(rank + 1) * 2] * (rank + 1)
    X = yield ("all_gather", x)
    g = yield ("all_reduce", X.sum(0) * (rank + 1))
    X2 = yield ("all_gather", X[rank:rank + 1] + g[:1])
    h = yield ("all_reduce", torch.full((3,), float(rank + 1)))
    return dict(X=X, g=g, X2=X2, h=h)


def _worker_b(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_dataset_distillation_amd import parallel as par
    base = torch.arange(12, dtype=torch.float32).view(4, 3)
    out[rank] = par.run_collectives(_toy_shard(rank, world, base))
    dist.destroy_process_group()


def test_mode_b_collective_driver_equals_the_lockstep_driver():
    """parallel.run_collectives (torch.distributed, 2 gloo ranks) and parallel.run_lockstep (all shards in
    one process; what the GPU parity test of mode B uses) must feed a shard generator the same tensors."""
    from multimodal_dataset_distillation_amd import parallel as par
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_b, args=(world, _free_port(), out), nprocs=world, join=True)
    base = torch.arange(12, dtype=torch.float32).view(4, 3)
    want = par.run_lockstep([_toy_shard(r, world, base) for r in range(world)])
    for r in range(world):
        for k in ("X", "g", "X2", "h"):
            assert torch.equal(out[r][k], want[r][k]), (r, k)
    assert torch.equal(out[0]["X"], out[1]["X"]) and torch.equal(out[0]["h"], torch.full((3,), 3.0))


def test_library_collective_has_no_cpu_path():
    """The library-issued all-reduce is RCCL only: asking for it on a CPU device fails loudly."""
    import pytest
    from multimodal_dataset_distillation_amd import parallel as par
    with pytest.raises(RuntimeError, match="needs a GPU"):
        par.LibraryCollective("cpu")
