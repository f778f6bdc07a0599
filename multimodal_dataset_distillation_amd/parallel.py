"""Multi-GPU plumbing of the hot path (SURVEY 8e mode A, expert-replica sharding): one process per
GPU, `torch.distributed` ("nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).  Every rank holds
the full synthetic set and matches ITS OWN expert trajectory; the only exchange per outer iteration
is one all-reduce (mean) of the fused buffer [d image_syn | d text_syn | d lr_img, d lr_txt]
(60.5 MB fp32 at 100 pairs), after which every rank applies the identical SGD step.  The reference
has no counterpart (its --distributed is single-process nn.DataParallel, distill.py:443-445)."""
import torch
import torch.distributed as dist


def fused_grad_buffer(image_syn, text_syn):
    """One flat fp32 buffer [d image_syn | d text_syn | d lr (2) | NaN flag] with views for the pieces:
    a single collective per iteration carries the gradients AND the stop decision."""
    n_img, n_txt = image_syn.numel(), text_syn.numel()
    flat = torch.zeros(n_img + n_txt + 3, dtype=torch.float32, device=image_syn.device)
    views = dict(image_syn=flat[:n_img].view_as(image_syn),
                 text_syn=flat[n_img:n_img + n_txt].view_as(text_syn),
                 lr=flat[n_img + n_txt:n_img + n_txt + 2], nan_flag=flat[n_img + n_txt + 2:],
                 grads=flat[:n_img + n_txt + 2])
    return flat, views


def reduce_gradients_and_stop_flag_(flat, views, losses3, reduce=True, group=None):
    """Mode A's one exchange per iteration, with the reference's NaN break (distill.py:599-600) made
    COLLECTIVE: every rank writes [its loss is NaN] into the flag slot of the fused buffer BEFORE the
    all-reduce and tests the REDUCED flag after it.  All ranks therefore stop at the same iteration (a
    rank leaving alone would pair its final barrier with the others' next all-reduce: mismatched
    collectives), and no rank applies a gradient that another rank's NaN poisoned.
    Returns True when the loop must stop.  One host sync (the flag read)."""
    views["nan_flag"].copy_(torch.isnan(losses3).any().to(torch.float32).reshape(1))
    if reduce:
        average_gradients_(flat, group)
    return bool(views["nan_flag"].item() > 0)


def average_gradients_(flat, group=None):
    """In-place mean over ranks; no-op for a single process.  `group` is a torch.distributed process group
    (or None = the default one) or a `LibraryCollective` (the library's own RCCL communicator)."""
    if isinstance(group, LibraryCollective):
        return group.all_reduce_mean_(flat)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


class LibraryCollective:
    """Mode A's all-reduce issued by libmdd_hip.so itself (`mdd_allreduce_syn_grads`, include/mdd_hip.h) on an
    RCCL communicator the library owns, in the caller's stream: no hop through torch's NCCL stream.
    torch.distributed (any backend) is used ONCE, to carry rank 0's 128-byte RCCL unique id to the other ranks;
    with no process group initialised the communicator has the single rank 0.  Same result as
    `all_reduce(SUM)` + `div_(world)`.  There is no fallback: without RCCL the constructor raises."""

    def __init__(self, device, group=None):
        import ctypes as C
        from . import _lib
        self._lib_mod, self._lib = _lib, _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("LibraryCollective needs a GPU: RCCL has no CPU path")
        multi = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if multi else 0
        self.world = dist.get_world_size(group) if multi else 1
        uid = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0:
            _lib.check(self._lib.mdd_comm_unique_id(C.c_void_p(uid.data_ptr())))
        if self.world > 1:
            carrier = uid.to(self.device) if dist.get_backend(group) == "nccl" else uid
            dist.broadcast(carrier, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = carrier.cpu()
        h = C.c_void_p()
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.mdd_comm_create(C.c_void_p(uid.data_ptr()), self.rank, self.world, index, C.byref(h)))
        self.h = h

    def all_reduce_mean_(self, flat):
        import ctypes as C
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self.device:
            raise RuntimeError("LibraryCollective: a contiguous fp32 tensor on %s is required" % self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._lib_mod.check(self._lib.mdd_allreduce_syn_grads(self.h, C.c_void_p(flat.data_ptr()), flat.numel(), 1, st))
        return flat

    def close(self):
        if getattr(self, "h", None):
            self._lib.mdd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeferredStop:
    """The NaN break of reference distill.py:599-600 without a host synchronisation per iteration.

    Every iteration: `update(flat, views, losses3, reduce)` writes this rank's NaN flag into the fused buffer,
    all-reduces it with the gradients (mode A) and folds the REDUCED flag into a sticky device scalar
    (`self.sticky`); the optimiser steps are enqueued GUARDED by that scalar (`mdd_flat_sgd_momentum_guarded`:
    a no-op once it is non-zero), so the synthetic set stops changing at exactly the iteration the reference
    would have left its loop at, on every rank.  The host learns about it one iteration later:
    `poll()` returns (stop, iteration, losses) for the PREVIOUS update, whose device->host copy was queued
    behind that iteration's work and has long finished when the next iteration has been enqueued.
    """

    def __init__(self, device):
        self.device = torch.device(device)
        self.sticky = torch.zeros(1, device=self.device)
        pin = self.device.type == "cuda"
        self._host = [torch.zeros(4, pin_memory=pin) for _ in range(2)]
        self._event = [torch.cuda.Event() if pin else None for _ in range(2)]
        self._pending = None
        self._n = 0

    def update(self, flat, views, losses3, reduce=True, group=None, iteration=None):
        views["nan_flag"].copy_(torch.isnan(losses3).any().to(torch.float32).reshape(1))
        if reduce:
            average_gradients_(flat, group)
        torch.maximum(self.sticky, (views["nan_flag"] > 0).to(torch.float32), out=self.sticky)
        slot = self._n % 2
        self._host[slot].copy_(torch.cat([self.sticky, losses3.reshape(3).to(torch.float32)]), non_blocking=True)
        if self._event[slot] is not None:
            self._event[slot].record()
        prev, self._pending = self._pending, (slot, self._n if iteration is None else iteration)
        self._n += 1
        return prev

    def read(self, pending):
        """(stop, iteration, [grand, img, txt]) of a pending record (waits for ITS copy only)."""
        slot, it = pending
        if self._event[slot] is not None:
            self._event[slot].synchronize()
        h = self._host[slot].tolist()
        return h[0] > 0, it, h[1:]

    def last(self):
        return self.read(self._pending) if self._pending is not None else (False, -1, [0.0, 0.0, 0.0])


def expert_for_rank(iteration, rank, world, num_experts):
    """Rank r takes expert (iteration*world + r) mod E: the ranks of one iteration cover `world`
    consecutive experts of the reference's rotation (distill.py:450-465)."""
    return (iteration * world + rank) % num_experts


def shared_permutations(num_queries, batch, syn_steps, iteration, seed=0):
    """Minibatch indices (distill.py:510-511) drawn identically on every rank."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + iteration)
    return torch.stack([torch.randperm(num_queries, generator=g)[:batch] for _ in range(syn_steps)])


# ----------------------------------------------------------------------------------------------
# Mode B: synthetic-sample sharding = what the reference's `--distributed` does with
# nn.DataParallel (distill.py:443-445, 515-517): every rank runs the SAME unrolled loop on its chunk of
# each step's minibatch with replicated theta, and the ranks exchange
#   per step, forward sweep : all-gather of the two [n_loc, D] feature blocks (the N x N logits need every
#                             row), all-reduce(sum) of the inner gradient (P_img + P_txt floats)
#   per step, reverse sweep : the same two exchanges for the tangent features and the Hessian-vector rows
#   per iteration           : all-reduce(sum) of d image_syn / d text_syn (each rank holds its rows' part)
# Results equal the single-GPU iteration up to summation order.  Costlier than mode A (2*K all-reduces of
# 159 MB per iteration at C2) -- it exists for parity with the reference's flag, not for speed.
#
# The iteration is a GENERATOR that yields ("all_gather" | "all_reduce", tensor) at every exchange and is
# sent the result back, so the same code runs under torch.distributed (`run_collectives`) and, for tests on
# one GPU, as several lock-stepped shards in one process (`run_lockstep`).
def _op_contrastive(lib, x, y, scale, xd=None, yd=None):
    import ctypes as C
    from . import _lib
    n, d = x.shape
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    work = torch.empty(lib.mdd_op_contrastive_workspace_floats(n, d), device=x.device)
    loss = torch.zeros(1, device=x.device)
    xb, yb, sb = torch.empty_like(x), torch.empty_like(y), torch.zeros(1, device=x.device)
    sdev, sconst = (scale, 0.0) if torch.is_tensor(scale) else (None, float(scale))
    _lib.check(lib.mdd_op_contrastive(n, d, P(x), P(y), P(xd), P(yd), P(sdev), sconst, P(work), P(loss),
                                      P(xb), P(yb), P(sb), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return loss, xb, yb, sb


def sharded_unrolled_match(eng, rank, world, image_syn, text_syn, lr, theta0_img, theta0_txt, target_img,
                           target_txt, perms, drop_masks=None, logit_scale=None):
    """One outer iteration (reference distill.py:509-606) on shard `rank` of `world`.
    eng: UnrollEngine(batch = minibatch/world, num_queries = len(image_syn)); image_syn/text_syn/theta/lr
    replicated; perms [K, minibatch] and drop_masks [K, minibatch, D] are the FULL minibatch's, identical
    on every rank.  Returns dict(grand_loss, img_loss, txt_loss, contrastive [K], image_syn, text_syn, lr)
    with the synthetic-set gradients already summed over ranks."""
    from . import _lib
    lib = _lib.load()
    K, B = perms.shape
    nl = eng.batch
    assert B == nl * world and eng.syn_steps >= K
    lo, hi = rank * nl, (rank + 1) * nl
    lr_img, lr_txt = lr[0:1], lr[1:2]
    scale = lr_img if logit_scale is None else float(logit_scale)
    thI, thT, gI, gT, feats, ces = [theta0_img], [theta0_txt], [], [], [], []
    # ---- unrolled student training
    for k in range(K):
        idx = perms[k, lo:hi].contiguous()
        mask = None if drop_masks is None else drop_masks[k, lo:hi].contiguous()
        x = eng.img_forward(k, thI[k], image_syn, idx=idx)
        y = eng.txt_forward(k, thT[k], text_syn, idx=idx, drop_mask=mask)
        X = yield ("all_gather", x)
        Y = yield ("all_gather", y)
        loss, xb, yb, _ = _op_contrastive(lib, X, Y, scale)
        ces.append(loss)
        feats.append((X, Y))
        g_i = yield ("all_reduce", eng.img_backward(k, thI[k], xb[lo:hi].contiguous()))
        g_t = yield ("all_reduce", eng.txt_backward(k, thT[k], yb[lo:hi].contiguous()))
        gI.append(g_i), gT.append(g_t)
        thI.append(thI[k] - lr_img * g_i)          # distill.py:582-583
        thT.append(thT[k] - lr_txt * g_t)
    # ---- trajectory-matching loss (distill.py:584-598), fp64 accumulation like the fused path
    d0i = (theta0_img.double() - target_img.double()).pow(2).sum()
    d0t = (theta0_txt.double() - target_txt.double()).pow(2).sum()
    img_loss = (thI[K].double() - target_img.double()).pow(2).sum() / d0i
    txt_loss = (thT[K].double() - target_txt.double()).pow(2).sum() / d0t
    # ---- outer backward as an explicit reverse sweep
    lamI = (2.0 * (thI[K].double() - target_img.double()) / d0i).float()
    lamT = (2.0 * (thT[K].double() - target_txt.double()) / d0t).float()
    g_image, g_text = torch.zeros_like(image_syn), torch.zeros_like(text_syn)
    dlr = torch.zeros(2, dtype=torch.float64, device=image_syn.device)
    for k in range(K - 1, -1, -1):
        idx = perms[k, lo:hi].contiguous()
        dlr[0] -= torch.dot(gI[k].double(), lamI.double())
        dlr[1] -= torch.dot(gT[k].double(), lamT.double())
        nuI, nuT = (lr_img * lamI).contiguous(), (lr_txt * lamT).contiguous()
        xd = eng.img_tangent_forward(k, thI[k], nuI)
        yd = eng.txt_tangent_forward(k, thT[k], nuT)
        Xd = yield ("all_gather", xd)
        Yd = yield ("all_gather", yd)
        X, Y = feats[k]
        _, xbd, ybd, sbd = _op_contrastive(lib, X, Y, scale, Xd, Yd)
        hI = yield ("all_reduce", eng.img_tangent_backward(k, thI[k], nuI, xbd[lo:hi].contiguous(),
                                                           dimage=g_image, idx=idx, mul=-1.0))
        hT = yield ("all_reduce", eng.txt_tangent_backward(k, thT[k], nuT, ybd[lo:hi].contiguous(),
                                                           dtext=g_text, idx=idx, mul=-1.0))
        if logit_scale is None:
            dlr[0] -= sbd[0].double()              # the fork's logit scale IS syn_lr_img (distill.py:548)
        lamI, lamT = lamI - hI, lamT - hT
    g_image = yield ("all_reduce", g_image)
    g_text = yield ("all_reduce", g_text)
    return dict(grand_loss=(img_loss + txt_loss).float(), img_loss=img_loss.float(), txt_loss=txt_loss.float(),
                contrastive=torch.cat(ces), image_syn=g_image, text_syn=g_text, lr=dlr.float())


def run_collectives(gen, group=None):
    """Drive one shard's generator with torch.distributed collectives (nccl = RCCL on the GPUs)."""
    world = dist.get_world_size(group)
    res = None
    try:
        while True:
            op, t = gen.send(res)
            if op == "all_gather":
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t.contiguous(), group=group)
                res = torch.cat(parts, 0)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                res = t
    except StopIteration as e:
        return e.value


def run_lockstep(gens):
    """Drive the generators of ALL shards in one process (tests): every shard reaches the same exchange,
    the collective is evaluated on their tensors, each is sent the result."""
    res = [None] * len(gens)
    outs = [None] * len(gens)
    live = True
    while live:
        reqs = []
        for i, g in enumerate(gens):
            try:
                reqs.append(g.send(res[i]))
            except StopIteration as e:
                outs[i] = e.value
                live = False
        if not live:
            assert all(o is not None for o in outs), "shards left the iteration at different exchanges"
            return outs
        ops = {r[0] for r in reqs}
        assert len(ops) == 1, "shards disagree on the collective"
        if ops.pop() == "all_gather":
            full = torch.cat([r[1] for r in reqs], 0)
            res = [full] * len(gens)
        else:
            tot = torch.stack([r[1] for r in reqs], 0).sum(0)
            res = [tot.clone() for _ in gens]
