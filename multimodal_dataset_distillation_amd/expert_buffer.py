"""Expert-trajectory buffers: the hot path's on-disk input.

Reference format (buffer.py:67-68, 94-95, 104-112; read at distill.py:255-283, 450-476):
  {buffer_path}/img_replay_buffer_{n}.pt, txt_replay_buffer_{n}.pt  =
  torch.save( list[expert] of list[epoch snapshot] of list[Tensor cpu fp32 in parameters() order] )
The reference re-`cat`s ~225 small tensors and H2D-copies them 6x per iteration (distill.py:469-476).

Here a file is flattened ONCE into a contiguous fp32 block [experts, snapshots, P] that lives in
HBM (E=20, T=10 -> 35 GB of the 288 GB); picking (start, target) for an iteration is two pointer
offsets -- zero copies, fully coalesced reads by the engine's streaming kernels.
`torch.load(weights_only=True)` only: nothing in a buffer file is ever unpickled as code.
"""
import os

import numpy as np
import torch


def list_expert_files(expert_dir):
    """reference distill.py:255-261."""
    img, txt, n = [], [], 0
    while os.path.exists(os.path.join(expert_dir, "img_replay_buffer_%d.pt" % n)):
        img.append(os.path.join(expert_dir, "img_replay_buffer_%d.pt" % n))
        txt.append(os.path.join(expert_dir, "txt_replay_buffer_%d.pt" % n))
        n += 1
    return img, txt


def shuffle_files(img_files, txt_files, rng=np.random):
    """reference distill.py:78-87."""
    assert len(img_files) == len(txt_files) and len(img_files) != 0
    order = rng.permutation(len(img_files))
    return [img_files[i] for i in order], [txt_files[i] for i in order]


def flatten_trajectories(trajs, expect_numel=None, expect_shapes=None):
    """list[expert][epoch][tensor] -> fp32 tensor [E, T, P] (reference distill.py:469-476 does the
    same `cat(p.reshape(-1))` per use)."""
    E, T = len(trajs), len(trajs[0])
    P = sum(int(p.numel()) for p in trajs[0][0])
    if expect_numel is not None and P != expect_numel:
        raise ValueError("expert buffer has %d parameters per snapshot, the student has %d" % (P, expect_numel))
    if expect_shapes is not None:
        got = [tuple(p.shape) for p in trajs[0][0]]
        if got != [tuple(s) for s in expect_shapes]:
            raise ValueError("expert buffer parameter shapes/order differ from the student's flatten order")
    out = torch.empty(E, T, P, dtype=torch.float32)
    for e, traj in enumerate(trajs):
        if len(traj) != T:
            raise ValueError("experts with different numbers of snapshots in one file")
        for t, snap in enumerate(traj):
            torch.cat([p.detach().reshape(-1).float() for p in snap], 0, out=out[e, t])
    return out


def load_expert_file(path, **kw):
    trajs = torch.load(path, map_location="cpu", weights_only=True)
    return flatten_trajectories(trajs, **kw)


def save_expert_file(path, flat, shapes):
    """Write [E, T, P] back in the reference's nested-list format (what buffer.py:104-112 saves)."""
    numels = [int(np.prod(s)) for s in shapes]
    trajs = [[[piece.reshape(s).clone() for piece, s in zip(flat[e, t].split(numels), shapes)]
              for t in range(flat.shape[1])] for e in range(flat.shape[0])]
    torch.save(trajs, path)


class ExpertBuffer:
    """HBM-resident [E, T, P] pair (image encoder / text projection) with the reference's
    expert rotation (distill.py:450-465) and start-epoch sampling (:466-470)."""

    def __init__(self, img_flat, txt_flat, device="cuda"):
        assert img_flat.shape[:2] == txt_flat.shape[:2]
        self.img = img_flat.to(device, non_blocking=True).contiguous()
        self.txt = txt_flat.to(device, non_blocking=True).contiguous()
        self.num_experts, self.num_snapshots = self.img.shape[:2]

    @classmethod
    def from_files(cls, img_path, txt_path, P_img, P_txt, device="cuda"):
        return cls(load_expert_file(img_path, expect_numel=P_img),
                   load_expert_file(txt_path, expect_numel=P_txt), device)

    def pick(self, expert_idx, start_epoch, expert_epochs):
        """(theta0_img, theta0_txt, target_img, target_txt): views, no copy."""
        t1 = start_epoch + expert_epochs
        if t1 >= self.num_snapshots:
            raise IndexError("start_epoch + expert_epochs = %d exceeds the trajectory (%d snapshots)"
                             % (t1, self.num_snapshots))
        return (self.img[expert_idx, start_epoch], self.txt[expert_idx, start_epoch],
                self.img[expert_idx, t1], self.txt[expert_idx, t1])


def synthetic_buffer(engine, num_experts, num_snapshots, seed=0, step_scale=1e-3, device="cuda"):
    """Synthetic expert trajectories of the engine's student (random-walk snapshots around a
    trained-like init): no dataset / pretrained weights exist offline."""
    from .networks import synthetic_expert_params
    img = torch.empty(num_experts, num_snapshots, engine.P_img, device=device)
    txt = torch.empty(num_experts, num_snapshots, engine.P_txt, device=device)
    g = torch.Generator(device=device).manual_seed(seed + 17)
    for e in range(num_experts):
        ti, tt = synthetic_expert_params(engine, seed * 1000 + e, device=device)
        img[e, 0], txt[e, 0] = ti, tt
        for t in range(1, num_snapshots):
            img[e, t] = img[e, t - 1] + step_scale * torch.randn(engine.P_img, device=device, generator=g)
            txt[e, t] = txt[e, t - 1] + step_scale * torch.randn(engine.P_txt, device=device, generator=g)
    buf = ExpertBuffer.__new__(ExpertBuffer)
    buf.img, buf.txt = img, txt
    buf.num_experts, buf.num_snapshots = num_experts, num_snapshots
    return buf
