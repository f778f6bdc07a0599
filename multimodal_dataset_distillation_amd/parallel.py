"""Multi-GPU plumbing of the hot path (SURVEY 8e mode A, expert-replica sharding): one process per
GPU, `torch.distributed` ("nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).  Every rank holds
the full synthetic set and matches ITS OWN expert trajectory; the only exchange per outer iteration
is one all-reduce (mean) of the fused buffer [d image_syn | d text_syn | d lr_img, d lr_txt]
(60.5 MB fp32 at 100 pairs), after which every rank applies the identical SGD step.  The reference
has no counterpart (its --distributed is single-process nn.DataParallel, distill.py:443-445)."""
import torch
import torch.distributed as dist


def fused_grad_buffer(image_syn, text_syn):
    """One flat fp32 buffer with views for the three gradient pieces (a single collective)."""
    n_img, n_txt = image_syn.numel(), text_syn.numel()
    flat = torch.zeros(n_img + n_txt + 2, dtype=torch.float32, device=image_syn.device)
    views = dict(image_syn=flat[:n_img].view_as(image_syn),
                 text_syn=flat[n_img:n_img + n_txt].view_as(text_syn), lr=flat[n_img + n_txt:])
    return flat, views


def average_gradients_(flat, group=None):
    """In-place mean over ranks; no-op for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


def expert_for_rank(iteration, rank, world, num_experts):
    """Rank r takes expert (iteration*world + r) mod E: the ranks of one iteration cover `world`
    consecutive experts of the reference's rotation (distill.py:450-465)."""
    return (iteration * world + rank) % num_experts


def shared_permutations(num_queries, batch, syn_steps, iteration, seed=0):
    """Minibatch indices (distill.py:510-511) drawn identically on every rank."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + iteration)
    return torch.stack([torch.randperm(num_queries, generator=g)[:batch] for _ in range(syn_steps)])
