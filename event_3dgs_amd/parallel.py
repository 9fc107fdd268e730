"""View-parallel data parallelism: one process per GPU, every rank renders its own camera
triplet of the replicated model, and the gradient of the 59 floats/Gaussian (+ the threshold c) is
averaged with ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests).  The reference has
no distributed path (utils/general_utils.py:133 pins cuda:0); this is the new capability north_star asks for.
"""
import torch
import torch.distributed as dist


def allreduce_mean_(flat_grad, group=None):
    """In-place mean over ranks of one flat gradient buffer."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    world = dist.get_world_size(group)
    if world == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    flat_grad.div_(world)
    return flat_grad


def rank_camera_indices(rank, world, n_cameras, iteration, seed=0, exclude=(5, 25, 45, 65, 85)):
    """Deterministic per-rank camera draw mirroring train.py:116-131: index in [2, n-4], the held-out
    evaluation views shifted down by one.  Every rank can recompute every other rank's draw."""
    g = torch.Generator().manual_seed(seed * 1000003 + iteration * 131 + rank)
    idx = int(torch.randint(2, max(3, n_cameras - 3), (1,), generator=g))
    if idx in exclude:
        idx -= 1
    return idx
