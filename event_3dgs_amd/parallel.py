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
    op, divide = _mean_op(group)
    dist.all_reduce(flat_grad, op=op, group=group)
    if divide:
        flat_grad.div_(world)
    return flat_grad


def _mean_op(group):
    """NCCL/RCCL averages inside the collective (ncclAvg); gloo has no AVG, so the caller divides."""
    try:
        backend = dist.get_backend(group)
    except Exception:
        backend = "gloo"
    return (dist.ReduceOp.AVG, False) if backend == "nccl" else (dist.ReduceOp.SUM, True)


class PendingMean:
    """Handle of allreduce_mean_async_: wait() makes the CURRENT stream wait for the collective (the host does not
    block with RCCL) and finishes the mean where the backend cannot average."""

    def __init__(self, tensor, work, divide_by):
        self.tensor, self.work, self.divide_by = tensor, work, divide_by

    def wait(self):
        if self.work is not None:
            self.work.wait()
            if self.divide_by:
                self.tensor.div_(self.divide_by)
        return self.tensor


def allreduce_mean_async_(chunk, group=None):
    """Starts the in-place mean over ranks of one contiguous chunk of the gradient buffer and returns at once.
    Issuing the chunks of the buffer back to back and waiting for chunk k only before the optimizer step of chunk k
    overlaps the optimizer (HBM-bound) with the remaining collectives (xGMI-bound)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return PendingMean(chunk, None, 0)
    op, divide = _mean_op(group)
    work = dist.all_reduce(chunk, op=op, group=group, async_op=True)
    return PendingMean(chunk, work, dist.get_world_size(group) if divide else 0)


class PendingGather:
    def __init__(self, out, work):
        self.out, self.work = out, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.out


def allgather_async_(out, local, group=None):
    """Starts the all-gather of one contiguous `local` block per rank into the rows of `out` (world, numel)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        out[0].copy_(local)
        return PendingGather(out, None)
    try:
        work = dist.all_gather_into_tensor(out.view(-1), local, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):           # backends without the flat form (older gloo)
        work = dist.all_gather(list(out.unbind(0)), local, group=group, async_op=True)
    return PendingGather(out, work)


def rank_camera_indices(rank, world, n_cameras, iteration, seed=0, exclude=(5, 25, 45, 65, 85), mode="event"):
    """Deterministic per-rank camera draw mirroring train.py:116-131: index in [2, n-4] in event mode ([2, n-3]
    otherwise), the held-out evaluation views shifted down by one in event / gray mode.  Every rank can recompute
    every other rank's draw."""
    g = torch.Generator().manual_seed(seed * 1000003 + iteration * 131 + rank)
    hi = n_cameras - 4 if mode == "event" else n_cameras - 3          # inclusive, as random.randint
    idx = int(torch.randint(2, max(3, hi + 1), (1,), generator=g))
    if mode in ("event", "gray") and idx in exclude:
        idx -= 1
    return idx
