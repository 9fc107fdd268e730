"""View-parallel data parallelism: one process per GPU, every rank renders its own camera
triplet of the replicated model, and the gradient of the 59 floats/Gaussian (+ the threshold c) is
averaged with ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests).  The reference has
no distributed path (utils/general_utils.py:133 pins cuda:0); this is the new capability north_star asks for.
"""
import torch
import torch.distributed as dist

# Test hook: issue the collectives also in a ONE-rank process group (they are short-circuited there otherwise), so that
# the real RCCL entry points -- ncclAvg all-reduce, reduce_scatter_tensor, all_gather_into_tensor -- execute on a box with a
# single GPU (tests/test_hip_rccl.py; EventTrainer(force_distributed=True) sets it for its own calls).
# A count of the live trainers that asked for it (released when such a trainer is collected): the hook does not outlive them.
FORCE_SINGLE_RANK_COLLECTIVES = 0


def force_single_rank_collectives(owner):
    """Switch the hook on for as long as `owner` is alive."""
    import weakref
    global FORCE_SINGLE_RANK_COLLECTIVES
    FORCE_SINGLE_RANK_COLLECTIVES += 1
    weakref.finalize(owner, _release_forced)


def _release_forced():
    global FORCE_SINGLE_RANK_COLLECTIVES
    FORCE_SINGLE_RANK_COLLECTIVES = max(0, FORCE_SINGLE_RANK_COLLECTIVES - 1)


def _collective_needed(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_SINGLE_RANK_COLLECTIVES > 0


def allreduce_mean_(flat_grad, group=None):
    """In-place mean over ranks of one flat gradient buffer."""
    if not _collective_needed(group):
        return flat_grad
    world = dist.get_world_size(group)
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
    if not _collective_needed(group):
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
    if not _collective_needed(group):
        out[0].copy_(local)
        return PendingGather(out, None)
    try:
        work = dist.all_gather_into_tensor(out.view(-1), local, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):           # backends without the flat form (older gloo)
        work = dist.all_gather(list(out.unbind(0)), local, group=group, async_op=True)
    return PendingGather(out, work)


def sh_exchange_bytes(world, n_gaussians, views_per_rank=3, sh_floats=48):
    """Bytes a rank RECEIVES per iteration for the SH coefficients' gradient (48 of the 59 floats per Gaussian):
      factorised  all-gather of the per-view colour gradients: (world - 1) blocks of views * 3 floats per Gaussian
                  (+ every rank then reads world blocks from HBM to rebuild the mean gradient);
      allreduce   ring all-reduce of the SH gradient itself: 2 (world - 1) / world * 48 floats per Gaussian.
    The factorised exchange moves fewer bytes while world < 2 * 48 / (3 * views) (= 10.67 for an event triplet)."""
    colour = 4 * 3 * views_per_rank * n_gaussians
    fact = (world - 1) * colour
    ar = int(2 * (world - 1) / max(world, 1) * 4 * sh_floats * n_gaussians)
    return {"factorised_link_bytes": fact, "factorised_rebuild_read_bytes": world * colour, "allreduce_link_bytes": ar}


def choose_sh_exchange(world, views_per_rank=3, sh_floats=48):
    """'factorised' or 'allreduce' by the bytes that cross xGMI (sh_exchange_bytes), not by `world > 1`."""
    b = sh_exchange_bytes(world, 1, views_per_rank, sh_floats)
    return "factorised" if world > 1 and b["factorised_link_bytes"] < b["allreduce_link_bytes"] else "allreduce"


class PendingShard:
    def __init__(self, out, work, divide_by):
        self.out, self.work, self.divide_by = out, work, divide_by

    def wait(self):
        if self.work is not None:
            self.work.wait()
            if self.divide_by:
                self.out.div_(self.divide_by)
        return self.out


def reduce_scatter_mean_async_(out_shard, staging, group=None):
    """Mean over the ranks of `staging` (world * shard elements), rank r keeping elements [r * shard, (r + 1) * shard):
    the first half of the direct reduce-scatter + all-gather schedule -- every rank then runs the optimizer
    on its shard only and the updated parameters are all-gathered."""
    op, divide = _mean_op(group)
    work = dist.reduce_scatter_tensor(out_shard, staging, op=op, group=group, async_op=True)
    return PendingShard(out_shard, work, dist.get_world_size(group) if divide else 0)


def allgather_flat_async_(full, shard, group=None):
    """All-gather of one equally sized shard per rank into the flat tensor `full` (world * shard elements)."""
    return PendingShard(full, dist.all_gather_into_tensor(full, shard, group=group, async_op=True), 0)


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
