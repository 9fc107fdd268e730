"""Adaptive density control: clone / split / prune / opacity reset.

Mirrors scene/gaussian_model.py:258-407 and the schedule of train.py:317-327 on the tensors of the fused
trainer.  The core works on a plain dict  name -> [param, exp_avg, exp_avg_sq]  in the REFERENCE layout and on
any device, so it is pinned against the reference's own GaussianModel (tests/golden/densify.npz, same
torch.manual_seed -> identical torch.normal draws); EventTrainer.export_groups()/import_groups() convert
from/to the flat, coefficient-major training buffers.

Faithfully reproduced quirks: every append (even an empty one) zeroes xyz_gradient_accum, denom AND
max_radii2D for all Gaussians (densification_postfix :329-347), so the `max_radii2D > max_screen_size` prune
criterion only ever sees zeros (:396-401).
"""
import torch

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def build_rotation(r):
    """utils/general_utils.py:78-99 (normalises the quaternion)."""
    q = r / torch.sqrt((r * r).sum(1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(-1, 3, 3)


class DensifyStats:
    """max_radii2D (N), xyz_gradient_accum (N,1), denom (N,1) -- scene/gaussian_model.py:53-55."""

    def __init__(self, n, device):
        self.max_radii2D = torch.zeros(n, device=device)
        self.xyz_gradient_accum = torch.zeros(n, 1, device=device)
        self.denom = torch.zeros(n, 1, device=device)

    def update(self, viewspace_grad, radii):
        """train.py:319-320 + gaussian_model.py:405-407, without boolean-mask gathers: `viewspace_grad` is the
        (N,3) NDC-unit screen-space gradient of render #1, `radii` its int32 radii (visible <=> radii > 0)."""
        if viewspace_grad.is_cuda:        # one kernel instead of ten elementwise launches
            from . import _lib
            g = viewspace_grad if viewspace_grad.is_contiguous() else viewspace_grad.contiguous()
            r = radii if (radii.dtype == torch.int32 and radii.is_contiguous()) else radii.to(torch.int32).contiguous()
            with torch.cuda.device(g.device):
                rc = _lib.lib().e3dgs_densify_stats_update(g.shape[0], _lib.ptr(g), _lib.ptr(r), _lib.ptr(self.max_radii2D),
                                                           _lib.ptr(self.xyz_gradient_accum), _lib.ptr(self.denom),
                                                           _lib.current_stream())
            _lib.check(rc, "e3dgs_densify_stats_update")
            return
        vis = radii > 0
        self.max_radii2D = torch.where(vis, torch.maximum(self.max_radii2D, radii.to(self.max_radii2D.dtype)),
                                       self.max_radii2D)
        visf = vis.to(self.denom.dtype).unsqueeze(1)
        self.xyz_gradient_accum = self.xyz_gradient_accum + visf * torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True)
        self.denom = self.denom + visf

    def sync(self, group=None):
        """Multi-rank training: every rank accumulated the statistics of ITS camera triplets; before a
        densification step they are combined -- sum of the gradient-norm accumulators and view counts, max of the
        screen radii -- so that all ranks take the identical clone / split / prune decisions.  (Every densification
        step resets the statistics, so combining the accumulators equals combining the per-iteration deltas.)"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        both = torch.cat((self.xyz_gradient_accum, self.denom), dim=1).contiguous()
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        self.xyz_gradient_accum, self.denom = both[:, :1].contiguous(), both[:, 1:].contiguous()
        r = self.max_radii2D.contiguous()
        dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
        self.max_radii2D = r

    def _zero(self, n, device):
        self.xyz_gradient_accum = torch.zeros(n, 1, device=device)
        self.denom = torch.zeros(n, 1, device=device)
        self.max_radii2D = torch.zeros(n, device=device)

    def _mask(self, keep):
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]


def _append(groups, stats, new):
    """cat_tensors_to_optimizer + densification_postfix (:307-347): new rows get zero Adam moments."""
    for name in GROUPS:
        p, m, v = groups[name]
        ext = new[name]
        groups[name] = [torch.cat((p, ext), 0), torch.cat((m, torch.zeros_like(ext)), 0),
                        torch.cat((v, torch.zeros_like(ext)), 0)]
    n = groups["xyz"][0].shape[0]
    stats._zero(n, groups["xyz"][0].device)


def _prune(groups, stats, mask):
    """prune_points / _prune_optimizer (:273-305).  The reference indexes 21 tensors with the boolean mask (a
    nonzero + host sync each); here the row list is built once and gathered."""
    keep = torch.nonzero(~mask).squeeze(1)
    for name in GROUPS:
        groups[name] = [t.index_select(0, keep) for t in groups[name]]
    stats._mask(keep)


def densify_and_prune(groups, stats, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01, N=2):
    """scene/gaussian_model.py:389-403 (clone :374-387, split :349-372).  Mutates `groups` and `stats`."""
    grads = stats.xyz_gradient_accum / stats.denom
    grads[grads.isnan()] = 0.0
    scaling = lambda: torch.exp(groups["scaling"][0])
    # ---- clone
    sel = (torch.norm(grads, dim=-1) >= max_grad) & (scaling().max(dim=1).values <= percent_dense * extent)
    rows = torch.nonzero(sel).squeeze(1)
    _append(groups, stats, {k: groups[k][0].index_select(0, rows) for k in GROUPS})
    # ---- split
    n_init = groups["xyz"][0].shape[0]
    padded = torch.zeros(n_init, device=grads.device)
    padded[:grads.shape[0]] = grads.squeeze()
    sel = (padded >= max_grad) & (scaling().max(dim=1).values > percent_dense * extent)
    rows = torch.nonzero(sel).squeeze(1)
    pick = lambda k: groups[k][0].index_select(0, rows)
    sc_sel = torch.exp(pick("scaling"))
    stds = sc_sel.repeat(N, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
    rot_sel = pick("rotation")
    rots = build_rotation(rot_sel).repeat(N, 1, 1)
    new = {
        "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + pick("xyz").repeat(N, 1),
        "scaling": torch.log(sc_sel.repeat(N, 1) / (0.8 * N)),
        "rotation": rot_sel.repeat(N, 1),
        "f_dc": pick("f_dc").repeat(N, 1, 1),
        "f_rest": pick("f_rest").repeat(N, 1, 1),
        "opacity": pick("opacity").repeat(N, 1),
    }
    _append(groups, stats, new)
    # ---- prune: the split parents (prune_points at :372) and the opacity / size test (:396-402) in ONE gather.  Both
    # masks are row-wise functions of the post-split model, so applying them together equals applying them in turn;
    # max_radii2D was just zeroed by the postfix (:347), as in the reference.
    drop = torch.cat((sel, torch.zeros(N * rows.shape[0], device=sel.device, dtype=torch.bool)))
    prune_mask = (torch.sigmoid(groups["opacity"][0]) < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = stats.max_radii2D > max_screen_size
        big_ws = scaling().max(dim=1).values > 0.1 * extent
        prune_mask = prune_mask | big_vs | big_ws
    _prune(groups, stats, drop | prune_mask)
    return groups["xyz"][0].shape[0]


def reset_opacity(groups):
    """scene/gaussian_model.py:210-213 + replace_tensor_to_optimizer (:258-271): the opacity moments are zeroed."""
    o = groups["opacity"][0]
    x = torch.min(torch.sigmoid(o), torch.ones_like(o) * 0.01)
    new = torch.log(x / (1 - x))
    groups["opacity"] = [new, torch.zeros_like(new), torch.zeros_like(new)]


def densification_schedule(iteration, opt_densify_until_iter=15000, densify_from_iter=500, densification_interval=100,
                           opacity_reset_interval=3000, white_background=False):
    """train.py:317-327 -> (update_stats, do_densify, size_threshold, do_reset)."""
    update = iteration < opt_densify_until_iter
    dens = update and iteration > densify_from_iter and iteration % densification_interval == 0
    size_threshold = 20 if iteration > opacity_reset_interval else None
    reset = update and (iteration % opacity_reset_interval == 0 or (white_background and iteration == densify_from_iter))
    return update, dens, size_threshold, reset
