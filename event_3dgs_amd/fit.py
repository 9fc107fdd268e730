"""The reference's `--event` training loop (train.py:45-336) on the fused trainer: camera sampling,
SH-degree ramp, per-iteration event step, densification schedule, opacity reset.  No GUI / TensorBoard /
argparse (SURVEY section 2 marks those out of scope); this is the loop body a user of the reference needs.
"""
from random import randint

import torch

import torch.distributed as dist

from . import densify, parallel
from .train_step import EventTrainer

HELD_OUT = (5, 25, 45, 65, 85)       # evaluation views, train.py:129-131 / eval.py:118


def sample_index(n_cameras, mode="event", rng=randint):
    """train.py:116-131: randint(2, n-4) in event mode, randint(2, n-3) otherwise; the held-out views (5, 25, 45, 65, 85)
    shift down by one in event and gray mode only (:129-131) -- plain RGB training samples them."""
    index = rng(2, n_cameras - 4) if mode == "event" else rng(2, n_cameras - 3)
    if mode in ("event", "gray") and index in HELD_OUT:
        index -= 1
    return index


def fit_event_scene(params, train_cameras, event_cameras, bg, device, iterations, cameras_extent=1.0,
                    blurry_cameras=None, densify_until_iter=15000, densify_from_iter=500, densification_interval=100,
                    opacity_reset_interval=None, densify_grad_threshold=0.0002, percent_dense=0.01, white_background=False,
                    sh_ramp_interval=1000, max_sh_degree=3, start_sh_degree=0, rng=randint, on_iteration=None,
                    seed=0, mode="event", lambda_dssim=0.2, **trainer_kw):
    """Returns the trained EventTrainer.  opacity_reset_interval defaults to the value the reference uses for the
    mode (10000 forced in event mode at train.py:119, otherwise the 3000 of arguments/__init__.py).  `params` = pre-activation dict (synth.make_scene / scene_io.create_from_pcd).

    With an initialised process group (one process per GPU) every rank draws its own camera triplet
    (parallel.rank_camera_indices), gradients are averaged inside EventTrainer.step, and before each densification
    step the statistics are combined over the ranks and the split sampler is seeded identically, so the replicas
    stay bit-identical through clone / split / prune.

    mode: "event" (train.py:149-212, the north-star path), "gray" (train.py:213-223) or "rgb" (train.py:292-296);
    the last two render one camera per iteration and need no event cameras (event_cameras may be None).

    Order inside an iteration, as train.py:144-332: forward + loss + backward; optimizer_c.step() (event mode, :212);
    densification statistics; densify_and_prune / reset_opacity; THEN the Gaussian optimizer step.  Densification and
    the opacity reset replace the parameters before optimizer.step(), so torch finds no gradient on them: on a
    densification iteration none of the six Gaussian groups is updated (their step counts stall too), on a reset
    iteration the opacity group is not, and the last iteration (`iteration < opt.iterations`, :330) has no Gaussian
    step at all.  The contrast threshold c steps on every event iteration."""
    if mode not in ("event", "gray", "rgb"):
        raise ValueError("mode must be 'event', 'gray' or 'rgb'")
    if opacity_reset_interval is None:              # train.py:119 forces 10000 in event mode; arguments/__init__.py: 3000
        opacity_reset_interval = 10000 if mode == "event" else 3000
    # (a dataset whose frames are not all of one size needs no special case: a triplet that mixes resolutions runs as two
    # multi-view passes and hands out the colour-gradient block every exchange expects, EventTrainer._compute_gradients_two_sizes)
    tr = EventTrainer(params, device, spatial_lr_scale=cameras_extent, active_sh_degree=start_sh_degree,
                      track_densification_stats=True, **trainer_kw)
    stats = densify.DensifyStats(tr.N, device)
    world = dist.get_world_size(tr.pg) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(tr.pg) if world > 1 else 0
    for iteration in range(1, iterations + 1):
        if iteration % sh_ramp_interval == 0 and tr.active_sh_degree < max_sh_degree:      # train.py:99-100
            tr.active_sh_degree += 1
        if world > 1:
            index = parallel.rank_camera_indices(rank, world, len(train_cameras), iteration, seed, HELD_OUT, mode=mode)
        else:
            index = sample_index(len(train_cameras), mode, rng)
        cam = train_cameras[index]
        upd, dens, size_thr, reset = densify.densification_schedule(
            iteration, densify_until_iter, densify_from_iter, densification_interval, opacity_reset_interval,
            white_background)
        if mode == "event":
            now, nxt = event_cameras[index], event_cameras[index + 1]
            blur = blurry_cameras[index].original_image if blurry_cameras else None         # train.py:197-203
            scalars = tr.compute_gradients(cam, now, nxt, cam.original_image, now.original_image, nxt.original_image,
                                           bg, gt_blur=blur, sh_via_colour=tr.sh_via_colour and not tr.overlap_features,
                                           viewspace_grad=upd)       # (statistics iterations need render #1's own gradient)
        else:
            scalars = tr.compute_gradients_image(cam, cam.original_image, bg, mode=mode, lambda_dssim=lambda_dssim,
                                                 sh_via_colour=tr.sh_via_colour and not tr.overlap_features)
        scalars = scalars.clone()
        skip = set() if mode == "event" else {"c"}
        if dens or iteration == iterations:
            skip.add("gaussians")
        if reset:
            skip.add("opacity")
        if "gaussians" in skip and "c" not in skip:
            # optimizer_c.step() comes right after backward (train.py:212), before the parameters are replaced; the
            # Gaussian gradients of this iteration are dropped, as torch drops them with the replaced tensors
            tr.apply_update(skip=("gaussians",))
        if upd:                                                                             # train.py:317-327
            stats.update(tr.viewspace_grad, tr.last_radii)
            if dens:
                if world > 1:
                    stats.sync(tr.pg)
                    torch.manual_seed(seed * 1000003 + iteration)       # identical torch.normal draws in the split
                tr.densify_and_prune(stats, densify_grad_threshold, 0.005, cameras_extent, size_thr, percent_dense)
            if reset:
                tr.reset_opacity()
        if "gaussians" not in skip:
            tr.apply_update(skip=tuple(skip))                                               # train.py:330-332
        elif "c" in skip:
            tr.iteration += 1                                    # nothing steps, the learning-rate clock still advances
        if on_iteration is not None:
            on_iteration(iteration, tr, scalars)
    return tr
