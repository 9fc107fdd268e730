"""Randomised check of the fused training iteration (EventTrainer.step: one multi-view pass, loss kernel, fused
backward) against the reference-style composition (torch activations + drop-in operator + autograd loss,
EventTrainer.step_autograd) from identical parameters: loss, gradients of every group, dL/dc.
Random Gaussian counts (1..20000), frames (from 8x8, not tile multiples), SH degree, deblur term, backgrounds,
camera distances, scale boosts.  Usage (GPU box, repo root):  python tools/fuzz_step.py [cases] [first_seed]
FUZZ_MODE=image: the one-render gray / RGB iterations (step_image vs step_image_autograd) instead.
FUZZ_MODE=shared: the shared-pose iteration (two views, with / without the statistics chain) vs three renders.
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from event_3dgs_amd import synth                     # noqa: E402
from event_3dgs_amd.cameras import orbit_camera      # noqa: E402
from event_3dgs_amd.train_step import EventTrainer   # noqa: E402

DEV = torch.device("cuda", 0)


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def check(seed):
    r = np.random.default_rng(seed)
    N = int(r.choice([1, 3, 50, 700, 5000, 20000]))
    W, H = int(r.integers(8, 260)), int(r.integers(8, 200))
    deg = int(r.integers(0, 4))
    deblur = bool(r.random() < 0.4)
    boost = float(np.exp(r.uniform(math.log(0.2), math.log(12.0))))
    radius = float(r.choice([1.0, 2.5, 4.0, 8.0]))
    bgv = float(r.choice([0.0, 0.5, 1.0]))
    params = synth.make_scene(N, "trained", seed=seed, device=DEV)
    params["scaling"] = params["scaling"] + math.log(boost)
    cams = [orbit_camera(int(r.integers(0, 16)), 16, W, H, device=DEV, radius=radius, daz=d) for d in (0.0, 0.004, 0.012)]
    bg = torch.full((3,), bgv, device=DEV)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(params["xyz"].shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = EventTrainer(gp, DEV, active_sh_degree=deg)
    q8 = lambda x: (torch.round(x.clamp(0, 1) * 255) / 255).contiguous()
    gts = [q8(t.render_raw(c, bg)["color"]) for c in cams]
    blur = (0.5 * (gts[0] + gts[2])).contiguous() if deblur else None
    a, b = EventTrainer(params, DEV, active_sh_degree=deg), EventTrainer(params, DEV, active_sh_degree=deg)
    a2 = EventTrainer(params, DEV, active_sh_degree=deg)
    # a: gradients in memory, then the optimizer; a2: step() as training runs it (SH gradient rebuilt inside the SH
    # optimizer kernel, never stored) -- must land on the same parameters bit for bit
    sa = a.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur).clone()
    a.apply_update()
    a2.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)
    lb = b.step_autograd(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)
    torch.cuda.synchronize()
    problems = []
    if not (torch.equal(a.flat, a2.flat) and torch.equal(a.exp_avg, a2.exp_avg) and torch.equal(a.exp_avg_sq, a2.exp_avg_sq)):
        problems.append("step() differs from compute_gradients() + apply_update()")
    # The two trainers differ by <= 1 ulp in the activations (kernel vs torch); the rasteriser is discontinuous at its
    # alpha >= 1/255 threshold, so a few pixels on a splat's outline may flip -- with a handful of Gaussians that is
    # visible at the 1e-3 level, with hundreds it is not.
    # On a black background ln(Y + 1e-8) amplifies any difference at the dark pixels by up to 1e8 (the reference's loss
    # is that ill-conditioned there), so only gross errors are flagged for bg = 0.
    # L1's gradient is sign(x): on a frame of a few thousand pixels, three pixel-channels whose |image - gt| is below the
    # 1e-7 by which the two paths' images differ flip their sign and move every gradient by 2 % (seed 817: 21 x 115, images
    # equal to 1.2e-7, no radius or threshold flip) -- small frames are compared loosely too.
    # (seeds 10078 / 10343: 700 Gaussians on 159 x 102 / 164 x 157 frames: loss to 2e-4, every gradient group to 3e-3 but
    # rotation 0.8 % and dL/dc 0.4 % / 3.6 % -- dL/dc is a sum of sign(e) D / c over the pixels)
    loose = N < 500 or bgv == 0.0 or W * H < 30000
    sa0, lb0 = float(sa[0].detach()), float(lb.detach())
    if not math.isfinite(sa0) or abs(sa0 - lb0) > (2e-3 if loose else 2e-4) * max(abs(lb0), 1e-6):
        problems.append("loss %.7g vs %.7g" % (sa0, lb0))
    for name in ("xyz", "features", "opacity", "scaling", "rotation"):
        ga, gb = a.grads[name].cpu().numpy(), b.grads[name].cpu().numpy()
        if not np.isfinite(ga).all():
            problems.append("non-finite grad " + name)
            continue
        sc = float(np.linalg.norm(gb))
        # (a group whose whole gradient is below 1e-6 is compared absolutely: seed 4149 -- one Gaussian with a saturated
        # opacity, sigmoid = 1.0f in the kernel and 1 - 2^-24 in torch, gradient 0 vs -4e-8)
        err = rel_l2(ga, gb) if sc > 1e-6 else float(np.abs(ga - gb).max())
        if err > (3e-2 if loose else 3e-3):
            problems.append("grad %s err %.3g (|ref| %.3g)" % (name, err, sc))
    cg, cr = float(a.c_grad), float(b.c_grad)
    # (dL/dc is a signed sum over the pixels -- typically 0.1; where it nearly cancels, seed 10343: 0.0055, the 2e-4 by
    # which the two paths differ is 3.6 % of it: the floor keeps the comparison at the scale of the terms)
    if abs(cg - cr) > (3e-2 if loose else 3e-3) * max(abs(cr), 2e-2):
        problems.append("dc %.6g vs %.6g" % (cg, cr))
    return "step seed %d: N=%d %dx%d deg=%d deblur=%s boost %.2f r=%.1f bg=%.1f" % (
        seed, N, W, H, deg, deblur, boost, radius, bgv), problems


def check_image(seed):
    """The one-render iterations (--gray / RGB): fused step_image against step_image_autograd."""
    r = np.random.default_rng(50_000 + seed)
    N = int(r.choice([1, 50, 700, 5000, 20000]))
    W, H = int(r.integers(8, 260)), int(r.integers(8, 200))
    deg = int(r.integers(0, 4))
    mode = str(r.choice(["gray", "rgb"]))
    lam = float(r.choice([0.2, 0.2, 0.0, 0.8]))
    boost = float(np.exp(r.uniform(math.log(0.2), math.log(12.0))))
    bgv = float(r.choice([0.0, 0.5, 1.0]))
    params = synth.make_scene(N, "trained", seed=seed, device=DEV)
    params["scaling"] = params["scaling"] + math.log(boost)
    cam = orbit_camera(int(r.integers(0, 16)), 16, W, H, device=DEV, radius=float(r.choice([1.0, 2.5, 4.0])))
    bg = torch.full((3,), bgv, device=DEV)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(params["xyz"].shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    gt = (torch.round(EventTrainer(gp, DEV, active_sh_degree=deg).render_raw(cam, bg)["color"].clamp(0, 1) * 255) / 255).contiguous()
    a, b = EventTrainer(params, DEV, active_sh_degree=deg), EventTrainer(params, DEV, active_sh_degree=deg)
    a2 = EventTrainer(params, DEV, active_sh_degree=deg)
    # a: gradients in memory (single-view per-Gaussian kernel), then the generic optimizer; a2: step_image() as training
    # runs it (multi-view per-Gaussian kernel with one view + SH gradient rebuilt inside the SH optimizer kernel): same
    # mathematics through different kernels -- the moments agree to rounding
    la = a.compute_gradients_image(cam, gt, bg, mode=mode, lambda_dssim=lam).clone()
    a.apply_update(skip=("c",))
    a2.step_image(cam, gt, bg, mode=mode, lambda_dssim=lam)
    lb = b.step_image_autograd(cam, gt, bg, mode=mode, lambda_dssim=lam)
    torch.cuda.synchronize()
    problems = []
    for nm, x, y in (("exp_avg", a.exp_avg, a2.exp_avg), ("exp_avg_sq", a.exp_avg_sq, a2.exp_avg_sq)):
        e = rel_l2(x.cpu().numpy(), y.cpu().numpy())
        if not e <= 2e-5:
            problems.append("step_image() vs compute_gradients_image() + apply_update(): %s rel. L2 %.3g" % (nm, e))
    loose = N < 500 or W * H < 6000       # (small frames: L1 sign flips, as in the event iteration -- seed 20298: 16 x 109)
    tiny = N <= 3                         # one outline pixel of a lone Gaussian is percents of its gradient (seed 10128)
    la0, lb0 = float(la.detach()), float(lb.detach())
    # 1 - SSIM of two nearly equal images is a difference of fp32 numbers close to 1: 1e-6 absolute on the loss
    if not math.isfinite(la0) or abs(la0 - lb0) > (2e-3 if loose else 2e-4) * max(abs(lb0), 1e-6) + 1e-6:
        problems.append("loss %.7g vs %.7g" % (la0, lb0))
    for name in ("xyz", "features", "opacity", "scaling", "rotation"):
        ga, gb = a.grads[name].cpu().numpy(), b.grads[name].cpu().numpy()
        if not np.isfinite(ga).all():
            problems.append("non-finite grad " + name)
            continue
        sc = float(np.linalg.norm(gb))
        err = rel_l2(ga, gb) if sc > 1e-9 else float(np.abs(ga).max())
        if err > (1e-1 if tiny else 3e-2 if loose else 3e-3):
            problems.append("grad %s err %.3g (|ref| %.3g)" % (name, err, sc))
    return "image seed %d: N=%d %dx%d deg=%d mode=%s lambda=%.1f boost %.2f bg=%.1f" % (
        seed, N, W, H, deg, mode, lam, boost, bgv), problems


def check_shared(seed):
    """Shared pose (cam_now at cam_int's pose, as on the reference's datasets): the two-view iteration -- with and without
    the statistics chain -- against the same trainer forced to three renders: loss bits, gradients, screen-space
    statistics input."""
    r = np.random.default_rng(90_000 + seed)
    N = int(r.choice([1, 3, 50, 700, 5000, 20000]))
    W, H = int(r.integers(8, 260)), int(r.integers(8, 200))
    deg = int(r.integers(0, 4))
    deblur = bool(r.random() < 0.4)
    stats = bool(r.random() < 0.6)
    boost = float(np.exp(r.uniform(math.log(0.2), math.log(12.0))))
    radius = float(r.choice([1.0, 2.5, 4.0, 8.0]))
    bgv = float(r.choice([0.0, 0.5, 1.0]))
    params = synth.make_scene(N, "trained", seed=seed, device=DEV)
    params["scaling"] = params["scaling"] + math.log(boost)
    k = int(r.integers(0, 16))
    cams = [orbit_camera(k, 16, W, H, device=DEV, radius=radius, daz=d) for d in (0.0, 0.0, 0.012)]
    bg = torch.full((3,), bgv, device=DEV)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(params["xyz"].shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = EventTrainer(gp, DEV, active_sh_degree=deg)
    q8 = lambda x: (torch.round(x.clamp(0, 1) * 255) / 255).contiguous()
    gts = [q8(t.render_raw(c, bg)["color"]) for c in (cams[0], orbit_camera(k, 16, W, H, device=DEV, radius=radius, daz=0.004), cams[2])]
    blur = (0.5 * (gts[0] + gts[2])).contiguous() if deblur else None
    a = EventTrainer(params, DEV, active_sh_degree=deg, track_densification_stats=stats)
    b = EventTrainer(params, DEV, active_sh_degree=deg, track_densification_stats=stats)
    a.SHARE_STATS_MIN_INSTANCES = a.SHARE_STATS_MIN_TILES = 0
    b.share_coincident_views = False
    sa = a.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur).clone()
    sb = b.compute_gradients(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur).clone()
    torch.cuda.synchronize()
    problems = []
    if a.shared_pose_iterations != 1 or b.shared_pose_iterations != 0:
        problems.append("sharing not taken / taken: %d %d" % (a.shared_pose_iterations, b.shared_pose_iterations))
    if not torch.equal(sa[:6], sb[:6]):
        problems.append("loss scalars differ: %s vs %s" % (sa[:6].tolist(), sb[:6].tolist()))
    pairs = [(n, a.grads[n], b.grads[n]) for n in ("xyz", "features", "opacity", "scaling", "rotation")]
    if stats:
        pairs.append(("viewspace", a.viewspace_grad, b.viewspace_grad))
    for name, x, y in pairs:
        ga, gb = x.cpu().numpy(), y.cpu().numpy()
        if not np.isfinite(ga).all():
            problems.append("non-finite grad " + name)
            continue
        sc = float(np.linalg.norm(gb))
        err = rel_l2(ga, gb) if sc > 1e-6 else float(np.abs(ga - gb).max())
        # one backward on the summed pixel gradient vs the sum of two backwards: both are fp32 evaluations of the same
        # sum -- against the float64 PyTorch oracle they sit equally far (seeds 252 / 292: 2e-5 .. 5e-5 each, 4e-5 apart; scaling / rotation of 4 seeds in 400: 3e-4 .. 1.2e-3 apart)
        if err > (5e-3 if N < 500 else 2e-3):
            problems.append("grad %s err %.3g (|ref| %.3g)" % (name, err, sc))
    return "shared seed %d: N=%d %dx%d deg=%d deblur=%s stats=%s boost %.2f r=%.1f bg=%.1f" % (
        seed, N, W, H, deg, deblur, stats, boost, radius, bgv), problems


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + cases):
        if os.environ.get("FUZZ_VERBOSE") == "1":
            print("seed", seed, flush=True)
        desc, problems = {"image": check_image, "shared": check_shared}.get(os.environ.get("FUZZ_MODE"), check)(seed)
        if problems:
            bad += 1
            print("FAIL", desc, "|", "; ".join(problems), flush=True)
    print("fuzz_step: %d cases, %d failing" % (cases, bad))
    sys.exit(1 if bad else 0)
