"""Soak run of the training loop on the GPU at the size of BASELINE.json configs[1] (200 k-Gaussian class, 800x800,
event iterations): point-cloud initialisation -> thousands of event iterations with the densification schedule,
SH ramp and an opacity reset -> gray PSNR / SSIM on the held-out views (eval.py:118-152 protocol) before and after.

Checks on the way: parameters and loss finite every 100 iterations, Gaussian count trajectory, iterations/s.
Usage (GPU box, repo root):  python tools/soak_fit.py [iterations] [gt_gaussians] [init_points] [size | WxH] [background] [mode]
BASELINE.json configs[2] scale (round 6): python tools/soak_fit.py 5000 1500000 700000 1920x1080
Writes one JSON line (also to gpurun_out/soak_fit.json when that directory exists).
"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from event_3dgs_amd import fit, io_formats, scene_io, synth          # noqa: E402
from event_3dgs_amd.cameras import orbit_camera                      # noqa: E402
from event_3dgs_amd.train_step import EventTrainer                   # noqa: E402
from simple_knn._C import distCUDA2                                  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N_GT = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
N_INIT = int(sys.argv[3]) if len(sys.argv) > 3 else 30_000
SIZE = sys.argv[4] if len(sys.argv) > 4 else "800"          # "800" (square) or "1920x1080"
W_, H_ = (int(x) for x in SIZE.split("x")) if "x" in SIZE else (int(SIZE), int(SIZE))
BG = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5        # black backgrounds put ln(1e-8) into the contrast GT
MODE = sys.argv[6] if len(sys.argv) > 6 else "event"          # event | gray | rgb (fit.fit_event_scene)
K = 100
dev = torch.device("cuda", 0)
bg = torch.full((3,), BG, device=dev)

gt_params = synth.make_scene(N_GT, "trained", seed=4, device=dev)
gt_tr = EventTrainer(gt_params, dev)
q8 = lambda t: (torch.round(t.clamp(0, 1) * 255) / 255).contiguous()
train, events = [], []
for k in range(K):
    # SOAK_SHARED=1: the event camera k carries the pose of the training camera k, as in the reference's datasets
    # (scene/dataset_readers.py:157) -- the event pair is then (view k, view k + 1) and EventTrainer renders two views per
    # iteration once the densification statistics are no longer collected
    for lst, daz in ((train, 0.0), (events, 0.0 if os.environ.get("SOAK_SHARED") == "1" else 0.003)):
        c = orbit_camera(k, K, W_, H_, device=dev, daz=daz)
        c.original_image = q8(gt_tr.render_raw(c, bg)["color"])
        lst.append(c)
del gt_tr
xyz = gt_params["xyz"].cpu().numpy()
sel = np.random.default_rng(0).permutation(N_GT)[:N_INIT]
pcd = io_formats.BasicPointCloud(xyz[sel], np.full((N_INIT, 3), 0.5), np.zeros((N_INIT, 3)))
params = scene_io.create_from_pcd(pcd, 1.0, distCUDA2, device=dev)

probe = EventTrainer({k: v.clone() for k, v in params.items()}, dev)
before = scene_io.evaluate_views(lambda cam: probe.render_raw(cam, bg)["color"], train)
del probe

log = dict(n=[], loss=[], nonfinite=0)
t_start = [None]


def watch(it, tr, scalars):
    if it == 10:
        torch.cuda.synchronize(); t_start[0] = time.perf_counter()
        torch.cuda.reset_peak_memory_stats(dev)
    if it % 100 == 0:
        loss = float(scalars.reshape(-1)[0])                    # scalars[0] = loss (losses.event_loss_raw)
        g = tr.export_groups()
        ok = all(bool(torch.isfinite(v[0]).all()) for v in g.values())
        log["n"].append(tr.N); log["loss"].append(round(loss, 5))
        if not ok or not np.isfinite(loss):
            log["nonfinite"] += 1


rnd = random.Random(0)
tr = fit.fit_event_scene(params, train, events, bg, dev, iterations=ITERS, cameras_extent=4.4,
                         densify_from_iter=500, densification_interval=100, densify_until_iter=int(ITERS * 0.7),
                         opacity_reset_interval=max(ITERS // 2, 1000), sh_ramp_interval=max(ITERS // 6, 100),
                         rng=rnd.randint, on_iteration=watch, mode=MODE,
                         densify_grad_threshold=float(os.environ.get("SOAK_GRAD_THRESHOLD", "0.0002")))
torch.cuda.synchronize()
elapsed = time.perf_counter() - t_start[0]
after = scene_io.evaluate_views(lambda cam: tr.render_raw(cam, bg)["color"], train)
out = dict(iterations=ITERS, size=SIZE, background=BG, mode=MODE, gt_gaussians=N_GT, init_points=N_INIT, final_gaussians=tr.N,
           iters_per_s=round((ITERS - 10) / elapsed, 1), psnr_before=round(float(before["psnr"]), 2),
           psnr_after=round(float(after["psnr"]), 2), ssim_before=round(float(before["ssim"]), 4),
           ssim_after=round(float(after["ssim"]), 4), nonfinite_checks=log["nonfinite"],
           gaussians_every_100=log["n"], loss_every_100=log["loss"], sh_degree=tr.active_sh_degree,
           shared_pose_iterations=tr.shared_pose_iterations, count_retries=tr.count_retries,
           peak_memory_gb=round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
           densify_grad_threshold=float(os.environ.get("SOAK_GRAD_THRESHOLD", "0.0002")),
           what="iterations/s over ALL iterations from the 10th on: densification (every 100 until 70 %), the opacity reset "
                "and the statistics iterations included; held-out views 5/25/45/65/85 scored with eval.py:118-152's protocol")
line = json.dumps(out)
print(line)
if os.path.isdir("gpurun_out"):
    open("gpurun_out/soak_fit_%s_%d.json" % (SIZE, N_GT), "w").write(line + "\n")
