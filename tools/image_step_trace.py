#!/usr/bin/env python
"""Timing of the one-render iterations (train.py --gray :213-223 / RGB :292-296: L1 + SSIM) on the fused path at the
benchmark scene size: ms per iteration; run under `rocprofv3 --kernel-trace` for the per-kernel split.
Usage (GPU box): python tools/image_step_trace.py [gray|rgb] [N W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
mode = sys.argv[1] if len(sys.argv) > 1 else "gray"
N, W, H = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (1_000_000, 1920, 1080)
dev = torch.device("cuda:0")
params = synth.make_scene(N, "trained", seed=0, device=dev)
gp = dict(params); gp["xyz"] = params["xyz"] + 0.01 * torch.randn_like(params["xyz"])
bg = torch.zeros(3, device=dev)
cams = [orbit_camera(k, 64, W, H, device=dev) for k in range(4)]
gts = [EventTrainer(gp, dev).render_raw(c, bg)["color"].clamp(0, 1).contiguous() for c in cams]
tr = EventTrainer(params, dev)
for k in range(4):
    tr.step_image(cams[k], gts[k], bg, mode=mode)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for k in range(K):
    tr.step_image(cams[k % 4], gts[k % 4], bg, mode=mode)
torch.cuda.synchronize()
print("IMAGE_STEP", mode, N, W, H, "ms per iteration %.3f" % (1e3 * (time.perf_counter() - t0) / K))
