#!/usr/bin/env python
"""N steps of the event iteration at a given size (for rocprofv3 timelines of small / mid scenes).
Usage: python tools/small_step.py [N] [W] [H] [kind] [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
kind = sys.argv[4] if len(sys.argv) > 4 else "trained"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
from simple_knn._C import distCUDA2
params = synth.make_scene(N, kind, seed=0, device=dev, dist2_fn=distCUDA2)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gt = EventTrainer(params, dev)
gts = [gt.render_raw(c, bg)["color"].clone() for c in cams]
tr = EventTrainer(params, dev)
for _ in range(10): tr.step(*cams, *gts, bg)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): tr.step(*cams, *gts, bg)
torch.cuda.synchronize()
print("N=%d %dx%d %s: %.3f ms/step" % (N, W, H, kind, (time.perf_counter() - t0) / steps * 1e3))
