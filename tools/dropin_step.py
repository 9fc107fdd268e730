#!/usr/bin/env python
"""The event iteration through the DROP-IN operator (three GaussianRasterizer calls + torch autograd + torch activations,
EventTrainer.step_autograd) next to the fused iteration (EventTrainer.step), same scene and cameras.
Usage: python tools/dropin_step.py [N] [W] [H]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gt = EventTrainer(params, dev)
gts = [gt.render_raw(c, bg)["color"].clone() for c in cams]
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = EventTrainer(params, dev); b = EventTrainer(params, dev)
print("DROPIN N=%d %dx%d: fused step %.3f ms, drop-in operator + autograd step %.3f ms" % (
    N, W, H, timed(lambda: a.step(*cams, *gts, bg)), timed(lambda: b.step_autograd(*cams, *gts, bg))))
