#!/usr/bin/env python
"""Where the host time of one training iteration goes: wall per step back to back, host time until step() returns,
and a cProfile of 200 steps.  Usage: python tools/host_profile.py [N] [W] [H] [kind]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
kind = sys.argv[4] if len(sys.argv) > 4 else "trained"
params = synth.make_scene(N, kind, seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gt = EventTrainer(params, dev)
gts = [gt.render_raw(c, bg)["color"].clone() for c in cams]
tr = EventTrainer(params, dev)
for _ in range(20): tr.step(*cams, *gts, bg)
torch.cuda.synchronize()
host = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(*cams, *gts, bg)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): tr.step(*cams, *gts, bg)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("N=%d %dx%d %s: wall %.3f ms/step back to back, host-only (GPU idle at entry) %.3f ms" % (N, W, H, kind, wall * 1e3, sorted(host)[len(host) // 2] * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): tr.step(*cams, *gts, bg)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
