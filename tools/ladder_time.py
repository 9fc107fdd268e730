#!/usr/bin/env python
"""The adoption ladder (event_3dgs_amd/adopt.py) timed on its own: ms per iteration of the reference's event iteration at
the given rungs.  Usage: python tools/ladder_time.py [N] [W] [H] [rungs, e.g. 0123 4]   (A/B switches through the
environment: E3DGS_COUNT_POLL=0, E3DGS_CPP_AUTOGRAD=0, ...)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import adopt, synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
rungs = [int(c) for c in (sys.argv[4] if len(sys.argv) > 4 else "01234")]
params = synth.make_scene(N, "trained", seed=0, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
gt = EventTrainer(params, dev)
gts = [(torch.round(gt.render_raw(c, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous() for c in cams]
del gt
for rung in rungs:
    loop = adopt.LadderLoop(rung, params, dev)
    for _ in range(3):
        loop.step(cams, gts, bg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8 if rung == 0 else 20
    for _ in range(n):
        loop.step(cams, gts, bg)
    torch.cuda.synchronize()
    print("LADDER rung %d N=%d %dx%d: %.3f ms per iteration" % (rung, N, W, H, (time.perf_counter() - t0) / n * 1e3))
    del loop
