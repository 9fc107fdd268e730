#!/usr/bin/env python
"""Event iteration on a scene whose event camera `now` shares the training camera's pose (the reference's datasets):
three renders / shared pose / shared pose + densification statistics (second dL/dalpha chain).  ms per iteration and
the compositing backward's share.  Usage (GPU box): python tools/shared_pose_time.py [N W H]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import synth, _lib
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
N, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1_000_000, 1920, 1080)
dev = torch.device("cuda:0")
params = synth.make_scene(N, "trained", seed=0, device=dev)
bg = torch.zeros(3, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.0, 0.015)]
gp = dict(params); gp["xyz"] = params["xyz"] + 0.01 * torch.randn_like(params["xyz"])
gts = [EventTrainer(gp, dev).render_raw(c, bg)["color"].clamp(0, 1).contiguous() for c in cams]
L = _lib.lib()
import gc; gc.disable()
for tag, share, stats in (("three renders", False, False), ("three renders + statistics", False, True),
                          ("shared pose", True, False), ("shared pose + statistics", True, True)):
    tr = EventTrainer(params, dev, track_densification_stats=stats)
    tr.share_coincident_views = share
    step = lambda: (tr.compute_gradients(*cams, *gts, bg, sh_via_colour=True, viewspace_grad=stats), tr.apply_update())
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 30
    L.e3dgs_profile_enable(1 << 6)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    b, n = C.c_double(0), C.c_int(0)
    L.e3dgs_profile_query(6, C.byref(b), C.byref(n))
    L.e3dgs_profile_enable(0)
    print("SHARED_POSE_TIME %-28s %.3f ms per iteration (render_bwd %.3f ms)" % (tag, ms, b.value / max(n.value, 1)))
