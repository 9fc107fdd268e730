#!/usr/bin/env python
"""Occupancy profile over time of the 3-view compositing launches (per-tile wall_clock stamps written by the kernels
when e3dgs_debug_set_trace is armed): how long is the tail of the LPT-ordered launch?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_3dgs_amd import _lib, synth, rasterizer
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
dev = torch.device("cuda:0")
L = _lib.lib()
L.e3dgs_debug_set_trace.argtypes = [C.c_void_p]
N, W, H = 1_000_000, 1920, 1080
tr = EventTrainer(synth.make_scene(N, "trained", seed=0, device=dev), dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
bg = torch.zeros(3, device=dev)
v = tr.views
T = 3 * ((W + 15) // 16) * ((H + 15) // 16)
settings = [tr._settings(c, bg) for c in cams]
names = dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"], scales=v["scaling"], rots=v["rotation"])
dpix = torch.randn(3, 3, H, W, device=dev)

def profile(which):
    buf = torch.zeros(T * 4, dtype=torch.int64, device=dev)
    for rep in range(3):
        arm_f = buf.data_ptr() if (rep == 2 and which == "fwd") else None
        L.e3dgs_debug_set_trace(arm_f)
        raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                       flags=tr.FWD_FLAGS)
        torch.cuda.synchronize()
        L.e3dgs_debug_set_trace(buf.data_ptr() if (rep == 2 and which == "bwd") else None)
        out = {n: torch.empty_like(t) for n, t in names.items()}
        rasterizer.backward_multi(raw, dpix, out)
        torch.cuda.synchronize()
        L.e3dgs_debug_set_trace(None)
    t = buf.cpu().numpy().reshape(T, 4)
    t = t[t[:, 1] > 0]
    T0, span = t[:, 0].min(), t[:, 1].max() - t[:, 0].min()
    edges = np.linspace(0, span, 21)
    act = [int(((t[:, 0] - T0) < edges[i + 1]).sum() - ((t[:, 1] - T0) < edges[i]).sum()) for i in range(20)]
    dur = t[:, 1] - t[:, 0]
    print(which, "tiles", len(t), "span(10ns)", span, "mean active waves", round(dur.sum() / span), "of 5120/8192 slots; profile", act)

profile("fwd")
profile("bwd")
