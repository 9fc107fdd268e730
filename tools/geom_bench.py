#!/usr/bin/env python
"""Times the per-Gaussian half of the backward (run_reduce + geom_bwd_multi: E3DGS_FLAG_BWD_ONLY_GEOM) and the compositing
half (E3DGS_FLAG_BWD_ONLY_RENDER) alone on the benchmark iteration, with HIP events on the launch stream.
Usage: python tools/geom_bench.py [N] [W] [H]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_3dgs_amd import _lib, rasterizer, synth
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
tr = EventTrainer(params, dev)
cap = {}
orig = rasterizer.backward_multi
def spy(raw, dpix, out, flags=None, grad_acc=None, **kw):
    cap["a"] = (raw, dpix, out)
    return orig(raw, dpix, out, flags, grad_acc, **kw)
rasterizer.backward_multi = spy
tr.compute_gradients(*cams, *gts, bg, sh_via_colour=True)
rasterizer.backward_multi = orig
raw, dpix, out = cap["a"]
def timed(flags, n=20):
    for _ in range(3): orig(raw, dpix, out, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): orig(raw, dpix, out, flags)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("GEOMBENCH N=%d %dx%d: geom half %.4f ms, render half %.4f ms" % (N, W, H, timed(raw["flags"] | _lib.FLAG_BWD_ONLY_GEOM), timed(raw["flags"] | _lib.FLAG_BWD_ONLY_RENDER)))
