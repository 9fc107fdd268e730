#!/usr/bin/env python
"""Do the three renders of an event iteration overlap when they are enqueued as three single-view passes on three HIP
streams (latency-bound list building of one view under the issue-bound compositing of another) -- against the ONE
three-view pass the trainer uses?  Forward only (projection, sorts, binning, compositing), buffers pre-sized.
Usage (GPU box): python tools/stream_overlap.py [N W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import synth, _lib, rasterizer
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.train_step import EventTrainer
N, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1_000_000, 1920, 1080)
dev = torch.device("cuda:0")
params = synth.make_scene(N, "trained", seed=0, device=dev)
bg = torch.zeros(3, device=dev)
cams = [orbit_camera(0, 64, W, H, device=dev, daz=d) for d in (0.0, 0.01, 0.025)]
tr = EventTrainer(params, dev)
v = tr.views
flags = tr.FWD_FLAGS
settings = [tr._settings(c, bg) for c in cams]
args = (v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"])
import gc; gc.disable()


def pinned():
    return torch.zeros(1, dtype=torch.int32).pin_memory()


# counts
raw = rasterizer.forward_multi(*args, settings, flags=flags)
total = raw["num_rendered"]
per_view = [rasterizer.forward_multi(*args, [s], flags=flags)["num_rendered"] for s in settings]
print("instances: 3-view", total, "single", per_view)
ref = raw["color"].clone()

REPS = 30


def timeit(fn, sync):
    for _ in range(3):
        fn()
    sync(); t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    sync()
    return 1e3 * (time.perf_counter() - t0) / REPS


pool3, cnt3 = rasterizer.ScratchPool(dev), pinned()
def multi():
    return rasterizer.forward_multi_capacity(*args, settings, int(total * 1.25), cnt3, flags=flags, pool=pool3)
t_multi = timeit(multi, torch.cuda.synchronize)

pools, cnts = [rasterizer.ScratchPool(dev) for _ in range(3)], [pinned() for _ in range(3)]
streams = [torch.cuda.Stream(dev) for _ in range(3)]
outs = [None] * 3
def single_one_stream():
    for i, s in enumerate(settings):
        outs[i] = rasterizer.forward_multi_capacity(*args, [s], int(per_view[i] * 1.25), cnts[i], flags=flags, pool=pools[i])
t_seq = timeit(single_one_stream, torch.cuda.synchronize)
chk = max(float((outs[i]["color"][0] - ref[i]).abs().max()) for i in range(3))

def single_three_streams():
    cur = torch.cuda.current_stream(dev)
    for i, s in enumerate(settings):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs[i] = rasterizer.forward_multi_capacity(*args, [s], int(per_view[i] * 1.25), cnts[i], flags=flags,
                                                        pool=pools[i])
    for st in streams:
        cur.wait_stream(st)
t_par = timeit(single_three_streams, torch.cuda.synchronize)
chk2 = max(float((outs[i]["color"][0] - ref[i]).abs().max()) for i in range(3))
print("STREAM_OVERLAP forward of three views: one 3-view pass %.3f ms | three single-view passes, one stream %.3f ms | "
      "three streams %.3f ms   (image diff %.1e %.1e)" % (t_multi, t_seq, t_par, chk, chk2))
