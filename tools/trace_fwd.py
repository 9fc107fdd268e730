#!/usr/bin/env python
"""Debug: per-tile timeline of render_fwd_kernel (start/end clocks, entries, CU placement)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_3dgs_amd import _lib, synth, rasterizer
from event_3dgs_amd.cameras import orbit_camera
dev = torch.device("cuda:0")
L = _lib.lib()
N, W, H = 1_000_000, 1920, 1080
act = synth.activate(synth.make_scene(N, "trained", seed=0, device=dev))
cam = orbit_camera(0, 64, W, H, device=dev)
rs = rasterizer.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                              1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
T = ((W + 15) // 16) * ((H + 15) // 16)
run = lambda: rasterizer.forward_raw(act["means3D"], act["shs"], None, act["opacities"], act["scales"], act["rotations"], None, rs)
for _ in range(3): run()
buf = torch.zeros(T * 6, dtype=torch.int64, device=dev)
L.e3dgs_debug_set_trace.argtypes = [C.c_void_p]
L.e3dgs_debug_set_trace(buf.data_ptr())
run(); torch.cuda.synchronize()
L.e3dgs_debug_set_trace(None)
t = buf.cpu().numpy().reshape(T, 6)
start, end = t[:, 0], t[:, 1]
n, proc = t[:, 2] >> 32, t[:, 2] & 0xFFFFFFFF
xcc, hw = (t[:, 3] >> 32) & 0xF, t[:, 3] & 0xFFFFFFFF
cu, se, simd = (hw >> 8) & 0xF, (hw >> 13) & 0x7, (hw >> 4) & 0x3
t0 = start.min()
dur = end - start
print("kernel span (clk)", end.max() - t0, "mean wave dur", dur.mean(), "max", dur.max(), "start spread", (start - t0).max())
print("entries: list mean", n.mean(), "processed mean", proc.mean(), "max", proc.max())
print("cycles per processed entry: mean", (dur / np.maximum(proc, 1)).mean(), "p10/p50/p90", np.percentile(dur / np.maximum(proc, 1), [10, 50, 90]))
key = xcc * 1000 + se * 100 + cu
import collections
load = collections.Counter(); cnt = collections.Counter()
for k, p in zip(key, proc): load[k] += p; cnt[k] += 1
v = np.array(list(load.values())); c = np.array(list(cnt.values()))
print("CUs seen", len(v), "tiles/CU min/mean/max", c.min(), c.mean(), c.max(), "entries/CU min/mean/max", v.min(), v.mean(), v.max())
# end time per CU
endcu = collections.defaultdict(int)
for k, e in zip(key, end - t0): endcu[k] = max(endcu[k], e)
e = np.array(list(endcu.values())); print("CU finish time min/mean/max", e.min(), e.mean(), e.max())
order = np.argsort(end)
print("last 5 tiles to finish: proc", proc[order[-5:]], "dur", dur[order[-5:]], "start", (start - t0)[order[-5:]])
