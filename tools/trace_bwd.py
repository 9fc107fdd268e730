#!/usr/bin/env python
"""Debug: per-tile timeline of render_bwd_kernel."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_3dgs_amd import _lib, synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians
dev = torch.device("cuda:0")
L = _lib.lib()
N, W, H = 1_000_000, 1920, 1080
act = synth.activate(synth.make_scene(N, "trained", seed=0, device=dev))
leaves = {k: v.detach().clone().requires_grad_(True) for k, v in act.items()}
cam = orbit_camera(0, 64, W, H, device=dev)
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                   1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
T = ((W + 15) // 16) * ((H + 15) // 16)
gw = torch.randn(3, H, W, device=dev)
def run(trace=None):
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    img, _ = rasterize_gaussians(leaves["means3D"], m2, leaves["shs"], None, leaves["opacities"], leaves["scales"], leaves["rotations"], None, rs)
    torch.cuda.synchronize()
    L.e3dgs_debug_set_trace(trace)
    (img * gw).sum().backward()
    torch.cuda.synchronize()
    L.e3dgs_debug_set_trace(None)
L.e3dgs_debug_set_trace.argtypes = [C.c_void_p]
for _ in range(3): run()
buf = torch.zeros(T * 6, dtype=torch.int64, device=dev)
run(buf.data_ptr())
t = buf.cpu().numpy().reshape(T, 6)
dur = t[:, 1] - t[:, 0]
n_list, n = t[:, 2] >> 32, t[:, 2] & 0xFFFFFFFF
xcc, hw = (t[:, 3] >> 32) & 0xF, t[:, 3] & 0xFFFFFFFF
cu, se, simd = (hw >> 8) & 0xF, (hw >> 13) & 0x7, (hw >> 4) & 0x3
print("mean wave dur", dur.mean(), "max", dur.max(), "walked mean", n.mean(), "max", n.max(), "cycles/entry p10/50/90", np.percentile(dur / np.maximum(n, 1), [10, 50, 90]))
print("global span (10ns ticks)", t[:, 1].max() - t[:, 0].min())
T0 = t[:, 0].min(); SPAN = t[:, 1].max() - T0
for x in range(8):
    m = xcc == x
    s0, e0 = t[m, 0], t[m, 1]
    t0 = T0; span = SPAN
    # active waves over time (20 buckets)
    edges = np.linspace(0, span, 21)
    act = [int(((s0 - t0) < edges[i + 1]) .sum() - ((e0 - t0) < edges[i]).sum()) for i in range(20)]
    busy = dur[m].sum()
    print(f"xcc{x}: waves {m.sum()} span {span} avg active {busy / span:.0f} (slots 640) profile {act}")
key = xcc * 10000 + se * 1000 + cu * 10 + simd
import collections
per = collections.defaultdict(list)
for k, s_, e_ in zip(key, t[:, 0], t[:, 1]): per[k].append((s_, e_))
print("SIMDs seen", len(per), "waves per SIMD min/mean/max", min(map(len, per.values())), np.mean(list(map(len, per.values()))), max(map(len, per.values())))
