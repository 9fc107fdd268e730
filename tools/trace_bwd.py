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
buf = torch.zeros(T * 4, dtype=torch.int64, device=dev)
run(buf.data_ptr())
t = buf.cpu().numpy().reshape(T, 4)
dur = t[:, 1] - t[:, 0]
n_list, n = t[:, 2] >> 32, t[:, 2] & 0xFFFFFFFF
pro = (t[:, 3] & 0xFFFF) << 8
segk = ((t[:, 3] >> 40) & 0xFFFFFF) << 4
segr = ((t[:, 3] >> 16) & 0xFFFFFF) << 4
print('per-entry cycles: k-blocks', (segk / np.maximum(t[:, 2] & 0xFFFFFFFF, 1)).mean(), 'reduce+atomics', (segr / np.maximum(t[:, 2] & 0xFFFFFFFF, 1)).mean())
print("mean wave dur", dur.mean(), "max", dur.max(), "prologue mean/max", pro.mean(), pro.max())
print("list mean", n_list.mean(), "walked mean", n.mean(), "max", n.max())
cpe = (dur - pro) / np.maximum(n, 1)
print("cycles per walked entry mean", cpe.mean(), "p10/p50/p90", np.percentile(cpe, [10, 50, 90]))
