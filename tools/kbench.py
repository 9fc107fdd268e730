#!/usr/bin/env python
"""Kernel micro-bench: per-stage HIP-event timings of forward+backward on the bench scene (GPU box)."""
import argparse, ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import _lib, synth
from event_3dgs_amd.cameras import orbit_camera
from event_3dgs_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--kind", default="trained")
a = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()
p = synth.make_scene(a.n, "trained", seed=0, device=dev)
act = synth.activate(p)
cam = orbit_camera(0, 64, a.w, a.h, device=dev)
bg = torch.zeros(3, device=dev)
rs = GaussianRasterizationSettings(a.h, a.w, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0,
                                   cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
leaves = {k: v.detach().clone().requires_grad_(True) for k, v in act.items()}
gw = torch.randn(3, a.h, a.w, device=dev)
def step():
    m2 = torch.zeros(a.n, 3, device=dev, requires_grad=True)
    img, radii = rasterize_gaussians(leaves["means3D"], m2, leaves["shs"], None, leaves["opacities"], leaves["scales"],
                                     leaves["rotations"], None, rs)
    (img * gw).sum().backward()
    return img
for _ in range(3):
    step()
torch.cuda.synchronize()
L.e3dgs_profile_enable(0xFF)
for _ in range(a.iters):
    img = step()
torch.cuda.synchronize()
out = {}
for s in range(8):
    ms, n = C.c_double(0), C.c_int(0)
    L.e3dgs_profile_query(s, C.byref(ms), C.byref(n))
    if n.value:
        out[L.e3dgs_profile_slot_name(s).decode()] = round(ms.value / n.value * (2 if s == 2 else 1), 4)
L.e3dgs_profile_enable(0)
print("KBENCH", out, "sum", round(sum(out.values()), 3), "checksum", float(img.double().sum()))
