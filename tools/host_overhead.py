#!/usr/bin/env python
"""Host microseconds per operator call (forward + backward of a tiny scene, GPU time negligible) through the compiled
torch extension and through ctypes (GPU box)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import rasterizer, synth
from event_3dgs_amd.cameras import orbit_camera
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
N, W, H = 200, 48, 32
act = synth.activate(synth.make_scene(N, "trained", seed=0, device=dev))
cam = orbit_camera(0, 8, W, H, device=dev)
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
leaves = {k: v.detach().clone().requires_grad_(True) for k, v in act.items()}
m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
R = GaussianRasterizer(rs)


def call():
    img, _ = R(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], shs=leaves["shs"],
               scales=leaves["scales"], rotations=leaves["rotations"])
    img.sum().backward()


for name, native in (("compiled extension", True), ("ctypes", False), ("compiled extension", True), ("ctypes", False)):
    saved = rasterizer._NATIVE
    if not native:
        rasterizer.native_ext()
        rasterizer._NATIVE = False
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    print("%-20s %7.1f us per forward+backward (wall, incl. the instance-count wait)" % (name, 1e6 * (time.perf_counter() - t0) / n))
    rasterizer._NATIVE = saved
