"""Shared scene builders for the parity tests (seeded, small enough for the CPU oracle)."""
import math

import numpy as np
import torch

from event_3dgs_amd import synth
from event_3dgs_amd.cameras import orbit_camera


def scene(N, W, H, seed=0, kind="trained", k=0, K=8, scale_boost=1.0, radius=4.0):
    params = synth.make_scene(N, kind, seed=seed)
    act = synth.activate(params)
    act["scales"] = act["scales"] * scale_boost
    cam = orbit_camera(k, K, W, H, radius=radius)
    g = torch.Generator().manual_seed(seed + 100)
    act["colors"] = torch.rand(N, 3, generator=g)
    return act, cam


def oracle_kwargs(act, cam, bg, use_sh, use_cov, sh_degree=3, scale_modifier=1.0):
    from oracle import torch_oracle
    kw = dict(means3D=act["means3D"].numpy(), opacities=act["opacities"].numpy(),
              viewmatrix=cam.world_view_transform.contiguous().numpy(), projmatrix=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.contiguous().numpy(), bg=np.asarray(bg, np.float32), width=cam.image_width,
              height=cam.image_height, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
              scale_modifier=scale_modifier)
    if use_sh:
        kw.update(shs=act["shs"].numpy(), sh_degree=sh_degree)
    else:
        kw.update(colors_precomp=act["colors"].numpy())
    if use_cov:
        kw.update(cov3D_precomp=torch_oracle.build_cov3d(act["scales"], act["rotations"], scale_modifier).numpy())
    else:
        kw.update(scales=act["scales"].numpy(), rotations=act["rotations"].numpy())
    return kw


from oracle.metrics import per_gaussian_err, rel_l2  # noqa: E402,F401
