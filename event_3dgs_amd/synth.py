"""Seeded synthetic scenes of SURVEY 8(d) (there is no dataset in the tree and no network).

"trained-like": means ~ U[-1.3,1.3]^3 (the reference's own random init, scene/dataset_readers.py:291),
log-normal anisotropic scales around 0.35x the expected nearest-neighbour spacing, random
quaternions, sigmoid(N(0,2^2)) opacities, SH DC from U[0,1] colours, rest ~ N(0,0.05^2).
"init-like": isotropic scales sqrt(distCUDA2), identity quaternions, opacity 0.1
(scene/gaussian_model.py:124-147).
"""
import math

import torch

C0 = 0.28209479177387814


def RGB2SH(rgb):
    """utils/sh_utils.py:114"""
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):
    """utils/general_utils.py:18"""
    return torch.log(x / (1 - x))


def make_scene(N, kind="trained", seed=0, device="cpu", dist2_fn=None):
    """Returns a dict of PRE-activation parameters laid out like scene/gaussian_model.py:126-147:
    xyz (N,3), features_dc (N,1,3), features_rest (N,15,3), scaling (N,3) log, rotation (N,4), opacity (N,1) logit."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(N, 3, generator=g) * 2.0 - 1.0) * 1.3
    if kind == "trained":
        s0 = 0.554 * (17.576 / N) ** (1.0 / 3.0)
        scaling = torch.randn(N, 3, generator=g) * 0.6 + math.log(0.35 * s0)
        rotation = torch.randn(N, 4, generator=g)
        opacity = torch.randn(N, 1, generator=g) * 2.0
        dc = RGB2SH(torch.rand(N, 1, 3, generator=g))
        rest = torch.randn(N, 15, 3, generator=g) * 0.05
    elif kind == "init":
        if dist2_fn is None:
            raise ValueError("init-like scenes need dist2_fn (distCUDA2)")
        dist2 = torch.clamp_min(dist2_fn(xyz.to(device)).cpu(), 1e-7)
        scaling = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3)
        rotation = torch.zeros(N, 4)
        rotation[:, 0] = 1
        opacity = inverse_sigmoid(0.1 * torch.ones(N, 1))
        dc = RGB2SH(torch.rand(N, 1, 3, generator=g))
        rest = torch.zeros(N, 15, 3)
    else:
        raise ValueError(kind)
    out = dict(xyz=xyz, features_dc=dc, features_rest=rest, scaling=scaling, rotation=rotation, opacity=opacity)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def activate(params):
    """Getters of scene/gaussian_model.py:95-118."""
    return dict(
        means3D=params["xyz"],
        scales=torch.exp(params["scaling"]),
        rotations=torch.nn.functional.normalize(params["rotation"]),
        opacities=torch.sigmoid(params["opacity"]),
        shs=torch.cat((params["features_dc"], params["features_rest"]), dim=1),
    )
