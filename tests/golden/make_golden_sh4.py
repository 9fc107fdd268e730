#!/usr/bin/env python
"""Golden vectors for SH degree 4 (utils/sh_utils.py:38-48,97-110), made by IMPORTING the reference in this container:
    python tests/golden/make_golden_sh4.py      ->  tests/golden/sh_deg4.npz
Same recipe as G1 in make_golden.py (colours as gaussian_renderer/__init__.py:74-81 computes them, + 0.5, clamp, autograd
gradients), with 25 coefficients per channel.  A separate file so that the seeded stream of make_golden.py stays as is."""
import os

import numpy as np
import torch

from make_golden import OUT, _import_reference


def main():
    _import_reference()
    from utils.sh_utils import eval_sh
    g = torch.Generator().manual_seed(4321)
    N = 64
    feats = torch.randn(N, 25, 3, generator=g) * 0.3
    xyz = torch.randn(N, 3, generator=g)
    campos = torch.tensor([0.3, -0.2, 2.5])
    gcol = torch.randn(N, 3, generator=g)
    out = {"features": feats.numpy(), "xyz": xyz.numpy(), "campos": campos.numpy(), "grad_colors": gcol.numpy()}
    f = feats.clone().requires_grad_(True)
    p = xyz.clone().requires_grad_(True)
    shs_view = f.transpose(1, 2).view(-1, 3, 25)
    dir_pp = p - campos.repeat(N, 1)
    dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    col = torch.clamp_min(eval_sh(4, shs_view, dirn) + 0.5, 0.0)
    (col * gcol).sum().backward()
    out["colors_deg4"], out["dfeatures_deg4"], out["dxyz_deg4"] = col.detach().numpy(), f.grad.numpy(), p.grad.numpy()
    np.savez(os.path.join(OUT, "sh_deg4.npz"), **out)
    print("wrote sh_deg4.npz; clamped channels:", int((col == 0).sum()))


if __name__ == "__main__":
    main()
