"""Densification against the reference's own GaussianModel (golden G8, CPU)."""
import os

import numpy as np
import torch

from conftest import GOLDEN

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def _load(tag, g):
    groups = {n: [torch.tensor(g[f"{tag}_{n}"]), torch.tensor(g[f"{tag}_{n}_m"]), torch.tensor(g[f"{tag}_{n}_v"])] for n in NAMES}
    return groups


def test_g8_densify_and_prune_matches_reference_model():
    from event_3dgs_amd import densify
    g = np.load(os.path.join(GOLDEN, "densify.npz"))
    groups = _load("in", g)
    stats = densify.DensifyStats(256, "cpu")
    stats.xyz_gradient_accum = torch.tensor(g["in_accum"])
    stats.denom = torch.tensor(g["in_denom"])
    stats.max_radii2D = torch.tensor(g["in_maxr"])
    max_grad, min_opacity, extent, max_screen, pdense = (float(v) for v in g["args"])
    torch.manual_seed(77)                                   # the golden run seeded torch.normal the same way
    n = densify.densify_and_prune(groups, stats, max_grad, min_opacity, extent, max_screen, pdense)
    assert n == g["out_xyz"].shape[0] and n != 256
    for name in NAMES:
        for j, suf in enumerate(("", "_m", "_v")):
            ref = g[f"out_{name}{suf}"]
            assert groups[name][j].shape == ref.shape, (name, suf)
            assert np.allclose(groups[name][j].numpy(), ref, rtol=1e-6, atol=1e-7), (name, suf)
    assert np.array_equal(stats.xyz_gradient_accum.numpy(), g["out_accum"])
    assert np.array_equal(stats.denom.numpy(), g["out_denom"])
    assert np.array_equal(stats.max_radii2D.numpy(), g["out_maxr"])
    densify.reset_opacity(groups)
    assert np.allclose(groups["opacity"][0].numpy(), g["reset_opacity"], rtol=1e-6, atol=1e-7)
    assert float(groups["opacity"][1].abs().sum()) == 0.0 and np.array_equal(groups["opacity"][1].numpy(), g["reset_opacity_m"])
    assert np.allclose(groups["xyz"][1].numpy(), g["reset_xyz_m"])       # other groups keep their moments


def test_schedule_matches_train_py():
    from event_3dgs_amd.densify import densification_schedule as S
    assert S(100) == (True, False, None, False)
    assert S(600) == (True, True, None, False)
    assert S(3000) == (True, True, None, True)
    assert S(3100) == (True, True, 20, False)
    assert S(15000) == (False, False, 20, False)
    assert S(500, white_background=True)[3] is True


def test_stats_update_equals_masked_reference_form():
    from event_3dgs_amd.densify import DensifyStats
    g = torch.Generator().manual_seed(0)
    n = 100
    st = DensifyStats(n, "cpu")
    acc, den, mr = torch.zeros(n, 1), torch.zeros(n, 1), torch.zeros(n)
    for _ in range(3):
        grad = torch.randn(n, 3, generator=g)
        radii = torch.randint(0, 30, (n,), generator=g, dtype=torch.int32) * (torch.rand(n, generator=g) > 0.4)
        st.update(grad, radii)
        vis = radii > 0                                              # train.py:319-320, gaussian_model.py:405-407
        mr[vis] = torch.max(mr[vis], radii[vis].float())
        acc[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
        den[vis] += 1
    assert torch.allclose(st.xyz_gradient_accum, acc) and torch.equal(st.denom, den) and torch.equal(st.max_radii2D, mr)
