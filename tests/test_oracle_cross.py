"""The two independent CPU restatements agree: the C oracle's hand-derived backward (SURVEY
Appendix B) against PyTorch autograd of oracle/torch_oracle.py, all four input variants."""
import numpy as np
import pytest
import torch

from helpers import oracle_kwargs, rel_l2, scene
from oracle import c_oracle, torch_oracle


@pytest.mark.parametrize("use_sh,use_cov,dtype,radius", [
    (False, False, torch.float64, 4.0), (True, False, torch.float32, 4.0), (False, True, torch.float64, 4.0),
    (True, True, torch.float64, 1.2),      # camera inside the cloud: guard-band + near-plane cases
])
def test_c_oracle_backward_matches_autograd(use_sh, use_cov, dtype, radius):
    N, W, H = 500, 96, 64
    act, cam = scene(N, W, H, seed=11, radius=radius)
    bg = (0.2, 0.4, 0.1)
    kw = oracle_kwargs(act, cam, bg, use_sh, use_cov)
    f = c_oracle.Forward(**kw)
    L = {k: torch.tensor(kw[k]).to(dtype).requires_grad_(True)
         for k in ("means3D", "opacities", "colors_precomp", "shs", "scales", "rotations", "cov3D_precomp") if k in kw}
    m2 = torch.zeros(N, 3, dtype=dtype, requires_grad=True)
    img, radii, aux = torch_oracle.rasterize(
        L["means3D"], L["opacities"], viewmatrix=torch.tensor(kw["viewmatrix"]), projmatrix=torch.tensor(kw["projmatrix"]),
        campos=torch.tensor(kw["campos"]), bg=torch.tensor(kw["bg"]), width=W, height=H, tanfovx=kw["tanfovx"],
        tanfovy=kw["tanfovy"], colors_precomp=L.get("colors_precomp"), shs=L.get("shs"), sh_degree=3,
        scales=L.get("scales"), rotations=L.get("rotations"), cov3D_precomp=L.get("cov3D_precomp"), means2D=m2,
        return_aux=True)
    assert np.array_equal(radii.numpy(), f.radii)
    assert np.array_equal(aux["point_list"].astype(np.uint32), f.point_list)
    assert np.abs(img.detach().numpy() - f.out_color).max() <= 2e-6
    assert (aux["n_contrib"].numpy() == f.n_contrib).mean() >= 0.999
    gw = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5))
    (img * gw.to(dtype)).sum().backward()
    gb = f.backward(gw.numpy())
    pairs = {"means3D": "means3D", "opacities": "opacities", "colors_precomp": "colors", "shs": "shs", "scales": "scales",
             "rotations": "rotations", "cov3D_precomp": "cov3D"}
    for k, ok in pairs.items():
        if k in L:
            assert rel_l2(gb[ok].reshape(L[k].shape), L[k].grad.numpy()) <= 2e-5, k
    assert rel_l2(gb["means2D"], m2.grad.numpy()) <= 2e-5


def test_config1_5k_gaussians_256px_cpu_forward():
    """BASELINE config #1: 5k Gaussians, 256x256, single-view forward on the pure-PyTorch CPU path."""
    N, W, H = 5000, 256, 256
    act, cam = scene(N, W, H, seed=0)
    kw = oracle_kwargs(act, cam, (0, 0, 0), True, False)
    t = {k: torch.tensor(v) for k, v in kw.items() if isinstance(v, np.ndarray)}
    img, radii = torch_oracle.rasterize(t["means3D"], t["opacities"], viewmatrix=t["viewmatrix"],
                                        projmatrix=t["projmatrix"], campos=t["campos"], bg=t["bg"], width=W, height=H,
                                        tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"], shs=t["shs"], sh_degree=3,
                                        scales=t["scales"], rotations=t["rotations"])
    f = c_oracle.Forward(**kw)
    assert np.array_equal(radii.numpy(), f.radii)
    assert np.abs(img.numpy() - f.out_color).max() <= 1e-4
    assert (f.radii > 0).sum() > 3000
