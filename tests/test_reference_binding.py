"""The reference's OWN boundary code against the drop-in packages (CPU, build container only).

`gaussian_renderer/__init__.py` of the reference is imported unmodified, with this repository's
`diff_gaussian_rasterization` / `simple_knn` first on the path, and its `render()`, `render_depth()` and
`render_point()` are called on CPU tensors: everything on the reference's side of the boundary runs (settings
construction with its twelve keywords, `GaussianRasterizer(raster_settings=...)`, the forced Python SH branch, the
operator call with its eight keywords) and the call ends in the operator's own "no CPU path" error -- i.e. every name
and keyword the reference binds exists with the meaning it expects.  The numerics behind the boundary are the GPU
tests' business.  Skipped where /root/reference does not exist (the GPU box); runs in a subprocess so that the stub
modules for the reference's absent third-party imports stay out of this test session."""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys, types
    sys.dont_write_bytecode = True
    sys.path.insert(0, %r); sys.path.insert(1, %r)
    for m in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
              "torchgeometry", "lpips", "plyfile"):
        sys.modules[m] = types.ModuleType(m)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    import numpy as np, torch
    import diff_gaussian_rasterization, simple_knn._C
    assert diff_gaussian_rasterization.__file__.startswith(%r) and simple_knn._C.__file__.startswith(%r)
    import gaussian_renderer as gr                     # the reference's module, unmodified
    assert gr.__file__.startswith(%r)
    assert gr.GaussianRasterizer is diff_gaussian_rasterization.GaussianRasterizer
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix
    zeros_like = torch.zeros_like                      # the reference allocates its screen-space tensor on "cuda"
    torch.zeros_like = lambda *a, **k: zeros_like(*a, **{**k, "device": "cpu"}) if k.get("device") == "cuda" else zeros_like(*a, **k)
    zeros = torch.zeros
    torch.zeros = lambda *a, **k: zeros(*a, **{**k, "device": "cpu"}) if k.get("device") == "cuda" else zeros(*a, **k)
    torch.Tensor.cuda = lambda self, *a, **k: self

    class Cam:
        FoVx = FoVy = 0.69
        image_height, image_width = 48, 64
    cam = Cam()
    cam.world_view_transform = torch.tensor(getWorld2View2(np.eye(3), np.array([0.0, 0.0, 4.0]))).transpose(0, 1)
    cam.projection_matrix = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=0.69, fovY=0.69).transpose(0, 1)
    cam.full_proj_transform = cam.world_view_transform.unsqueeze(0).bmm(cam.projection_matrix.unsqueeze(0)).squeeze(0)
    cam.camera_center = cam.world_view_transform.inverse()[3, :3]

    class PC:
        active_sh_degree = max_sh_degree = 3
        def __init__(self, n=12):
            g = torch.Generator().manual_seed(0)
            self.get_xyz = torch.randn(n, 3, generator=g).requires_grad_(True)
            self.get_opacity = torch.rand(n, 1, generator=g)
            self.get_scaling = torch.rand(n, 3, generator=g) * 0.1
            self.get_rotation = torch.nn.functional.normalize(torch.randn(n, 4, generator=g))
            self.get_features = torch.randn(n, 16, 3, generator=g)
        def get_covariance(self, scaling_modifier=1.0):
            return torch.zeros(self.get_xyz.shape[0], 6)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False

    reached = []
    for name in ("render", "render_depth", "render_point"):
        for cov in (False, True):
            pipe = Pipe(); pipe.compute_cov3D_python = cov
            try:
                getattr(gr, name)(cam, PC(), pipe, torch.zeros(3))
            except RuntimeError as e:
                assert "no CPU path" in str(e), (name, str(e))
                reached.append((name, cov))
    print("REACHED", len(reached), reached)
''') % (ROOT, REF, ROOT, ROOT, REF)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")), reason="reference tree not present")
def test_reference_render_functions_bind_to_the_drop_in_packages():
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "REACHED 6" in r.stdout, r.stdout[-2000:]
