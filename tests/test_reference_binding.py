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


ADAM_SCRIPT = textwrap.dedent('''
    import sys, types
    sys.dont_write_bytecode = True
    sys.path.insert(0, %r); sys.path.insert(1, %r)
    for m in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
              "torchgeometry", "lpips", "plyfile"):
        sys.modules[m] = types.ModuleType(m)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    import math, torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    def _wrap(fn):
        def inner(*a, **k):
            if k.get("device") is not None and "cuda" in str(k["device"]):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    for name in ["zeros", "ones", "tensor", "empty", "rand", "zeros_like", "ones_like", "full"]:
        setattr(torch, name, _wrap(getattr(torch, name)))
    from argparse import ArgumentParser
    from arguments import OptimizationParams
    from scene.gaussian_model import GaussianModel            # the reference's class, unmodified
    from event_3dgs_amd import adopt

    def model(fused):
        gm = GaussianModel(3)
        n = 200
        g = torch.Generator().manual_seed(5)
        P = lambda t: torch.nn.Parameter(t.requires_grad_(True))
        gm._xyz = P(torch.randn(n, 3, generator=g))
        gm._features_dc = P(torch.randn(n, 1, 3, generator=g))
        gm._features_rest = P(torch.randn(n, 15, 3, generator=g) * 0.1)
        gm._scaling = P(torch.randn(n, 3, generator=g) * 0.8 + math.log(0.04))
        gm._rotation = P(torch.randn(n, 4, generator=g))
        gm._opacity = P(torch.randn(n, 1, generator=g) * 3.0)
        gm.max_radii2D = torch.zeros(n)
        gm.spatial_lr_scale = 1.0
        gm.training_setup(OptimizationParams(ArgumentParser()))
        if fused:   # rung 3: the one line of scene/gaussian_model.py:163, with the groups training_setup built
            gm.optimizer = adopt.FusedAdam([dict(params=grp["params"], lr=grp["lr"], name=grp["name"])
                                            for grp in gm.optimizer.param_groups], lr=0.0, eps=1e-15)
        for grp in gm.optimizer.param_groups:      # the state one optimizer step leaves (FusedAdam.step itself needs the GPU)
            p = grp["params"][0]
            gm.optimizer.state[p] = {"step": torch.tensor(1.0), "exp_avg": torch.randn(p.shape, generator=g) * 1e-3,
                                     "exp_avg_sq": torch.rand(p.shape, generator=g) * 1e-6}
        gm.xyz_gradient_accum = torch.rand(n, 1, generator=g) * 6e-4
        gm.denom = torch.randint(0, 3, (n, 1), generator=g).float()
        gm.max_radii2D = torch.rand(n, generator=g) * 40
        return gm

    def snapshot(gm):
        out = {}
        for grp in gm.optimizer.param_groups:
            p = grp["params"][0]
            st = gm.optimizer.state[p]
            out[grp["name"]] = (p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), float(grp["lr"]))
        return out

    a, b = model(False), model(True)
    assert adopt._reference_activations(b) and all(hasattr(b, n) for n in adopt._RAW_ATTRS)
    for gm in (a, b):
        torch.manual_seed(77)
        gm.densify_and_prune(0.0002, 0.005, 4.0, 20)         # cat_tensors_to_optimizer / _prune_optimizer on the state
        gm.reset_opacity()                                   # replace_tensor_to_optimizer
        gm.update_learning_rate(1234)
    sa, sb = snapshot(a), snapshot(b)
    assert sa.keys() == sb.keys() and len(sa) == 6
    for k in sa:
        assert sa[k][0].shape[0] != 200
        for x, y in zip(sa[k][:3], sb[k][:3]):
            assert torch.equal(x, y), k
        assert sa[k][3] == sb[k][3], k
    assert isinstance(b.optimizer, adopt.FusedAdam)
    # rung 1 on the reference's own model, CPU tensors: the fast path is not taken (no GPU) and the reference's sequence
    # ends at the operator's "no CPU path" error
    class Cam:
        FoVx = FoVy = 0.69
        image_height, image_width = 48, 64
        world_view_transform = torch.eye(4); full_proj_transform = torch.eye(4); camera_center = torch.zeros(3)
    class Pipe:
        convert_SHs_python = False; compute_cov3D_python = False; debug = False
    try:
        adopt.render(Cam(), b, Pipe(), torch.zeros(3))
        raise SystemExit("adopt.render ran on CPU tensors")
    except RuntimeError as e:
        assert "no CPU path" in str(e), str(e)
    print("FUSED_ADAM_STATE_OK", {k: tuple(v[0].shape) for k, v in sb.items()})
''') % (ROOT, REF)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")), reason="reference tree not present")
def test_reference_densification_code_runs_on_the_fused_optimizer():
    """Rung 3 of the adoption ladder against the reference's OWN `GaussianModel` (scene/gaussian_model.py, unmodified): with
    `adopt.FusedAdam` in place of `torch.optim.Adam` (:163), `densify_and_prune` (-> cat_tensors_to_optimizer, _prune_optimizer
    :273-347), `reset_opacity` (-> replace_tensor_to_optimizer :258-271) and `update_learning_rate` (:165-171) leave the same
    parameters, Adam moments and learning rates as with torch's optimizer, bit for bit -- the state layout is torch's."""
    r = subprocess.run([sys.executable, "-c", ADAM_SCRIPT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "FUSED_ADAM_STATE_OK" in r.stdout, r.stdout[-2000:]
