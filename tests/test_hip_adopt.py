"""The adoption ladder (event_3dgs_amd/adopt.py): the reference's event iteration with its render() / loss block /
optimizers swapped one at a time for this repo's, every rung pinned to the fused EventTrainer.step -- and the C++ autograd
node of the drop-in operator against the Python one."""
import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(N=3000, W=176, H=128):
    from event_3dgs_amd import synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params = synth.make_scene(N, "trained", seed=0, device=DEV)
    cams = [orbit_camera(0, 16, W, H, device=DEV, daz=d) for d in (0.0, 0.004, 0.012)]
    bg = torch.zeros(3, device=DEV)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = EventTrainer(gp, DEV)
    gts = [t.render_raw(c, bg)["color"].clone() for c in cams]
    return params, cams, gts, bg


@pytest.mark.parametrize("deblur", [False, True])
@pytest.mark.parametrize("rung", [0, 1, 2, 3, 4])
def test_every_rung_of_the_ladder_equals_the_fused_trainer(rung, deblur):
    """Three iterations of train.py:144-212,330-332 at each rung against EventTrainer.step: the loss of every iteration to
    1e-6 relative (rung 0 runs torch's exp / sigmoid instead of the kernels' deterministic forms: 1e-5), Adam's first
    moments -- i.e. the gradients, which the sign-like first updates hide -- to fp32 summation order (rung 0: to the
    activations' 2e-6), the threshold c, and the parameters within what three steps can move them apart."""
    from event_3dgs_amd import adopt
    from event_3dgs_amd.train_step import EventTrainer
    params, cams, gts, bg = _scene()
    blur = (0.5 * (gts[0] + gts[2])).contiguous() if deblur else None
    ref = EventTrainer(params, DEV)
    loop = adopt.LadderLoop(rung, params, DEV)
    # (rungs 0 and 1 keep the reference's torch loss: on the black background the log-contrast has 1 / (Y + 1e-8) factors
    # of 1e8, so the fp32 order of the luminance sum shows in the gradient -- 1e-3 on the first moments of xyz)
    tol_loss, tol_m = (1e-5, 2e-3) if rung == 0 else ((1e-6, 2e-3) if rung == 1 else (1e-6, 1e-4))
    for it in range(3):
        lr = float(ref.step(cams[0], cams[1], cams[2], gts[0], gts[1], gts[2], bg, gt_blur=blur)[0])
        ll = float(loop.step(cams, gts, bg, gt_blur=blur))
        assert abs(ll - lr) <= tol_loss * abs(lr) * (1 + 20 * it), (it, ll, lr)      # (later iterations: the paths drift)
    torch.cuda.synchronize()
    g = ref.export_groups()
    name = dict(xyz="xyz", features_dc="f_dc", features_rest="f_rest", opacity="opacity", scaling="scaling", rotation="rotation")
    for k, gk in name.items():
        p = loop.P[k]
        st = loop.optimizer.state[p]
        assert int(st["step"]) == 3
        m_ref = g[gk][1].cpu().numpy()
        assert np.abs(m_ref).max() > 0
        assert rel_l2(st["exp_avg"].cpu().numpy().reshape(m_ref.shape), m_ref) <= tol_m * 3, k
        assert rel_l2(st["exp_avg_sq"].cpu().numpy().reshape(m_ref.shape), g[gk][2].cpu().numpy()) <= tol_m * 6, k
        assert float((p.detach() - g[gk][0]).abs().max()) <= 6.5 * loop.LR[k] + 1e-7, k      # <= 2 lr per step apart
    assert abs(float(loop.c) - float(ref.c)) <= 1e-4


def test_fast_render_matches_the_reference_render_and_falls_back():
    """adopt.render on the raw parameters == renderer.render (the reference's sequence) to the activations' rounding,
    same dict; viewspace_points.grad is filled; override_color / compute_cov3D_python take the reference's sequence."""
    from event_3dgs_amd import adopt, renderer
    params, cams, gts, bg = _scene()
    P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    pc = renderer.GaussianView(P, 3, 3)
    pipe = renderer.PipelineParams()
    a = adopt.render(cams[0], pc, pipe, bg)
    b = renderer.render(cams[0], pc, pipe, bg, force_python_sh=True)
    assert set(a) == set(b) == {"render", "viewspace_points", "visibility_filter", "radii"}
    assert (a["radii"] != b["radii"]).sum().item() <= 4
    assert (a["render"] - b["render"]).abs().mean().item() <= 1e-6
    (a["render"] * gts[0]).sum().backward()
    ga = {k: v.grad.clone() for k, v in P.items()}
    vs_a = a["viewspace_points"].grad.clone()
    for v in P.values():
        v.grad = None
    (b["render"] * gts[0]).sum().backward()
    assert vs_a.abs().max() > 0 and rel_l2(vs_a.cpu().numpy(), b["viewspace_points"].grad.cpu().numpy()) <= 1e-3
    for k, v in P.items():
        assert rel_l2(ga[k].cpu().numpy(), v.grad.cpu().numpy()) <= 1e-3, k
    # the branches the fast path does not serve
    col = torch.rand(P["xyz"].shape[0], 3, device=DEV)
    c = adopt.render(cams[0], pc, pipe, bg, override_color=col)
    d = renderer.render(cams[0], pc, pipe, bg, override_color=col)
    assert torch.equal(c["render"], d["render"])
    pipe2 = renderer.PipelineParams(compute_cov3D_python=True)
    e = adopt.render(cams[0], pc, pipe2, bg)
    assert (e["render"] - b["render"]).abs().mean().item() <= 1e-5
    with torch.no_grad():
        f = adopt.render(cams[1], pc, pipe, bg)
    assert not f["render"].requires_grad


def test_cpp_autograd_node_equals_the_python_function(monkeypatch):
    """The operator through csrc/ext.cpp's RasterizeFunction and through the Python autograd.Function (E3DGS_CPP_AUTOGRAD=0):
    the same image, radii and gradients, bit for bit, for SH / precomputed colours and scale+rotation / covariance."""
    import math
    from event_3dgs_amd import rasterizer, synth
    from event_3dgs_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians
    from oracle import torch_oracle
    assert rasterizer.cpp_autograd_ext() is not None
    params, cams, gts, bg = _scene(N=2000)
    act = synth.activate(params)
    cam = cams[0]
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                       bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                       False, False)
    cov = torch_oracle.build_cov3d(act["scales"].cpu(), act["rotations"].cpu(), 1.0).to(DEV)
    cols = torch.rand(2000, 3, device=DEV)
    for use_sh, use_cov in ((True, False), (False, False), (True, True)):
        res = []
        for cpp in ("1", "0"):
            monkeypatch.setenv("E3DGS_CPP_AUTOGRAD", cpp)
            leaves = {k: v.detach().clone().requires_grad_(True) for k, v in act.items()}
            m2 = torch.zeros(2000, 3, device=DEV, requires_grad=True)
            cv = cov.clone().requires_grad_(True)
            cl = cols.clone().requires_grad_(True)
            img, radii = rasterize_gaussians(leaves["means3D"], m2, leaves["shs"] if use_sh else None,
                                             None if use_sh else cl, leaves["opacities"],
                                             None if use_cov else leaves["scales"], None if use_cov else leaves["rotations"],
                                             cv if use_cov else None, rs)
            (img * gts[0]).sum().backward()
            grads = [m2.grad, leaves["means3D"].grad, leaves["opacities"].grad,
                     leaves["shs"].grad if use_sh else cl.grad, cv.grad if use_cov else leaves["scales"].grad]
            assert (leaves["scales"].grad is None) == use_cov
            res.append((img.detach(), radii, grads))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        for ga, gb in zip(res[0][2], res[1][2]):
            assert ga is not None and torch.equal(ga, gb)


def test_fused_adam_is_torch_adam():
    """adopt.FusedAdam against torch.optim.Adam on the reference's group structure: same parameters and state after steps
    with a skipped parameter (no gradient), a learning-rate change and a state replaced the way the reference's
    densification does it (scene/gaussian_model.py:258-271)."""
    from event_3dgs_amd import adopt
    gen = torch.Generator().manual_seed(0)
    shapes = dict(xyz=(500, 3), f_dc=(500, 1, 3), f_rest=(500, 15, 3), opacity=(500, 1))
    init = {k: torch.randn(*s, generator=gen).to(DEV) for k, s in shapes.items()}
    opts, params = [], []
    for Opt in (torch.optim.Adam, adopt.FusedAdam):
        P = {k: torch.nn.Parameter(v.clone()) for k, v in init.items()}
        groups = [{"params": [P[k]], "lr": 1e-3 * (i + 1), "name": k} for i, k in enumerate(P)]
        opts.append(Opt(groups, lr=0.0, eps=1e-15)); params.append(P)
    for it in range(5):
        gs = {k: torch.randn(*s, generator=gen).to(DEV) for k, s in shapes.items()}
        for opt, P in zip(opts, params):
            for k, p in P.items():
                p.grad = None if (k == "opacity" and it == 2) else gs[k].clone()
            if it == 3:
                opt.param_groups[0]["lr"] = 5e-4
                st = opt.state[P["f_dc"]]                                    # replace_tensor_to_optimizer
                st["exp_avg"] = torch.zeros_like(P["f_dc"]); st["exp_avg_sq"] = torch.zeros_like(P["f_dc"])
            opt.step()
    for k in shapes:
        a, b = params[0][k], params[1][k]
        assert torch.allclose(a, b, rtol=1e-6, atol=2e-6), k
        sa, sb = opts[0].state[a], opts[1].state[b]
        assert int(sa["step"]) == int(sb["step"]) == (4 if k == "opacity" else 5)
        # (torch's lerp / addcmul fuse their multiply-adds, the kernel does not: one rounding of O(|g|) per step)
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-6, atol=2e-7)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-9)
        assert a._version == b._version                      # the kernel's writes are visible to autograd's version counters


def test_deferred_renders_are_one_multi_view_pass_and_equal_three_immediate_renders():
    """adopt.render(defer=True): the three renders of an iteration run when the first image is used, as ONE multi-view pass
    (one launch of every rasteriser stage, counted with the library's profiler); images bit-identical to immediate renders,
    parameter gradients equal to fp32 summation order, render #1's screen-space gradient equal; shape / dtype / device of a
    pending result are answered without rendering; a first use under no_grad still builds the graph; a parameter update
    between render() and the first use raises."""
    import ctypes as C
    from event_3dgs_amd import _lib, adopt, renderer
    params, cams, gts, bg = _scene()
    L = _lib.lib()
    pipe = renderer.PipelineParams()
    w = [torch.randn(3, 128, 176, generator=torch.Generator().manual_seed(k)).to(DEV) for k in range(3)]

    def launches(slot):
        ms, n = C.c_double(0), C.c_int(0)
        L.e3dgs_profile_query(slot, C.byref(ms), C.byref(n))
        return n.value

    def run(defer):
        P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
        pc = renderer.GaussianView(P, 3, 3)
        L.e3dgs_profile_enable(0xFF)
        pk = [adopt.render(c, pc, pipe, bg, defer=defer) for c in cams]
        if defer:
            assert launches(0) == 0, "nothing may have been rendered yet"
            assert tuple(pk[1]["render"].shape) == (3, 128, 176) and pk[1]["render"].dtype == torch.float32
            assert pk[2]["radii"].shape[0] == params["xyz"].shape[0] and pk[0]["render"].is_cuda
            assert launches(0) == 0, "metadata of a pending result is answered without rendering"
            with torch.no_grad():
                _ = float(pk[2]["render"].mean())               # first use inside no_grad (a logging line)
        loss = sum((p["render"] * wk).sum() for p, wk in zip(pk, w))
        loss.backward()
        torch.cuda.synchronize()
        n_pre, n_bwd = launches(0), launches(6)
        L.e3dgs_profile_enable(0)
        return pk, P, n_pre, n_bwd
    pk_d, P_d, pre_d, bwd_d = run(True)
    pk_i, P_i, pre_i, bwd_i = run(False)
    assert (pre_d, bwd_d) == (1, 1) and (pre_i, bwd_i) == (3, 3)
    for a, b in zip(pk_d, pk_i):
        assert torch.equal(adopt.materialize(a["render"]), b["render"])
        assert torch.equal(adopt.materialize(a["radii"]), b["radii"])
        assert torch.equal(adopt.materialize(a["visibility_filter"]), b["visibility_filter"])
    for k in P_d:
        ga, gb = P_d[k].grad.cpu().numpy(), P_i[k].grad.cpu().numpy()
        assert rel_l2(ga, gb) <= 2e-5, k
    assert rel_l2(pk_d[0]["viewspace_points"].grad.cpu().numpy(), pk_i[0]["viewspace_points"].grad.cpu().numpy()) <= 2e-5
    for k in (1, 2):              # every render's own screen-space gradient (E3DGS_FLAG_MEAN2D_VIEWS)
        assert rel_l2(pk_d[k]["viewspace_points"].grad.cpu().numpy(), pk_i[k]["viewspace_points"].grad.cpu().numpy()) <= 2e-5
    # visibility_filter as an index (train.py:318-320) and an image used after the optimizer moved the parameters
    P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    pc = renderer.GaussianView(P, 3, 3)
    pkg = adopt.render(cams[0], pc, pipe, bg, defer=True)
    acc = torch.zeros(params["xyz"].shape[0], device=DEV)
    acc[pkg["visibility_filter"]] = torch.max(acc[pkg["visibility_filter"]], pkg["radii"][pkg["visibility_filter"]].float())
    assert float(acc.max()) > 0
    stale = adopt.render(cams[1], pc, pipe, bg, defer=True)
    with torch.no_grad():
        P["xyz"].add_(1e-3)
    with pytest.raises(RuntimeError, match="modified in place"):
        stale["render"].sum()
