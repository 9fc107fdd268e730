"""Rank-1 pixel gradients (ABI 16: e3dgs_rasterize_backward_multi_rank1, e3dgs_event_loss_rank1, e3dgs_image_loss_rank1).

A render that enters the loss only through a luminance has dL/dC(pixel) = s(pixel) * w: the two contrast renders of an event
iteration (rgb_to_LUVscale, utils/loss_utils.py:24-28,234-249; train.py:159-176) and the --gray iteration
(rgb_to_grayscale, :18-23,40-48; train.py:213-223).  The loss kernels then hand out the scalar field s and the compositing
backward runs one colour chain and seven sums per (pixel, entry) instead of three and nine.  These tests pin the rank-1
form to the general one (which the oracle tests pin): the oracle itself is compared in
test_hip_configs.py::test_cfg3_the_path_the_benchmark_times_against_the_oracle[rank1]."""
import numpy as np
import pytest
import torch

from helpers import per_gaussian_err, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(N=4000, W=183, H=131, seed=0):
    from event_3dgs_amd import synth
    from event_3dgs_amd.cameras import orbit_camera
    params = synth.make_scene(N, "trained", seed=seed, device=DEV)
    cams = [orbit_camera(0, 16, W, H, device=DEV, daz=d) for d in (0.0, 0.004, 0.012)]
    return params, cams


def _outs(tr, with_m2d=True):
    v = tr.views
    names = dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"], scales=v["scaling"], rots=v["rotation"])
    out = {n: torch.full_like(t, float("nan")) for n, t in names.items()}
    if with_m2d:
        out["means2D"] = torch.full((tr.N, 3), float("nan"), device=DEV)
    return out


@pytest.mark.parametrize("views", [(1, 2), (0,), (0, 1, 2), (2,)])
def test_rank1_backward_equals_the_general_backward_on_the_expanded_gradient(views):
    """backward_multi(rank1={v: w}) on (s in plane 0, NaN in planes 1-2) == backward_multi on s * w: every output within
    fp32 summation order, per Gaussian; deterministic; the frame has partial tiles in both directions."""
    from event_3dgs_amd import rasterizer
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene()
    W, H = cams[0].image_width, cams[0].image_height
    bg = torch.tensor([0.2, 0.1, 0.3], device=DEV)
    tr = EventTrainer(params, DEV)
    v = tr.views
    settings = [tr._settings(c, bg) for c in cams]
    gen = torch.Generator().manual_seed(5)
    full = torch.randn(3, 3, H, W, generator=gen).to(DEV)
    r1_in = full.clone()
    weights = {1: rasterizer.LUV_WEIGHTS, 2: rasterizer.LUV_WEIGHTS, 0: rasterizer.GRAY_WEIGHTS}
    for k in views:
        s = full[k, 0].clone()
        for ch in range(3):
            full[k, ch] = s * weights[k][ch]
        r1_in[k, 0] = s
        r1_in[k, 1:] = float("nan")                       # must not be read
    raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                   flags=tr.FWD_FLAGS)
    ref, got, again = _outs(tr), _outs(tr), _outs(tr)
    rasterizer.backward_multi(raw, full, ref)
    rasterizer.backward_multi(raw, r1_in, got, rank1={k: weights[k] for k in views})
    rasterizer.backward_multi(raw, r1_in, again, rank1={k: weights[k] for k in views})
    torch.cuda.synchronize()
    for n in ref:
        a, b = got[n].cpu().numpy(), ref[n].cpu().numpy()
        assert np.isfinite(a).all(), n
        assert np.abs(b).max() > 0, n
        if n == "sh":
            a, b = a.T, b.T                                # rows = Gaussians
        assert rel_l2(a, b) <= 2e-6, (n, rel_l2(a, b))
        assert per_gaussian_err(a.reshape(tr.N, -1), b.reshape(tr.N, -1)) <= 1e-4, n
        assert torch.equal(got[n], again[n]), n           # no atomics: bit-reproducible


def test_rank1_argument_checks():
    from event_3dgs_amd import _lib, rasterizer
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=500)
    bg = torch.zeros(3, device=DEV)
    tr = EventTrainer(params, DEV)
    v = tr.views
    settings = [tr._settings(c, bg) for c in cams[:2]]
    raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                   flags=tr.FWD_FLAGS)
    H, W = cams[0].image_height, cams[0].image_width
    g = torch.zeros(2, 3, H, W, device=DEV)
    with pytest.raises(ValueError):
        rasterizer.backward_multi(raw, g, _outs(tr), rank1={2: rasterizer.LUV_WEIGHTS})      # no such view
    with pytest.raises(_lib.HipLibraryError, match="second gradient chain"):
        rasterizer.backward_multi(raw, g, _outs(tr), stats_grad_view0=g[0].contiguous(), rank1={0: rasterizer.GRAY_WEIGHTS})


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("cached", [False, True])
def test_event_loss_rank1_is_the_general_gradient_divided_by_the_weights(shared, cached):
    """e3dgs_event_loss_rank1: scalars and dL/dc bit-identical to e3dgs_event_loss; plane 0 of d_next (and of d_now when it
    is a render of its own) times the luminance weights reproduces the general gradient to rounding; planes 1, 2 of those
    outputs are not written; d_image (and the shared render's total) unchanged bit for bit."""
    from event_3dgs_amd import losses, rasterizer
    H, W = 96, 132
    gen = torch.Generator().manual_seed(3)
    r = lambda: (torch.rand(3, H, W, generator=gen) * 0.9 + 0.05).to(DEV)
    image, nxt, gi, gn, gx = r(), r(), r(), r(), r()
    gn = (gn * 20).round() / 20                     # quantised targets: some D* == 0 exactly
    gx = torch.where(torch.rand(1, H, W, generator=gen).to(DEV) < 0.3, gn, (gx * 20).round() / 20)
    now = image if shared else r()
    c = torch.tensor([0.21], device=DEV)
    pc_a, pc_b = (losses.PairCounts(), losses.PairCounts()) if cached else (None, None)
    for _ in range(2 if cached else 1):             # second round: the one-sweep kernel
        sa, da_i, da_n, da_x = losses.event_loss_raw(image, now, nxt, c, gi, gn, gx, pair_counts=pc_a)
        mark = torch.full((3, H, W), 7.0, device=DEV)
        out = (torch.empty(8, device=DEV), mark.clone(), mark.clone(), mark.clone(),
               torch.empty(losses._lib.lib().e3dgs_event_loss_scratch_bytes(W, H), dtype=torch.uint8, device=DEV))
        sb, db_i, db_n, db_x = losses.event_loss_raw(image, now, nxt, c, gi, gn, gx, out=out, pair_counts=pc_b, rank1=True)
        torch.cuda.synchronize()
        assert torch.equal(sa[:6], sb[:6])
        assert torch.equal(da_i, db_i)
        w = torch.tensor(rasterizer.LUV_WEIGHTS, device=DEV).view(3, 1, 1)
        assert torch.all(db_x[1:] == 7.0)
        assert torch.allclose(db_x[0:1] * w, da_x, rtol=3e-7, atol=0)
        if shared:
            assert torch.equal(da_n, db_n)          # the shared render's total: a full gradient
        else:
            assert torch.all(db_n[1:] == 7.0)
            assert torch.allclose(db_n[0:1] * w, da_n, rtol=3e-7, atol=0)


def test_gray_loss_rank1():
    from event_3dgs_amd import losses, rasterizer
    H, W = 75, 101
    gen = torch.Generator().manual_seed(4)
    img, gt = (torch.rand(3, H, W, generator=gen).to(DEV) for _ in range(2))
    sa, da = losses.image_loss_raw(img, gt, True, 0.2)
    mark = torch.full((3, H, W), 7.0, device=DEV)
    sb, db = losses.image_loss_raw(img, gt, True, 0.2,
                                   out=(torch.empty(4, device=DEV), mark,
                                        torch.empty(losses._lib.lib().e3dgs_image_loss_scratch_bytes(3, H, W),
                                                    dtype=torch.uint8, device=DEV)), rank1=True)
    torch.cuda.synchronize()
    assert torch.equal(sa[:3], sb[:3])
    assert torch.all(db[1:] == 7.0)
    w = torch.tensor(rasterizer.GRAY_WEIGHTS, device=DEV).view(3, 1, 1)
    assert torch.allclose(db[0:1] * w, da, rtol=3e-7, atol=0)
    with pytest.raises(ValueError):
        losses.image_loss_raw(img, gt, False, 0.2, rank1=True)


@pytest.mark.parametrize("mode", ["event", "event_shared", "event_shared_stats", "gray"])
def test_trainer_iterations_with_rank1_gradients_equal_the_general_form(mode):
    """EventTrainer (rank1 on by default; E3DGS_RANK1=0 / trainer.rank1 = False: general form): the same loss bits and the
    same gradients to fp32 summation order, for the three-render iteration, the shared-pose iteration (only `next` is rank
    1 there), the shared-pose iteration with the statistics chain in view 0, and the --gray iteration."""
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, cams = _scene(N=3000, W=176, H=128)
    bg = torch.tensor([0.2, 0.2, 0.2], device=DEV)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(params["xyz"].shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    gts = [EventTrainer(gp, DEV).render_raw(c, bg)["color"].clone() for c in cams]
    stats = mode == "event_shared_stats"
    a, b = (EventTrainer(params, DEV, track_densification_stats=stats) for _ in range(2))
    assert a.rank1
    b.rank1 = False
    for t in (a, b):
        t.SHARE_STATS_MIN_INSTANCES = t.SHARE_STATS_MIN_TILES = 0
    if mode == "gray":
        la = a.compute_gradients_image(cams[0], gts[0], bg, "gray").clone()
        lb = b.compute_gradients_image(cams[0], gts[0], bg, "gray").clone()
        assert torch.equal(la, lb)
    else:
        cam_now = orbit_camera(0, 16, 176, 128, device=DEV, daz=0.0) if mode != "event" else cams[1]
        sa = a.compute_gradients(cams[0], cam_now, cams[2], gts[0], gts[1], gts[2], bg).clone()
        sb = b.compute_gradients(cams[0], cam_now, cams[2], gts[0], gts[1], gts[2], bg).clone()
        assert torch.equal(sa[:6], sb[:6])
        assert a.shared_pose_iterations == (0 if mode == "event" else 1)
    torch.cuda.synchronize()
    for name in ("xyz", "features", "opacity", "scaling", "rotation"):
        ga, gb = a.grads[name].cpu().numpy(), b.grads[name].cpu().numpy()
        assert np.abs(gb).max() > 0 and np.isfinite(ga).all()
        assert rel_l2(ga, gb) <= 3e-5, (name, rel_l2(ga, gb))      # (s w summed as w (sum s): fp32 rounding, measured 1e-5)
    if stats:
        assert rel_l2(a.viewspace_grad.cpu().numpy(), b.viewspace_grad.cpu().numpy()) <= 3e-5
